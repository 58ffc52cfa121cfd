#!/bin/bash
# f4: what the device line pass buys end to end (line-printing modes), and what k_lines costs on the device
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY' 2>&1 | tee gpurun_out/s_line_pass.txt
import os, subprocess, sys, time, shutil
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
from fullsize_parity import gen_files
from grab_amd import bin_path, synth
base = "/dev/shm/grab_s_%d" % os.getpid()
os.makedirs(base)
try:
    gen_files(base, 128, 64 << 20, 64)
    nbytes = 128 * (64 << 20)
    ref = os.path.join(os.getcwd(), "oracle", "_ref", "grab_jit")
    def timed(argv, env=None):
        best = None
        for it in range(3):
            t0 = time.perf_counter(); r = subprocess.run(argv, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env); dt = time.perf_counter() - t0
            if it: best = dt if best is None else min(best, dt)
        return best
    for pat, name in ((synth.IDENT_RE, "identifier"), ("e+", "e+")):
        for flags in (["-O"], []):
            dt = timed([ref, "-n", "64", "-r"] + flags + [pat, base])
            print("%-10s %-4s reference -n 64        : %.3f s = %5.1f GB/s" % (name, " ".join(flags), dt, nbytes / dt / 1e9), flush=True)
            for w in (8,):
                a = timed([bin_path(), "-n", str(w), "-r"] + flags + [pat, base], dict(os.environ, GRAB_LINE_PASS="1"))
                b = timed([bin_path(), "-n", str(w), "-r"] + flags + [pat, base])
                print("%-10s %-4s grab -n %-2d line pass on : %.3f s = %5.1f GB/s   off: %.3f s = %5.1f GB/s" % (name, " ".join(flags), w, a, nbytes / a / 1e9, b, nbytes / b / 1e9), flush=True)
finally:
    shutil.rmtree(base, ignore_errors=True)
PY
