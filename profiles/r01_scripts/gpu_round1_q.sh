#!/bin/bash
# reader-thread count vs end-to-end rate on the 64 GiB cfg2 corpus (one corpus, several GSCAN_READERS settings)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY' 2>&1 | tee gpurun_out/q_readers.txt
import os, subprocess, sys, time, shutil
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
from fullsize_parity import gen_files
from grab_amd import bin_path, synth
base = "/dev/shm/grab_q_%d" % os.getpid()
os.makedirs(base)
try:
    t0 = time.perf_counter(); gen_files(base, 1024, 64 << 20, 64); print("gen %.1f s" % (time.perf_counter() - t0), flush=True)
    nbytes = 1024 * (64 << 20)
    ref = os.path.join(os.getcwd(), "oracle", "_ref", "grab_jit")
    def timed(argv, env=None):
        best = None
        for it in range(3):
            t0 = time.perf_counter(); r = subprocess.run(argv, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env); dt = time.perf_counter() - t0
            if it: best = dt if best is None else min(best, dt)
        return best
    dt = timed([ref, "-n", "64", "-r", "-O", "-l", synth.NEEDLE.decode(), base]); print("reference -n 64: %.3f s = %.1f GB/s" % (dt, nbytes / dt / 1e9), flush=True)
    for readers in (8, 12, 16, 24, 32):
        for workers in (2, 4, 8):
            env = dict(os.environ, GSCAN_READERS=str(readers))
            dt = timed([bin_path(), "-n", str(workers), "-r", "-O", "-l", synth.NEEDLE.decode(), base], env)
            print("readers %2d workers %d: %.3f s = %.1f GB/s" % (readers, workers, dt, nbytes / dt / 1e9), flush=True)
    for pat, name in ((synth.IDENT_RE, "ident"),):
        dt = timed([ref, "-n", "64", "-r", "-O", "-l", pat, base]); print("reference -n 64 %s: %.3f s = %.1f GB/s" % (name, dt, nbytes / dt / 1e9), flush=True)
        for workers in (8, 16, 32):
            dt = timed([bin_path(), "-n", str(workers), "-r", "-O", "-l", pat, base], dict(os.environ, GSCAN_READERS="16"))
            print("%s readers 16 workers %d: %.3f s = %.1f GB/s" % (name, workers, dt, nbytes / dt / 1e9), flush=True)
finally:
    shutil.rmtree(base, ignore_errors=True)
PY
