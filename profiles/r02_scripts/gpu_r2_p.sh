#!/bin/bash
# Round 2, GPU session P: device-wide scan and copy streams against per-context ones (cfg2 at 32 GiB, 8 workers; cfg4-like small files).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY' > gpurun_out/p_shared_streams.txt 2>&1
import os, subprocess, sys, time, shutil
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
import e2e_sweep
from grab_amd import bin_path, synth
EXTRAS = [{}, {"GSCAN_SHARED_COPY": "1"}, {"GSCAN_SHARED_COPY": "1", "GSCAN_SHARED_COMPUTE": "1"}, {"GSCAN_SHARED_COPY": "1", "GSCAN_SHARED_COMPUTE": "2"},
          {"GSCAN_SHARED_COPY": "2", "GSCAN_SHARED_COMPUTE": "2"}, {}, {"GSCAN_SHARED_COPY": "1", "GSCAN_SHARED_COMPUTE": "1"}]
def run(tag, d, nbytes, argv, ref_lines=None):
    for extra in EXTRAS:
        runs = []
        for rep in range(4):
            t0 = time.perf_counter()
            r = subprocess.run(argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, GRAB_TIMING="1", **extra))
            runs.append((time.perf_counter() - t0, r.stderr.decode(), r.stdout.count(b"\n"), r.returncode))
        runs.sort()
        best = runs[0]
        marks = [ln for ln in best[1].splitlines() if ln.startswith("[grab timing] +")]
        print("%s %s: wall min %.3f median %.3f s = %.2f GB/s rc %d lines %d | %s" % (tag, extra, best[0], runs[2][0], nbytes / best[0] / 1e9, best[3], best[2], " | ".join(m[14:] for m in marks)), flush=True)
d = "/dev/shm/p_cfg2"
os.makedirs(d)
e2e_sweep.gen_files(d, 512, 64 << 20, 1)
run("cfg2 32 GiB -n 8", d, 512 * (64 << 20), [bin_path(), "-n", "8", "-r", synth.NEEDLE.decode(), d])
run("cfg3 32 GiB -n 8 -O -l ident", d, 512 * (64 << 20), [bin_path(), "-n", "8", "-r", "-O", "-l", "[A-Za-z_][A-Za-z0-9_]{15,}", d])
shutil.rmtree(d)
d = "/dev/shm/p_cfg4"
os.makedirs(d)
e2e_sweep.gen_files(d, 16 * 2048, 512 << 10, 32)
run("cfg4 16 GiB -n 8", d, 16 * 2048 * (512 << 10), [bin_path(), "-n", "8", "-r", synth.NEEDLE.decode(), d])
shutil.rmtree(d)
PY
cat gpurun_out/p_shared_streams.txt
