#!/bin/bash
# Round 2, GPU session T: cfg5 (one big file, -O -l, 1 GiB windows, one context) -- where its time goes.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY' > gpurun_out/t_cfg5_timing.txt 2>&1
import os, subprocess, sys, time, shutil
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
import e2e_sweep
from grab_amd import bin_path, synth
d = "/dev/shm/t_cfg5"
os.makedirs(d)
e2e_sweep.gen_files(d, 1, 16 << 30, 1, needles_every=32768 + 77)
f = os.path.join(d, "f000000.txt")
for flags, extra in ((["-O", "-l"], {}), (["-O", "-l"], {"GRAB_CLOSE": "1", "GSCAN_TIMING": "1"}), (["-O", "-l", "-L", "-L"], {}), (["-O", "-l", "-L", "-L"], {"GRAB_CLOSE": "1", "GSCAN_TIMING": "1"})):
    for rep in range(2):
        t0 = time.monotonic()
        r = subprocess.run([bin_path()] + flags + [synth.NEEDLE.decode(), f], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, GRAB_TIMING="1", **extra))
        t1 = time.monotonic()
    print("##", " ".join(flags), extra, "wall %.3f s = %.2f GB/s" % (t1 - t0, (16 << 30) / (t1 - t0) / 1e9))
    print("\n".join(l for l in r.stderr.decode().splitlines() if "timing]" in l and "gscan_open" not in l))
shutil.rmtree(d)
PY
cat gpurun_out/t_cfg5_timing.txt
