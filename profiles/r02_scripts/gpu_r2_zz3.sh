#!/bin/bash
# Round 2: the scan in a child that hands its status back and leaves the teardown behind (GRAB_DETACH, default on) against one
# process (GRAB_DETACH=0): cfg2 at 32 GiB, wall clock as the caller sees it; golden CLI cases through both.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY' > gpurun_out/zz3_detach.txt 2>&1
import os, subprocess, sys, time, shutil
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
import e2e_sweep
from grab_amd import bin_path, synth
d = "/dev/shm/zz3"
os.makedirs(d)
e2e_sweep.gen_files(d, 512, 64 << 20, 1)
argv = [bin_path(), "-n", "8", "-r", synth.NEEDLE.decode(), d]
for rnd in range(2):
    for det in ("1", "0"):
        ts = []
        for rep in range(4):
            with open("/dev/shm/zz3_out.txt", "wb") as o:
                t0 = time.monotonic()
                r = subprocess.run(argv, stdout=o, stderr=subprocess.PIPE, env=dict(os.environ, GRAB_DETACH=det))
                ts.append(time.monotonic() - t0)
            n = open("/dev/shm/zz3_out.txt", "rb").read().count(b"\n")
        ts.sort()
        print("GRAB_DETACH=%s: wall min %.3f median %.3f s = %.2f GB/s, rc %d, %d lines" % (det, ts[0], ts[1], 512 * (64 << 20) / ts[0] / 1e9, r.returncode, n), flush=True)
r = subprocess.run(argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
print("piped: rc", r.returncode, "lines", r.stdout.count(b"\n"))
r = subprocess.run([bin_path(), "a(", d + "/f000000.txt"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
print("bad regex: rc", r.returncode, r.stderr[:60])
shutil.rmtree(d)
PY
cat gpurun_out/zz3_detach.txt
timeout 200 python -m pytest tests/test_gpu_filegrep.py -m gpu -q -k "t1_ or tree_ or q5 or bad_regex or missing or interface or flag" 2>&1 | tail -2
