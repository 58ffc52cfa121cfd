#!/bin/bash
# Round 2, GPU session O: what the exit of a HIP process costs by what it still owns; contexts opened one at a time against all at once.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY' > gpurun_out/o_exit_and_open_order.txt 2>&1
import os, subprocess, sys, time, shutil
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
for args in (["x"], ["x"], ["x", "6"], ["x", "24"], ["x", "0", "1024"], ["x", "0", "4096"], ["x", "0", "0", "16"], ["x", "6", "1024", "16"], ["x", "6", "1024", "16"]):
    t0 = time.monotonic()
    r = subprocess.run(["grab_amd/bin/init_probe"] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    t1 = time.monotonic()
    out = r.stdout.decode().splitlines()
    at = [float(l.split()[1]) for l in out if l.startswith("exit_at")][0]
    left = [l for l in out if l.startswith("left allocated")][0].split()[-2]
    print("init_probe %-18s wall %.3f s   allocating what is left %s s   _exit -> parent sees it %.3f s" % (" ".join(args), t1 - t0, left, t1 - at))
import e2e_sweep
from grab_amd import bin_path, synth
d = "/dev/shm/o_cfg2"
os.makedirs(d)
e2e_sweep.gen_files(d, 512, 64 << 20, 1)
for workers in (8,):
    argv = [bin_path(), "-n", str(workers), "-r", synth.NEEDLE.decode(), d]
    for extra in [{}, {"GSCAN_OPEN_UNORDERED": "1"}, {}, {"GSCAN_OPEN_UNORDERED": "1"}, {"GSCAN_SHARED_COPY": "1"}, {"GSCAN_SHARED_COPY": "2"}]:
        runs = []
        for rep in range(4):
            t0 = time.perf_counter()
            r = subprocess.run(argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, GRAB_TIMING="1", **extra))
            runs.append((time.perf_counter() - t0, r.stderr.decode()))
        runs.sort()
        best = runs[0]
        marks = [ln for ln in best[1].splitlines() if ln.startswith("[grab timing] +")]
        print("workers %d %s: wall min %.3f median %.3f s = %.2f GB/s | %s" % (workers, extra, best[0], runs[2][0], 512 * (64 << 20) / best[0] / 1e9, " | ".join(m[14:] for m in marks)))
shutil.rmtree(d)
PY
cat gpurun_out/o_exit_and_open_order.txt
