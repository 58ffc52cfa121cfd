#!/bin/bash
# Round 2: cfg3 (16 GiB, -n 8 / 16, output discarded) after the host walk's prefetch -- the same measurement as session U.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY' > gpurun_out/zz2_cfg3.txt 2>&1
import os, subprocess, sys, time, shutil
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
import e2e_sweep
from grab_amd import bin_path
d = "/dev/shm/zz2_cfg3"
os.makedirs(d)
e2e_sweep.gen_files(d, 256, 64 << 20, 1)
ident = "[A-Za-z_][A-Za-z0-9_]{15,}"
for n in (8, 16):
    best = None
    for rep in range(3):
        t0 = time.monotonic()
        r = subprocess.run([bin_path(), "-n", str(n), "-r", "-O", "-l", ident, d], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, GRAB_TIMING="1"))
        dt = time.monotonic() - t0
        if best is None or dt < best[0]: best = (dt, r.stderr.decode())
    lines = [l for l in best[1].splitlines() if "device 0:" in l][:2] + [l for l in best[1].splitlines() if "workers joined" in l or "runtime up" in l]
    print("## cfg3 16 GiB -n %d: wall %.3f s = %.2f GB/s" % (n, best[0], (16 << 30) / best[0] / 1e9)); print("\n".join(lines))
shutil.rmtree(d)
PY
cat gpurun_out/zz2_cfg3.txt
