#!/bin/bash
# Round 3, session I: how the printed lines of the line pass come about end to end (diagnostic tallies).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY' 2>&1 | tee gpurun_out/i_lines_diag.txt
import os, subprocess, sys, time, shutil
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
import e2e_sweep
from grab_amd import bin_path
d = "/dev/shm/r3i_cfg3"
os.makedirs(d)
e2e_sweep.gen_files(d, 256, 64 << 20, 1)
ident = "[A-Za-z_][A-Za-z0-9_]{15,}"
for n in (8, 16):
    for env_extra, label in (({}, "host walk"), ({"GRAB_LINE_PASS": "1"}, "device line pass + gather")):
        argv = [bin_path()] + (["-n", str(n)] if n > 1 else []) + ["-r", "-O", ident, d]
        best = None
        for rep in range(2):
            t0 = time.monotonic()
            r = subprocess.run(argv, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, GRAB_TIMING="1", **env_extra))
            dt = time.monotonic() - t0
            if best is None or dt < best[0]: best = (dt, r.stderr.decode())
        lines = [l for l in best[1].splitlines() if "device 0:" in l][:1] + [l for l in best[1].splitlines() if "printed so far" in l][-1:]
        print("## 16 GiB -n %d -O (%s): wall %.3f s" % (n, label, best[0])); print("\n".join(lines))
shutil.rmtree(d)
PY
