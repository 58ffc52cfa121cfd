#!/bin/bash
# Round 3, session L: where gscan_wait's time goes in the dense modes (GSCAN_TIMING per context).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY' 2>&1 | tee gpurun_out/l_wait_split.txt
import os, subprocess, sys, time, shutil
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
import e2e_sweep
from grab_amd import bin_path
d = "/dev/shm/r3l"
os.makedirs(d)
e2e_sweep.gen_files(d, 256, 64 << 20, 1)
ident = "[A-Za-z_][A-Za-z0-9_]{15,}"
for flags, env_extra, label in ((["-O"], {"GRAB_LINE_PASS": "1"}, "device line pass + gather"), (["-O", "-l"], {}, "device match ends"), (["-O"], {}, "host walk")):
    argv = [bin_path(), "-n", "8", "-r"] + flags + [ident, d]
    best = None
    for rep in range(2):
        t0 = time.monotonic()
        r = subprocess.run(argv, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, GRAB_TIMING="1", GSCAN_TIMING="1", GRAB_CLOSE="1", **env_extra))
        dt = time.monotonic() - t0
        if best is None or dt < best[0]: best = (dt, r.stderr.decode())
    print("## 16 GiB -n 8 %s (%s): wall %.3f s" % (" ".join(flags), label, best[0]))
    print("\n".join([l for l in best[1].splitlines() if "device 0:" in l][:1] + [l for l in best[1].splitlines() if "context on device" in l][:3]))
shutil.rmtree(d)
PY
