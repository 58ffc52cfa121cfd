#!/bin/bash
# Round 2, GPU session Q: cfg3 end to end -- where the second after "scan done" goes (memory the process holds when it leaves).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY' > gpurun_out/q_cfg3_exit.txt 2>&1
import os, subprocess, sys, time, shutil
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
import e2e_sweep
from grab_amd import bin_path, synth
d = "/dev/shm/q_cfg3"
os.makedirs(d)
e2e_sweep.gen_files(d, 256, 64 << 20, 1)
ident = "[A-Za-z_][A-Za-z0-9_]{15,}"
for tag, argv, out in [("ident -> pipe", [bin_path(), "-n", "8", "-r", "-O", "-l", ident, d], subprocess.PIPE),
                       ("ident -> /dev/null", [bin_path(), "-n", "8", "-r", "-O", "-l", ident, d], subprocess.DEVNULL),
                       ("ident -> file in /dev/shm", [bin_path(), "-n", "8", "-r", "-O", "-l", ident, d], "file"),
                       ("needle -> pipe", [bin_path(), "-n", "8", "-r", synth.NEEDLE.decode(), d], subprocess.PIPE)]:
    for extra in ({}, {"GRAB_CLOSE": "1"}):
        for rep in range(2):
            o = open("/dev/shm/q_out.txt", "wb") if out == "file" else out
            t0 = time.monotonic()
            r = subprocess.run(argv, stdout=o, stderr=subprocess.PIPE, env=dict(os.environ, GRAB_TIMING="1", **extra))
            t1 = time.monotonic()
            if out == "file":
                o.close()
        marks = [ln[14:] for ln in r.stderr.decode().splitlines() if ln.startswith("[grab timing] +") or ln.startswith("[grab timing] memory")]
        print("%s %s: wall %.3f s | %s" % (tag, extra, t1 - t0, " | ".join(marks)), flush=True)
shutil.rmtree(d)
PY
cat gpurun_out/q_cfg3_exit.txt
