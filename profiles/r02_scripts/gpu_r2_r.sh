#!/bin/bash
# Round 2, GPU session R: whose the 2 GB of anonymous memory of a run are (resident set by phase; the bare runtime's for comparison).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
{
grab_amd/bin/init_probe x 1 1024 2
python - <<'PY'
import os, subprocess, sys, shutil
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
import e2e_sweep
from grab_amd import bin_path, synth
d = "/dev/shm/r_cfg2"
os.makedirs(d)
e2e_sweep.gen_files(d, 64, 64 << 20, 1)
for w in ("2", "8"):
    r = subprocess.run([bin_path(), "-n", w, "-r", synth.NEEDLE.decode(), d], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, GRAB_TIMING="1"))
    print("## -n", w)
    print("\n".join(l for l in r.stderr.decode().splitlines() if l.startswith("[grab timing] +") or "memory" in l))
r = subprocess.run([bin_path(), synth.NEEDLE.decode(), os.path.join(d, "f000000.txt")], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, GRAB_TIMING="1"))
print("## one file, serial")
print("\n".join(l for l in r.stderr.decode().splitlines() if l.startswith("[grab timing] +") or "memory" in l))
shutil.rmtree(d)
PY
} > gpurun_out/r_resident_memory.txt 2>&1
cat gpurun_out/r_resident_memory.txt
