#!/bin/bash
# Round 2, GPU session M: where the fixed part of a cfg2 run goes (GRAB_TIMING marks around context open / close).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY' > gpurun_out/m_fixed_cost.txt 2>&1
import os, subprocess, sys, time
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
import e2e_sweep
from grab_amd import bin_path, synth
d = "/dev/shm/m_cfg2"
os.makedirs(d)
e2e_sweep.gen_files(d, 256, 64 << 20, 1)
argv = [bin_path(), "-n", "8", "-r", synth.NEEDLE.decode(), d]
for env_extra in [{}, {}, {"GSCAN_SLOTS_HINT": "1"}, {"AMD_LOG_LEVEL": "0", "HIP_ENABLE_DEFERRED_LOADING": "1"}]:
    t0 = time.perf_counter()
    r = subprocess.run(argv, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, GRAB_TIMING="1", GSCAN_TIMING="1", **env_extra))
    print("wall %.3f s" % (time.perf_counter() - t0), env_extra)
    print(r.stderr.decode())
for n in (1, 2, 4):
    t0 = time.perf_counter()
    r = subprocess.run([bin_path(), "-n", str(max(n, 2)), "-r", synth.NEEDLE.decode(), d], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, GRAB_TIMING="1", GSCAN_TIMING="1"))
    print("workers %d wall %.3f s" % (max(n, 2), time.perf_counter() - t0))
    print(r.stderr.decode())
import shutil
shutil.rmtree(d)
PY
cat gpurun_out/m_fixed_cost.txt
