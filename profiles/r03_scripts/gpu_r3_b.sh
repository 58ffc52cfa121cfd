#!/bin/bash
# Round 3, session B: the pipelined lane kernel (next tile's loads before the epilogue, scalar descriptors, NR 3/4 programs),
# pinned dense readback, the reworked bench line.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest: engine-level parity =="
timeout 1200 python -m pytest tests/test_gpu_engine.py -m gpu -x -q --durations=5 2>&1 | tail -15 | tee gpurun_out/b_pytest_engine.txt
echo "== pytest: CLI subset =="
timeout 900 python -m pytest tests/test_gpu_filegrep.py tests/test_gpu_geometry.py -m gpu -x -q -k "without_the_text or multichunk or tree_differential or geometry or window" 2>&1 | tail -6 | tee gpurun_out/b_pytest_cli.txt
SW=$R/grab_amd/bin/gscan_sweep
echo "== kernel sweep, 16 GiB: variant 6 vs 38 =="
timeout 600 $SW --gib 16 --iters 6 --variants 6,38 --bpc 0 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --pattern '[0-9]{16}' --pattern '[a-z][0-9][A-Z]{3}' --pattern '[a-z]{3}[0-9][A-Z]{2}[a-z_]{6}' --pattern '[a-z][0-9][a-z]{9}' --pattern '[0-9a-f]{8}[g-z]' 2>&1 | grep -E "^variant|^#" | tee gpurun_out/b_sweep.txt
echo "== cfg3 end to end, 16 GiB =="
python - <<'PY' > gpurun_out/b_cfg3_e2e.txt 2>&1
import os, subprocess, sys, time, shutil
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
import e2e_sweep
from grab_amd import bin_path
d = "/dev/shm/r3b_cfg3"
os.makedirs(d)
e2e_sweep.gen_files(d, 256, 64 << 20, 1)
ident = "[A-Za-z_][A-Za-z0-9_]{15,}"
for env_extra, label in (({}, "device ends"), ({"GRAB_NO_ENDS": "1"}, "host walk")):
    for n in (8, 16):
        best = None
        for rep in range(3):
            t0 = time.monotonic()
            r = subprocess.run([bin_path(), "-n", str(n), "-r", "-O", "-l", ident, d], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, GRAB_TIMING="1", **env_extra))
            dt = time.monotonic() - t0
            if best is None or dt < best[0]: best = (dt, r.stderr.decode())
        lines = [l for l in best[1].splitlines() if "device 0:" in l][:1] + [l for l in best[1].splitlines() if "workers joined" in l or "runtime up" in l]
        print("## cfg3 16 GiB -n %d (%s): wall %.3f s = %.2f GB/s" % (n, label, best[0], (16 << 30) / best[0] / 1e9)); print("\n".join(lines))
shutil.rmtree(d)
PY
cat gpurun_out/b_cfg3_e2e.txt
echo "== bench.py (default) =="
( time timeout 900 python bench.py ) > gpurun_out/b_bench.json 2> gpurun_out/b_bench.err
tail -4 gpurun_out/b_bench.err
python - <<'PY'
import json
r = json.loads(open('gpurun_out/b_bench.json').read().strip().splitlines()[-1])
print("value", r['value'], "roofline", r['roofline'])
print({k: (v['frac'], v['kernel_ms'], v['traffic']) for k, v in r['kernels'].items()})
for k in ("e2e", "cpu_baseline", "e2e_cfg3", "e2e_cfg5"):
    print(k, json.dumps(r.get(k))[:900])
PY
