#!/bin/bash
# Round 3, session D: the whole GPU suite on the new defaults, the kernel sweep of the table kernels (K2 lane forms with sorted
# run programs, K3 with the deferred confirm), the bench line.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest -m gpu =="
timeout 2400 python -m pytest tests -m gpu -q -x --durations=6 2>&1 | tail -14 | tee gpurun_out/d_pytest.txt
SW=$R/grab_amd/bin/gscan_sweep
echo "== kernel sweep, 16 GiB =="
timeout 900 $SW --gib 16 --iters 6 --variants 6,38 --bpc 0 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --pattern '[0-9]{16}' --pattern '[a-z][0-9][A-Z]{3}' --pattern '[a-z]{3}[0-9][A-Z]{2}[a-z_]{6}' --pattern '[a-z][0-9][a-z]{9}' 2>&1 | grep -E "^variant|^#" | tee gpurun_out/d_sweep_k2.txt
timeout 900 $SW --gib 16 --iters 6 --variants 38 --bpc 0 --pattern '[0-9]+\.[0-9]+' --pattern 'foobardoesnotexist|Linus|555-1234' --pattern '(?i)foobar|k7Q,;q|[0-9]{12}x?' --pattern '[a-z][0-9][A-Z][.,][;:]' --pattern 'foo.*bar' --pattern 'foobardoesnotexist' 2>&1 | grep -E "^variant|^#" | tee gpurun_out/d_sweep_k3.txt
echo "== bench.py (default) =="
( time timeout 900 python bench.py ) > gpurun_out/d_bench.json 2> gpurun_out/d_bench.err
tail -4 gpurun_out/d_bench.err
python - <<'PY'
import json
r = json.loads(open('gpurun_out/d_bench.json').read().strip().splitlines()[-1])
print("value", r['value'], "roofline", {k: r['roofline'][k] for k in ('frac', 'kernel_ms', 'traffic')})
print({k: (v['frac'], v['kernel_ms'], v['traffic']) for k, v in r['kernels'].items()})
for k in ("e2e", "cpu_baseline", "e2e_cfg3", "e2e_cfg5"):
    v = r.get(k) or {}
    print(k, {x: v.get(x) for x in ("value", "wall_s", "detached_GBps", "scan_phase_GBps", "frac", "lines", "lines_ok", "vs_cpu_baseline", "cores", "GBps_by_threads", "parity_subset", "same_as_reference", "error")}, (v.get("cpu_baseline") or {}).get("value"), (v.get("cpu_baseline") or {}).get("GBps_by_threads"))
PY
