#!/bin/bash
# Round 3, session N: the GPU suite with the line pass on by default, then the bench line.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/n_pytest.txt
echo "== smoke =="
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/n_smoke.txt
echo "== bench.py (default) =="
( time timeout 900 python bench.py ) > gpurun_out/n_bench.json 2> gpurun_out/n_bench.err
tail -4 gpurun_out/n_bench.err
python - <<'PY'
import json
r = json.loads(open('gpurun_out/n_bench.json').read().strip().splitlines()[-1])
print("value", r['value'], "roofline", {k: r['roofline'][k] for k in ('frac', 'kernel_ms', 'traffic')})
print({k: (v['frac'], v['kernel_ms'], v['traffic']) for k, v in r['kernels'].items()})
for k in ("e2e", "cpu_baseline", "e2e_cfg3", "e2e_cfg5"):
    v = r.get(k) or {}
    print(k, {x: v.get(x) for x in ("value", "wall_s", "detached_GBps", "scan_phase_GBps", "frac", "lines", "lines_ok", "vs_cpu_baseline", "cores", "GBps_by_threads", "parity_subset", "same_as_reference", "error")}, (v.get("cpu_baseline") or {}).get("value"), (v.get("cpu_baseline") or {}).get("GBps_by_threads"))
PY
