#!/bin/bash
# Round 2, GPU session S: GPU suite and bench line with the device-wide streams as the default.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee gpurun_out/s_pytest.txt
( time timeout 900 python bench.py ) > gpurun_out/s_bench.json 2> gpurun_out/s_bench.err
tail -4 gpurun_out/s_bench.err
python - <<'PY'
import json
r = json.loads(open('gpurun_out/s_bench.json').read().strip().splitlines()[-1])
print(r['value'], r['roofline'], {k: (v['frac'], v['kernel_ms']) for k, v in r['kernels'].items()})
print(r['e2e'])
print(r.get('cpu_baseline'))
PY
