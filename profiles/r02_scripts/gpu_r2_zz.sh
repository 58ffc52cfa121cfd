#!/bin/bash
# Round 2, last GPU run: the GPU suite (incl. the round-2 golden cases and the calls-and-conditions CLI test), smoke, and cfg3 /
# the inexact patterns end to end after the host walk's prefetch.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee gpurun_out/zz_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/zz_smoke.txt
rm -f gpurun_out/zz_e2e.jsonl
timeout 200 python scripts/e2e_cli.py --files 128 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --flags "-O -l" --workers 8 --tag cfg3 >> gpurun_out/zz_e2e.jsonl 2>> gpurun_out/zz_e2e.err
timeout 200 python scripts/e2e_cli.py --files 128 --pattern '(\w)\1{3,}x|foobardoes(?=not)' --flags "-O -l" --workers 8 --tag vm >> gpurun_out/zz_e2e.jsonl 2>> gpurun_out/zz_e2e.err
python - <<'PY'
import json
for ln in open('gpurun_out/zz_e2e.jsonl'):
    d = json.loads(ln); g = d['grab']['8']
    print(d['tag'], d['pattern'][:30], 'grab', g['s'], 's', g['GBps'], 'GB/s same', g['same_as_reference'], 'lines', g['lines'], '| reference', d['reference'])
PY
