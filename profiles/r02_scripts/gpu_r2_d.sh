#!/bin/bash
# Round 2, GPU session D: the GPU suite with per-wave descriptors (barrier-free epilogue), the pipelined K2 look-ups, the VM;
# readers through a mapping + non-temporal copy vs pread; the kernel sweep again; the bench line.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -m gpu -q --durations=5 2>&1 | tail -40 | tee gpurun_out/d_pytest.txt
echo "== e2e: reader modes =="
timeout 700 python scripts/e2e_sweep.py --gib 64 --small-gib 8 --single-gib 8 --blocks 16 --readers 8,12,16 --streams 1 \
   --extra-env "GSCAN_READ_MODE=0;GSCAN_READ_MODE=0,GSCAN_READERS=12" > gpurun_out/d_e2e.jsonl 2> gpurun_out/d_e2e.err
python - <<'PY'
import json
for l in open('gpurun_out/d_e2e.jsonl'):
    r = json.loads(l)
    t = [x for x in r.get('timing', []) if 'device' in x]
    print({k: r[k] for k in r if k not in ('timing',)})
    if t: print('      ', t[0][15:])
PY
echo "== kernel sweep =="
S=grab_amd/bin/gscan_sweep
{
$S --gib 16 --iters 6 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --variants 6,13,21 --bpc 0
$S --gib 16 --iters 6 --pattern '[0-9]{16}' --variants 6,13,21 --bpc 0
$S --gib 16 --iters 6 --pattern 'foobardoesnotexist|Linus|555-1234' --variants 6,13 --bpc 0
$S --gib 16 --iters 6 --pattern '[0-9]+\.[0-9]+' --variants 6,13 --bpc 0
$S --gib 16 --iters 6 --pattern '[a-z][0-9][A-Z]{3}' --variants 6,13 --bpc 0
$S --gib 16 --iters 6 --pattern 'foobardoesnotexist' --variants 6,2 --bpc 0,8
} > gpurun_out/d_kernel_sweep.txt 2>&1
grep -E "^#|variant" gpurun_out/d_kernel_sweep.txt
echo "== bench =="
( time timeout 900 python bench.py ) > gpurun_out/d_bench.json 2> gpurun_out/d_bench.err
tail -3 gpurun_out/d_bench.err
python - <<'PY'
import json
r = json.loads(open('gpurun_out/d_bench.json').read().strip().splitlines()[-1])
print(r['value'], r['roofline']['frac'], {k: v['frac'] for k, v in r['kernels'].items()}, r['e2e'].get('value'), r['e2e'].get('frac'), r.get('cpu_baseline', {}).get('value'))
PY
