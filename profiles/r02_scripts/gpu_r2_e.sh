#!/bin/bash
# Round 2, GPU session E: 64 record shards under the per-wave epilogue; the VM with first-byte prediction; readers through an
# L2-sized bounce buffer + non-temporal copy; the bench line.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -m gpu -q --durations=5 2>&1 | tail -30 | tee gpurun_out/e_pytest.txt
echo "== kernel sweep =="
S=grab_amd/bin/gscan_sweep
{
$S --gib 16 --iters 6 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --variants 6,13,21 --bpc 0
$S --gib 16 --iters 6 --pattern '[0-9]{16}' --variants 6,13,21 --bpc 0
$S --gib 16 --iters 6 --pattern 'foobardoesnotexist|Linus|555-1234' --variants 6,13 --bpc 0
$S --gib 16 --iters 6 --pattern '[0-9]+\.[0-9]+' --variants 6,13 --bpc 0
$S --gib 16 --iters 6 --pattern '[a-z][0-9][A-Z]{3}' --variants 6,13 --bpc 0
$S --gib 16 --iters 6 --pattern 'foobardoesnotexist' --variants 6 --bpc 0
$S --gib 8 --iters 3 --pattern '(\w)\1{3,}x|foobardoes(?=not)' --variants 6 --bpc 0
$S --gib 8 --iters 3 --pattern '[a-z]+\([a-z0-9, ]*\);' --variants 6 --bpc 0
} > gpurun_out/e_kernel_sweep.txt 2>&1
grep -E "^#|variant|overflow" gpurun_out/e_kernel_sweep.txt
echo "== e2e: bounce + non-temporal copy =="
timeout 500 python scripts/e2e_sweep.py --gib 64 --small-gib 0 --single-gib 0 --blocks 16 --readers 8 --streams 1 \
   --extra-env "GSCAN_READ_MODE=2;GSCAN_READ_MODE=2,GSCAN_READERS=12;GSCAN_READ_MODE=2,GSCAN_READERS=16" > gpurun_out/e_e2e.jsonl 2> gpurun_out/e_e2e.err
python - <<'PY'
import json
for l in open('gpurun_out/e_e2e.jsonl'):
    r = json.loads(l)
    t = [x for x in r.get('timing', []) if 'device' in x]
    print({k: r[k] for k in r if k not in ('timing',)})
    if t: print('      ', t[0][15:])
PY
echo "== inexact patterns end to end =="
for P in '(\w)\1{3,}x|foobardoes(?=not)' '[a-z]+\([a-z0-9, ]*\);'; do
  timeout 300 python scripts/e2e_cli.py --files 128 --pattern "$P" --flags "-O -l" --workers 8 --tag vm >> gpurun_out/e_vm_e2e.jsonl 2>> gpurun_out/e_vm_e2e.err
done
python - <<'PY'
import json
for l in open('gpurun_out/e_vm_e2e.jsonl'):
    r = json.loads(l)
    print(r['tag'], r['pattern'], r['bytes'] >> 30, 'GiB ref', r['reference'], {w: (v['s'], v['GBps'], v['same_as_reference'], v['lines']) for w, v in r['grab'].items()})
PY
echo "== bench =="
( time timeout 900 python bench.py ) > gpurun_out/e_bench.json 2> gpurun_out/e_bench.err
tail -3 gpurun_out/e_bench.err
python - <<'PY'
import json
r = json.loads(open('gpurun_out/e_bench.json').read().strip().splitlines()[-1])
print(r['value'], r['roofline']['frac'], {k: v['frac'] for k, v in r['kernels'].items()}, r['e2e'].get('value'), r['e2e'].get('frac'), r.get('cpu_baseline', {}).get('value'))
PY
