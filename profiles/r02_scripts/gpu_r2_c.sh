#!/bin/bash
# Round 2, GPU session C: the GPU suite with the device VM in the K3 cold path; what holds the ingest at ~40 GB/s (shared copy
# streams? one pinned slab?); inexact patterns end to end with and without the VM; the bench line.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -60 | tee gpurun_out/c_pytest.txt
echo "== e2e: ingest experiments =="
timeout 700 python scripts/e2e_sweep.py --gib 64 --small-gib 0 --single-gib 0 --blocks 16 --readers 8 --streams 1 \
   --extra-env "GSCAN_READERS=16;GSCAN_SHARED_COPY=2;GSCAN_SHARED_COPY=2,GSCAN_READERS=16;GSCAN_SLAB=1;GSCAN_SLAB=1,GSCAN_READERS=16;GSCAN_SLAB=1,GSCAN_SHARED_COPY=2,GSCAN_READERS=16;GSCAN_SLAB=1,GSCAN_SHARED_COPY=1,GSCAN_READERS=12;GSCAN_SLAB=1,GSCAN_READERS=24,GSCAN_BLOCK_MIB=8" > gpurun_out/c_e2e.jsonl 2> gpurun_out/c_e2e.err
python - <<'PY'
import json
for l in open('gpurun_out/c_e2e.jsonl'):
    r = json.loads(l)
    t = [x for x in r.get('timing', []) if 'device' in x]
    print({k: r[k] for k in r if k not in ('timing',)})
    if t: print('      ', t[0][15:])
PY
echo "== inexact patterns end to end, VM on / off =="
for P in '(\w)\1{3,}x|foobardoes(?=not)' '[a-z]+\([a-z0-9, ]*\);' '[a-z]+_[0-9]+\.[a-z]+' '(?:foo|bar|ab)+baz'; do
  timeout 300 python scripts/e2e_cli.py --files 128 --pattern "$P" --flags "-O -l" --workers 8 --tag vm >> gpurun_out/c_vm_e2e.jsonl 2>> gpurun_out/c_vm_e2e.err
  GSCAN_NO_VM=1 timeout 300 python scripts/e2e_cli.py --files 32 --pattern "$P" --flags "-O -l" --workers 8 --tag novm >> gpurun_out/c_vm_e2e.jsonl 2>> gpurun_out/c_vm_e2e.err
done
python - <<'PY'
import json
for l in open('gpurun_out/c_vm_e2e.jsonl'):
    r = json.loads(l)
    print(r['tag'], r['pattern'], r['bytes'] >> 30, 'GiB ref', r['reference'], {w: (v['s'], v['GBps'], v['same_as_reference'], v['lines']) for w, v in r['grab'].items()})
PY
echo "== VM kernel rate (HBM-resident) =="
S=grab_amd/bin/gscan_sweep
{
$S --gib 8 --iters 3 --pattern '(\w)\1{3,}x|foobardoes(?=not)' --variants 6 --bpc 0
$S --gib 8 --iters 3 --pattern '[a-z]+\([a-z0-9, ]*\);' --variants 6 --bpc 0
$S --gib 8 --iters 3 --pattern '(?:foo|bar|ab)+baz' --variants 6 --bpc 0
} > gpurun_out/c_vm_sweep.txt 2>&1
grep -E "^#|variant|overflow" gpurun_out/c_vm_sweep.txt
echo "== bench =="
( time timeout 900 python bench.py ) > gpurun_out/c_bench.json 2> gpurun_out/c_bench.err
tail -3 gpurun_out/c_bench.err
python - <<'PY'
import json
r = json.loads(open('gpurun_out/c_bench.json').read().strip().splitlines()[-1])
print(r['value'], r['roofline']['frac'], {k: v['frac'] for k, v in r['kernels'].items()}, r['e2e'].get('value'), r['e2e'].get('frac'), r.get('cpu_baseline', {}).get('value'))
PY
