#!/bin/bash
# Round 2, GPU session G: the hybrid epilogue (K2 per tile, K1/K3 per wave), the VM with register slots; quick numbers only.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
S=grab_amd/bin/gscan_sweep
{
$S --gib 16 --iters 6 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --variants 6,13 --bpc 0
$S --gib 16 --iters 6 --pattern '[0-9]{16}' --variants 6,13 --bpc 0
$S --gib 16 --iters 6 --pattern 'foobardoesnotexist|Linus|555-1234' --variants 6 --bpc 0
$S --gib 16 --iters 6 --pattern '[0-9]+\.[0-9]+' --variants 6 --bpc 0
$S --gib 16 --iters 6 --pattern '[a-z][0-9][A-Z]{3}' --variants 6,13 --bpc 0
$S --gib 16 --iters 6 --pattern 'foobardoesnotexist' --variants 6 --bpc 0
$S --gib 8 --iters 3 --pattern '(\w)\1{3,}x|foobardoes(?=not)' --variants 6 --bpc 0
$S --gib 8 --iters 3 --pattern '[a-z]+\([a-z0-9, ]*\);' --variants 6 --bpc 0
$S --gib 8 --iters 3 --pattern '(?:foo|bar|ab)+baz' --variants 6 --bpc 0
} > gpurun_out/g_kernel_sweep.txt 2>&1
grep -E "^#|variant|overflow" gpurun_out/g_kernel_sweep.txt
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_filegrep.py -m gpu -q -k "kernels_against or parity_patterns or ragged or dense or inx or look or bref or random_patterns or syn8_inx or tree_differential" 2>&1 | tail -5
for P in '(\w)\1{3,}x|foobardoes(?=not)'; do
  timeout 300 python scripts/e2e_cli.py --files 128 --pattern "$P" --flags "-O -l" --workers 8 --tag vm >> gpurun_out/g_vm_e2e.jsonl 2>> gpurun_out/g_vm_e2e.err
done
python - <<'PY'
import json
for l in open('gpurun_out/g_vm_e2e.jsonl'):
    r = json.loads(l)
    print(r['tag'], r['pattern'], r['bytes'] >> 30, 'GiB ref', r['reference'], {w: (v['s'], v['GBps'], v['same_as_reference'], v['lines']) for w, v in r['grab'].items()})
PY
( time timeout 900 python bench.py --no-e2e --no-cpu-baseline ) > gpurun_out/g_bench.json 2> gpurun_out/g_bench.err
python - <<'PY'
import json
r = json.loads(open('gpurun_out/g_bench.json').read().strip().splitlines()[-1])
print(r['value'], r['roofline']['frac'], {k: v['frac'] for k, v in r['kernels'].items()})
PY
