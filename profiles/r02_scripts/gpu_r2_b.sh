#!/bin/bash
# Round 2, GPU session B: the whole GPU suite (no -x), the DMA probe with and without the SDMA engines, the e2e run under the
# candidate fixes, the kernel sweep of the new workgroup shapes (variant 13 / 21) and of K3's two-operation filter, the bench line.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -m gpu -q --durations=12 2>&1 | tail -40 | tee gpurun_out/b_pytest.txt
echo "== dma probe =="
( grab_amd/bin/dma_probe; HSA_ENABLE_SDMA=0 grab_amd/bin/dma_probe ) > gpurun_out/b_dma_probe.txt 2>&1
cat gpurun_out/b_dma_probe.txt
echo "== e2e under candidate fixes =="
timeout 600 python scripts/e2e_sweep.py --gib 64 --small-gib 0 --single-gib 0 --blocks 16 --readers 8,12 --streams 1 \
   --extra-env "HSA_ENABLE_SDMA=0;GSCAN_PIN_FLAGS=1;GSCAN_PIN_FLAGS=2;HSA_ENABLE_SDMA=0,GSCAN_PIN_FLAGS=1;HSA_ENABLE_SDMA=0,GSCAN_READERS=16" > gpurun_out/b_e2e.jsonl 2> gpurun_out/b_e2e.err
python - <<'PY'
import json
for l in open('gpurun_out/b_e2e.jsonl'):
    r = json.loads(l)
    print({k: r[k] for k in r if k not in ('timing',)})
PY
echo "== kernel sweep =="
S=grab_amd/bin/gscan_sweep
{
$S --gib 16 --iters 6 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --variants 6,5,13,21 --bpc 0
$S --gib 16 --iters 6 --pattern '[0-9]{16}' --variants 6,13,21 --bpc 0
$S --gib 16 --iters 6 --pattern 'foobardoesnotexist|Linus|555-1234' --variants 6,5,13 --bpc 0
GSCAN_K3_DEPTH=4 $S --gib 16 --iters 6 --pattern 'foobardoesnotexist|Linus|555-1234' --variants 6,13 --bpc 0
$S --gib 16 --iters 6 --pattern '[0-9]+\.[0-9]+' --variants 6,13 --bpc 0
$S --gib 16 --iters 6 --pattern '(?i)foobar|k7Q,;q|[0-9]{12}x?' --variants 6,13 --bpc 0
$S --gib 16 --iters 6 --pattern '[a-z][0-9][A-Z]{3}' --variants 6,5,13 --bpc 0,4
$S --gib 16 --iters 6 --pattern '[ab][cd][ef][gh]{20}' --variants 6,13 --bpc 0,4
$S --gib 16 --iters 6 --pattern 'foobardoesnotexist' --variants 6 --bpc 0
} > gpurun_out/b_kernel_sweep.txt 2>&1
grep -E "^#|variant" gpurun_out/b_kernel_sweep.txt
echo "== bench =="
( time timeout 900 python bench.py ) > gpurun_out/b_bench.json 2> gpurun_out/b_bench.err
tail -3 gpurun_out/b_bench.err
python - <<'PY'
import json
r = json.loads(open('gpurun_out/b_bench.json').read().strip().splitlines()[-1])
print(r['value'], r['roofline']['frac'], {k: v['frac'] for k, v in r['kernels'].items()}, r['e2e'].get('value'), r['e2e'].get('frac'), r.get('cpu_baseline', {}).get('value'))
PY
