#!/bin/bash
# K3 with the per-lane table (one v_perm per look-up address, 512-thread workgroups): GPU suite, sweep, bench alt.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SW=$R/grab_amd/bin/gscan_sweep
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/y_pytest.txt
{
for pat in 'foobardoes(?:not)?exist|[0-9A-F]{7}[a-z]?|(?i:xyzzy)' 'foo|bar' '(?i)linus' '[a-z][0-9][A-Z][.,][;:]' '\bfoo\b' '[a-z]+_[0-9]+\.[a-z]+' '(?:ab|cd|ef|gh|ij|kl|mn|op|qr|st)x'; do
  echo "== pattern $pat"
  timeout 300 $SW --gib 8 --iters 6 --variants 6,4,5 --bpc 0 --pattern "$pat" | grep -v "^overflow" | tail -3
done
} 2>&1 | tee gpurun_out/y_k3_sweep.txt
echo "== bench alt"
timeout 900 python bench.py --config alt --steps 5 --warmup 1 2>&1 | tail -1 | tee gpurun_out/y_bench_alt.json
