#!/bin/bash
# K2 two-class form with the branch-free run program (NR template) + v_lshl_or chain: GPU suite, sweep, bench cfg3.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SW=$R/grab_amd/bin/gscan_sweep
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/x_pytest.txt
{
for pat in '[A-Za-z_][A-Za-z0-9_]{15,}' '[0-9]{16}' '[a-z]{5}' '[0-9]{3}-[0-9]{4}' '[a-z][0-9][a-z][0-9][a-z]'; do
  echo "== pattern $pat"
  timeout 300 $SW --gib 8 --iters 6 --variants 6,4,5 --bpc 0 --pattern "$pat" | grep -v "^overflow" | tail -3
done
} 2>&1 | tee gpurun_out/x_k2_sweep.txt
echo "== bench cfg3"
timeout 900 python bench.py --config cfg3 --steps 5 --warmup 1 2>&1 | tail -1 | tee gpurun_out/x_bench_cfg3.json
