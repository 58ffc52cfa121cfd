#!/bin/bash
# Round 3, session AK: record slices, the slice records counted once per wave and launch (no memory operation per flush),
# against GSCAN_NO_SLICES=1: S S N N S S N N on one box.  Engine tests first.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SW=$R/grab_amd/bin/gscan_sweep
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/ak_pytest.txt
{
for M in slices slices none none slices slices none none; do
  echo "## $M"
  if [ $M = none ]; then export GSCAN_NO_SLICES=1; else unset GSCAN_NO_SLICES; fi
  timeout 300 $SW --gib 16 --iters 8 --variants 38 --bpc 0 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --pattern '[a-z][0-9][A-Z]{3}' --pattern '[0-9]+\.[0-9]+' 2>&1 | grep -E "^variant"
done
} | tee gpurun_out/ak_slices_sweep.txt
