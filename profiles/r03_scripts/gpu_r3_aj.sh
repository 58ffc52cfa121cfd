#!/bin/bash
# Round 3, session AJ: record slices for the lane-table form (a sub-tile with <= 64 records writes them into its own slice: no
# returning atomic) against GSCAN_NO_SLICES=1, same library, same box, interleaved.  Engine + filegrep tests first.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SW=$R/grab_amd/bin/gscan_sweep
timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_filegrep.py -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/aj_pytest.txt
{
for M in slices none slices none slices none; do
  echo "## $M"
  if [ $M = none ]; then export GSCAN_NO_SLICES=1; else unset GSCAN_NO_SLICES; fi
  timeout 300 $SW --gib 16 --iters 8 --variants 38 --bpc 0 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --pattern '[0-9]{16}' --pattern '[a-z][0-9][A-Z]{3}' --pattern '[0-9]+\.[0-9]+' --pattern '[a-z]{2,5}' 2>&1 | grep -E "^variant"
done
} | tee gpurun_out/aj_slices_sweep.txt
