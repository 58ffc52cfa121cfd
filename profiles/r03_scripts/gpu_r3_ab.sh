#!/bin/bash
# Round 3, session AB: alternatives that share ONE device window ([0-9]+\.[0-9]+ and the other leading-repeat shapes) on K1 / K2
# instead of the bucket filter (GSCAN_SAME_WINDOW_K3=1: the round-2 choice).  GPU suite first, then the sweep, interleaved.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SW=$R/grab_amd/bin/gscan_sweep
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee gpurun_out/ab_pytest.txt
{
for M in new old new old new old; do
  echo "## $M"
  if [ $M = old ]; then export GSCAN_SAME_WINDOW_K3=1; else unset GSCAN_SAME_WINDOW_K3; fi
  timeout 300 $SW --gib 16 --iters 8 --variants 38 --bpc 0 --pattern '[0-9]+\.[0-9]+' --pattern '[a-z]+@[a-z]+' --pattern '\w+\(' --pattern 'x+foobar' --pattern '[0-9]+px' 2>&1 | grep -E "^variant"
done
} | tee gpurun_out/ab_same_window_sweep.txt
