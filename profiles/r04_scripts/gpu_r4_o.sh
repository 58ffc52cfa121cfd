#!/bin/bash
# Round 4, session O: K2 lane form with a scheduling barrier behind a step's look-ups (they stay ABOVE the step's arithmetic
# instead of sinking to their first use): lib against lib_ab (the same source without the barrier), interleaved; engine tests first.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/o_pytest_engine.txt
SW=$R/grab_amd/bin/gscan_sweep
{
for L in lib lib_ab lib lib_ab lib lib_ab lib lib_ab; do
  echo "## $L"
  LD_LIBRARY_PATH=$R/grab_amd/$L timeout 300 $SW --gib 16 --iters 8 --variants 38 --bpc 0 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --pattern '[0-9]{16}' --pattern '[a-z][0-9][A-Z]{3}' --pattern '[a-z]{2,5}' --pattern '[0-9]+\.[0-9]+' --pattern '[a-z][0-9][A-Z]{2}[.,;]' 2>&1 | grep -E "^variant"
done
} | tee gpurun_out/o_lane_sched_barrier_sweep.txt
