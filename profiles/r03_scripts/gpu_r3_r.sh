#!/bin/bash
# Round 3, session R: one reservation per TWO sub-tiles (GSCAN_LANE_BATCH=2, the default) against one each (1), same box.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SW=$R/grab_amd/bin/gscan_sweep
echo "== pytest: engine parity =="
timeout 1200 python -m pytest tests/test_gpu_engine.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r_pytest.txt
for round in 1 2 3; do for G in 2 3; do
  echo "## round $round GSCAN_LANE_BATCH=$G"
  GSCAN_LANE_BATCH=$G timeout 300 $SW --gib 16 --iters 8 --variants 38 --bpc 0 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --pattern '[0-9]{16}' --pattern '[a-z][0-9][A-Z]{3}' --pattern '[a-z]{2,5}' 2>&1 | grep -E "^variant"
done; done | tee gpurun_out/r_batch_sweep.txt
