#!/bin/bash
# Round 3, session C: where the lane kernel's next-tile loads go (GSCAN_LANE_PF 0 none / 1 before the epilogue / 2 behind the
# reserving atomic) x KiB per wave (12 / 16), same box, the two benchmark programs.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SW=$R/grab_amd/bin/gscan_sweep
echo "== pytest: engine-level parity ==" 
timeout 1200 python -m pytest tests/test_gpu_engine.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/c_pytest_engine.txt
for round in 1 2; do
for IT in 12 16; do for PF in 0 1 2; do
  echo "## round $round GSCAN_LANE_ITER=$IT GSCAN_LANE_PF=$PF"
  GSCAN_LANE_ITER=$IT GSCAN_LANE_PF=$PF timeout 300 $SW --gib 16 --iters 8 --variants 38 --bpc 0 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --pattern '[0-9]{16}' 2>&1 | grep -E "^variant"
done; done; done | tee gpurun_out/c_pf_iter_sweep.txt
