#!/bin/bash
# Round 3, session O: look-ups one step ahead (GSCAN_LANE_LA=1) vs two (2), same box, interleaved rounds.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SW=$R/grab_amd/bin/gscan_sweep
echo "== pytest: engine parity (default = LA 2) =="
timeout 1200 python -m pytest tests/test_gpu_engine.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/o_pytest.txt
for round in 1 2 3; do for LA in 1 2; do
  echo "## round $round GSCAN_LANE_LA=$LA"
  GSCAN_LANE_LA=$LA timeout 300 $SW --gib 16 --iters 8 --variants 38 --bpc 0 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --pattern '[0-9]{16}' --pattern '[a-z][0-9][A-Z]{3}' 2>&1 | grep -E "^variant"
done; done | tee gpurun_out/o_la_sweep.txt
