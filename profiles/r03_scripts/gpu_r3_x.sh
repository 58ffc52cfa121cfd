#!/bin/bash
# Round 3, session X: rolling refill (GSCAN_LANE_PF=3: the next tile's piece k is requested as soon as this tile's piece k has
# been looked up) against PF 0 and PF 1, same library, same box, interleaved.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SW=$R/grab_amd/bin/gscan_sweep
GSCAN_LANE_PF=3 timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -x -q -k "lane or variant or libpcre or cfg3 or ident" 2>&1 | tail -5 | tee gpurun_out/x_pytest.txt
{
for PF in 0 3 1 0 3 1 0 3; do
  echo "## PF $PF"
  GSCAN_LANE_PF=$PF timeout 300 $SW --gib 16 --iters 8 --variants 38 --bpc 0 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --pattern '[0-9]{16}' 2>&1 | grep -E "^variant"
done
} | tee gpurun_out/x_rolling_sweep.txt
