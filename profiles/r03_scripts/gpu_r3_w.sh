#!/bin/bash
# Round 3, session W: records through the wave's strip (one coalesced store per 64) against: straight from the bit loops
# (libnostage), no record stores at all (libnostore, timing only), no reservation (libslice, timing only).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SW=$R/grab_amd/bin/gscan_sweep
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/w_pytest.txt
{
for L in lib libnostage libnostore libslice lib libnostage libnostore libslice lib libnostage; do
  echo "## $L"
  LD_LIBRARY_PATH=$R/grab_amd/$L timeout 300 $SW --gib 16 --iters 8 --variants 38 --bpc 0 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --pattern '[0-9]{16}' --pattern '[a-z][0-9][A-Z]{3}' --pattern '[a-z]{2,5}' 2>&1 | grep -E "^variant"
done
} | tee gpurun_out/w_stage_sweep.txt
