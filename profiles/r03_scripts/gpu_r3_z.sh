#!/bin/bash
# Round 3, session Z: runs of <= 64 records written without a loop (lane_write_few): lib; against the bit loops (libloop);
# both once more without the reservation (libslice, libsliceloop: timing only).  Engine tests first (they use lib).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SW=$R/grab_amd/bin/gscan_sweep
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/z_pytest.txt
{
for L in lib libloop libslice libsliceloop lib libloop libslice libsliceloop lib libloop libslice; do
  echo "## $L"
  LD_LIBRARY_PATH=$R/grab_amd/$L timeout 300 $SW --gib 16 --iters 8 --variants 38 --bpc 0 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --pattern '[0-9]{16}' --pattern '[a-z][0-9][A-Z]{3}' --pattern '[a-z]{2,5}' 2>&1 | grep -E "^variant"
done
} | tee gpurun_out/z_few_sweep.txt
