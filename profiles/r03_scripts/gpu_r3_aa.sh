#!/bin/bash
# Round 3, session AA: 10 waves per workgroup (5 per SIMD, <= 96 VGPRs, 79 KiB LDS x 2) against 8 (4 per SIMD).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SW=$R/grab_amd/bin/gscan_sweep
LD_LIBRARY_PATH=$R/grab_amd/libw10 timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/aa_pytest.txt
{
for L in lib libw10 lib libw10 lib libw10 lib libw10; do
  echo "## $L"
  LD_LIBRARY_PATH=$R/grab_amd/$L timeout 300 $SW --gib 16 --iters 8 --variants 38 --bpc 0 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --pattern '[0-9]{16}' --pattern '[a-z][0-9][A-Z]{3}' --pattern '[a-z]{2,5}' 2>&1 | grep -E "^variant"
done
} | tee gpurun_out/aa_w10_sweep.txt
