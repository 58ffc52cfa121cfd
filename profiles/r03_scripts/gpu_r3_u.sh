#!/bin/bash
# Round 3, session U: the halo by ballot (one look-up + a ballot per class instead of 16 look-ups and a merge), against the
# build before it: engine tests first, then A A B B A B on the same box.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SW=$R/grab_amd/bin/gscan_sweep
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/u_pytest.txt
{
for L in lib lib libprev libprev lib libprev lib libprev; do
  echo "## $L"
  LD_LIBRARY_PATH=$R/grab_amd/$L timeout 300 $SW --gib 16 --iters 8 --variants 38 --bpc 0 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --pattern '[0-9]{16}' --pattern '[a-z][0-9][A-Z]{3}' --pattern '[0-9]+\.[0-9]+' --pattern '[a-z]{2,5}' 2>&1 | grep -E "^variant"
done
} | tee gpurun_out/u_halo_sweep.txt
