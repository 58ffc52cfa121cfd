#!/bin/bash
# Round 3, session V: what would records WITHOUT a reservation buy?  (experiment build -DGSCAN_LANE_SLICE=768: sub-tile pair d
# writes at d * 768, no atomic; timing only -- the sweep's match count is not meaningful for it)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SW=$R/grab_amd/bin/gscan_sweep
{
for L in lib libslice lib libslice lib libslice; do
  echo "## $L"
  LD_LIBRARY_PATH=$R/grab_amd/$L timeout 300 $SW --gib 16 --iters 8 --variants 38 --bpc 0 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --pattern '[0-9]{16}' --pattern '[a-z][0-9][A-Z]{3}' --pattern '[0-9]+\.[0-9]+' 2>&1 | grep -E "^variant"
done
} | tee gpurun_out/v_slice_sweep.txt
