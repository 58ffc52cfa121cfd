#!/bin/bash
# Round 3, session Y: the identifier scan with neither reservation nor record stores (libx1, timing only), with stores but no
# reservation (libx2), and as shipped (lib) -- is its distance to [0-9]{16} the records or the arithmetic?
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SW=$R/grab_amd/bin/gscan_sweep
{
for L in lib libx1 libx2 lib libx1 libx2 lib libx1; do
  echo "## $L"
  LD_LIBRARY_PATH=$R/grab_amd/$L timeout 300 $SW --gib 16 --iters 8 --variants 38 --bpc 0 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --pattern '[0-9]{16}' --pattern '[A-Za-z_][0-9]{15,}' --pattern '[a-z][0-9][A-Z]{3}' 2>&1 | grep -E "^variant"
done
} | tee gpurun_out/y_norecords_sweep.txt
