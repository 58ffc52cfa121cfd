#!/bin/bash
# Round 3, session S: sub-tiles per reservation (compile-time: 1 / 2 / 3), same box, interleaved rounds.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SW=$R/grab_amd/bin/gscan_sweep
for round in 1 2 3; do for N in 1 2 3; do
  echo "## round $round sub-tiles per reservation $N"
  L=$R/grab_amd/lib; [ $N != 2 ] && L=$R/grab_amd/libb$N
  LD_LIBRARY_PATH=$L timeout 300 $SW --gib 16 --iters 8 --variants 38 --bpc 0 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --pattern '[0-9]{16}' --pattern '[a-z][0-9][A-Z]{3}' --pattern '[a-z]{2,5}' 2>&1 | grep -E "^variant"
done; done | tee gpurun_out/s_batch_sweep.txt
