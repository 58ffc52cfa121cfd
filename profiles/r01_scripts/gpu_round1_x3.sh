#!/bin/bash
# Same-box A/B of the K2 general form: previous commit's engine (grab_amd/_ab_old, 32-copy table, 256 threads) against
# the per-lane table (512 threads).  Interleaved, three rounds.
set -u
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
{
for round in 1 2 3; do
for pat in '[a-z][0-9][A-Z]{3}' '[a-z]{3}[0-9]{3}[A-F]{2}_' '[a-z][0-9][A-Z][a-z]{20}'; do
  for which in old new; do
    SW=$R/grab_amd/bin/gscan_sweep; [ $which = old ] && SW=$R/grab_amd/_ab_old/bin/gscan_sweep
    echo "== $which round $round pattern $pat"
    timeout 300 $SW --gib 8 --iters 6 --variants 6,4 --bpc 0 --pattern "$pat" | grep -v "^overflow" | tail -2
  done
done
done
} 2>&1 | tee gpurun_out/x3_k2_general_ab.txt
