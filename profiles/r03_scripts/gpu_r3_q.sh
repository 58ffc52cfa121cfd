#!/bin/bash
# Round 3, session Q: record-buffer shards (reservation counters, one 128-byte line each): 64 / 256 / 512, same box.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SW=$R/grab_amd/bin/gscan_sweep
for round in 1 2; do for N in 64 256 512; do
  echo "## round $round shards $N"
  L=$R/grab_amd/lib; [ $N != 64 ] && L=$R/grab_amd/lib$N
  LD_LIBRARY_PATH=$L timeout 300 $SW --gib 16 --iters 8 --variants 38 --bpc 0 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --pattern '[0-9]{16}' --pattern '[a-z]{2,5}' --pattern '[0-9]+\.[0-9]+' --pattern 'foobardoesnotexist' 2>&1 | grep -E "^variant"
done; done | tee gpurun_out/q_shards_sweep.txt
