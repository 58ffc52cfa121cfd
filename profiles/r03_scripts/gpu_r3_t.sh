#!/bin/bash
# Round 3, session T: the same sweep command eight times in a row (does the rate alternate from process to process?), then with
# a pause between the runs.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SW=$R/grab_amd/bin/gscan_sweep
{
for i in 1 2 3 4 5 6 7 8; do
  echo "## back to back, run $i"
  timeout 300 $SW --gib 16 --iters 8 --variants 38 --bpc 0 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --pattern '[0-9]{16}' 2>&1 | grep -E "^variant"
done
for i in 1 2 3 4; do
  sleep 3
  echo "## 3 s apart, run $i"
  timeout 300 $SW --gib 16 --iters 8 --variants 38 --bpc 0 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --pattern '[0-9]{16}' 2>&1 | grep -E "^variant"
done
echo "## 64 GiB arena, twice"
for i in 1 2; do timeout 300 $SW --gib 64 --iters 5 --variants 38 --bpc 0 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --pattern '[0-9]{16}' 2>&1 | grep -E "^variant"; done
} | tee gpurun_out/t_alternation.txt
