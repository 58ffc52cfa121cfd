#!/bin/bash
# Round 2, GPU session K: the two-class table kernel with a per-lane single-byte table (GSCAN_K2_LANETBL=1) against the pair table.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
S=grab_amd/bin/gscan_sweep
{
for e in 0 1; do
  if [ $e = 1 ]; then export GSCAN_K2_LANETBL=1; echo "## GSCAN_K2_LANETBL=1"; else unset GSCAN_K2_LANETBL; echo "## pair table"; fi
  $S --gib 16 --iters 8 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --variants 6 --bpc 0
  $S --gib 16 --iters 8 --pattern '[0-9]{16}' --variants 6 --bpc 0
  $S --gib 16 --iters 8 --pattern '[0-9]+\.[0-9]+' --variants 6 --bpc 0
  $S --gib 64 --iters 4 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --variants 6 --bpc 0
done
} > gpurun_out/k_kernel_sweep.txt 2>&1
grep -E "^#|variant|overflow" gpurun_out/k_kernel_sweep.txt
GSCAN_K2_LANETBL=1 timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_geometry.py -m gpu -q 2>&1 | tail -4
