#!/bin/bash
# Round 2, GPU session H: shard counters one per cache line; K2's epilogue per tile vs per wave under them.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
S=grab_amd/bin/gscan_sweep
{
for W in 0 1; do
  if [ $W = 1 ]; then export GSCAN_K2_EMIT_WAVE=1; echo "## GSCAN_K2_EMIT_WAVE=1"; fi
  $S --gib 16 --iters 6 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --variants 6 --bpc 0
  $S --gib 16 --iters 6 --pattern '[0-9]{16}' --variants 6 --bpc 0
  $S --gib 16 --iters 6 --pattern '[a-z][0-9][A-Z]{3}' --variants 6 --bpc 0
done
unset GSCAN_K2_EMIT_WAVE
$S --gib 16 --iters 6 --pattern 'foobardoesnotexist|Linus|555-1234' --variants 6 --bpc 0
$S --gib 16 --iters 6 --pattern '[0-9]+\.[0-9]+' --variants 6 --bpc 0
$S --gib 16 --iters 6 --pattern 'e|ee|eee' --variants 6 --bpc 0
$S --gib 16 --iters 6 --pattern 'foobardoesnotexist' --variants 6 --bpc 0
$S --gib 16 --iters 6 --pattern 'foo' --variants 6 --bpc 0
} > gpurun_out/h_kernel_sweep.txt 2>&1
grep -E "^#|variant|overflow" gpurun_out/h_kernel_sweep.txt
timeout 600 python -m pytest tests/test_gpu_engine.py -m gpu -q 2>&1 | tail -3
GSCAN_K2_EMIT_WAVE=1 timeout 600 python -m pytest tests/test_gpu_engine.py -m gpu -q -k "kernels_against or parity_patterns or dense" 2>&1 | tail -3
