#!/bin/bash
# Round 2, GPU session I: the table kernels with the next tile's text prefetched (variant 14) against variant 6.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
S=grab_amd/bin/gscan_sweep
{
$S --gib 16 --iters 8 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --variants 6,14 --bpc 0
$S --gib 16 --iters 8 --pattern '[0-9]{16}' --variants 6,14 --bpc 0
$S --gib 16 --iters 8 --pattern '[0-9a-f]{32}' --variants 6,14 --bpc 0
$S --gib 16 --iters 8 --pattern 'foobardoesnotexist|Linus|555-1234' --variants 6,14 --bpc 0
GSCAN_K3_DEPTH=4 $S --gib 16 --iters 8 --pattern 'foobardoesnotexist|Linus|555-1234' --variants 6,14 --bpc 0
$S --gib 16 --iters 8 --pattern '[0-9]+\.[0-9]+' --variants 6,14 --bpc 0
$S --gib 16 --iters 8 --pattern '(?i)foobar|k7Q,;q|[0-9]{12}x?' --variants 6,14 --bpc 0
$S --gib 64 --iters 4 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --variants 6,14 --bpc 0
$S --gib 64 --iters 4 --pattern 'foobardoesnotexist|Linus|555-1234' --variants 6,14 --bpc 0
} > gpurun_out/i_kernel_sweep.txt 2>&1
grep -E "^#|variant|overflow" gpurun_out/i_kernel_sweep.txt
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -q 2>&1 | tail -4
