#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu engine =="
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -4
SW=grab_amd/bin/gscan_sweep
{
echo "== K1 literal: 4-wave workgroups (6) vs single-wave (14), and ITER 8/16 solo"; timeout 600 $SW --gib 32 --iters 8 --variants 6,14,13,12 --bpc 0 | tail -5
echo "== K1 with word boundaries"; timeout 300 $SW --gib 16 --iters 6 --variants 6,14 --bpc 0 --pattern '\bfoobardoesnotexist\b' | tail -3
} 2>&1 | tee gpurun_out/n_sweep_k1_solo.txt
