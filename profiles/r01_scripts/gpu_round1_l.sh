#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu engine =="
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -4
SW=grab_amd/bin/gscan_sweep
{
echo "== K2 pair, identifier (dense)"; timeout 300 $SW --gib 16 --iters 6 --variants 6,14 --bpc 0 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' | tail -3
echo "== K2 pair, no match"; timeout 300 $SW --gib 16 --iters 6 --variants 6,14 --bpc 0 --pattern '[0-9]{16}' | tail -3
echo "== K2 general 3 classes dense"; timeout 300 $SW --gib 16 --iters 6 --variants 6,14 --bpc 0 --pattern '[a-z][0-9a-z][ a-z]{3,}' | tail -3
echo "== K2 pair very dense [a-z]{2,5}"; timeout 300 $SW --gib 8 --iters 4 --variants 6,14 --bpc 0 --pattern '[a-z]{2,5}' | tail -3
} 2>&1 | tee gpurun_out/l_sweep_k2_xp.txt
