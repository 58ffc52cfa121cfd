#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu (engine) =="
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -25 | tee gpurun_out/pytest_engine.txt
SW=grab_amd/bin/gscan_sweep
echo "== K2 pair, dense (ident) =="
timeout 300 $SW --gib 16 --iters 6 --variants 0,2,4,6 --bpc 0 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' 2>&1 | tee gpurun_out/sweep_k2_ident.txt
echo "== K2 pair, no match =="
timeout 300 $SW --gib 16 --iters 6 --variants 0,2,4,6 --bpc 0 --pattern '[0-9]{16}' 2>&1 | tee gpurun_out/sweep_k2_nomatch.txt
echo "== K2 pair wide, no match =="
timeout 300 $SW --gib 16 --iters 6 --variants 4,6 --bpc 0 --pattern '[0-9a-f]{32}' 2>&1 | tee gpurun_out/sweep_k2_wide.txt
echo "== K2 general (3 classes) =="
timeout 300 $SW --gib 16 --iters 6 --variants 0,2,4,6 --bpc 0,8 --pattern '[a-z][0-9][A-Z_]{4}' 2>&1 | tee gpurun_out/sweep_k2_general.txt
echo "== K1 =="
timeout 300 $SW --gib 16 --iters 6 --variants 2,6 --bpc 0 2>&1 | tee gpurun_out/sweep_k1.txt
echo "== pytest gpu (filegrep) =="
timeout 900 python -m pytest tests/test_gpu_filegrep.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/pytest_filegrep.txt
