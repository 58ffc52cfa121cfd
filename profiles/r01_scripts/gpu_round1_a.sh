#!/bin/bash
# First GPU session: does it run, is it right, how fast is it.  Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocminfo =="; /opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6
echo "== nproc =="; nproc; free -g | head -2
echo "== sweep K1 (8 GiB) =="
timeout 300 grab_amd/bin/gscan_sweep --gib 8 --iters 6 --variants 0,1,2,4,5,6 --bpc 0,4,8,16 2>&1 | tee gpurun_out/sweep_k1.txt
echo "== sweep K2 (8 GiB) =="
timeout 300 grab_amd/bin/gscan_sweep --gib 8 --iters 6 --variants 0,1,2,4,5,6 --bpc 0,4,8 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' 2>&1 | tee gpurun_out/sweep_k2.txt
echo "== pytest gpu (engine) =="
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -40 | tee gpurun_out/pytest_engine.txt
echo "== pytest gpu (filegrep) =="
timeout 1200 python -m pytest tests/test_gpu_filegrep.py -x -q -m gpu 2>&1 | tail -40 | tee gpurun_out/pytest_filegrep.txt
echo "== smoke =="
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 | tee gpurun_out/smoke.txt
echo "== bench small =="
timeout 600 python bench.py --files 64 --steps 10 --warmup 2 2>&1 | tail -3 | tee gpurun_out/bench_small.txt
