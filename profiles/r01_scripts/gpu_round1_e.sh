#!/bin/bash
# Fresh-box re-check: full GPU parity suite, smoke, headline bench line.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== host =="; nproc; free -g | head -2; df -h /dev/shm /tmp | tail -2
echo "== pytest gpu =="
timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -16 | tee gpurun_out/e_pytest_gpu.txt
echo "== smoke =="
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
echo "== bench cfg2 (default) =="
timeout 900 python bench.py 2>&1 | tail -1 | tee gpurun_out/e_bench_cfg2.json
