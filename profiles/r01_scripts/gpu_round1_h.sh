#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu =="
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/h_pytest_gpu.txt
bash scripts/gpu_round1_g.sh
