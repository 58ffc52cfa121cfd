#!/bin/bash
# Round-1 closing run: the GPU suite, then the evidence script (bench lines, rocprofv3 kernel stats, PMC traffic, full-size cfg3).
set -u
mkdir -p gpurun_out
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/final_pytest.txt
bash scripts/gpu_round1_z.sh
