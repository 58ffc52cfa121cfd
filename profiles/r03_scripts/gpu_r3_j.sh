#!/bin/bash
# Round 3, session J: the line pass with gather + resync: parity, then end to end against the host walk.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest: line pass parity =="
timeout 1200 python -m pytest tests/test_gpu_engine.py tests/test_gpu_filegrep.py -m gpu -q -x -k "line_extents or line_pass or match_ends or without_the_text or multichunk" 2>&1 | tail -4 | tee gpurun_out/j_pytest.txt
bash profiles/r03_scripts/gpu_r3_i.sh
cp gpurun_out/i_lines_diag.txt gpurun_out/j_lines_e2e.txt
