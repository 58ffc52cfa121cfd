#!/bin/bash
# Round 3, session M: the whole GPU suite on the ordered-result path, then the dense modes end to end with the wait split.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/m_pytest.txt
bash profiles/r03_scripts/gpu_r3_l.sh
cp gpurun_out/l_wait_split.txt gpurun_out/m_wait_split.txt
