#!/bin/bash
# Round 5, session P: random differential campaigns at the final code (k_lines' 32-byte steps and LDS tail table came after
# session D's): random patterns of the supported grammar through `grab` in all three output modes against the oracle.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
timeout 200 python scripts/gpu_random_campaign.py --seed 70707 --seconds 110 2>&1 | tail -2 | tee gpurun_out/r5p_random_campaign.txt
timeout 200 python scripts/gpu_random_campaign.py --seed 80808 --seconds 80 --lead-repeat 2>&1 | tail -2 | tee -a gpurun_out/r5p_random_campaign.txt
timeout 200 python scripts/gpu_random_campaign.py --seed 90909 --seconds 80 --tree 2>&1 | tail -2 | tee -a gpurun_out/r5p_random_campaign.txt
