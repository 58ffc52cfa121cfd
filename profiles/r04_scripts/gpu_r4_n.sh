#!/bin/bash
# Round 4, session N: the random campaign through the small-file path (comparator fixed: threaded -O output compared file by file).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
{
python scripts/gpu_random_campaign.py --seed 4101 --seconds 140 --tree
python scripts/gpu_random_campaign.py --seed 4102 --seconds 50 --tree --lead-repeat
} 2>&1 | tee gpurun_out/n_random_campaign.txt
