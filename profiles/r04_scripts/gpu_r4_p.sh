#!/bin/bash
# Round 4, session P: the random campaign through the small-file path with the comparator that compares threaded -O output
# file by file; one 8 GiB file at 32 MiB windows (-L x 5): one context (three windows in flight) against several on the device.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
{
python scripts/gpu_random_campaign.py --seed 4201 --seconds 120 --tree
python scripts/gpu_random_campaign.py --seed 4202 --seconds 40 --tree --lead-repeat
} 2>&1 | tee gpurun_out/p_random_campaign.txt
python - <<'PY'
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "scripts")
import fullsize_parity
fullsize_parity.gen_big("/dev/shm/big8.bin", 8 << 30, 1 << 30, 250000)
PY
G=grab_amd/bin/grab
{
echo "--- 8 GiB, default chunk (1 GiB)"
python scripts/ab_run.py --sleep 0.5 --reps 3 --bytes $((8 << 30)) --env "" -- $G -O -l foobardoesnotexist /dev/shm/big8.bin
echo "--- 8 GiB, -L x 5 (32 MiB windows): contexts on the one device"
python scripts/ab_run.py --sleep 0.5 --reps 3 --bytes $((8 << 30)) --interleave --env "" --env "GRAB_DEVICES=2" --env "GRAB_DEVICES=4" --env "GRAB_DEVICES=8" -- $G -L -L -L -L -L -O -l foobardoesnotexist /dev/shm/big8.bin
echo "--- 8 GiB, -L -L (256 MiB windows)"
python scripts/ab_run.py --sleep 0.5 --reps 3 --bytes $((8 << 30)) --interleave --env "" --env "GRAB_DEVICES=2" -- $G -L -L -O -l foobardoesnotexist /dev/shm/big8.bin
} 2>&1 | tee gpurun_out/p_small_windows.txt
rm -f /dev/shm/big8.bin
