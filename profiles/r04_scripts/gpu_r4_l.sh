#!/bin/bash
# Round 4, session L: random patterns through the small-file path against the oracle (scripts/gpu_random_campaign.py --tree);
# what the exit of an OLD process (asleep 0.6 s before _exit) costs against what it holds: workers, readers, streams, VRAM.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
{
python scripts/gpu_random_campaign.py --seed 4001 --seconds 150 --tree
python scripts/gpu_random_campaign.py --seed 4002 --seconds 60 --tree --lead-repeat
python scripts/gpu_random_campaign.py --seed 4003 --seconds 45
} 2>&1 | tee gpurun_out/l_random_campaign.txt
python - <<'PY'
import os, sys
sys.path.insert(0, ".")
import torch
from grab_amd import synth
dev = torch.device("cuda", 0)
os.makedirs("/dev/shm/c2_16")
for i in range(16):
    synth.torch_text(64 << 20, i, dev).cpu().numpy().tofile("/dev/shm/c2_16/f%04d.txt" % i)
PY
G=grab_amd/bin/grab
{
echo "--- exit of a process that is 0.7 s old (1 GiB scanned, 0.6 s asleep): what it holds"
python scripts/ab_run.py --sleep 0.5 --reps 3 --bytes $((16 * 67108864)) --interleave \
  --env "GRAB_EXIT_SLEEP_MS=600" --env "GRAB_EXIT_SLEEP_MS=600 GSCAN_READERS=2" --env "GRAB_EXIT_SLEEP_MS=600 GSCAN_ONE_STREAM_COPIES=1" --env "GRAB_EXIT_SLEEP_MS=600 GRAB_LINE_PASS=0" \
  --env "GRAB_EXIT_SLEEP_MS=600 GRAB_CLOSE=1" --env "GRAB_EXIT_SLEEP_MS=600 GSCAN_ONE_STREAM=0" \
  -- $G -n 8 -r foobardoesnotexist /dev/shm/c2_16
python scripts/ab_run.py --sleep 0.5 --reps 3 --bytes $((16 * 67108864)) --env "GRAB_EXIT_SLEEP_MS=600" -- $G -n 2 -r foobardoesnotexist /dev/shm/c2_16
python scripts/ab_run.py --sleep 0.5 --reps 3 --bytes $((16 * 67108864)) --env "GRAB_EXIT_SLEEP_MS=600" -- $G -r foobardoesnotexist /dev/shm/c2_16
python scripts/ab_run.py --sleep 0.5 --reps 3 --bytes 67108864 --env "GRAB_EXIT_SLEEP_MS=600" --env "GRAB_EXIT_SLEEP_MS=0" -- $G foobardoesnotexist /dev/shm/c2_16/f0000.txt
} 2>&1 | tee gpurun_out/l_exit_of_an_old_process.txt
rm -rf /dev/shm/c2_16
