#!/bin/bash
# Round 4, session R: staging blocks mapped and touched while the HIP runtime starts, then only registered (gscan_prefault),
# against blocks from hipHostMalloc made while the pipe fills (GSCAN_PREFAULT=0): cfg1, 16 GiB, 64 GiB (16 GiB x 4 names).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_filegrep.py -m gpu -q -x -k "reader_pool or tree_differential or multichunk" 2>&1 | tail -3
python - <<'PY'
import os, sys
sys.path.insert(0, ".")
import torch
from grab_amd import synth
dev = torch.device("cuda", 0)
for i in range(256):
    sub = "/dev/shm/c2_64g/d0_%02d" % (i % 16)
    os.makedirs(sub, exist_ok=True)
    synth.torch_text(64 << 20, i, dev).cpu().numpy().tofile(sub + "/f%04d.txt" % i)
for k in range(1, 4):
    for i in range(256):
        sub = "/dev/shm/c2_64g/d%d_%02d" % (k, i % 16)
        os.makedirs(sub, exist_ok=True)
        os.link("/dev/shm/c2_64g/d0_%02d/f%04d.txt" % (i % 16, i), sub + "/f%04d.txt" % i)
os.makedirs("/dev/shm/c2_16g")
for i in range(256):
    os.link("/dev/shm/c2_64g/d0_%02d/f%04d.txt" % (i % 16, i), "/dev/shm/c2_16g/f%04d.txt" % i)
synth.torch_text(256 << 20, 0, dev).cpu().numpy().tofile("/dev/shm/c1probe.txt")
PY
G=grab_amd/bin/grab
{
echo "--- cfg1"
python scripts/ab_run.py --sleep 0.5 --reps 6 --bytes 268435456 --interleave --env "" --env "GSCAN_PREFAULT=0" -- $G foobardoesnotexist /dev/shm/c1probe.txt
echo "--- 16 GiB, -n 8"
python scripts/ab_run.py --sleep 0.5 --reps 4 --bytes $((256 * 67108864)) --interleave --env "" --env "GSCAN_PREFAULT=0" -- $G -n 8 -r foobardoesnotexist /dev/shm/c2_16g
echo "--- 64 GiB, -n 8"
python scripts/ab_run.py --sleep 0.5 --reps 3 --bytes $((1024 * 67108864)) --interleave --env "" --env "GSCAN_PREFAULT=0" -- $G -n 8 -r foobardoesnotexist /dev/shm/c2_64g
echo "--- time line, cfg1"
sleep 0.6
GRAB_TIMING=1 GSCAN_TRACE=1 $G foobardoesnotexist /dev/shm/c1probe.txt 2>&1 >/dev/null | grep -v "grab bytes\|printed so far" | grep -v "reader: task\|reader: block in hand\|bytes read" | head -50
} 2>&1 | tee gpurun_out/r_prefault.txt
rm -rf /dev/shm/c2_64g /dev/shm/c2_16g /dev/shm/c1probe.txt
