#!/bin/bash
# Round 4, session K: is the pipe (49 of the link's 57.6 GB/s) the host's DRAM?  Per byte: page cache read + pinned write +
# the write's allocate + DMA read.  The readers' copy modes (pread / map + non-temporal copy / bounce + non-temporal copy),
# write-combined and non-coherent blocks, reader counts -- at 64 GiB (16 GiB x 4 names), two copy streams (the default).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
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
PY
G=grab_amd/bin/grab
NB=$((1024 * 67108864))
{
python scripts/ab_run.py --sleep 0.5 --reps 2 --bytes $NB --interleave \
  --env "" --env "GSCAN_READ_MODE=2" --env "GSCAN_READ_MODE=2 GSCAN_READERS=12" --env "GSCAN_READ_MODE=2 GSCAN_READERS=16" --env "GSCAN_READ_MODE=1 GSCAN_READERS=12" \
  --env "GSCAN_PIN_FLAGS=2" --env "GSCAN_PIN_FLAGS=2 GSCAN_READERS=12" --env "GSCAN_PIN_FLAGS=1" --env "GSCAN_READERS=6" \
  -- $G -n 8 -r foobardoesnotexist /dev/shm/c2_64g
for e in "GSCAN_READ_MODE=0" "GSCAN_READ_MODE=2 GSCAN_READERS=12" "GSCAN_PIN_FLAGS=2"; do
  env $e GRAB_CLOSE=1 GRAB_TIMING=1 GSCAN_TIMING=1 $G -n 8 -r foobardoesnotexist /dev/shm/c2_64g 2>&1 >/dev/null | grep "gscan timing\] device" | head -1
done
echo "--- cfg3 (dense output): where gscan_wait spends its time"
GRAB_CLOSE=1 GRAB_TIMING=1 GSCAN_TIMING=1 $G -n 8 -r -O -l '[A-Za-z_][A-Za-z0-9_]{15,}' /dev/shm/c2_64g 2>&1 >/dev/null | grep "gscan timing\] context\|grab timing\] device\|workers joined\|runtime up" | head -12
numactl --hardware 2>/dev/null | head -8; lscpu | grep -i "model name\|socket\|numa\|l3" | head -8; dmidecode -t memory 2>/dev/null | grep -i "speed\|size" | sort | uniq -c | head -8
} 2>&1 | tee gpurun_out/k_copy_modes.txt
rm -rf /dev/shm/c2_64g
