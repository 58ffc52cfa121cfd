#!/bin/bash
# Round 4, session F: the readers wait for blocks half of their time (r04_e): the DMA side is what is slow, 183 us per 8 MiB
# piece where the link alone needs 145.  Copy streams 1 / 2 / 3, block sizes, at 64 GiB (16 GiB x 4 names).
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
  --env "" --env "GSCAN_SHARED_COPY=2 GSCAN_SHARED_COMPUTE=1" --env "GSCAN_SHARED_COPY=3 GSCAN_SHARED_COMPUTE=1" --env "GSCAN_SHARED_COPY=2 GSCAN_SHARED_COMPUTE=1 GSCAN_BLOCK_MIB=16" \
  --env "GSCAN_SHARED_COPY=1 GSCAN_SHARED_COMPUTE=1" --env "GSCAN_BLOCK_MIB=32" --env "GSCAN_SHARED_COPY=2 GSCAN_SHARED_COMPUTE=1 GSCAN_READERS=12" --env "GSCAN_SHARED_COPY=4 GSCAN_SHARED_COMPUTE=1 GSCAN_READERS=12" \
  -- $G -n 8 -r foobardoesnotexist /dev/shm/c2_64g
for e in "GSCAN_SHARED_COPY=2 GSCAN_SHARED_COMPUTE=1" "GSCAN_SHARED_COPY=2 GSCAN_SHARED_COMPUTE=1 GSCAN_READERS=12"; do
  env $e GRAB_CLOSE=1 GRAB_TIMING=1 GSCAN_TIMING=1 $G -n 8 -r foobardoesnotexist /dev/shm/c2_64g 2>&1 >/dev/null | grep "gscan timing\] device" | head -1
done
} 2>&1 | tee gpurun_out/f_copy_streams.txt
rm -rf /dev/shm/c2_64g
