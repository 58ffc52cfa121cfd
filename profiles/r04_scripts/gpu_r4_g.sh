#!/bin/bash
# Round 4, session G: an event behind every K-th DMA only (GSCAN_MARK_EVERY): 1 (round 3's scheme) / 4 / 8, one stream and
# two copy streams, 8 and 16 MiB blocks, at 64 GiB (16 GiB x 4 names); before that the GPU suite (the ingest changed).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
( time timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 ) > gpurun_out/g_pytest.txt 2>&1
tail -6 gpurun_out/g_pytest.txt
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
C2="GSCAN_SHARED_COPY=2 GSCAN_SHARED_COMPUTE=1"
{
python scripts/ab_run.py --sleep 0.5 --reps 2 --bytes $NB --interleave \
  --env "GSCAN_MARK_EVERY=1" --env "" --env "GSCAN_MARK_EVERY=8" --env "GSCAN_BLOCK_MIB=16" \
  --env "$C2 GSCAN_MARK_EVERY=1" --env "$C2" --env "$C2 GSCAN_MARK_EVERY=8" --env "$C2 GSCAN_BLOCK_MIB=16" --env "$C2 GSCAN_READERS=10" \
  -- $G -n 8 -r foobardoesnotexist /dev/shm/c2_64g
for e in "GSCAN_MARK_EVERY=4" "$C2"; do
  env $e GRAB_CLOSE=1 GRAB_TIMING=1 GSCAN_TIMING=1 $G -n 8 -r foobardoesnotexist /dev/shm/c2_64g 2>&1 >/dev/null | grep "gscan timing\] device" | head -1
done
} 2>&1 | tee gpurun_out/g_mark_every.txt
rm -rf /dev/shm/c2_64g
