#!/bin/bash
# Round 5, session I: the host half of the pipe once more, now that the GPU half is known not to be the limiter -- where the
# readers run (the device's NUMA node / both sockets), how many, the staging block's size (do 2 MiB blocks that stay in the
# cores' caches help the DMA engine or hurt it?), the non-temporal copy: 64 GiB of distinct files, interleaved A/B.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
python - <<'PY'
import os, sys
sys.path.insert(0, ".")
import torch, bench
from grab_amd import synth
bench.interleave_page_placement()
dev = torch.device("cuda", 0)
for i in range(1024):
    sub = "/dev/shm/c64/d%02d" % (i % 16)
    os.makedirs(sub, exist_ok=True)
    synth.torch_text(64 << 20, i, dev).cpu().numpy().tofile(sub + "/f%04d.txt" % i)
PY
G=grab_amd/bin/grab
NB=$((1024 * 67108864))
{
python scripts/ab_run.py --sleep 0.5 --reps 2 --bytes $NB --interleave --env "GSCAN_TIMING=1" --env "GSCAN_TIMING=1 GSCAN_NUMA=0" --env "GSCAN_TIMING=1 GSCAN_NUMA=0 GSCAN_READERS=16" \
   --env "GSCAN_TIMING=1 GSCAN_READERS=16" --env "GSCAN_TIMING=1 GSCAN_NT_COPY=1" --env "GSCAN_TIMING=1 GSCAN_NT_COPY=1 GSCAN_READERS=12" \
   --env "GSCAN_TIMING=1 GSCAN_BLOCK_MIB=2 GSCAN_POOL_CAP=48" --env "GSCAN_TIMING=1 GSCAN_BLOCK_MIB=4 GSCAN_POOL_CAP=32" --env "GSCAN_TIMING=1 GSCAN_BLOCK_MIB=16" \
   --env "GSCAN_TIMING=1 GSCAN_POOL_CAP=32" --env "GSCAN_TIMING=1 GSCAN_NUMA=0 GSCAN_READERS=16 GSCAN_NT_COPY=1" \
   -- $G -n 8 -r foobardoesnotexist /dev/shm/c64
numactl --hardware 2>/dev/null | head -12
} 2>&1 | tee gpurun_out/r5i_host_side.txt
rm -rf /dev/shm/c64
