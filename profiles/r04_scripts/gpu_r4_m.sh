#!/bin/bash
# Round 4, session M: the random campaign through the small-file path again (comparator fixed), and the time line of the
# first 0.2 s of `-n 8` with this round's defaults.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
{
python scripts/gpu_random_campaign.py --seed 4101 --seconds 140 --tree
python scripts/gpu_random_campaign.py --seed 4102 --seconds 50 --tree --lead-repeat
} 2>&1 | tee gpurun_out/m_random_campaign.txt
python - <<'PY'
import os, sys
sys.path.insert(0, ".")
import torch
from grab_amd import synth
dev = torch.device("cuda", 0)
os.makedirs("/dev/shm/c2_256")
for i in range(256):
    synth.torch_text(64 << 20, i, dev).cpu().numpy().tofile("/dev/shm/c2_256/f%04d.txt" % i)
PY
G=grab_amd/bin/grab
{
for n in 8 4; do
echo "--- time line, -n $n over 16 GiB: until 0.2 s"
sleep 0.6
GRAB_TIMING=1 GSCAN_TRACE=1 $G -n $n -r foobardoesnotexist /dev/shm/c2_256 2>&1 >/dev/null | grep "trace\|timing\] +" | awk '{ if ($4+0 < 0.2 || $0 ~ /timing/) print }' | grep -v "reader: task\|reader: block in hand\|bytes read" | head -150
done
} 2>&1 | tee gpurun_out/m_ramp_timeline.txt
rm -rf /dev/shm/c2_256
