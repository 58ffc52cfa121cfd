#!/bin/bash
# Round 4, session E: engine tests on the fixed deferred-reservation build; the A/B sweep again; the steady state at 64 GiB
# (16 GiB of distinct files under four names each): reader-thread split, reader counts, block sizes; time lines.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/e_pytest_engine.txt
SW=$R/grab_amd/bin/gscan_sweep
{
for L in lib lib_ab lib lib_ab lib lib_ab lib lib_ab; do
  echo "## $L"
  LD_LIBRARY_PATH=$R/grab_amd/$L timeout 300 $SW --gib 16 --iters 8 --variants 38 --bpc 0 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --pattern '[0-9]{16}' --pattern '[a-z][0-9][A-Z]{3}' --pattern '[a-z]{2,5}' --pattern '[0-9]+\.[0-9]+' 2>&1 | grep -E "^variant"
done
} | tee gpurun_out/e_lane_defer_sweep.txt
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
synth.torch_text(256 << 20, 0, dev).cpu().numpy().tofile("/dev/shm/c1probe.txt")
PY
G=grab_amd/bin/grab
NB=$((1024 * 67108864))
{
echo "--- 64 GiB (16 GiB x 4 names), -n 8: readers, blocks"
python scripts/ab_run.py --sleep 0.5 --reps 2 --bytes $NB --interleave --env "" --env "GSCAN_READERS=10" --env "GSCAN_READERS=12" --env "GSCAN_READERS=16" --env "GSCAN_BLOCK_MIB=16" --env "GSCAN_BLOCK_MIB=4" -- $G -n 8 -r foobardoesnotexist /dev/shm/c2_64g
echo "--- workers"
python scripts/ab_run.py --sleep 0.5 --reps 2 --bytes $NB --env "" -- $G -n 4 -r foobardoesnotexist /dev/shm/c2_64g
python scripts/ab_run.py --sleep 0.5 --reps 2 --bytes $NB --env "" -- $G -r foobardoesnotexist /dev/shm/c2_64g
echo "--- reader split (contexts closed: GRAB_CLOSE)"
for r in 8 12; do GSCAN_READERS=$r GRAB_CLOSE=1 GRAB_TIMING=1 GSCAN_TIMING=1 $G -n 8 -r foobardoesnotexist /dev/shm/c2_64g 2>&1 >/dev/null | grep "gscan timing\] device\|workers joined\|runtime up" | head -4; done
echo "--- the reference"
python scripts/ab_run.py --sleep 0.5 --reps 2 --bytes $NB --env "" -- oracle/_ref/grab_jit -n 32 -r foobardoesnotexist /dev/shm/c2_64g
echo "--- cfg1"
python scripts/ab_run.py --sleep 0.5 --reps 5 --bytes 268435456 --interleave --env "" --env "GSCAN_BLOCK_MIB=16" --env "GRAB_LINE_PASS=0" -- $G foobardoesnotexist /dev/shm/c1probe.txt
echo "--- time line, cfg1"
sleep 0.5
GRAB_TIMING=1 GSCAN_TRACE=1 $G foobardoesnotexist /dev/shm/c1probe.txt 2>&1 >/dev/null | grep -v "grab bytes\|printed so far" | grep -v "reader: task\|reader: block in hand\|bytes read" | head -60
echo "--- time line, -n 8: the first 0.2 s"
sleep 0.5
GRAB_TIMING=1 GSCAN_TRACE=1 $G -n 8 -r foobardoesnotexist /dev/shm/c2_64g 2>&1 >/dev/null | grep "trace\|timing\] +" | awk '{ if ($4+0 < 0.2 || $0 ~ /timing/) print }' | grep -v "reader: task\|reader: block in hand\|bytes read\|piece queued" | head -150
} 2>&1 | tee gpurun_out/e_steady_state.txt
rm -rf /dev/shm/c2_64g /dev/shm/c1probe.txt
