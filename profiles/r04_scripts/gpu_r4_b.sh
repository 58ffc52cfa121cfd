#!/bin/bash
# Round 4, session B: the GPU suite again (session A stopped at a test bug of mine), then the fixed cost of a run taken apart
# on a 16 GiB cfg2 corpus: what the exit costs against the number of workers / contexts / VRAM / streams, pinned blocks made
# ahead, one stream per device; cfg1 (one 256 MiB file) with the GSCAN_TRACE time line, against the reference on one core.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
( time timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > gpurun_out/b_pytest.txt 2>&1
tail -6 gpurun_out/b_pytest.txt
python - <<'PY'
import os, sys, time
sys.path.insert(0, ".")
import torch
from grab_amd import synth
d = "/dev/shm/c2probe"
dev = torch.device("cuda", 0)
for i in range(256):
    sub = os.path.join(d, "%02d" % (i % 16)); os.makedirs(sub, exist_ok=True)
    synth.torch_text(64 << 20, i, dev).cpu().numpy().tofile(os.path.join(sub, "f%04d.txt" % i))
synth.torch_text(256 << 20, 0, dev).cpu().numpy().tofile("/dev/shm/c1probe.txt")
PY
G=grab_amd/bin/grab
N2=$((256 * 67108864))
{
echo "--- workers"
for n in 2 4 8 16; do python scripts/ab_run.py --reps 3 --bytes $N2 --env "" -- $G -n $n -r foobardoesnotexist /dev/shm/c2probe; done
echo "--- serial -r"
python scripts/ab_run.py --reps 3 --bytes $N2 --env "" -- $G -r foobardoesnotexist /dev/shm/c2probe
echo "--- -n 8 variants (interleaved)"
python scripts/ab_run.py --reps 4 --bytes $N2 --interleave \
  --env "" --env "GRAB_CLOSE=1" --env "GRAB_LINE_PASS=0" --env "GSCAN_ONE_STREAM=1" --env "GSCAN_PREALLOC=0" --env "GSCAN_BLOCK_MIB=8" \
  --env "GSCAN_ONE_STREAM=1 GSCAN_BLOCK_MIB=8 GRAB_LINE_PASS=0" --env "GSCAN_READERS=4" --env "GSCAN_READERS=12" \
  -- $G -n 8 -r foobardoesnotexist /dev/shm/c2probe
echo "--- the same, each environment on its own (no other process's exit in front)"
for e in "" "GSCAN_ONE_STREAM=1" "GSCAN_PREALLOC=0" "GSCAN_ONE_STREAM=1 GSCAN_BLOCK_MIB=8 GRAB_LINE_PASS=0"; do sleep 1; python scripts/ab_run.py --reps 3 --bytes $N2 --env "$e" -- $G -n 8 -r foobardoesnotexist /dev/shm/c2probe; done
} 2>&1 | tee gpurun_out/b_fixed_cost.txt
{
echo "--- cfg1: one 256 MiB file"
python scripts/ab_run.py --reps 5 --bytes 268435456 --interleave --env "" --env "GSCAN_ONE_STREAM=1" --env "GSCAN_PREALLOC=0" --env "GSCAN_ONE_STREAM=1 GSCAN_BLOCK_MIB=8" --env "GRAB_LINE_PASS=0 GSCAN_ONE_STREAM=1" -- $G foobardoesnotexist /dev/shm/c1probe.txt
echo "--- reference, one core"
python - <<'PY'
import subprocess, time
for b in ("oracle/_ref/grab_jit", "oracle/grab_oracle"):
    ts = []
    for i in range(6):
        t0 = time.perf_counter(); subprocess.run([b, "foobardoesnotexist", "/dev/shm/c1probe.txt"], stdout=subprocess.DEVNULL); ts.append(time.perf_counter() - t0)
    print(b, "min %.4f s median %.4f s" % (min(ts[1:]), sorted(ts[1:])[2]))
PY
echo "--- time line, cfg1, default"
GRAB_TIMING=1 GSCAN_TRACE=1 GSCAN_TIMING=1 $G foobardoesnotexist /dev/shm/c1probe.txt 2>&1 >/dev/null | grep -v "grab bytes\|printed so far" | head -150
echo "--- time line, cfg1, one stream"
GRAB_TIMING=1 GSCAN_TRACE=1 GSCAN_ONE_STREAM=1 $G foobardoesnotexist /dev/shm/c1probe.txt 2>&1 >/dev/null | grep -v "grab bytes\|printed so far" | grep -v "reader: task\|reader: block in hand" | head -100
echo "--- time line, -n 8 over 16 GiB: the first 0.25 s"
GRAB_TIMING=1 GSCAN_TRACE=1 $G -n 8 -r foobardoesnotexist /dev/shm/c2probe 2>&1 >/dev/null | grep "trace\|timing\] +" | awk '{ if ($4+0 < 0.26 || $0 ~ /timing/) print }' | grep -v "reader: task\|reader: block in hand" | head -150
} 2>&1 | tee gpurun_out/b_cfg1.txt
rm -rf /dev/shm/c2probe /dev/shm/c1probe.txt
