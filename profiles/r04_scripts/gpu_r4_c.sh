#!/bin/bash
# Round 4, session C: what the exit of a run costs against what the run did (bytes through the pipe, workers), with half a
# second of quiet before every run; cfg1's ramp with smaller / fewer pinned blocks.
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
for n in (4, 16, 64, 256):
    d = "/dev/shm/c2_%d" % n
    os.makedirs(d, exist_ok=True)
    for i in range(n):
        p = "/dev/shm/c2_256/f%04d.txt" % i
        if n == 256 or not os.path.exists(p):
            synth.torch_text(64 << 20, i, dev).cpu().numpy().tofile(os.path.join(d, "f%04d.txt" % i))
        else:
            os.link(p, os.path.join(d, "f%04d.txt" % i))
synth.torch_text(256 << 20, 0, dev).cpu().numpy().tofile("/dev/shm/c1probe.txt")
PY
ls /dev/shm
G=grab_amd/bin/grab
{
echo "--- serial -r, corpus size"
for n in 4 16 64 256; do python scripts/ab_run.py --sleep 0.5 --reps 3 --bytes $((n * 67108864)) --env "" -- $G -r foobardoesnotexist /dev/shm/c2_$n; done
echo "--- -n 8, corpus size"
for n in 4 16 64 256; do python scripts/ab_run.py --sleep 0.5 --reps 3 --bytes $((n * 67108864)) --env "" -- $G -n 8 -r foobardoesnotexist /dev/shm/c2_$n; done
echo "--- -n 8, 16 GiB, variants (half a second of quiet before every run)"
python scripts/ab_run.py --sleep 0.5 --reps 3 --bytes $((256 * 67108864)) --interleave \
  --env "" --env "GRAB_NORMAL_EXIT=1" --env "GSCAN_BLOCK_MIB=8" --env "GSCAN_PREALLOC=0" --env "GSCAN_ONE_STREAM=1" --env "GRAB_LINE_PASS=0" --env "GSCAN_READERS=6" --env "GSCAN_PIN_FLAGS=1" \
  -- $G -n 8 -r foobardoesnotexist /dev/shm/c2_256
python scripts/ab_run.py --sleep 0.5 --reps 3 --bytes $((256 * 67108864)) --interleave --env "" --env "GSCAN_ONE_STREAM=1" -- $G -n 2 -r foobardoesnotexist /dev/shm/c2_256
echo "--- cfg1"
python scripts/ab_run.py --sleep 0.5 --reps 5 --bytes 268435456 --interleave --env "" --env "GSCAN_ONE_STREAM=1" --env "GSCAN_PREALLOC=0" --env "GSCAN_PREALLOC=4" --env "GSCAN_BLOCK_MIB=8" --env "GSCAN_BLOCK_MIB=8 GSCAN_ONE_STREAM=1" --env "GSCAN_BLOCK_MIB=4 GSCAN_ONE_STREAM=1" --env "GRAB_LINE_PASS=0 GSCAN_ONE_STREAM=1 GSCAN_BLOCK_MIB=8" -- $G foobardoesnotexist /dev/shm/c1probe.txt
} 2>&1 | tee gpurun_out/c_exit_cost.txt
g++ -O2 -std=c++17 scripts/probes/report_probe.cc -Igrab_amd/csrc -Iinclude -Lgrab_amd/lib -lgrabhost -lgscan -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/grab_amd/lib -Wl,-rpath,/opt/rocm/lib -o /tmp/report_probe && /tmp/report_probe 64 2>&1 | tee gpurun_out/c_report_probe.txt
rm -rf /dev/shm/c2_* /dev/shm/c1probe.txt
