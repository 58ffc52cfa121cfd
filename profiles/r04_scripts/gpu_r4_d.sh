#!/bin/bash
# Round 4, session D: (1) engine tests on the new kernel build; (2) K2 lane form: deferred reservation (lib) against round 3's
# reserve-and-wait (lib_ab), interleaved, with LDS-only fences in both; (3) what the exit costs against the process's age;
# (4) the ramp after the changes (one stream, no blocks made ahead, program upload not waited for): cfg1, cfg2 at 16 GiB,
# cfg4 at 16 GiB; block size 8 / 16 MiB.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/d_pytest_engine.txt
SW=$R/grab_amd/bin/gscan_sweep
{
for L in lib lib_ab lib lib_ab lib lib_ab; do
  echo "## $L"
  LD_LIBRARY_PATH=$R/grab_amd/$L timeout 300 $SW --gib 16 --iters 8 --variants 38 --bpc 0 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' --pattern '[0-9]{16}' --pattern '[a-z][0-9][A-Z]{3}' --pattern '[a-z]{2,5}' --pattern '[0-9]+\.[0-9]+' 2>&1 | grep -E "^variant"
done
} | tee gpurun_out/d_lane_defer_sweep.txt
python - <<'PY'
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "scripts")
import torch
from grab_amd import synth
import fullsize_parity
dev = torch.device("cuda", 0)
os.makedirs("/dev/shm/c2_256")
for i in range(256):
    synth.torch_text(64 << 20, i, dev).cpu().numpy().tofile("/dev/shm/c2_256/f%04d.txt" % i)
os.makedirs("/dev/shm/c2_16")
for i in range(16):
    os.link("/dev/shm/c2_256/f%04d.txt" % i, "/dev/shm/c2_16/f%04d.txt" % i)
synth.torch_text(256 << 20, 0, dev).cpu().numpy().tofile("/dev/shm/c1probe.txt")
os.makedirs("/dev/shm/c4probe")
fullsize_parity.gen_files("/dev/shm/c4probe", 32768, 512 << 10, 1, tree=(64, 64, 32))
PY
G=grab_amd/bin/grab
{
echo "--- exit against age: 1 GiB scanned, then asleep before _exit"
python scripts/ab_run.py --sleep 0.5 --reps 3 --bytes $((16 * 67108864)) --env "" --env "GRAB_EXIT_SLEEP_MS=200" --env "GRAB_EXIT_SLEEP_MS=500" --env "GRAB_EXIT_SLEEP_MS=1000" -- $G -n 8 -r foobardoesnotexist /dev/shm/c2_16
echo "--- cfg1"
python scripts/ab_run.py --sleep 0.5 --reps 5 --bytes 268435456 --interleave --env "" --env "GSCAN_BLOCK_MIB=8" --env "GSCAN_ONE_STREAM=0" --env "GRAB_LINE_PASS=0" -- $G foobardoesnotexist /dev/shm/c1probe.txt
echo "--- cfg2, 16 GiB"
python scripts/ab_run.py --sleep 0.5 --reps 3 --bytes $((256 * 67108864)) --interleave --env "" --env "GSCAN_BLOCK_MIB=8" --env "GSCAN_ONE_STREAM=0" -- $G -n 8 -r foobardoesnotexist /dev/shm/c2_256
python scripts/ab_run.py --sleep 0.5 --reps 3 --bytes $((256 * 67108864)) --interleave --env "" --env "GSCAN_BLOCK_MIB=8" -- $G -n 4 -r foobardoesnotexist /dev/shm/c2_256
python scripts/ab_run.py --sleep 0.5 --reps 3 --bytes $((256 * 67108864)) --interleave --env "" --env "GSCAN_BLOCK_MIB=8" -- $G -r foobardoesnotexist /dev/shm/c2_256
echo "--- cfg4, 16 GiB"
python scripts/ab_run.py --sleep 0.5 --reps 3 --bytes $((32768 * 524288)) --interleave --env "" --env "GSCAN_BLOCK_MIB=8" --env "GRAB_BATCH_READ=worker" --env "GRAB_BATCH_MIB=64" -- $G -n 8 -r -O -l foobardoesnotexist /dev/shm/c4probe
python scripts/ab_run.py --sleep 0.5 --reps 3 --bytes $((32768 * 524288)) --interleave --env "" --env "GRAB_BATCH_MIB=64" -- $G -n 4 -r -O -l foobardoesnotexist /dev/shm/c4probe
echo "--- time line, -n 8 over 16 GiB: the first 0.2 s"
GRAB_TIMING=1 GSCAN_TRACE=1 $G -n 8 -r foobardoesnotexist /dev/shm/c2_256 2>&1 >/dev/null | grep "trace\|timing\] +" | awk '{ if ($4+0 < 0.2 || $0 ~ /timing/) print }' | grep -v "reader: task\|reader: block in hand\|bytes read" | head -120
echo "--- time line, cfg1"
GRAB_TIMING=1 GSCAN_TRACE=1 $G foobardoesnotexist /dev/shm/c1probe.txt 2>&1 >/dev/null | grep -v "grab bytes\|printed so far" | grep -v "reader: task\|reader: block in hand\|bytes read" | head -60
} 2>&1 | tee gpurun_out/d_ramp.txt
rm -rf /dev/shm/c2_* /dev/shm/c1probe.txt /dev/shm/c4probe
