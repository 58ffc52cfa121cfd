#!/bin/bash
# Round 5, session C: the stream refactor (second copy stream made on demand; GSCAN_SCAN_STREAM) under the pool and engine
# tests; at 64 GiB: what the scans and read-backs that ride on the first copy stream cost the DMA side -- as shipped, no reads
# (DIAG=2), no reads and no scans (DIAG=3), the scans on a stream of their own; cfg1 with the second stream made on demand:
# read-ahead helper threads 8 / 4 / 2 / none.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
{
timeout 900 python -m pytest tests/test_gpu_pool.py -m gpu -q -x 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -q -x -k "fd or pipelined or batch or submit or files" 2>&1 | tail -4
GSCAN_SECOND_STREAM_MIB=0 timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -q -x -k "fd or pipelined or batch or submit or files" 2>&1 | tail -4
GSCAN_SCAN_STREAM=1 GSCAN_SECOND_STREAM_MIB=0 timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_filegrep.py -m gpu -q -x -k "fd or pipelined or batch or submit or files or reader_pool or read_ahead" 2>&1 | tail -4
} | tee gpurun_out/r5c_pytest.txt
python - <<'PY'
import os, sys
sys.path.insert(0, ".")
import torch, bench
from grab_amd import synth
bench.interleave_page_placement()
dev = torch.device("cuda", 0)
synth.torch_text(256 << 20, 0, dev).cpu().numpy().tofile("/dev/shm/one256.txt")
for i in range(256):
    sub = "/dev/shm/c64/d0_%02d" % (i % 16)
    os.makedirs(sub, exist_ok=True)
    synth.torch_text(64 << 20, i, dev).cpu().numpy().tofile(sub + "/f%04d.txt" % i)
for k in range(1, 4):
    for i in range(256):
        sub = "/dev/shm/c64/d%d_%02d" % (k, i % 16)
        os.makedirs(sub, exist_ok=True)
        os.link("/dev/shm/c64/d0_%02d/f%04d.txt" % (i % 16, i), sub + "/f%04d.txt" % i)
PY
G=grab_amd/bin/grab
NB=$((1024 * 67108864))
{
echo "--- 64 GiB"
python scripts/ab_run.py --sleep 0.5 --reps 2 --bytes $NB --interleave --env "GSCAN_TIMING=1" --env "GSCAN_TIMING=1 GSCAN_DIAG=2" --env "GSCAN_TIMING=1 GSCAN_DIAG=3" \
   --env "GSCAN_TIMING=1 GSCAN_SCAN_STREAM=1" --env "GSCAN_TIMING=1 GSCAN_DIAG=2 GSCAN_SCAN_STREAM=1" --env "GSCAN_TIMING=1 GSCAN_COPY_STREAMS=1" --env "GSCAN_TIMING=1 GSCAN_SECOND_STREAM_MIB=0" \
   --env "GSCAN_TIMING=1 GSCAN_DIAG=3 GSCAN_COPY_STREAMS=1" --env "GSCAN_TIMING=1 GSCAN_READERS=12 GSCAN_SCAN_STREAM=1" \
   -- $G -n 8 -r foobardoesnotexist /dev/shm/c64
echo "--- cfg3 at 64 GiB, as shipped vs scans on their own stream"
python scripts/ab_run.py --sleep 0.5 --reps 2 --bytes $NB --interleave --env "" --env "GSCAN_SCAN_STREAM=1" -- $G -n 8 -r -O -l '[A-Za-z_][A-Za-z0-9_]{15,}' /dev/shm/c64
} 2>&1 | tee gpurun_out/r5c_pipe.txt
{
echo "--- cfg1: one 256 MiB file"
python scripts/ab_run.py --sleep 0.5 --reps 8 --bytes $((256 << 20)) --interleave --env "" --env "GRAB_NO_READ_AHEAD=1" --env "GSCAN_AHEAD_THREADS=4" --env "GSCAN_AHEAD_THREADS=2" --env "GSCAN_SECOND_STREAM_MIB=0" -- $G foobardoesnotexist /dev/shm/one256.txt
GSCAN_TRACE=1 GRAB_TIMING=1 $G foobardoesnotexist /dev/shm/one256.txt 2>&1 >/dev/null | grep "grab timing\] +\|submit_fd\|slot:\|wait:\|launching" | head -40
} 2>&1 | tee gpurun_out/r5c_cfg1.txt
rm -rf /dev/shm/c64 /dev/shm/one256.txt
