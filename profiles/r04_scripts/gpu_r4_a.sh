#!/bin/bash
# Round 4, session A: (1) the whole GPU suite with the new tests (shipped cfg3 kernel in every matrix, gscan_submit_files,
# eight device indices); (2) cfg4 at 16 GiB: small files through the reader pool against round 3's worker-read path, batch
# sizes; (3) where the fixed 0.3 s goes: HW queue count, stream sharing, block size, on a 16 GiB cfg2 corpus; cfg1.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/a_pytest.txt 2>&1
tail -5 gpurun_out/a_pytest.txt
nproc; free -g | head -2
python - <<'PY'
import os, sys, time
sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
import fullsize_parity
t0 = time.time()
d = "/dev/shm/c4probe"
os.makedirs(d, exist_ok=True)
fullsize_parity.gen_files(d, 32768, 512 << 10, 1, tree=(64, 64, 32))
print("cfg4 tree: %.1f s" % (time.time() - t0))
PY
G=grab_amd/bin/grab
N4=$((32768 * 524288))
{
python scripts/ab_run.py --reps 3 --bytes $N4 --interleave \
  --env "" --env "GRAB_BATCH_READ=worker" --env "GRAB_BATCH_MIB=8" --env "GRAB_BATCH_MIB=16" --env "GRAB_BATCH_MIB=64" --env "GRAB_BATCH_MIB=128" \
  -- $G -n 8 -r -O -l foobardoesnotexist /dev/shm/c4probe
python scripts/ab_run.py --reps 2 --bytes $N4 --env "" --env "GRAB_BATCH_READ=worker" -- $G -n 16 -r -O -l foobardoesnotexist /dev/shm/c4probe
python scripts/ab_run.py --reps 2 --bytes $N4 --env "GSCAN_TIMING=1" --env "GSCAN_TIMING=1 GRAB_BATCH_READ=worker" -- $G -n 8 -r -O -l foobardoesnotexist /dev/shm/c4probe
} 2>&1 | tee gpurun_out/a_cfg4_reader_pool.txt
$G -n 8 -r -O -l foobardoesnotexist /dev/shm/c4probe | sort | md5sum > gpurun_out/a_cfg4_md5.txt
GRAB_BATCH_READ=worker $G -n 8 -r -O -l foobardoesnotexist /dev/shm/c4probe | sort | md5sum >> gpurun_out/a_cfg4_md5.txt
oracle/_ref/grab_jit -n 32 -r -O -l foobardoesnotexist /dev/shm/c4probe | sort | md5sum >> gpurun_out/a_cfg4_md5.txt
cat gpurun_out/a_cfg4_md5.txt
rm -rf /dev/shm/c4probe
# cfg2 at 16 GiB: 256 x 64 MiB
python - <<'PY'
import os, sys, time
sys.path.insert(0, ".")
import torch
from grab_amd import synth
d = "/dev/shm/c2probe"
dev = torch.device("cuda", 0)
t0 = time.time()
for i in range(256):
    sub = os.path.join(d, "%02d" % (i % 16)); os.makedirs(sub, exist_ok=True)
    synth.torch_text(64 << 20, i, dev).cpu().numpy().tofile(os.path.join(sub, "f%04d.txt" % i))
synth.torch_text(256 << 20, 0, dev).cpu().numpy().tofile("/dev/shm/c1probe.txt")
print("cfg2 16 GiB: %.1f s" % (time.time() - t0))
PY
N2=$((256 * 67108864))
{
python scripts/ab_run.py --reps 4 --bytes $N2 --interleave \
  --env "" --env "GPU_MAX_HW_QUEUES=1" --env "GPU_MAX_HW_QUEUES=2" --env "GSCAN_SHARED_COMPUTE=1" --env "GSCAN_SHARED_COMPUTE=1 GPU_MAX_HW_QUEUES=2" \
  --env "GSCAN_BLOCK_MIB=8" --env "GSCAN_BLOCK_MIB=8 GSCAN_SHARED_COMPUTE=1 GPU_MAX_HW_QUEUES=2" --env "HSA_ENABLE_SDMA=0" \
  -- $G -n 8 -r foobardoesnotexist /dev/shm/c2probe
echo "--- cfg1: one 256 MiB file"
python scripts/ab_run.py --reps 5 --bytes 268435456 --interleave --env "" --env "GPU_MAX_HW_QUEUES=1" --env "GSCAN_SHARED_COMPUTE=1 GPU_MAX_HW_QUEUES=2" --env "GSCAN_READERS=16" -- $G foobardoesnotexist /dev/shm/c1probe.txt
echo "--- reference, one core"
for i in 1 2 3; do /usr/bin/time -f "%e s" oracle/_ref/grab_jit foobardoesnotexist /dev/shm/c1probe.txt; done
echo "--- GSCAN_TIMING phases, cfg2"
GRAB_TIMING=1 GSCAN_TIMING=1 $G -n 8 -r foobardoesnotexist /dev/shm/c2probe 2>&1 >/dev/null | grep -v "^\[gscan timing\] gscan_open" | head -60
echo "--- GSCAN_TIMING phases, cfg1"
GRAB_TIMING=1 GSCAN_TIMING=1 $G foobardoesnotexist /dev/shm/c1probe.txt 2>&1 >/dev/null | head -40
} 2>&1 | tee gpurun_out/a_startup.txt
rm -rf /dev/shm/c2probe /dev/shm/c1probe.txt
