#!/bin/bash
# Round 5, session A: (1) the refactored engine (one stream layout, pool hooks) under the pool stress test, the ABI tests and
# a slice of the engine / CLI tests; (2) the device VM with the wave-wide survivor queue: the three rows of r03_e_vm_sweep;
# (3) WHERE DOES THE LINK IDLE: rocprofv3 --memory-copy-trace of `grab -n 8 -r` over 16 GiB, the ingest pipe with the DMA
# stubbed out (GSCAN_DIAG=1) and with the reads stubbed out (GSCAN_DIAG=2) at 64 GiB; (4) the N = 8 model's measurable terms
# (scripts/n8_model.py): the fixed cost with eight device indices, the host copy ceiling at 8..64 readers, pread vs NT copy.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
{
timeout 900 python -m pytest tests/test_gpu_pool.py -m gpu -q -x 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -q -x 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_filegrep.py -m gpu -q -x -k "reader_pool or errors_surface or returned_once or tree_differential or offsets_without or golden" 2>&1 | tail -3
} | tee gpurun_out/a_pytest.txt
SW=$R/grab_amd/bin/gscan_sweep
{
for p in '(\w)\1{3,}x|foobardoes(?=not)' 'a+b+c' '[a-z]+\([a-z0-9, ]*\);' '\w+@\w+\.com' '(?:foo|bar)+baz'; do
  timeout 300 $SW --gib 8 --iters 3 --variants 38 --bpc 0 --pattern "$p"
done
} 2>&1 | tee gpurun_out/a_vm_sweep.txt
# corpus: 256 x 64 MiB = 16 GiB, pages interleaved over the NUMA nodes; x 4 names = 64 GiB
python - <<'PY'
import os, sys
sys.path.insert(0, ".")
import torch, bench
from grab_amd import synth
bench.interleave_page_placement()
dev = torch.device("cuda", 0)
for i in range(256):
    sub = "/dev/shm/c16/d0_%02d" % (i % 16)
    os.makedirs(sub, exist_ok=True)
    synth.torch_text(64 << 20, i, dev).cpu().numpy().tofile(sub + "/f%04d.txt" % i)
for k in range(1, 4):
    for i in range(256):
        sub = "/dev/shm/c64/d%d_%02d" % (k, i % 16)
        os.makedirs(sub, exist_ok=True)
        os.link("/dev/shm/c16/d0_%02d/f%04d.txt" % (i % 16, i), sub + "/f%04d.txt" % i)
for i in range(256):
    sub = "/dev/shm/c64/d0_%02d" % (i % 16)
    os.makedirs(sub, exist_ok=True)
    os.link("/dev/shm/c16/d0_%02d/f%04d.txt" % (i % 16, i), sub + "/f%04d.txt" % i)
PY
G=grab_amd/bin/grab
NB=$((1024 * 67108864))
{
echo "--- copy trace, 16 GiB"
python scripts/copy_trace.py --out gpurun_out/a_copytrace -- $G -n 8 -r foobardoesnotexist /dev/shm/c16
echo "--- 64 GiB: as shipped / no DMA no scan (DIAG=1) / no read (DIAG=2) / NT copy"
python scripts/ab_run.py --sleep 0.5 --reps 2 --bytes $NB --interleave --env "GSCAN_TIMING=1" --env "GSCAN_TIMING=1 GSCAN_DIAG=1" --env "GSCAN_TIMING=1 GSCAN_DIAG=2" --env "GSCAN_TIMING=1 GSCAN_NT_COPY=1" \
   --env "GSCAN_TIMING=1 GSCAN_COPY_STREAMS=1" --env "GSCAN_TIMING=1 GSCAN_DIAG=2 GSCAN_COPY_STREAMS=1" --env "GSCAN_TIMING=1 GSCAN_DIAG=2 GSCAN_BLOCK_MIB=32" \
   -- $G -n 8 -r foobardoesnotexist /dev/shm/c64
for e in "GSCAN_DIAG=0" "GSCAN_DIAG=1" "GSCAN_DIAG=2"; do
  env $e GRAB_CLOSE=1 GRAB_TIMING=1 GSCAN_TIMING=1 $G -n 8 -r foobardoesnotexist /dev/shm/c64 2>&1 >/dev/null | grep "gscan timing\] device 0 readers" | head -1
done
} 2>&1 | tee gpurun_out/a_pipe.txt
rm -rf gpurun_out/a_copytrace
{
echo "--- n8 model, 64 GiB"
python scripts/n8_model.py --dir /dev/shm/c64 --bytes $NB
echo "--- n8 model, 16 GiB (fixed cost only)"
python scripts/n8_model.py --dir /dev/shm/c16 --bytes $((256 * 67108864)) --no-host-copy
} 2>&1 | tee gpurun_out/a_n8.txt
numactl --hardware 2>/dev/null | head -4; lscpu | grep -i "model name\|socket\|numa node" | head -6
rm -rf /dev/shm/c16 /dev/shm/c64
