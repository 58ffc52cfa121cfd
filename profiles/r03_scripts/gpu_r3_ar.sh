#!/bin/bash
# Round 3, session AR: cfg4 at 16 GiB (32 768 files of 512 KiB): a worker spends 0.15 s of its 0.5 s in gscan_submit_segs, 1.25 ms
# per 16 MiB batch (r03_aq_*).  Bigger pinned blocks = fewer, larger batches: GSCAN_BLOCK_MIB 16 / 32 / 64.
set -u
mkdir -p gpurun_out
python - <<'PY'
import os, sys, time
sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
import fullsize_parity
d = "/dev/shm/c4probe"
os.makedirs(d, exist_ok=True)
fullsize_parity.gen_files(d, 32768, 512 << 10, 1, tree=(64, 64, 32))
PY
{
for rep in 1 2; do for mib in 16 32 64; do
  GSCAN_BLOCK_MIB=$mib GRAB_TIMING=1 grab_amd/bin/grab -n 8 -r -O -l foobardoesnotexist /dev/shm/c4probe 2> gpurun_out/ar.err | wc -l
  echo "block $mib MiB: $(grep leaving gpurun_out/ar.err | tail -1) | $(grep 'device 0: files' gpurun_out/ar.err | head -1 | cut -d'|' -f2)"
done; done
} | tee gpurun_out/ar_cfg4_block_size.txt
rm -rf /dev/shm/c4probe
