#!/bin/bash
# Round 5, session R: the pool stress test's statistics as numbers (how far are the slow-path counts from the test's thresholds?)
set -u
mkdir -p gpurun_out /tmp/pool
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
{
for k in 1 2 3; do
GSCAN_VIRTUAL_DEVICES=4 GSCAN_BLOCK_MIB=1 GSCAN_READERS=16 GSCAN_SECOND_STREAM_MIB=16 GSCAN_POOL_CAP=4 python tests/pool_driver.py --dir /tmp/pool --mib 256 --per-device 2 2>/dev/null | tail -1
done
GSCAN_VIRTUAL_DEVICES=4 GSCAN_BLOCK_MIB=1 GSCAN_READERS=16 GSCAN_SECOND_STREAM_MIB=16 GSCAN_POOL_CAP=8 GSCAN_FAIL_ALLOC_AFTER=2 python tests/pool_driver.py --dir /tmp/pool --mib 96 --per-device 2 2>/dev/null | tail -1
GSCAN_VIRTUAL_DEVICES=4 GSCAN_BLOCK_MIB=1 GSCAN_READERS=16 GSCAN_SECOND_STREAM_MIB=16 GSCAN_POOL_CAP=6 GSCAN_NT_COPY=1 python tests/pool_driver.py --dir /tmp/pool --mib 96 --per-device 2 2>/dev/null | tail -1
} | tee gpurun_out/r5r_pool_stats.txt
