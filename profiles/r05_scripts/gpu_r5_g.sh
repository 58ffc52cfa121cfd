#!/bin/bash
# Round 5, session G: BASELINE configs[4] at its full 32 GiB, standalone (ab_run, interleaved) and the way bench.py measures it
# (its e2e_cfg5 block called from a process that holds a HIP context, like the bench does): session E's bench line had it at
# 1.02 s against round 4's 0.84 s, session F's 8 GiB runs show no such loss.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
python - <<'PY'
import sys
sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
import bench, fullsize_parity
bench.interleave_page_placement()
print("plants", fullsize_parity.gen_big("/dev/shm/one32g.bin", 32 << 30, 1 << 30, 1000000))
PY
G=grab_amd/bin/grab
{
python scripts/ab_run.py --sleep 0.5 --reps 3 --bytes $((32 << 30)) --interleave --env "GSCAN_TIMING=1" --env "GSCAN_TIMING=1 GRAB_NO_READ_AHEAD=1" --env "GSCAN_TIMING=1 GSCAN_SECOND_STREAM_MIB=0" --env "GSCAN_TIMING=1 GRAB_NO_READ_AHEAD=1 GSCAN_SECOND_STREAM_MIB=0" \
  -- $G -O -l foobardoesnotexist /dev/shm/one32g.bin
env GRAB_TIMING=1 GSCAN_TIMING=1 $G -O -l foobardoesnotexist /dev/shm/one32g.bin 2>&1 >/dev/null | grep "grab timing\] +\|gscan timing\] device\|grab timing\] device" | cut -c1-330
} 2>&1 | tee gpurun_out/r5g_cfg5_standalone.txt
rm -f /dev/shm/one32g.bin
python - <<'PY' 2>&1 | tee gpurun_out/r5g_cfg5_in_bench_context.txt
import json, sys
sys.path.insert(0, ".")
import torch, bench
dev = torch.device("cuda", 0)
x = torch.empty(1 << 30, dtype=torch.uint8, device=dev)   # (the bench process holds a context and memory while its e2e blocks run)
bench.interleave_page_placement()
for k in range(2):
    e = bench.e2e_cfg5("/dev/shm", 32, k == 0)
    print(json.dumps({q: e.get(q) for q in ("value", "wall_s", "back_to_back_wall_s", "same_as_reference", "at_8GiB", "corpus_write_s")}), flush=True)
PY
