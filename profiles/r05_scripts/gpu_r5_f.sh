#!/bin/bash
# Round 5, session F: BASELINE configs[4] (one big file, -O -l) came out at 1.02 s in session E's bench line against round 4's
# 0.84 s: which of this round's changes did it -- the read-ahead (a pool of 36 staging blocks instead of 16), the second copy
# stream made on demand?  One 8 GiB file, interleaved A/B.  And the clock sampler reading the card the kernels run on.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
python - <<'PY'
import sys
sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
import bench, fullsize_parity
bench.interleave_page_placement()
print("plants", fullsize_parity.gen_big("/dev/shm/one8g.bin", 8 << 30, 1 << 30, 250000))
PY
G=grab_amd/bin/grab
{
python scripts/ab_run.py --sleep 0.5 --reps 3 --bytes $((8 << 30)) --interleave --env "GSCAN_TIMING=1" --env "GSCAN_TIMING=1 GRAB_NO_READ_AHEAD=1" --env "GSCAN_TIMING=1 GSCAN_COPY_STREAMS=1" \
  --env "GSCAN_TIMING=1 GSCAN_SECOND_STREAM_MIB=0" --env "GSCAN_TIMING=1 GRAB_NO_READ_AHEAD=1 GSCAN_SECOND_STREAM_MIB=0" --env "GSCAN_TIMING=1 GRAB_NO_READ_AHEAD=1 GSCAN_POOL_CAP=36" \
  -- $G -O -l foobardoesnotexist /dev/shm/one8g.bin
for e in "GSCAN_DIAG=0" "GRAB_NO_READ_AHEAD=1" "GRAB_NO_READ_AHEAD=1 GSCAN_SECOND_STREAM_MIB=0"; do
  echo "== $e"
  env $e GRAB_CLOSE=1 GRAB_TIMING=1 GSCAN_TIMING=1 $G -O -l foobardoesnotexist /dev/shm/one8g.bin 2>&1 >/dev/null | grep "grab timing\] +\|gscan timing\] device\|gscan timing\] context\|grab timing\] device" | cut -c1-330
done
} 2>&1 | tee gpurun_out/r5f_cfg5.txt
rm -f /dev/shm/one8g.bin
timeout 600 python bench.py --no-e2e --no-cpu-baseline --no-live-traffic --steps 10 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline', r['roofline']['frac'], r['clocks'])
for k, v in r['kernels'].items():
    print(k, v['frac'], v.get('implied_sclk_ghz'), [p['clocks'] for p in v['passes']], v['sustained'])
" | tee gpurun_out/r5f_clocks.txt
ls -la /sys/class/drm/ | head -30 >> gpurun_out/r5f_clocks.txt
