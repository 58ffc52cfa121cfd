#!/bin/bash
# Round 2, closing run 2 (after the start-up / stream / walk / VM changes): the bench line, rocprofv3 kernel-trace stats of the
# same command, full-size parity with the reference (BASELINE configs 2-5).  (PMC traffic and SQ counters: the scan kernels
# K1 / K2 are those of gpu_r2_final.sh; K3's per-hit confirm moved into a lambda, same instructions.)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( time timeout 900 python bench.py ) > gpurun_out/z_bench.json 2> gpurun_out/z_bench.err
tail -3 gpurun_out/z_bench.err
python - <<'PY'
import json
r = json.loads(open('gpurun_out/z_bench.json').read().strip().splitlines()[-1])
print(r['value'], r['roofline']['frac'], {k: (v['frac'], v['kernel_ms']) for k, v in r['kernels'].items()}, {k: r['e2e'].get(k) for k in ('value', 'frac', 'scan_phase_GBps', 'scan_phase_frac', 'wall_s')}, r.get('cpu_baseline', {}).get('value'))
PY
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/z_prof -- python $R/bench.py --no-e2e --no-cpu-baseline > $R/gpurun_out/z_prof.log 2>&1
cd $R; f=$(find gpurun_out/z_prof -name "*kernel_stats.csv" | head -1); grep -E "gscan|Name" "$f" | cut -c1-260; cp "$f" gpurun_out/z_prof_kernel_stats.csv
find gpurun_out/z_prof -name "*kernel_trace.csv" -size +1M -delete
timeout 1200 python scripts/fullsize_parity.py --workers 8 2>&1 | tail -1 | tee gpurun_out/z_fullsize_parity.txt | cut -c1-1800
