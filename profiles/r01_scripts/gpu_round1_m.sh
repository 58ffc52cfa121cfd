#!/bin/bash
# Round-1 evidence refresh: headline bench lines (cfg2 = default, cfg3), rocprofv3 kernel-trace stats of the same commands,
# PMC traffic of the same commands in separate passes (counters only), K3 sweep.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== bench cfg2 (default) =="
timeout 900 python bench.py 2>&1 | tail -1 | tee gpurun_out/m_bench_cfg2.json
echo "== bench cfg3 =="
timeout 900 python bench.py --config cfg3 --steps 5 --warmup 1 2>&1 | tail -1 | tee gpurun_out/m_bench_cfg3.json
echo "== rocprofv3 kernel-trace stats: python bench.py (default workload) =="
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/m_prof_cfg2 -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/m_prof_cfg2.log 2>&1
tail -1 $R/gpurun_out/m_prof_cfg2.log | cut -c1-400
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/m_prof_cfg3 -- python $R/bench.py --config cfg3 --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/m_prof_cfg3.log 2>&1
tail -1 $R/gpurun_out/m_prof_cfg3.log | cut -c1-400
cd $R; for d in m_prof_cfg2 m_prof_cfg3; do f=$(find gpurun_out/$d -name "*kernel_stats.csv" | head -1); echo "-- $f"; head -4 "$f" | cut -c1-200; cp "$f" gpurun_out/${d}_kernel_stats.csv; done
echo "== PMC traffic of the same command (separate passes) =="
for cfg in cfg2 cfg3; do for ctr in FETCH_SIZE WRITE_SIZE; do
  extra=""; [ $cfg = cfg3 ] && extra="--config cfg3"
  cd /tmp && timeout 900 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/m_pmc_${ctr}_$cfg -- python $R/bench.py $extra --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  cd $R; f=$(find gpurun_out/m_pmc_${ctr}_$cfg -name "*counter_collection.csv" | head -1); python3 - "$f" "$cfg $ctr" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "scan" in r["Kernel_Name"]: agg[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in agg.items(): print("PMC", sys.argv[2], k[0], "launches", len(v), "mean", sum(v)/len(v))
PY
done; done | tee gpurun_out/m_pmc_summary.txt
find gpurun_out -name "*kernel_trace.csv" -size +1M -delete; find gpurun_out -name "*counter_collection.csv" -size +1M -delete
