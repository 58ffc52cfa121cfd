#!/bin/bash
# Round-1 final evidence: bench lines (cfg2 = default, cfg3, alt), rocprofv3 kernel-trace stats of the same commands,
# PMC traffic of the same commands in separate passes (counters only).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in cfg2 cfg3 alt; do
  extra=""; [ $cfg != cfg2 ] && extra="--config $cfg --steps 5 --warmup 1"
  echo "== bench $cfg =="
  timeout 900 python bench.py $extra 2>&1 | tail -1 | tee gpurun_out/u_bench_$cfg.json
  echo "== rocprofv3 kernel-trace stats: bench $cfg =="
  cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/u_prof_$cfg -- python $R/bench.py $extra --no-cpu-baseline > $R/gpurun_out/u_prof_$cfg.log 2>&1
  cd $R; f=$(find gpurun_out/u_prof_$cfg -name "*kernel_stats.csv" | head -1); grep "gscan" "$f" | cut -c1-220; cp "$f" gpurun_out/u_prof_${cfg}_kernel_stats.csv
done
echo "== PMC traffic (separate passes) =="
for cfg in cfg2 cfg3 alt; do for ctr in FETCH_SIZE WRITE_SIZE; do
  extra=""; [ $cfg != cfg2 ] && extra="--config $cfg"
  cd /tmp && timeout 900 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/u_pmc_${ctr}_$cfg -- python $R/bench.py $extra --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  cd $R; f=$(find gpurun_out/u_pmc_${ctr}_$cfg -name "*counter_collection.csv" | head -1); python3 - "$f" "$cfg $ctr" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "scan" in r["Kernel_Name"]: agg[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in agg.items(): print("PMC", sys.argv[2], k[0], "launches", len(v), "mean", sum(v)/len(v))
PY
done; done | tee gpurun_out/u_pmc_summary.txt
find gpurun_out -name "*kernel_trace.csv" -size +1M -delete; find gpurun_out -name "*counter_collection.csv" -size +1M -delete
