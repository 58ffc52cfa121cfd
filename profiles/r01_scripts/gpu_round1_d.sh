#!/bin/bash
# Parity re-check, headline bench lines (both configs, default kernel settings), rocprofv3 kernel
# stats of the same bench command, PMC traffic of the same command (separate passes).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest gpu (engine) =="
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/pytest_engine.txt
echo "== sweep ident =="
timeout 300 grab_amd/bin/gscan_sweep --gib 16 --iters 6 --variants 0,4,6 --bpc 0 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' 2>&1 | tee gpurun_out/sweep_k2_ident.txt
echo "== bench cfg2 (default) =="
timeout 900 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_cfg2.json
echo "== bench cfg3 =="
timeout 900 python bench.py --config cfg3 --steps 5 --warmup 1 2>&1 | tail -1 | tee gpurun_out/bench_cfg3.json
echo "== rocprofv3 kernel-trace stats: python bench.py (default workload) =="
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg2 -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_cfg2.log 2>&1
tail -1 $R/gpurun_out/prof_cfg2.log | cut -c1-600
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg3 -- python $R/bench.py --config cfg3 --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_cfg3.log 2>&1
tail -1 $R/gpurun_out/prof_cfg3.log | cut -c1-600
cd $R; for d in prof_cfg2 prof_cfg3; do f=$(find gpurun_out/$d -name "*kernel_stats.csv" | head -1); echo "-- $f"; head -6 "$f"; done
echo "== PMC traffic of the same command (separate passes) =="
cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_cfg2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_cfg2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_cfg3 -- python $R/bench.py --config cfg3 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_cfg3 -- python $R/bench.py --config cfg3 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $R; for d in pmc_fetch_cfg2 pmc_write_cfg2 pmc_fetch_cfg3 pmc_write_cfg3; do f=$(find gpurun_out/$d -name "*counter_collection.csv" | head -1); python3 - "$f" "$d" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "scan" in r["Kernel_Name"]: agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in agg.items(): print(sys.argv[2], k, "launches", len(v), "mean", sum(v)/len(v))
PY
done
# keep only the small csv summaries (the per-dispatch traces are big)
find gpurun_out -name "*kernel_trace.csv" -size +2M -delete
