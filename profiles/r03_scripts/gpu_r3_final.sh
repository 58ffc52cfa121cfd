#!/bin/bash
# Round 3, closing run: the GPU suite, smoke, the bench line, rocprofv3 kernel-trace stats of the same command (with the first,
# cold launch of every kernel dropped from a second summary), HBM traffic and SQ counters of the final kernels in separate --pmc
# passes (counters only), full-size parity (BASELINE configs 2-5, cfg3 at its full 64 GiB).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest -m gpu =="
timeout 2400 python -m pytest tests -m gpu -q --durations=5 2>&1 | tail -12 | tee gpurun_out/fin_pytest.txt
echo "== smoke =="
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/fin_smoke.txt
echo "== bench =="
( time timeout 900 python bench.py ) > gpurun_out/fin_bench.json 2> gpurun_out/fin_bench.err
tail -4 gpurun_out/fin_bench.err
python - <<'PY'
import json
r = json.loads(open('gpurun_out/fin_bench.json').read().strip().splitlines()[-1])
print("value", r['value'], "roofline", {k: r['roofline'][k] for k in ('frac', 'kernel_ms', 'traffic')})
print({k: (v['frac'], v['kernel_ms'], v['traffic']) for k, v in r['kernels'].items()})
for k in ("e2e", "cpu_baseline", "e2e_cfg3", "e2e_cfg5"):
    v = r.get(k) or {}
    print(k, {x: v.get(x) for x in ("value", "wall_s", "detached_GBps", "scan_phase_GBps", "frac", "lines", "lines_ok", "vs_cpu_baseline", "cores", "GBps_by_threads", "parity_subset", "same_as_reference", "error")}, (v.get("cpu_baseline") or {}).get("value"), (v.get("cpu_baseline") or {}).get("GBps_by_threads"))
PY
echo "== rocprofv3 kernel-trace stats of the bench command (all three kernels in one process) =="
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fin_prof -- python $R/bench.py --no-e2e --no-cpu-baseline --no-live-traffic > $R/gpurun_out/fin_prof.log 2>&1
cd $R; f=$(find gpurun_out/fin_prof -name "*kernel_stats.csv" | head -1); grep -E "gscan|Name" "$f" | cut -c1-260; cp "$f" gpurun_out/fin_prof_kernel_stats.csv
t=$(find gpurun_out/fin_prof -name "*kernel_trace.csv" | head -1); python3 - "$t" <<'PY' | tee gpurun_out/fin_prof_kernel_stats_warm.txt
# the same trace with the first launch of every kernel (cold: code object load, first touch of the tables) left out
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
print("kernel | launches | first launch us | warm launches: mean us, min us, max us")
for k, v in sorted(d.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    if "gscan" not in k: continue
    v.sort(); dur = [(e - s) / 1e3 for s, e in v]; w = dur[1:] or dur
    print(f"{k[:110]} | {len(dur)} | {dur[0]:.1f} | {sum(w)/len(w):.1f} {min(w):.1f} {max(w):.1f}")
PY
echo "== PMC: HBM traffic (separate passes) =="
for ctr in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 900 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/fin_pmc_$ctr -- python $R/bench.py --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --no-live-traffic > /dev/null 2>&1
  cd $R; f=$(find gpurun_out/fin_pmc_$ctr -name "*counter_collection.csv" | head -1); python3 - "$f" "$ctr" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "gscan" in r["Kernel_Name"]: agg[(r["Kernel_Name"][:90], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in agg.items(): print("PMC", sys.argv[2], k[0], "launches", len(v), "mean", sum(v)/len(v))
PY
done | tee gpurun_out/fin_pmc_traffic.txt
echo "== PMC: SQ counters of the final kernels (gscan_sweep, 4 GiB, counters only) =="
SW=$R/grab_amd/bin/gscan_sweep
run() { # name, pmc list, sweep args...
  name=$1; pmc=$2; shift 2
  cd /tmp && timeout 300 rocprofv3 --pmc $pmc -d $R/gpurun_out/fin_sq_$name --output-format csv -- $SW "$@" > $R/gpurun_out/fin_sq_$name.log 2>&1
  cd $R; f=$(find gpurun_out/fin_sq_$name -name "*counter_collection.csv" | head -1)
  echo "== $name ($pmc)"; python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:80]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "scan" not in k: continue
    print(" ", k, {c: round(v / n[(k, c)]) for c, v in d.items()})
PY
}
A="--gib 4 --iters 2 --variants 38 --bpc 0"
C1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
C2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD"
{
run k1_a "$C1" $A --pattern 'foobardoesnotexist'
run k2lane_a "$C1" $A --pattern '[A-Za-z_][A-Za-z0-9_]{15,}'
run k2lane_b "$C2" $A --pattern '[A-Za-z_][A-Za-z0-9_]{15,}'
run k2lane_digits_a "$C1" $A --pattern '[0-9]{16}'
run k2lane_digits_b "$C2" $A --pattern '[0-9]{16}'
run k3_a "$C1" $A --pattern 'foobardoesnotexist|Linus|555-1234'
run k3_b "$C2" $A --pattern 'foobardoesnotexist|Linus|555-1234'
run k2lane3_a "$C1" $A --pattern '[a-z][0-9][A-Z]{3}'
run k2flat_a "$C1" $A --pattern '[0-9]+\.[0-9]+'
run k2flat_b "$C2" $A --pattern '[0-9]+\.[0-9]+'
} 2>&1 | tee gpurun_out/fin_sq_counters.txt
find gpurun_out -name "*kernel_trace.csv" -size +1M -delete; find gpurun_out -name "*counter_collection.csv" -size +1M -delete
echo "== full-size parity (BASELINE configs 2-5; cfg3 at 1024 files = 64 GiB) =="
timeout 1500 python scripts/fullsize_parity.py --workers 8 --cfg3-files 1024 2>&1 | tail -6 | tee gpurun_out/fin_fullsize_parity.txt
