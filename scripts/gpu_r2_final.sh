#!/bin/bash
# Round 2, closing run: the GPU suite, the bench line, rocprofv3 kernel-trace stats of the same command, HBM traffic and SQ
# counters of the final kernels in separate --pmc passes (counters only), full-size parity (BASELINE configs 2-5).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -m gpu -q --durations=5 2>&1 | tail -12 | tee gpurun_out/f_pytest.txt
echo "== smoke =="
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/f_smoke.txt
echo "== bench =="
( time timeout 900 python bench.py ) > gpurun_out/f_bench.json 2> gpurun_out/f_bench.err
tail -3 gpurun_out/f_bench.err
python - <<'PY'
import json
r = json.loads(open('gpurun_out/f_bench.json').read().strip().splitlines()[-1])
print(r['value'], r['roofline'], {k: (v['frac'], v['kernel_ms']) for k, v in r['kernels'].items()}, r['e2e'], r.get('cpu_baseline'))
PY
echo "== rocprofv3 kernel-trace stats of the bench command (all three kernels in one process) =="
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/f_prof -- python $R/bench.py --no-e2e --no-cpu-baseline > $R/gpurun_out/f_prof.log 2>&1
cd $R; f=$(find gpurun_out/f_prof -name "*kernel_stats.csv" | head -1); grep -E "gscan|Name" "$f" | cut -c1-260; cp "$f" gpurun_out/f_prof_kernel_stats.csv
echo "== PMC: HBM traffic (separate passes) =="
for ctr in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 900 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/f_pmc_$ctr -- python $R/bench.py --steps 3 --warmup 1 --no-e2e --no-cpu-baseline > /dev/null 2>&1
  cd $R; f=$(find gpurun_out/f_pmc_$ctr -name "*counter_collection.csv" | head -1); python3 - "$f" "$ctr" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "scan" in r["Kernel_Name"]: agg[(r["Kernel_Name"][:90], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in agg.items(): print("PMC", sys.argv[2], k[0], "launches", len(v), "mean", sum(v)/len(v))
PY
done | tee gpurun_out/f_pmc_traffic.txt
echo "== PMC: SQ counters of the final kernels (gscan_sweep, 4 GiB, counters only) =="
SW=$R/grab_amd/bin/gscan_sweep
run() { # name, pmc list, sweep args...
  name=$1; pmc=$2; shift 2
  cd /tmp && timeout 300 rocprofv3 --pmc $pmc -d $R/gpurun_out/f_sq_$name --output-format csv -- $SW "$@" > $R/gpurun_out/f_sq_$name.log 2>&1
  cd $R; f=$(find gpurun_out/f_sq_$name -name "*counter_collection.csv" | head -1)
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
A="--gib 4 --iters 2 --variants 6 --bpc 0"
C1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
C2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD"
{
run k1_a "$C1" $A --pattern 'foobardoesnotexist'
run k1_b "$C2" $A --pattern 'foobardoesnotexist'
run k2pair_a "$C1" $A --pattern '[A-Za-z_][A-Za-z0-9_]{15,}'
run k2pair_b "$C2" $A --pattern '[A-Za-z_][A-Za-z0-9_]{15,}'
run k3_a "$C1" $A --pattern 'foobardoesnotexist|Linus|555-1234'
run k3_b "$C2" $A --pattern 'foobardoesnotexist|Linus|555-1234'
run k2gen_a "$C1" $A --pattern '[a-z][0-9][A-Z]{3}'
run k3vm_a "$C1" --gib 1 --iters 1 --variants 6 --bpc 0 --pattern '(\w)\1{3,}x|foobardoes(?=not)'
} 2>&1 | tee gpurun_out/f_sq_counters.txt
find gpurun_out -name "*kernel_trace.csv" -size +1M -delete; find gpurun_out -name "*counter_collection.csv" -size +1M -delete
echo "== full-size parity (BASELINE configs 2-5) =="
timeout 1200 python scripts/fullsize_parity.py --workers 8 2>&1 | tail -4 | tee gpurun_out/f_fullsize_parity.txt
echo "== inexact patterns end to end (8 GiB) =="
for P in '(\w)\1{3,}x|foobardoes(?=not)' '[a-z]+\([a-z0-9, ]*\);'; do
  timeout 300 python scripts/e2e_cli.py --files 128 --pattern "$P" --flags "-O -l" --workers 8 --tag vm >> gpurun_out/f_vm_e2e.jsonl 2>> gpurun_out/f_vm_e2e.err
done
cat gpurun_out/f_vm_e2e.jsonl | cut -c1-600
