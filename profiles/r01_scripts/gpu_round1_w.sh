#!/bin/bash
# K2 two-class form: A/B of the experiment switches (GSCAN_K2_OPT: 1 bank-spreading table, 2 branch-free run program,
# 4 two runs decoded once), parity of each form, PMC of the baseline and of the full form.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SW=$R/grab_amd/bin/gscan_sweep
ID='[A-Za-z_][A-Za-z0-9_]{15,}'
{
for pat in "$ID" '[0-9]{16}' '[a-z]{2,5}' '[a-z]+'; do
  for o in 0 1 2 3 6 7; do
    echo "== opt $o pattern $pat"
    GSCAN_K2_OPT=$o timeout 300 $SW --gib 8 --iters 6 --variants 6 --bpc 0 --pattern "$pat" | tail -1
  done
done
} 2>&1 | tee gpurun_out/w_k2_opt_sweep.txt
echo "== parity under each form"
for o in 1 2 7; do
  GSCAN_K2_OPT=$o timeout 600 python -m pytest tests/test_gpu_engine.py -m gpu -x -q -k "parity or ragged or dense or boundary" 2>&1 | tail -2 | tee -a gpurun_out/w_k2_opt_parity.txt
done
echo "== PMC"
cd /tmp
for o in 0 7; do
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
    tag=$(echo $set | cut -c1-12 | tr ' ' '_')
    GSCAN_K2_OPT=$o timeout 300 rocprofv3 --pmc $set -d $R/gpurun_out/w_pmc_${o}_$tag --output-format csv -- $SW --gib 4 --iters 2 --variants 6 --bpc 0 --pattern "$ID" > /dev/null 2>&1
    f=$(find $R/gpurun_out/w_pmc_${o}_$tag -name "*counter_collection.csv" | head -1)
    python3 - "$f" "opt $o" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:70]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "k2_" not in k: continue
    print("PMC", sys.argv[2], k, {c: round(v / n[(k, c)]) for c, v in d.items()})
PY
  done
done 2>&1 | tee $R/gpurun_out/w_k2_pmc.txt
cd $R; find gpurun_out -name "*counter_collection.csv" -size +1M -delete
