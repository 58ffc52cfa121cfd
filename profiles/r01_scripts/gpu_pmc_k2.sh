#!/bin/bash
# PMC diagnostic of the scan kernels (counters only: no trace domains alongside --pmc).
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
SW=$GRAFT_REPO_ROOT/grab_amd/bin/gscan_sweep
cd /tmp
run() { # name, pmc list, sweep args...
  name=$1; pmc=$2; shift 2
  timeout 300 rocprofv3 --pmc $pmc -d $OUT/$name --output-format csv -- $SW "$@" > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  echo "== $name ($pmc)"; python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "scan" not in k and "probe" not in k: continue
    print(" ", k, {c: round(v / n[(k, c)]) for c, v in d.items()})
PY
}
ID='[A-Za-z_][A-Za-z0-9_]{15,}'
A="--gib 4 --iters 2 --variants 4 --bpc 0"
run k2_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" $A --pattern "$ID"
run k2_b "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" $A --pattern "$ID"
run k2n_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" $A --pattern '[0-9]{16}'
run k2n_b "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" $A --pattern '[0-9]{16}'
run k1_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" --gib 4 --iters 2 --variants 6 --bpc 0
run k1_fetch "FETCH_SIZE" --gib 4 --iters 2 --variants 6 --bpc 0
run k1_write "WRITE_SIZE" --gib 4 --iters 2 --variants 6 --bpc 0
run k2_fetch "FETCH_SIZE" $A --pattern "$ID"
run k2_write "WRITE_SIZE" $A --pattern "$ID"
echo "== plain timing of the no-match K2 pattern"; $SW --gib 8 --iters 4 --variants 4,6 --bpc 0 --pattern '[0-9]{16}' | tail -3
