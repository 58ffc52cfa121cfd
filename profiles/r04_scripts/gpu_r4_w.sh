#!/bin/bash
# Round 4, session W: the last check at the round's final code (engine tests, a slice of the CLI tests, smoke) and the SQ
# counters of the three bench kernels re-taken (counters only, gscan_sweep, 4 GiB): the VALU-issue figures DESIGN.md 4 argues from.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_integration.py -m gpu -q 2>&1 | tail -2 | tee gpurun_out/w_pytest.txt
timeout 600 python -m pytest tests/test_gpu_filegrep.py -m gpu -q -k "reader_pool or errors_surface or tree_differential or offsets_without" 2>&1 | tail -2 | tee -a gpurun_out/w_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/w_pytest.txt
SW=$R/grab_amd/bin/gscan_sweep
run() { # name, pmc list, sweep args...
  name=$1; pmc=$2; shift 2
  cd /tmp && timeout 300 rocprofv3 --pmc $pmc -d $R/gpurun_out/w_sq_$name --output-format csv -- $SW "$@" > $R/gpurun_out/w_sq_$name.log 2>&1
  cd $R; f=$(find gpurun_out/w_sq_$name -name "*counter_collection.csv" | head -1)
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
run k1_b "$C2" $A --pattern 'foobardoesnotexist'
run k2lane_a "$C1" $A --pattern '[A-Za-z_][A-Za-z0-9_]{15,}'
run k2lane_b "$C2" $A --pattern '[A-Za-z_][A-Za-z0-9_]{15,}'
run k2lane_digits_a "$C1" $A --pattern '[0-9]{16}'
run k3_a "$C1" $A --pattern 'foobardoesnotexist|Linus|555-1234'
run k3_b "$C2" $A --pattern 'foobardoesnotexist|Linus|555-1234'
} 2>&1 | tee gpurun_out/w_sq_counters.txt
rm -rf gpurun_out/w_sq_k*
