#!/bin/bash
# Round 3, session A: the lane-table K2 kernel (variant 38) against the pair-table / general forms (variant 6): parity of
# the new kernel and of the device's match ends, then the kernel sweep, then SQ counters of the new kernel, then cfg3 end to
# end with and without the device's match ends.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest: engine-level parity (all variants incl. 38, libpcre checks, lane run programs, match ends) =="
timeout 1200 python -m pytest tests/test_gpu_engine.py -m gpu -x -q --durations=8 2>&1 | tail -25 | tee gpurun_out/a_pytest_engine.txt
echo "== pytest: CLI text-free walk A/B + multichunk + tree differential =="
timeout 900 python -m pytest tests/test_gpu_filegrep.py -m gpu -x -q -k "without_the_text or multichunk or tree_differential or line_pass" 2>&1 | tail -8 | tee gpurun_out/a_pytest_cli.txt
SW=$R/grab_amd/bin/gscan_sweep
echo "== kernel sweep, 16 GiB: variant 6 (pair / general form) vs 38 (lane form) =="
for P in '[A-Za-z_][A-Za-z0-9_]{15,}' '[0-9]{16}' '[a-z][0-9][A-Z]{3}' '[a-z]{3}[0-9][A-Z]{2}[a-z_]{6}' '[a-z]{2,5}' '[0-9a-f]{8}[g-z]'; do
  timeout 300 $SW --gib 16 --iters 6 --variants 6,38 --bpc 0 --pattern "$P" 2>&1 | grep -E "^variant|^#"
done | tee gpurun_out/a_sweep.txt
echo "== SQ counters of the lane kernel (4 GiB, counters only) =="
run() { # name, pmc list, sweep args...
  name=$1; pmc=$2; shift 2
  cd /tmp && timeout 300 rocprofv3 --pmc $pmc -d $R/gpurun_out/a_sq_$name --output-format csv -- $SW "$@" > $R/gpurun_out/a_sq_$name.log 2>&1
  cd $R; f=$(find gpurun_out/a_sq_$name -name "*counter_collection.csv" | head -1)
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
run k2lane_a "$C1" $A --pattern '[A-Za-z_][A-Za-z0-9_]{15,}'
run k2lane_b "$C2" $A --pattern '[A-Za-z_][A-Za-z0-9_]{15,}'
run k2lane4_a "$C1" $A --pattern '[a-z][0-9][A-Z]{3}'
run k2lane4_b "$C2" $A --pattern '[a-z][0-9][A-Z]{3}'
} 2>&1 | tee gpurun_out/a_sq_counters.txt
find gpurun_out -name "*counter_collection.csv" -size +1M -delete
echo "== cfg3 end to end, 16 GiB: match ends from the device vs the host walk over the text =="
python - <<'PY' > gpurun_out/a_cfg3_e2e.txt 2>&1
import os, subprocess, sys, time, shutil, hashlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
import e2e_sweep
from grab_amd import bin_path
d = "/dev/shm/r3a_cfg3"
os.makedirs(d)
e2e_sweep.gen_files(d, 256, 64 << 20, 1)
ident = "[A-Za-z_][A-Za-z0-9_]{15,}"
for env_extra, label in (({}, "device ends"), ({"GRAB_NO_ENDS": "1"}, "host walk")):
    for n in (8, 16):
        best = None
        for rep in range(3):
            t0 = time.monotonic()
            r = subprocess.run([bin_path(), "-n", str(n), "-r", "-O", "-l", ident, d], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, GRAB_TIMING="1", **env_extra))
            dt = time.monotonic() - t0
            if best is None or dt < best[0]: best = (dt, r.stderr.decode())
        lines = [l for l in best[1].splitlines() if "device 0:" in l][:2] + [l for l in best[1].splitlines() if "workers joined" in l or "runtime up" in l]
        print("## cfg3 16 GiB -n %d (%s): wall %.3f s = %.2f GB/s" % (n, label, best[0], (16 << 30) / best[0] / 1e9)); print("\n".join(lines))
# same bytes either way (sorted: -n output order is arbitrary)
outs = []
for env_extra in ({}, {"GRAB_NO_ENDS": "1"}):
    p = "/dev/shm/r3a_out.txt"
    with open(p, "wb") as o:
        subprocess.run([bin_path(), "-n", "8", "-r", "-O", "-l", ident, d], stdout=o, env=dict(os.environ, **env_extra))
    r = subprocess.run("LC_ALL=C sort %s | md5sum; wc -l < %s" % (p, p), shell=True, capture_output=True, text=True)
    outs.append(r.stdout.split())
    os.unlink(p)
print("sorted md5 / lines:", outs, "same" if outs[0] == outs[1] else "DIFFERENT")
shutil.rmtree(d)
PY
cat gpurun_out/a_cfg3_e2e.txt
