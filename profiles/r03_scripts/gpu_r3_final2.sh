#!/bin/bash
# Round 3, the state at the end of the round once more: GPU suite, smoke, bench line (e2e_cfg3 at 64 GiB, the extra kernels
# with the headline's launch counts), rocprofv3 kernel stats of the bench command with the cold launches left out.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/fin2_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/fin2_smoke.txt
( time timeout 900 python bench.py ) > gpurun_out/fin2_bench.json 2> gpurun_out/fin2_bench.err
tail -4 gpurun_out/fin2_bench.err
python - <<'PY'
import json
r = json.loads(open('gpurun_out/fin2_bench.json').read().strip().splitlines()[-1])
print("value", r['value'], "roofline", {k: r['roofline'][k] for k in ('frac', 'kernel_ms', 'traffic')})
print({k: (v['frac'], v['kernel_ms'], v['steps'], v['traffic']) for k, v in r['kernels'].items()})
for k in ("e2e", "cpu_baseline", "e2e_cfg3", "e2e_cfg5"):
    v = r.get(k) or {}
    print(k, {x: v.get(x) for x in ("value", "wall_s", "detached_GBps", "scan_phase_GBps", "lines", "lines_ok", "vs_cpu_baseline", "cores", "at_16GiB", "same_as_reference", "error")}, (v.get("parity_subset") or {}).get("same"), (v.get("cpu_baseline") or {}).get("value"))
PY
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fin2_prof -- python $R/bench.py --no-e2e --no-cpu-baseline --no-live-traffic > $R/gpurun_out/fin2_prof.log 2>&1
cd $R; f=$(find gpurun_out/fin2_prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/fin2_prof_kernel_stats.csv
t=$(find gpurun_out/fin2_prof -name "*kernel_trace.csv" | head -1); python3 - "$t" <<'PY' | tee gpurun_out/fin2_prof_kernel_stats_warm.txt
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
find gpurun_out -name "*kernel_trace.csv" -size +1M -delete
