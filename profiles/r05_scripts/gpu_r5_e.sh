#!/bin/bash
# Round 5, session E: the bench line at the round-5 code (digest parity, in-flight clocks, back-to-back, n8_model), rocprofv3
# kernel-trace stats of the same command with the cold launches set apart.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
{
timeout 900 python -m pytest "tests/test_gpu_geometry.py::test_queue_over_eight_device_indices" -m gpu -q -x 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_filegrep.py -m gpu -q -x -k "line or ends or extents or offsets_without" 2>&1 | tail -3
} | tee gpurun_out/r5e_pytest_slice.txt
echo "== bench =="
( time timeout 1200 python bench.py ) > gpurun_out/r5e_bench.json 2> gpurun_out/r5e_bench.err
tail -4 gpurun_out/r5e_bench.err
python - <<'PY'
import json
r = json.loads(open('gpurun_out/r5e_bench.json').read().strip().splitlines()[-1])
print("value", r['value'], "check", r['check'], "roofline", {k: r['roofline'][k] for k in ('frac', 'kernel_ms', 'traffic')})
print({k: (v['frac'], v['frac_min'], v['frac_max'], v['check'], v['kernel_ms'], v['traffic'], v.get('implied_sclk_ghz'), v.get('clocks')) for k, v in r['kernels'].items()})
print('clocks', r.get('clocks'), 'n8', json.dumps(r.get('n8_model'))[:3000])
for k in ("e2e", "cpu_baseline", "e2e_cfg3", "e2e_cfg1", "e2e_cfg5", "e2e_cfg4"):
    v = r.get(k) or {}
    print(k, {x: v.get(x) for x in ("value", "wall_s", "startup_s", "exit_s", "fixed_s", "scan_phase_GBps", "frac", "lines", "lines_ok", "digest", "back_to_back_wall_s", "vs_cpu_baseline", "cores", "same_as_reference", "at_16GiB", "at_8GiB", "error")}, (v.get("cpu_baseline") or {}).get("value"))
PY
echo "== rocprofv3 kernel-trace stats of the bench command (all three kernels in one process) =="
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5e_prof -- python $R/bench.py --no-e2e --no-cpu-baseline --no-live-traffic > $R/gpurun_out/r5e_prof.log 2>&1
cd $R; f=$(find gpurun_out/r5e_prof -name "*kernel_stats.csv" | head -1); grep -E "gscan|Name" "$f" | cut -c1-260; cp "$f" gpurun_out/r5e_prof_kernel_stats.csv
t=$(find gpurun_out/r5e_prof -name "*kernel_trace.csv" | head -1); python3 - "$t" <<'PY' | tee gpurun_out/r5e_prof_kernel_stats_warm.txt
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
rm -rf gpurun_out/r5e_prof

