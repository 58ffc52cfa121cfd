#!/bin/bash
# Round 5, session L (closing run at the final code): the whole GPU suite, the per-window passes with k_lines' 32-byte steps and LDS tail table, the bench line, rocprofv3
# kernel-trace stats of the same command with the cold launches set apart.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/r5l_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/r5l_pytest.txt
python - <<'PY'
import os, sys
sys.path.insert(0, ".")
import torch, bench
from grab_amd import synth
bench.interleave_page_placement()
dev = torch.device("cuda", 0)
for i in range(128):
    sub = "/dev/shm/c8/d%02d" % (i % 16)
    os.makedirs(sub, exist_ok=True)
    synth.torch_text(64 << 20, i, dev).cpu().numpy().tofile(sub + "/f%04d.txt" % i)
PY
G=$R/grab_amd/bin/grab
{
for flags in "-O" "-O -l"; do
  echo "== grab -n 8 -r $flags IDENT over 8 GiB: kernels per 64 MiB window"
  rm -rf /tmp/prof_w; cd /tmp
  GRAB_NORMAL_EXIT=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_w --output-format csv -- $G -n 8 -r $flags '[A-Za-z_][A-Za-z0-9_]{15,}' /dev/shm/c8 > /dev/null 2>/tmp/prof_w.err
  cd $R
  f=$(find /tmp/prof_w -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("  %-70s calls %6s  avg %9.1f us  total %8.2f ms  %5s %%" % (r["Name"].replace("(anonymous namespace)::", "").replace("void gscan::", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
done
} 2>&1 | tee gpurun_out/r5l_per_window_passes.txt
rm -rf /dev/shm/c8
echo "== bench =="
( time timeout 1200 python bench.py ) > gpurun_out/r5l_bench.json 2> gpurun_out/r5l_bench.err
tail -4 gpurun_out/r5l_bench.err
python - <<'PY'
import json
r = json.loads(open('gpurun_out/r5l_bench.json').read().strip().splitlines()[-1])
print("value", r['value'], "check", r['check'], "roofline", {k: r['roofline'][k] for k in ('frac', 'kernel_ms', 'traffic')})
print({k: (v['frac'], v['frac_min'], v['frac_max'], v['check'], v['kernel_ms'], v['traffic'], v.get('implied_sclk_ghz'), v.get('clocks')) for k, v in r['kernels'].items()})
print('clocks', r.get('clocks'), 'n8', json.dumps(r.get('n8_model'))[:3000])
for k in ("e2e", "cpu_baseline", "e2e_cfg3", "e2e_cfg1", "e2e_cfg5", "e2e_cfg4"):
    v = r.get(k) or {}
    print(k, {x: v.get(x) for x in ("value", "wall_s", "startup_s", "exit_s", "fixed_s", "scan_phase_GBps", "frac", "lines", "lines_ok", "digest", "back_to_back_wall_s", "vs_cpu_baseline", "cores", "same_as_reference", "at_16GiB", "at_8GiB", "error")}, (v.get("cpu_baseline") or {}).get("value"))
PY
echo "== rocprofv3 kernel-trace stats of the bench command (all three kernels in one process) =="
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5l_prof -- python $R/bench.py --no-e2e --no-cpu-baseline --no-live-traffic > $R/gpurun_out/r5l_prof.log 2>&1
cd $R; f=$(find gpurun_out/r5l_prof -name "*kernel_stats.csv" | head -1); grep -E "gscan|Name" "$f" | cut -c1-260; cp "$f" gpurun_out/r5l_prof_kernel_stats.csv
t=$(find gpurun_out/r5l_prof -name "*kernel_trace.csv" | head -1); python3 - "$t" <<'PY' | tee gpurun_out/r5l_prof_kernel_stats_warm.txt
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
rm -rf gpurun_out/r5l_prof

