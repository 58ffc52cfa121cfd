#!/bin/bash
# Round 5, session D: the whole GPU suite at the refactored engine (one stream layout, second stream on demand, read-ahead,
# survivor queue, experiment code removed from the kernels); the per-window passes re-taken (VERDICT r4 task 8): rocprofv3
# --kernel-trace --stats of `grab -n 8 -r -O` and `-O -l` over 16 GiB; the inexact (device VM) patterns end to end against the
# 64-core reference; a random differential campaign; cfg1 with and without the helper that makes the first transfer.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee gpurun_out/r5d_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/r5d_pytest.txt
python - <<'PY'
import os, sys
sys.path.insert(0, ".")
import torch, bench
from grab_amd import synth
bench.interleave_page_placement()
dev = torch.device("cuda", 0)
synth.torch_text(256 << 20, 0, dev).cpu().numpy().tofile("/dev/shm/one256.txt")
for i in range(256):
    sub = "/dev/shm/c16/d%02d" % (i % 16)
    os.makedirs(sub, exist_ok=True)
    synth.torch_text(64 << 20, i, dev).cpu().numpy().tofile(sub + "/f%04d.txt" % i)
PY
G=$R/grab_amd/bin/grab
{
for flags in "-O" "-O -l" ""; do
  echo "== grab -n 8 -r $flags IDENT over 16 GiB: kernels per 64 MiB window"
  rm -rf /tmp/prof_w; cd /tmp
  GRAB_NORMAL_EXIT=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_w --output-format csv -- $G -n 8 -r $flags '[A-Za-z_][A-Za-z0-9_]{15,}' /dev/shm/c16 > /dev/null 2>/tmp/prof_w.err
  cd $R
  f=$(find /tmp/prof_w -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("  %-70s calls %6s  avg %9.1f us  total %8.2f ms  %5s %%" % (r["Name"].replace("(anonymous namespace)::", "").replace("void gscan::", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
done
} 2>&1 | tee gpurun_out/r5d_per_window_passes.txt
{
for p in '(\w)\1{3,}x|foobardoes(?=not)' '[a-z]+\([a-z0-9, ]*\);' 'a+b+c'; do
  timeout 300 python scripts/e2e_cli.py --files 128 --file-kib 65536 --pattern "$p" --flags "-O -l" --workers 8 --ref-cores 64 --reps 2 --tag vm
done
} 2>/dev/null | tee gpurun_out/r5d_inexact_e2e.jsonl
timeout 200 python scripts/gpu_random_campaign.py --seed 50505 --seconds 100 2>&1 | tail -2 | tee gpurun_out/r5d_random_campaign.txt
timeout 200 python scripts/gpu_random_campaign.py --seed 60606 --seconds 80 --tree 2>&1 | tail -2 | tee -a gpurun_out/r5d_random_campaign.txt
{
echo "--- cfg1: one 256 MiB file"
python scripts/ab_run.py --sleep 0.5 --reps 8 --bytes $((256 << 20)) --interleave --env "" --env "GSCAN_WARM_COPY=0" --env "GRAB_NO_READ_AHEAD=1" --env "GRAB_NO_READ_AHEAD=1 GSCAN_WARM_COPY=0" -- $G foobardoesnotexist /dev/shm/one256.txt
GSCAN_TIMING=1 GRAB_TIMING=1 $G foobardoesnotexist /dev/shm/one256.txt 2>&1 >/dev/null | grep "gscan_open\|grab timing\] +" | head -20
echo "--- 16 GiB -n 8 -r, warm copy on / off"
python scripts/ab_run.py --sleep 0.5 --reps 3 --bytes $((256 * 67108864)) --interleave --env "" --env "GSCAN_WARM_COPY=0" -- $G -n 8 -r foobardoesnotexist /dev/shm/c16
} 2>&1 | tee gpurun_out/r5d_cfg1.txt
rm -rf /dev/shm/c16 /dev/shm/one256.txt
