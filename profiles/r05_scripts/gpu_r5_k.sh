#!/bin/bash
# Round 5, session K: k_lines with 32-byte steps in its newline searches and its gather copy (one wave per descriptor again):
# the line tests, the CLI suite, the per-window passes.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
{
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -q -x 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_filegrep.py tests/test_integration.py -m gpu -q -x 2>&1 | tail -3
} | tee gpurun_out/r5k_pytest.txt
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
for flags in "-O" ""; do
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
echo "== end to end, 8 GiB, -n 8 -r -O IDENT (lines printed), three runs"
for k in 1 2 3; do sleep 0.5; /usr/bin/time -f "%e s" $G -n 8 -r -O '[A-Za-z_][A-Za-z0-9_]{15,}' /dev/shm/c8 > /dev/null; done
} 2>&1 | tee gpurun_out/r5k_per_window_passes.txt
rm -rf /dev/shm/c8
