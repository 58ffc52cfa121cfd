#!/bin/bash
# Round 3, session K: kernel trace of the drop-in binary in line mode with the device's line pass: how long k_lines takes.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
import e2e_sweep
os.makedirs("/dev/shm/r3k")
e2e_sweep.gen_files("/dev/shm/r3k", 64, 64 << 20, 1)
PY
cd /tmp && GRAB_NORMAL_EXIT=1 GRAB_CLOSE=1 GRAB_LINE_PASS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/k_prof -- $R/grab_amd/bin/grab -n 8 -r -O '[A-Za-z_][A-Za-z0-9_]{15,}' /dev/shm/r3k > /dev/null 2> $R/gpurun_out/k_prof.err
cd $R; f=$(find gpurun_out/k_prof -name "*kernel_stats.csv" | head -1); cut -c1-200 "$f" | head -8 | tee gpurun_out/k_kernel_stats.txt
cd /tmp && GRAB_NORMAL_EXIT=1 GRAB_CLOSE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/k_prof2 -- $R/grab_amd/bin/grab -n 8 -r -O -l '[A-Za-z_][A-Za-z0-9_]{15,}' /dev/shm/r3k > /dev/null 2>> $R/gpurun_out/k_prof.err
cd $R; f=$(find gpurun_out/k_prof2 -name "*kernel_stats.csv" | head -1); cut -c1-200 "$f" | head -8 | tee -a gpurun_out/k_kernel_stats.txt
find gpurun_out -name "*kernel_trace.csv" -size +1M -delete
rm -rf /dev/shm/r3k
