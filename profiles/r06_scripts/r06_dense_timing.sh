# Round 6, session C: the dense patterns over 8 GiB in /dev/shm: where the time goes (GRAB_TIMING /
# GSCAN_TIMING) and the per-kernel profile of one of them.  Writes gpurun_out/r06_c_*.
D=/dev/shm/r06d; mkdir -p $D
R=$PWD; G=$R/grab_amd/bin/grab; O=$R/gpurun_out/r06_c_dense_timing.txt; : > $O
python - <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from grab_amd import synth
dev = torch.device('cuda', 0)
for i in range(128):
    synth.torch_text(64 << 20, i, dev).cpu().numpy().tofile('/dev/shm/r06d/f%04d.txt' % i)
PY
for p in '\b[A-Za-z_]\w*\s*\(' '\b[a-z]{3,}\b' '\([^()]*\)' '\s\w{8,}\s'; do
  echo "=== $p" >> $O
  $G -n 8 -r -O -l "$p" $D > /dev/shm/r06d.out   # warm
  for k in 1 2; do
    ( time env GRAB_TIMING=1 GSCAN_TIMING=1 $G -n 8 -r -O -l "$p" $D 2>/dev/shm/r06d.err > /dev/shm/r06d.out ) 2>&1 | tr '\n' ' ' >> $O; echo >> $O
  done
  grep -v "gscan_open" /dev/shm/r06d.err | cut -c1-900 >> $O
  wc -l < /dev/shm/r06d.out >> $O
done
cd /tmp && export TMPDIR=/tmp
for p in '\b[A-Za-z_]\w*\s*\(' '\([^()]*\)'; do
  rm -rf /tmp/prof_w
  GRAB_NORMAL_EXIT=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_w --output-format csv -- $G -n 8 -r -O -l "$p" $D > /dev/null 2>/tmp/prof_w.err || tail -5 /tmp/prof_w.err
  f=$(find /tmp/prof_w -name '*kernel_stats.csv' | head -1)
  echo "=== rocprofv3 --kernel-trace --stats: grab -n 8 -r -O -l '$p' over 8 GiB (128 x 64 MiB windows)" >> $R/gpurun_out/r06_c_dense_kernel_stats.txt
  cut -d, -f1-8 "$f" | head -14 >> $R/gpurun_out/r06_c_dense_kernel_stats.txt
done
cd $R
rm -rf $D /dev/shm/r06d.out /dev/shm/r06d.err
tail -80 $O; cat gpurun_out/r06_c_dense_kernel_stats.txt | cut -c1-200
