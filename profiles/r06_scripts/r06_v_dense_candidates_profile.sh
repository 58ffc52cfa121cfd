# Round 6, session V: patterns whose START windows list most of the text -- \w+(?=\() (three bytes in four) -- where does a 4 GiB
# run's time go: GRAB_TIMING of one run, rocprofv3 kernel stats.
D=/dev/shm/r06v; mkdir -p $D; R=$PWD; G=$R/grab_amd/bin/grab; O=$R/gpurun_out/r06_v_dense_candidates.txt; : > $O
python - <<'PY'
import sys, torch
sys.path.insert(0, '.')
from grab_amd import synth
for i in range(64):
    synth.torch_text(64 << 20, i, torch.device('cuda', 0)).cpu().numpy().tofile('/dev/shm/r06v/f%04d.txt' % i)
PY
for p in '\w+(?=\()' '\w+\s*=\s*\w+\s*\(' '(?<=\()[^()\n]+(?=\))'; do
  echo "=== $p" >> $O
  $G -n 8 -r -O -l "$p" $D > /dev/null
  for k in 1 2; do ( time $G -n 8 -r -O -l "$p" $D > /dev/null ) 2>&1 | tr '\n' ' ' >> $O; echo >> $O; done
  GRAB_TIMING=1 GSCAN_TIMING=1 $G -n 8 -r -O -l "$p" $D 2>&1 >/dev/null | grep "device 0: files 8 \|context on device\|workers joined" | head -4 | cut -c1-400 >> $O
  cd /tmp; rm -rf /tmp/prof_v
  GRAB_NORMAL_EXIT=1 TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_v --output-format csv -- $G -n 8 -r -O -l "$p" $D > /dev/null 2>/tmp/prof_v.err
  f=$(find /tmp/prof_v -name '*kernel_stats.csv' | head -1); cut -d, -f1-5 "$f" | head -8 | cut -c1-60,150-260 >> $O
  cd $R
done
rm -rf $D; cat $O
