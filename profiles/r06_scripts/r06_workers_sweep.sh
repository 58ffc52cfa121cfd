# Round 6, session G: the dense patterns over 16 GiB, output to /dev/null, workers 8 / 12 / 16 / 24; GRAB_TIMING of one run each
D=/dev/shm/r06g; mkdir -p $D
G=$PWD/grab_amd/bin/grab; O=$PWD/gpurun_out/r06_g_workers_sweep.txt; : > $O
python - <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from grab_amd import synth
dev = torch.device('cuda', 0)
for i in range(256):
    synth.torch_text(64 << 20, i, dev).cpu().numpy().tofile('/dev/shm/r06g/f%04d.txt' % i)
PY
for p in '\b[A-Z][a-z]+\b' '\b[a-z]{3,}\b' '\b[A-Za-z_]\w*\s*\(' '\([^()]*\)' '[A-Za-z_][A-Za-z0-9_]{15,}'; do
  $G -n 8 -r -O -l "$p" $D > /dev/null
  for w in 8 12 16 24; do
    best=99
    for k in 1 2 3; do
      sleep 0.5
      t0=$(date +%s.%N); $G -n $w -r -O -l "$p" $D > /dev/null; t1=$(date +%s.%N)
      dt=$(echo "$t1 - $t0" | bc); best=$(echo "if ($dt < $best) $dt else $best" | bc)
    done
    echo "$p  -n $w  best of 3: $best s  $(echo "scale=2; 17.179869184 / $best" | bc) GB/s" >> $O
  done
  GRAB_TIMING=1 $G -n 8 -r -O -l "$p" $D 2>&1 >/dev/null | grep "device 0: files 32\|workers joined" | head -3 | cut -c1-330 >> $O
done
rm -rf $D
cat $O
