#!/bin/bash
# Round 3, session AH: is the identifier scan in bench.py slower than in the closing run because of the box or because of a
# commit?  The library of the closing run's commit (libold) and HEAD's, swapped under the same bench command, same box, A B A B.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/grab_amd/lib/libgscan.so /tmp/libnew.so
{
for L in new old new old; do
  if [ $L = old ]; then cp $R/grab_amd/libold/libgscan.so $R/grab_amd/lib/libgscan.so; else cp /tmp/libnew.so $R/grab_amd/lib/libgscan.so; fi
  python bench.py --no-e2e --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$L', r['roofline']['frac'], {k:(v['frac'], v['kernel_ms'], (v.get('sustained') or {}).get('frac')) for k,v in r['kernels'].items()})"
done
cp /tmp/libnew.so $R/grab_amd/lib/libgscan.so
} | tee gpurun_out/ah_old_vs_new_lib_bench.txt
