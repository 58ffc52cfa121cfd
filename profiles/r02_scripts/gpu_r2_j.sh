#!/bin/bash
# Round 2, GPU session J: windows registered and DMA'd in place (GRAB_INGEST=register) against the reader pool.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 500 python scripts/e2e_sweep.py --gib 64 --small-gib 0 --single-gib 8 --blocks 16 --readers 8 --streams 1 \
   --extra-env "GRAB_INGEST=register;GRAB_INGEST=register,GSCAN_NUMA=0" > gpurun_out/j_e2e.jsonl 2> gpurun_out/j_e2e.err
python - <<'PY'
import json
for l in open('gpurun_out/j_e2e.jsonl'):
    r = json.loads(l)
    t = [x for x in r.get('timing', []) if 'grab timing' in x]
    print({k: r[k] for k in r if k not in ('timing',)})
    for x in t[-3:]: print('      ', x)
PY
cd /dev/shm && python - <<'PY'
import numpy as np, os
os.makedirs('/dev/shm/jj', exist_ok=True)
buf = np.full((1<<30)+5000, ord('.'), np.uint8); buf[79::80] = 10; buf[12345:12351] = np.frombuffer(b'NEEDLE', np.uint8); buf[-6:] = np.frombuffer(b'NEEDLE', np.uint8)
buf.tofile('/dev/shm/jj/f')
PY
cd $GRAFT_REPO_ROOT
for m in fd register; do GRAB_INGEST=$m grab_amd/bin/grab -O -l NEEDLE /dev/shm/jj/f | md5sum; GRAB_INGEST=$m grab_amd/bin/grab -L -L -L -O NEEDLE /dev/shm/jj/f | md5sum; done
oracle/grab_oracle -O -l NEEDLE /dev/shm/jj/f | md5sum; oracle/grab_oracle -L -L -L -O NEEDLE /dev/shm/jj/f | md5sum
rm -rf /dev/shm/jj
