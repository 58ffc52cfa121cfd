#!/bin/bash
# Round 5, session U: bench.py's IN-PROCESS path after the refactor (what runs under torch.distributed, N > 1: kernel blocks and
# end-to-end blocks in one process, the corpus written out of the HBM arena) at a reduced size, and as one rank of a
# torch.distributed.run launch (N = 1 worker, the N > 1 code path: WORLD_SIZE is set).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
{
echo "== python bench.py --in-process (reduced)"
timeout 600 python bench.py --in-process --files 128 --e2e-gib 8 --steps 3 --warmup 1 --no-live-traffic 2>/tmp/u1.err | python3 -c "
import sys, json
out = sys.stdin.read().strip().splitlines()
print('stdout lines', len(out))
r = json.loads(out[-1])
print({k: r.get(k) for k in ('value', 'n_gpus', 'steps', 'check', 'e2e_context')}, r['roofline']['frac'])
for k in ('e2e', 'e2e_cfg3', 'e2e_cfg1', 'e2e_cfg5', 'e2e_cfg4', 'n8_model'):
    v = r.get(k) or {}
    print(k, {x: v.get(x) for x in ('value', 'wall_s', 'lines_ok', 'same_as_reference', 'error', 'F8_minus_F1_measured_s')})
"
tail -3 /tmp/u1.err
echo "== torch.distributed.run --nproc-per-node 1 bench.py --gpus 1 (WORLD_SIZE set: the multi-GPU code path)"
WORLD_SIZE_HINT=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --in-process --files 128 --e2e-gib 8 --steps 3 --warmup 1 --no-live-traffic --no-e2e-extra 2>/tmp/u2.err | python3 -c "
import sys, json
out = [l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')]
print('json lines', len(out))
r = json.loads(out[-1])
print({k: r.get(k) for k in ('value', 'n_gpus', 'steps', 'check', 'scaling')}, r['roofline']['frac'], {x: r['e2e'].get(x) for x in ('value', 'wall_s', 'lines_ok', 'error')})
"
tail -3 /tmp/u2.err
} 2>&1 | tee gpurun_out/r5u_bench_in_process_paths.txt
