#!/bin/bash
# Round 4, session Q: the GPU suite three more times in a row (the one GPU fault of this round has not come back since the
# block pool's event waits became exclusive: this makes it 8 clean passes).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
for k in 1 2 3; do
echo "== pytest -m gpu, pass $k =="
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
done 2>&1 | tee gpurun_out/q_pytest_three_passes.txt
