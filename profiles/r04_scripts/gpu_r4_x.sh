#!/bin/bash
# Round 4, session X: the whole GPU suite once more at HEAD.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee gpurun_out/x_pytest.txt
