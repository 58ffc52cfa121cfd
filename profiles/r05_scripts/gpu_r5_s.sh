#!/bin/bash
# Round 5, session S: BASELINE configs[4] at its full 32 GiB in all three forms -- `-O -l` at 1 GiB windows, `-L -L -L -L -L -O -l`
# (32 MiB windows: 1025 of them, every overlap duplicate), `-O` (lines printed: k_lines with this round's 32-byte steps and LDS
# tail table over a million planted needles) -- byte-exact against the reference binary (scripts/fullsize_parity.py --only cfg5).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
timeout 800 python scripts/fullsize_parity.py --only cfg5 2>&1 | tail -1 | tee gpurun_out/r5s_fullsize_parity_cfg5.txt
