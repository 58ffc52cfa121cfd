#!/bin/bash
# K3 (bucket filter) bring-up: parity suite, then throughput of K3 on a few alternation shapes.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu =="
timeout 1500 python -m pytest tests -x -q -m gpu --durations=5 2>&1 | tail -25 | tee gpurun_out/f_pytest_gpu.txt
SW=grab_amd/bin/gscan_sweep
for P in 'foobardoesnotexist|Linus' 'foo|bar' '(?i)foobardoesnotexist' 'foobardoes(?:not)?exist|[0-9A-F]{7}[a-z]?|(?i:xyzzy)' '[a-z][0-9][A-Z][.,][;:]' 'alpha|beta|gamma|delta|epsilon|zeta|eta|theta|iota|kappa|lambda|mu'; do
  echo "== sweep K3: $P"
  timeout 300 $SW --gib 8 --iters 5 --variants 4,5,6 --bpc 0,4 --pattern "$P" 2>&1 | tail -7
done | tee gpurun_out/f_sweep_k3.txt
