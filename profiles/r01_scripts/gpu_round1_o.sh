#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu engine =="
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -4
SW=grab_amd/bin/gscan_sweep
for P in 'foo|bar' 'foobardoesnotexist|Linus' '(?i)foobardoesnotexist' 'foobardoes(?:not)?exist|[0-9A-F]{7}[a-z]?|(?i:xyzzy)' 'alpha|beta|gamma|delta|epsilon|zeta|eta|theta|iota|kappa|lambda|mu' '(?i)alpha|beta|gamma|delta|epsilon|zeta|theta|iota|kappa|lambda|omega|sigma|omicron|upsilon' '[0-9]+\.[0-9]+' 'foo.*bar' '\bfoo\b'; do
  echo "== sweep K3: $P"
  timeout 300 $SW --gib 16 --iters 8 --variants 6 --bpc 0 --pattern "$P" 2>&1 | tail -2
done | tee gpurun_out/o_sweep_k3.txt
