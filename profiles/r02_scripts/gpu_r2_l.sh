#!/bin/bash
# Round 2, GPU session L: does it matter which socket's memory holds the corpus (tmpfs pages are first-touched by the writer)?
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for where in local remote; do
  timeout 600 python scripts/e2e_sweep.py --gib 32 --small-gib 0 --single-gib 0 --blocks 16 --readers 8 --streams 1 --reps 2 --corpus-cpus $where
done > gpurun_out/l_e2e_corpus_numa.jsonl 2> gpurun_out/l_err.txt
cut -c1-700 gpurun_out/l_e2e_corpus_numa.jsonl; tail -3 gpurun_out/l_err.txt
