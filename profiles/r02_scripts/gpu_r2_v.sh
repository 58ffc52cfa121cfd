#!/bin/bash
# Round 2, GPU session V: full-size parity (BASELINE configs 2-5) and the inexact patterns end to end, after the start-up /
# stream / walk changes.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python scripts/fullsize_parity.py --workers 8 2>&1 | tail -4 | tee gpurun_out/v_fullsize_parity.txt
rm -f gpurun_out/v_vm_e2e.jsonl
for P in '(\w)\1{3,}x|foobardoes(?=not)' '[a-z]+\([a-z0-9, ]*\);'; do
  timeout 300 python scripts/e2e_cli.py --files 128 --pattern "$P" --flags "-O -l" --workers 8 --tag vm >> gpurun_out/v_vm_e2e.jsonl 2>> gpurun_out/v_vm_e2e.err
done
cat gpurun_out/v_vm_e2e.jsonl | cut -c1-600
