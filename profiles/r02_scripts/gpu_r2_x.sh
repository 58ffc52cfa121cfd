#!/bin/bash
# Round 2, GPU session Y: survivors of the two-byte table confirmed per sub-tile instead of per hit (session X: the table alone) (DevProgram::vm_pair): kernel rate and end to end,
# with the table and without (GSCAN_NO_VM_PAIRS=1); the engine tests.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
S=grab_amd/bin/gscan_sweep
{
for e in 0 1; do
  if [ $e = 1 ]; then export GSCAN_NO_VM_PAIRS=1; echo "## GSCAN_NO_VM_PAIRS=1"; else unset GSCAN_NO_VM_PAIRS; echo "## with the table"; fi
  $S --gib 8 --iters 3 --pattern '(\w)\1{3,}x|foobardoes(?=not)' --variants 6 --bpc 0
  $S --gib 8 --iters 3 --pattern '[a-z]+\([a-z0-9, ]*\);' --variants 6 --bpc 0
  $S --gib 8 --iters 3 --pattern 'a+b+c' --variants 6 --bpc 0
done
} > gpurun_out/x_vm_kernel_rate.txt 2>&1
grep -E "^#|variant" gpurun_out/x_vm_kernel_rate.txt
unset GSCAN_NO_VM_PAIRS
rm -f gpurun_out/x_vm_e2e.jsonl
for P in '(\w)\1{3,}x|foobardoes(?=not)' '[a-z]+\([a-z0-9, ]*\);'; do
  timeout 300 python scripts/e2e_cli.py --files 128 --pattern "$P" --flags "-O -l" --workers 8 --tag vm >> gpurun_out/x_vm_e2e.jsonl 2>> gpurun_out/x_vm_e2e.err
done
cut -c1-500 gpurun_out/x_vm_e2e.jsonl
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_filegrep.py -m gpu -q 2>&1 | tail -3
