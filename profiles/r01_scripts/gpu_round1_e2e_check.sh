#!/bin/bash
# End-to-end regression check after the host-side work of the round (matcher, cursor, back references): literal and
# identifier regex over 128 x 64 MiB, outputs compared with the reference binary.
set -u
mkdir -p gpurun_out
for spec in "foobardoesnotexist|-O -l" "[A-Za-z_][A-Za-z0-9_]{15,}|-O -l" "(\\w)\\1{3,}x|foobardoes(?=not)|-O -l"; do
  pat="${spec%%|*}"; rest="${spec#*|}"
  if [ "$pat" = '(\w)\1{3,}x' ]; then pat='(\w)\1{3,}x|foobardoes(?=not)'; rest='-O -l'; fi
  timeout 600 python scripts/e2e_cli.py --files 128 --file-kib 65536 --pattern "$pat" --flags "$rest" --workers 8 --ref-cores 64 --reps 1 --tag "e2e_check" 2>&1 | tail -1 | tee -a gpurun_out/e2e_check.jsonl
done
