#!/bin/bash
# Inexact patterns (gscan_info.exact == 0: kernels look for what a match must begin with, the host matcher confirms):
# the GPU suite, then end-to-end CLI runs next to the reference binary on a 2 GiB corpus (output compared, sorted).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/v_pytest.txt
echo "== e2e, inexact patterns =="
for pat in '[a-z]+_[0-9]+\.[a-z]+' '(?:foo|bar|ab)+baz' '[a-z]+\([a-z0-9, ]*\);' '(?m)^[a-z]+ = [0-9A-F]+;$' 'foobar[a-z]*does.*not.*exist'; do
  timeout 600 python scripts/e2e_cli.py --files 32 --file-kib 65536 --pattern "$pat" --flags "-O -l" --workers 1,8 --ref-cores 32 --reps 1 --tag "inexact" 2>&1 | tail -1 | tee -a gpurun_out/v_e2e_inexact.jsonl
done
