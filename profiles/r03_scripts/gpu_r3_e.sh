#!/bin/bash
# Round 3, session E: K3 with packed hit bytes (SDWA destination select) + dot-product hit masks + deferred confirm.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest -m gpu (engine + CLI) =="
timeout 2400 python -m pytest tests/test_gpu_engine.py tests/test_gpu_filegrep.py -m gpu -q -x 2>&1 | tail -6 | tee gpurun_out/e_pytest.txt
SW=$R/grab_amd/bin/gscan_sweep
echo "== kernel sweep, 16 GiB: K3 forms =="
timeout 900 $SW --gib 16 --iters 6 --variants 38 --bpc 0 --pattern '[0-9]+\.[0-9]+' --pattern 'foobardoesnotexist|Linus|555-1234' --pattern '(?i)foobar|k7Q,;q|[0-9]{12}x?' --pattern '[a-z][0-9][A-Z][.,][;:]' --pattern 'foo.*bar' --pattern 'foo|bar' --pattern '[a-z]{2,5}' --pattern 'foobardoesnotexist' 2>&1 | grep -E "^variant|^#" | tee gpurun_out/e_sweep_k3.txt
echo "== inexact patterns (device VM), 8 GiB =="
timeout 600 $SW --gib 8 --iters 3 --variants 38 --bpc 0 --pattern '(\w)\1{3,}x|foobardoes(?=not)' --pattern 'a+b+c' --pattern '[a-z]+\([a-z0-9, ]*\);' 2>&1 | grep -E "^variant|^#" | tee gpurun_out/e_sweep_vm.txt
