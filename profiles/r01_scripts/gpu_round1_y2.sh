#!/bin/bash
# K3 with three filter positions (DEPTH 3) against four: parity with the form forced on for every pattern, then the sweep.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SW=$R/grab_amd/bin/gscan_sweep
echo "== parity, depth 3 forced =="
GSCAN_K3_DEPTH=3 timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_filegrep.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/y2_pytest_depth3.txt
echo "== parity, heuristic =="
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/y2_pytest.txt
{
for pat in 'foobardoesnotexist|Linus|555-1234' 'foo|bar' '(?i)linus' '[a-z][0-9][A-Z][.,][;:]' '(?:ab|cd|ef|gh|ij|kl|mn|op|qr|st)x' 'foobardoes(?:not)?exist|[0-9A-F]{7}[a-z]?|(?i:xyzzy)'; do
  for d in 4 3; do
    echo "== depth $d pattern $pat"
    GSCAN_K3_DEPTH=$d timeout 300 $SW --gib 8 --iters 6 --variants 6 --bpc 0 --pattern "$pat" | grep -v "^overflow" | tail -1
  done
done
} 2>&1 | tee gpurun_out/y2_k3_depth_sweep.txt
