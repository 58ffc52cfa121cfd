#!/bin/bash
# Round 3, session AC: K3 with the look-ups of step k + 1 in flight while step k is computed (lib) against the build without
# (libk3old).  Engine tests first.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SW=$R/grab_amd/bin/gscan_sweep
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/ac_pytest.txt
{
for L in lib libk3old lib libk3old lib libk3old; do
  echo "## $L"
  LD_LIBRARY_PATH=$R/grab_amd/$L timeout 300 $SW --gib 16 --iters 8 --variants 38 --bpc 0 --pattern 'foobardoesnotexist|Linus|555-1234' --pattern 'foo|bar' --pattern '(?i)linus|torvalds|kernel|module|driver|device|buffer|socket|signal|thread|mutex|atomic|barrier|memory' --pattern '[a-z][0-9][A-Z][_]x' --pattern 'colou?r' 2>&1 | grep -E "^variant"
done
} | tee gpurun_out/ac_k3_ahead_sweep.txt
