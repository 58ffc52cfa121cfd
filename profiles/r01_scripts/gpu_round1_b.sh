#!/bin/bash
# Second GPU session: re-verify parity after the kernel/engine restructuring, measure the read ceiling,
# sweep variants, run the full 64 GiB bench for both configs, collect rocprofv3 kernel stats.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu (engine) =="
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -25 | tee gpurun_out/pytest_engine.txt
echo "== sweep K1 + ceiling (16 GiB) =="
timeout 300 grab_amd/bin/gscan_sweep --gib 16 --iters 6 --variants 0,1,2,4,5,6 --bpc 0,8,16 --ceiling 2>&1 | tee gpurun_out/sweep_k1.txt
echo "== sweep K2 (16 GiB) =="
timeout 300 grab_amd/bin/gscan_sweep --gib 16 --iters 6 --variants 0,1,2,4,5,6 --bpc 0,8 --pattern '[A-Za-z_][A-Za-z0-9_]{15,}' 2>&1 | tee gpurun_out/sweep_k2.txt
echo "== sweep K2 wide (8 GiB) =="
timeout 300 grab_amd/bin/gscan_sweep --gib 8 --iters 4 --variants 0,1,4 --bpc 0,8 --pattern '[0-9a-f]{32}' 2>&1 | tee gpurun_out/sweep_k2_wide.txt
echo "== pytest gpu (filegrep, fast subset) =="
timeout 900 python -m pytest tests/test_gpu_filegrep.py -x -q -m gpu -k "not syn256" 2>&1 | tail -8 | tee gpurun_out/pytest_filegrep.txt
echo "== bench cfg2 full (64 GiB) =="
timeout 900 python bench.py --steps 10 --warmup 2 2>&1 | tail -2 | tee gpurun_out/bench_cfg2.json
echo "== bench cfg3 full (64 GiB) =="
timeout 900 python bench.py --config cfg3 --steps 5 --warmup 1 2>&1 | tail -2 | tee gpurun_out/bench_cfg3.json
echo "== rocprofv3 kernel stats (cfg2, 16 GiB) =="
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_cfg2 -- python $GRAFT_REPO_ROOT/bench.py --files 256 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -3
cd $GRAFT_REPO_ROOT; find gpurun_out/prof_cfg2 -name "*stats*" | head; for f in $(find gpurun_out/prof_cfg2 -name "*kernel_stats.csv"); do head -8 $f; done
