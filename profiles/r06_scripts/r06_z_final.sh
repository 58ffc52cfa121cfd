# Round 6, closing session Z: the GPU suite, smoke, the bench line, rocprofv3 kernel stats of the bench command, the everyday-patterns table
python -m pytest tests -q -m gpu > gpurun_out/r06_z_pytest_gpu.txt 2>&1; tail -3 gpurun_out/r06_z_pytest_gpu.txt
python __graft_entry__.py smoke > gpurun_out/r06_z_smoke.txt 2>&1; tail -1 gpurun_out/r06_z_smoke.txt
( time python bench.py > gpurun_out/r06_z_bench.json 2> gpurun_out/r06_z_bench.err ) 2> gpurun_out/r06_z_bench_time.txt; cat gpurun_out/r06_z_bench_time.txt
R=$PWD; cd /tmp && TMPDIR=/tmp timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_z_prof -- python $R/bench.py --no-e2e --no-cpu-baseline --no-live-traffic > $R/gpurun_out/r06_z_prof.log 2>&1; cd $R
f=$(find gpurun_out/r06_z_prof -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r06_z_prof_kernel_stats.csv; rm -rf gpurun_out/r06_z_prof; head -8 gpurun_out/r06_z_prof_kernel_stats.csv | cut -c1-200
python scripts/everyday_patterns.py 4 > gpurun_out/r06_z_everyday_patterns.jsonl 2> gpurun_out/r06_z_everyday_err.txt
