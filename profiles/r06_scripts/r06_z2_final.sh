# Round 6, closing session Z2 (session Z again, after the reader pools' lost wake-up was fixed): the GPU suite, smoke, rocprofv3 kernel
# stats of the bench command, the bench line.  Every step under its own limit.
timeout 480 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r06_z2_pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r06_z2_pytest_gpu.txt
timeout 120 python __graft_entry__.py smoke > gpurun_out/r06_z2_smoke.txt 2>&1; echo "smoke rc $?"; tail -1 gpurun_out/r06_z2_smoke.txt
R=$PWD; cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_z2_prof -- python $R/bench.py --no-e2e --no-cpu-baseline --no-live-traffic > $R/gpurun_out/r06_z2_prof.log 2>&1; echo "rocprof rc $?"; cd $R
f=$(find gpurun_out/r06_z2_prof -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r06_z2_prof_kernel_stats.csv; rm -rf gpurun_out/r06_z2_prof; head -8 gpurun_out/r06_z2_prof_kernel_stats.csv | cut -c1-200
( time timeout 600 python bench.py > gpurun_out/r06_z2_bench.json 2> gpurun_out/r06_z2_bench.err ) 2> gpurun_out/r06_z2_bench_time.txt; cat gpurun_out/r06_z2_bench_time.txt; cut -c1-600 gpurun_out/r06_z2_bench.json
