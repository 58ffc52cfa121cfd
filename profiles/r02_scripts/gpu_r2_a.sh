#!/bin/bash
# Round 2, GPU session A: the GPU suite after the ingest rework (async submit_fd, 3 slots, parallel walk, multi-context spread)
# incl. the new real-geometry tests; the ingest knob sweep end to end; the new bench line.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
nproc > gpurun_out/a_box.txt; free -g >> gpurun_out/a_box.txt; df -h /dev/shm /tmp >> gpurun_out/a_box.txt
python -c "from grab_amd import engine; print(engine.device_cpulist(0)); print(engine.ingest_info())" >> gpurun_out/a_box.txt 2>&1
lscpu | grep -E "NUMA|Model name|Socket" >> gpurun_out/a_box.txt
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 2>&1 | tail -30 | tee gpurun_out/a_pytest.txt
echo "== e2e sweep =="
timeout 900 python scripts/e2e_sweep.py --gib 64 --small-gib 16 --single-gib 8 > gpurun_out/a_e2e_sweep.jsonl 2> gpurun_out/a_e2e_sweep.err
tail -5 gpurun_out/a_e2e_sweep.err
echo "== bench =="
( time timeout 900 python bench.py ) > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err
tail -3 gpurun_out/a_bench.err
cat gpurun_out/a_bench.json | head -c 3000
