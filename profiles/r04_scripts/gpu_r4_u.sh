#!/bin/bash
# Round 4, session U: what makes the exit of a GPU process older than ~0.5 s cost 0.1 s: a process that only initialised the
# runtime / + a stream / + one DMA / + one kernel, asleep 0 / 300 / 700 / 1500 ms, then _exit -- time from its last stamp
# until the parent has reaped it.  Then the prefault test and the smoke test at the round's final code.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/probes/exit_age_probe.hip -o /tmp/exit_age_probe
python - <<'PY' | tee gpurun_out/u_exit_age_probe.txt
import subprocess, time
print("what (0 init, 1 +stream, 2 +DMA, 3 +kernel) x sleep ms -> exit ms (min of 3; half a second of quiet before each)")
for what in (0, 1, 2, 3):
    row = []
    for ms in (0, 300, 700, 1500):
        best = None
        for rep in range(3):
            time.sleep(0.5)
            p = subprocess.Popen(["/tmp/exit_age_probe", str(what), str(ms)], stdout=subprocess.PIPE)
            stamp = float(p.stdout.readline())
            p.wait()
            dt = time.clock_gettime(time.CLOCK_MONOTONIC) - stamp
            best = dt if best is None else min(best, dt)
        row.append("%4d ms asleep: %6.1f" % (ms, best * 1e3))
    print("what %d | " % what + " | ".join(row))
PY
timeout 600 python -m pytest tests/test_gpu_engine.py -m gpu -q -k "prefault or submit_files or pipelined" 2>&1 | tail -2 | tee gpurun_out/u_pytest_subset.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/u_pytest_subset.txt
