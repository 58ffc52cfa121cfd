#!/bin/bash
# Round 4, session V: the exit of an old process once more, on ONE box: the bare HIP probe (no age effect seen in session U)
# next to `grab` on one 64 MiB file asleep 0 / 700 ms before it leaves, and the same with one empty kernel launched right
# before leaving (GRAB_EXIT_KICK=1).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/probes/exit_age_probe.hip -o /tmp/exit_age_probe
python - <<'PY'
import os, sys
sys.path.insert(0, ".")
import torch
from grab_amd import synth
synth.torch_text(64 << 20, 0, torch.device("cuda", 0)).cpu().numpy().tofile("/dev/shm/one.txt")
os.makedirs("/dev/shm/c2_16", exist_ok=True)
for i in range(16):
    synth.torch_text(64 << 20, i, torch.device("cuda", 0)).cpu().numpy().tofile("/dev/shm/c2_16/f%02d.txt" % i)
PY
{
python - <<'PY'
import subprocess, time
for what in (0, 3):
    row = []
    for ms in (0, 700):
        best = None
        for rep in range(3):
            time.sleep(0.5)
            p = subprocess.Popen(["/tmp/exit_age_probe", str(what), str(ms)], stdout=subprocess.PIPE)
            stamp = float(p.stdout.readline()); p.wait()
            dt = time.clock_gettime(time.CLOCK_MONOTONIC) - stamp
            best = dt if best is None else min(best, dt)
        row.append("%4d ms asleep: %6.1f ms" % (ms, best * 1e3))
    print("bare probe, what %d | " % what + " | ".join(row))
PY
G=grab_amd/bin/grab
python scripts/ab_run.py --sleep 0.5 --reps 3 --bytes 67108864 --interleave --env "GRAB_EXIT_SLEEP_MS=0" --env "GRAB_EXIT_SLEEP_MS=700" --env "GRAB_EXIT_SLEEP_MS=700 GRAB_EXIT_KICK=1" --env "GRAB_EXIT_SLEEP_MS=0 GRAB_EXIT_KICK=1" --env "GRAB_EXIT_SLEEP_MS=700 GSCAN_PREFAULT=0" -- $G foobardoesnotexist /dev/shm/one.txt
python scripts/ab_run.py --sleep 0.5 --reps 3 --bytes $((16 * 67108864)) --interleave --env "GRAB_EXIT_SLEEP_MS=700" --env "GRAB_EXIT_SLEEP_MS=700 GRAB_EXIT_KICK=1" -- $G -n 8 -r foobardoesnotexist /dev/shm/c2_16
} 2>&1 | tee gpurun_out/v_exit_kick.txt
rm -rf /dev/shm/one.txt /dev/shm/c2_16
