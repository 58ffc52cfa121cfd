#!/bin/bash
# Round 2, GPU session N: the HIP runtime's first calls one by one (init_probe), and a cfg2 run that no longer closes
# its contexts before _exit against one that does (GRAB_CLOSE=1).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for a in x h x; do echo "## init_probe $a"; grab_amd/bin/init_probe $a; done
for e in "GPU_MAX_HW_QUEUES=2" "HSA_ENABLE_SDMA=0" "HSA_ENABLE_INTERRUPT=0"; do echo "## $e"; env $e grab_amd/bin/init_probe x | head -3; done
} > gpurun_out/n_init_probe.txt 2>&1
python - <<'PY' > gpurun_out/n_close_or_not.txt 2>&1
import os, subprocess, sys, time, shutil
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
import e2e_sweep
from grab_amd import bin_path, synth
d = "/dev/shm/n_cfg2"
os.makedirs(d)
e2e_sweep.gen_files(d, 512, 64 << 20, 1)
for workers in (8, 3):
    argv = [bin_path(), "-n", str(workers), "-r", synth.NEEDLE.decode(), d]
    for extra in [{}, {"GRAB_CLOSE": "1"}, {}, {"GRAB_CLOSE": "1"}]:
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            r = subprocess.run(argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, GRAB_TIMING="1", **extra))
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, r.stderr.decode())
        marks = [ln for ln in best[1].splitlines() if ln.startswith("[grab timing] +")]
        print("workers %d %s: wall %.3f s = %.2f GB/s | %s" % (workers, extra, best[0], 512 * (64 << 20) / best[0] / 1e9, " | ".join(m[14:] for m in marks)))
shutil.rmtree(d)
PY
cat gpurun_out/n_init_probe.txt gpurun_out/n_close_or_not.txt
