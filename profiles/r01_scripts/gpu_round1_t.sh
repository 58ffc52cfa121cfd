#!/bin/bash
set -u
export TMPDIR=/tmp
python - <<'PY'
import os, subprocess, sys, time, shutil
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
from fullsize_parity import gen_files
from grab_amd import bin_path, synth
base = "/dev/shm/grab_t_%d" % os.getpid()
os.makedirs(base)
try:
    gen_files(base, 32, 64 << 20, 64)
    for env_extra in ({"GRAB_LINE_PASS": "1"}, {}):
        for w in (1, 4):
            argv = [bin_path()] + (["-n", str(w)] if w > 1 else []) + ["-r", "-O", synth.IDENT_RE, base]
            env = dict(os.environ, GRAB_TIMING="1", **env_extra)
            for it in range(2):
                t0 = time.perf_counter(); r = subprocess.run(argv, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env); dt = time.perf_counter() - t0
            lines = [l for l in r.stderr.decode().splitlines() if "device 0: files" in l]
            print("line pass %s, workers %d: %.3f s" % ("on " if env_extra else "off", w, dt)); print("   ", lines[0][14:] if lines else "")
finally:
    shutil.rmtree(base, ignore_errors=True)
PY
