#!/bin/bash
# Round 3, session G: the device's line pass WITH the line gather (the printed lines' text comes back from the device; the
# host formats and never reads the window) against the host walk, line-printing modes end to end; parity of the pass.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest: line pass parity =="
timeout 1200 python -m pytest tests/test_gpu_engine.py tests/test_gpu_filegrep.py -m gpu -q -x -k "line_extents or line_pass or match_ends or without_the_text or multichunk" 2>&1 | tail -4 | tee gpurun_out/g_pytest.txt
echo "== line-printing modes end to end, 16 GiB: device line pass + gather vs host walk =="
python - <<'PY' > gpurun_out/g_lines_e2e.txt 2>&1
import os, subprocess, sys, time, shutil, hashlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
import e2e_sweep
from grab_amd import bin_path
d = "/dev/shm/r3g_cfg3"
os.makedirs(d)
e2e_sweep.gen_files(d, 256, 64 << 20, 1)
ident = "[A-Za-z_][A-Za-z0-9_]{15,}"
for pattern in (ident, "foobardoesnotexist", "[0-9A-F]{6}[a-z]"):
  for flags in (["-O"], []):
    for env_extra, label in (({}, "host walk"), ({"GRAB_LINE_PASS": "1"}, "device line pass + gather")):
        best = None
        for rep in range(3):
            t0 = time.monotonic()
            r = subprocess.run([bin_path(), "-n", "8", "-r"] + flags + [pattern, d], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, GRAB_TIMING="1", **env_extra))
            dt = time.monotonic() - t0
            if best is None or dt < best[0]: best = (dt, r.stderr.decode())
        lines = [l for l in best[1].splitlines() if "device 0:" in l][:1]
        print("## 16 GiB -n 8 %s '%s' (%s): wall %.3f s = %.2f GB/s" % (" ".join(flags) or "(lines)", pattern, label, best[0], (16 << 30) / best[0] / 1e9)); print("\n".join(lines))
# same bytes (sorted) with and without the pass, 4 GiB subset
d4 = "/dev/shm/r3g_sub"; os.makedirs(d4)
for i in range(64): os.link(os.path.join(d, "f%06d.txt" % i), os.path.join(d4, "f%06d.txt" % i))
for flags in (["-O"], []):
    outs = []
    for env_extra in ({}, {"GRAB_LINE_PASS": "1"}):
        p = "/dev/shm/r3g_out.txt"
        with open(p, "wb") as o:
            subprocess.run([bin_path(), "-n", "8", "-r"] + flags + [ident, d4], stdout=o, env=dict(os.environ, **env_extra))
        r = subprocess.run("LC_ALL=C sort %s | md5sum; wc -l < %s" % (p, p), shell=True, capture_output=True, text=True)
        outs.append(r.stdout.split()); os.unlink(p)
    print("sorted md5 / lines", flags, outs, "same" if outs[0] == outs[1] else "DIFFERENT")
shutil.rmtree(d); shutil.rmtree(d4)
PY
cat gpurun_out/g_lines_e2e.txt
