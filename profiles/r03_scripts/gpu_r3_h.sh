#!/bin/bash
# Round 3, session H: why the line pass + gather does not show end to end: the host's per-line cost on this box in isolation
# (scripts/probes/report_probe.cc, incl. the gathered text in pinned memory), then the end-to-end A/B with the report stage
# split into formatting and writing out.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== report probe (one thread, 64 MiB window) =="
$R/grab_amd/bin/report_probe 64 2>&1 | tee gpurun_out/h_report_probe.txt
echo "== end to end, 16 GiB, -n 8 =="
python - <<'PY' > gpurun_out/h_lines_e2e.txt 2>&1
import os, subprocess, sys, time, shutil
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
import e2e_sweep
from grab_amd import bin_path
d = "/dev/shm/r3h_cfg3"
os.makedirs(d)
e2e_sweep.gen_files(d, 256, 64 << 20, 1)
ident = "[A-Za-z_][A-Za-z0-9_]{15,}"
for n in (8, 16):
  for flags in (["-O"], []):
    for env_extra, label in (({}, "host walk"), ({"GRAB_LINE_PASS": "1"}, "device line pass + gather")):
        best = None
        for rep in range(3):
            t0 = time.monotonic()
            r = subprocess.run([bin_path(), "-n", str(n), "-r"] + flags + [ident, d], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, GRAB_TIMING="1", GSCAN_TIMING="1", **env_extra))
            dt = time.monotonic() - t0
            if best is None or dt < best[0]: best = (dt, r.stderr.decode())
        lines = [l for l in best[1].splitlines() if "device 0:" in l][:2] + [l for l in best[1].splitlines() if "workers joined" in l or "runtime up" in l or "gscan timing" in l][-3:]
        print("## 16 GiB -n %d %s (%s): wall %.3f s = %.2f GB/s" % (n, " ".join(flags) or "(lines)", label, best[0], (16 << 30) / best[0] / 1e9)); print("\n".join(lines))
shutil.rmtree(d)
PY
cat gpurun_out/h_lines_e2e.txt
