#!/bin/bash
# Round 3, session F: K3's three-position filter as the whole pattern for windows of <= 3 bytes; the device's line pass
# (k_lines) against the host walk in the line-printing modes, end to end; cfg3 at BASELINE's full 64 GiB against the reference.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest -m gpu (engine) =="
timeout 1200 python -m pytest tests/test_gpu_engine.py -m gpu -q -x 2>&1 | tail -4 | tee gpurun_out/f_pytest.txt
SW=$R/grab_amd/bin/gscan_sweep
echo "== kernel sweep, 16 GiB: K3 short windows =="
timeout 900 $SW --gib 16 --iters 6 --variants 38 --bpc 0 --pattern '[0-9]+\.[0-9]+' --pattern 'foo|bar' --pattern 'a|ab' --pattern '[a-z][0-9][A-Z][.,][;:]' --pattern 'foobardoesnotexist|Linus|555-1234' 2>&1 | grep -E "^variant|^#" | tee gpurun_out/f_sweep_k3.txt
echo "== line-printing modes end to end, 16 GiB: device line pass vs host walk =="
python - <<'PY' > gpurun_out/f_lines_e2e.txt 2>&1
import os, subprocess, sys, time, shutil
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
import e2e_sweep
from grab_amd import bin_path
d = "/dev/shm/r3f_cfg3"
os.makedirs(d)
e2e_sweep.gen_files(d, 256, 64 << 20, 1)
ident = "[A-Za-z_][A-Za-z0-9_]{15,}"
for flags in (["-O"], []):
    for env_extra, label in (({}, "host walk"), ({"GRAB_LINE_PASS": "1"}, "device line pass")):
        best = None
        for rep in range(3):
            t0 = time.monotonic()
            r = subprocess.run([bin_path(), "-n", "8", "-r"] + flags + [ident, d], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, GRAB_TIMING="1", **env_extra))
            dt = time.monotonic() - t0
            if best is None or dt < best[0]: best = (dt, r.stderr.decode())
        lines = [l for l in best[1].splitlines() if "device 0:" in l][:1] + [l for l in best[1].splitlines() if "workers joined" in l or "runtime up" in l]
        print("## cfg3 16 GiB -n 8 %s (%s): wall %.3f s = %.2f GB/s" % (" ".join(flags) or "(lines)", label, best[0], (16 << 30) / best[0] / 1e9)); print("\n".join(lines))
shutil.rmtree(d)
PY
cat gpurun_out/f_lines_e2e.txt
echo "== cfg3 at 64 GiB against the reference (sorted md5) =="
timeout 1500 python scripts/fullsize_parity.py --only cfg3 --cfg3-files 1024 --workers 8 2>&1 | tail -3 | tee gpurun_out/f_fullsize_cfg3.txt
