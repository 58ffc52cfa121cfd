#!/bin/bash
# Round 5, session H: BASELINE configs[4] (one 32 GiB file, -O -l) the way bench.py measures it -- a child of a process that
# holds a HIP context -- with ROUND 4's binary (HEAD aa1670d, built into gpurun_ab/r04) and this round's, alternating on the
# same file: is the 0.94-1.02 s of sessions E and G (round 4's bench lines: 0.83) the code's or the context's?  Then the whole
# GPU suite.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
python - <<'PY' 2>&1 | tee gpurun_out/r5h_cfg5_old_vs_new_in_bench_context.txt
import json, os, subprocess, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "scripts")
import torch, bench, fullsize_parity
dev = torch.device("cuda", 0)
x = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
bench.interleave_page_placement()
path = "/dev/shm/one32g.bin"
fullsize_parity.gen_big(path, 32 << 30, 1 << 30, 1000000)
bins = {"r04": os.path.abspath("gpurun_ab/r04/bin/grab"), "r05": os.path.abspath("grab_amd/bin/grab")}
envs = {"r04": None, "r05": None, "r05 no read-ahead": dict(os.environ, GRAB_NO_READ_AHEAD="1"), "r05 second stream at open": dict(os.environ, GSCAN_SECOND_STREAM_MIB="0"),
        "r05 one stream": dict(os.environ, GSCAN_COPY_STREAMS="1"), "r05 8 readers fixed": dict(os.environ, GSCAN_READERS="8")}
res = {k: [] for k in envs}
for rep in range(4):
    for k in envs:
        time.sleep(0.5)
        with open("/dev/shm/out.txt", "wb") as out:
            t0 = time.perf_counter()
            r = subprocess.run([bins[k.split()[0]], "-O", "-l", "foobardoesnotexist", path], stdout=out, stderr=subprocess.DEVNULL, env=envs[k])
            dt = time.perf_counter() - t0
        if rep:
            res[k].append(round(dt, 4))
print(json.dumps(res))
del x
torch.cuda.empty_cache()
res2 = {k: [] for k in ("r04", "r05")}
for rep in range(3):
    for k in res2:
        time.sleep(0.5)
        t0 = time.perf_counter()
        subprocess.run([bins[k], "-O", "-l", "foobardoesnotexist", path], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        res2[k].append(round(time.perf_counter() - t0, 4))
print("after the parent freed its GiB (context still there), output to /dev/null:", json.dumps(res2))
os.unlink(path)
PY
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee gpurun_out/r5h_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/r5h_pytest.txt
