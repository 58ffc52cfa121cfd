#!/bin/bash
# Round 5, session T: the line-printing modes of BASELINE configs[2]'s regex over 16 GiB (256 x 64 MiB) at the final code -- `-n 8 -r -O`
# and `-n 8 -r` -- count + order-independent digest of every output line against the reference's (k_lines changed this round).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
python - <<'PY' 2>&1 | tee gpurun_out/r5t_line_modes_16GiB_digest.txt
import json, os, subprocess, sys, time
sys.path.insert(0, ".")
import bench
from grab_amd import bin_path, synth
bench.interleave_page_placement()
bench.gen_child("import bench\nbench.gen_corpus('/dev/shm/c16', 256, 64 << 20)\n")
ref = "oracle/_ref/grab_jit"
for flags in (["-O"], []):
    t0 = time.perf_counter()
    subprocess.run([bin_path(), "-n", "8", "-r"] + flags + [synth.IDENT_RE, "/dev/shm/c16"], stdout=subprocess.DEVNULL)
    wall = time.perf_counter() - t0
    n, d, s = bench.line_digest([bin_path(), "-n", "8", "-r"] + flags + [synth.IDENT_RE, "/dev/shm/c16"])
    rn, rd, rs = bench.line_digest([ref, "-n", "64", "-r"] + flags + [synth.IDENT_RE, "/dev/shm/c16"])
    print(json.dumps({"flags": " ".join(["-n 8 -r"] + flags), "lines": n, "reference_lines": rn, "digest": d, "reference_digest": rd, "same": n == rn and d == rd and d is not None,
                      "grab_wall_s_to_dev_null": round(wall, 3), "reference_s_through_the_digest": round(rs, 1)}), flush=True)
import shutil
shutil.rmtree("/dev/shm/c16", ignore_errors=True)
PY
