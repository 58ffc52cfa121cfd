#!/bin/bash
# Round 2, GPU session U: cfg5 after the walk stopped reading the text of listed literals; cfg3 per-worker time split and
# how it scales with -n (the reference runs it with -n 64).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY' > gpurun_out/u_cfg5_cfg3_host.txt 2>&1
import os, subprocess, sys, time, shutil
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
import e2e_sweep
from grab_amd import bin_path, synth
REF = os.path.join(os.getcwd(), "oracle", "_ref", "grab_jit")
def run(argv, extra={}, reps=2, keep=lambda l: "timing]" in l and "gscan_open" not in l and "memory" not in l):
    best = None
    for rep in range(reps):
        with open("/dev/shm/u_out.txt", "wb") as o:
            t0 = time.monotonic()
            r = subprocess.run(argv, stdout=o, stderr=subprocess.PIPE, env=dict(os.environ, **extra))
            dt = time.monotonic() - t0
        if best is None or dt < best[0]:
            best = (dt, r.stderr.decode(), os.path.getsize("/dev/shm/u_out.txt"))
    return best[0], "\n".join(l for l in best[1].splitlines() if keep(l)), best[2]
d = "/dev/shm/u_cfg5"
os.makedirs(d)
e2e_sweep.gen_files(d, 1, 16 << 30, 1, needles_every=32768 + 77)
f = os.path.join(d, "f000000.txt")
for flags in (["-O", "-l"], ["-O"], []):
    dt, marks, nout = run([bin_path()] + flags + [synth.NEEDLE.decode(), f], {"GRAB_TIMING": "1"})
    print("## cfg5 16 GiB", " ".join(flags), "wall %.3f s = %.2f GB/s, %d bytes out" % (dt, (16 << 30) / dt / 1e9, nout))
    print(marks)
shutil.rmtree(d)
d = "/dev/shm/u_cfg3"
os.makedirs(d)
e2e_sweep.gen_files(d, 256, 64 << 20, 1)
ident = "[A-Za-z_][A-Za-z0-9_]{15,}"
for n in (8, 16, 32, 64):
    dt, marks, nout = run([bin_path(), "-n", str(n), "-r", "-O", "-l", ident, d], {"GRAB_TIMING": "1"})
    print("## cfg3 16 GiB -n %d: wall %.3f s = %.2f GB/s, %d bytes out" % (n, dt, (16 << 30) / dt / 1e9, nout))
    lines = marks.splitlines()
    print("\n".join(lines[:3] + [l for l in lines if "device 0:" in l][:2] + lines[-3:]))
dt, marks, nout = run([REF, "-n", "64", "-r", "-O", "-l", ident, d], reps=1)
print("## cfg3 16 GiB reference -n 64: wall %.3f s = %.2f GB/s, %d bytes out" % (dt, (16 << 30) / dt / 1e9, nout))
for n in (8, 64):
    dt, marks, nout = run([bin_path(), "-n", str(n), "-r", ident, d], {"GRAB_TIMING": "1"})
    print("## cfg3 lines 16 GiB -n %d: wall %.3f s = %.2f GB/s, %d bytes out" % (n, dt, (16 << 30) / dt / 1e9, nout))
    lines = marks.splitlines()
    print("\n".join([l for l in lines if "device 0:" in l][:2] + lines[-2:]))
shutil.rmtree(d)
PY
cat gpurun_out/u_cfg5_cfg3_host.txt
