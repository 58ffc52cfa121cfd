#!/bin/bash
# Windows longer than the kernels take, through the CLI on the GPU: product vs the C oracle (libpcre under the reference's loop).
set -u
mkdir -p gpurun_out
python3 - <<'PY' | tee gpurun_out/long_window_check.txt
import subprocess, os, numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", ".")
rng = np.random.default_rng(9)
lit = "x" * 150 + "y" * 150
buf = bytearray(rng.choice(np.frombuffer(b"xy ab\n", np.uint8), 4 << 20).tobytes())
for at in (1000, 70000, 3 << 20, (4 << 20) - 300):
    buf[at:at + 300] = lit.encode()
buf[500000:500000 + 299] = lit.encode()[:299]
open("/tmp/lw.txt", "wb").write(buf)
ok = True
for pat, flags in [(lit, ["-O", "-l"]), ("x{150}y{150}", ["-O", "-l"]), ("[xy]{280,}", ["-O", "-l"]), (lit, ["-S", "-O", "-l"])]:
    a = subprocess.run([R + "/grab_amd/bin/grab"] + flags + [pat, "/tmp/lw.txt"], capture_output=True)
    b = subprocess.run([R + "/oracle/grab_oracle"] + [f for f in flags if f != "-S"] + [pat, "/tmp/lw.txt"], capture_output=True)
    same = a.stdout == b.stdout and a.returncode == b.returncode == 0
    ok = ok and same
    print(pat[:20], flags, "lines", a.stdout.count(b"\n"), "same" if same else ("DIFF %r %r" % (a.stdout[:200], a.stderr[:200])))
print("ALL SAME" if ok else "MISMATCH")
PY
