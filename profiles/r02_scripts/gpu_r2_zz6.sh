#!/bin/bash
# Round 2, last seconds of GPU time: the device VM with the new guard op -- conditional-group patterns and a reference at the head
# of an alternative, `grab` against the oracle on one file.
set -u
mkdir -p gpurun_out
python - <<'PY' 2>&1 | tee gpurun_out/zz6_vm_cond.txt
import os, subprocess, sys, numpy as np
sys.path.insert(0, os.getcwd())
from grab_amd import bin_path, engine
rng = np.random.default_rng(5)
alpha = np.frombuffer(b"abcdef 01\n", np.uint8)
data = alpha[rng.integers(0, alpha.size, 400_000)]
for k, w in enumerate([b"abde", b"ce", b"abeef", b"cdef", b"bcd", b"abccdd", b"adf", b"aab", b"acde", b"aabde"]):
    for r in range(20):
        at = 1000 + 997 * (k * 20 + r)
        data[at:at + len(w)] = np.frombuffer(w, np.uint8)
os.makedirs("/tmp/zz6", exist_ok=True)
data.tofile("/tmp/zz6/f")
ok = n = 0
for pat in [r"(a)?(?(1)b|c)d*e", r"x?(a)?(?(1)b|c)d", r"(?(?=a)ab|cd)e+f+", r"(?(?!a)b|ab)c+d+", r"(?:a|(b))(?(1)c|d)e?f", r"(?<n>a)?(?(<n>)b)c+d", r"a(?(?!(b))c)d*e|ab", r"(a)(?:\1b|c)d*e*", r"(\w)\1{3,}x|foobardoes(?=not)"]:
    db = engine.Database(pat)
    for flags in (["-O", "-l"], []):
        a = subprocess.run([bin_path()] + flags + [pat, "f"], cwd="/tmp/zz6", capture_output=True)
        b = subprocess.run([os.path.join(os.getcwd(), "oracle", "grab_oracle")] + flags + [pat, "f"], cwd="/tmp/zz6", capture_output=True)
        n += 1; same = (a.returncode, a.stdout) == (b.returncode, b.stdout); ok += same
        print(pat, flags, "vm", db.info.vm, "same" if same else "DIFF", a.returncode, len(a.stdout), len(b.stdout), a.stderr[:80])
print("same", ok, "of", n)
PY
