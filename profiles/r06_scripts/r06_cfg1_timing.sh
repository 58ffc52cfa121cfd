# Round 6, session J: BASELINE configs[0] -- one 256 MiB file, the literal that is not in it -- wall clock of `grab` x 7 and the
# engine's trace of one run (where the time between "context open" and "scan done" goes)
F=/dev/shm/r06_cfg1.txt
python - <<'PY'
import sys, torch
sys.path.insert(0, '.')
from grab_amd import synth
synth.torch_text(256 << 20, 0, torch.device('cuda', 0)).cpu().numpy().tofile('/dev/shm/r06_cfg1.txt')
PY
G=grab_amd/bin/grab
$G foobardoesnotexist $F > /dev/null
python - <<'PY'
import subprocess, time
ts = []
for k in range(9):
    time.sleep(0.5)
    t0 = time.perf_counter(); subprocess.run(['grab_amd/bin/grab', 'foobardoesnotexist', '/dev/shm/r06_cfg1.txt'], stdout=subprocess.DEVNULL); ts.append(time.perf_counter() - t0)
print("grab foobardoesnotexist <256 MiB>: wall s", " ".join("%.4f" % t for t in sorted(ts)), " min %.4f median %.4f" % (min(ts), sorted(ts)[len(ts)//2]))
ts = []
for k in range(5):
    t0 = time.perf_counter(); subprocess.run(['oracle/_ref/grab_jit', 'foobardoesnotexist', '/dev/shm/r06_cfg1.txt'], stdout=subprocess.DEVNULL); ts.append(time.perf_counter() - t0)
print("reference, one core: min %.4f" % min(ts))
PY
sleep 0.5
GRAB_TIMING=1 GSCAN_TRACE=1 $G foobardoesnotexist $F 2>&1 >/dev/null | grep -v "gscan_open:" | cut -c1-160
rm -f $F
