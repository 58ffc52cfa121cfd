set -u
mkdir -p gpurun_out
python - <<'PY'
import os, sys, time
sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
import fullsize_parity
d = "/dev/shm/c4probe"
os.makedirs(d, exist_ok=True)
t0 = time.time(); fullsize_parity.gen_files(d, 32768, 512 << 10, 1, tree=(64, 64, 32)); print("gen", round(time.time() - t0, 1))
PY
for n in 8 16; do
  for rep in 1 2; do
    env GRAB_TIMING=1 GSCAN_TIMING=1 grab_amd/bin/grab -n $n -r -O -l foobardoesnotexist /dev/shm/c4probe 2> gpurun_out/aq_cfg4_n${n}_$rep.err > /dev/null
    tail -1 gpurun_out/aq_cfg4_n${n}_$rep.err
  done
done
grep -v "gscan_open" gpurun_out/aq_cfg4_n8_2.err | head -60
rm -rf /dev/shm/c4probe
