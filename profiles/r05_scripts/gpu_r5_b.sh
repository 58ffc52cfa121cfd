#!/bin/bash
# Round 5, session B: (1) engine tests after the survivor queue's rank fix, the read-ahead test; (2) the VM kernel at four
# waves per SIMD (120 VGPRs) against one workgroup per CU (lib_ab: -DGSCAN_VM_WAVES=1); (3) pipe_probe: the engine's way of
# driving the copy streams rebuilt one ingredient at a time; (4) BASELINE configs[0] with and without the read-ahead during
# hipInit; (5) eight PROCESSES on one GPU (one per eighth of the corpus) against one process and against eight device
# indices in one process: does bring-up / teardown run in parallel across runtimes?  (6) the host copy ceiling on distinct
# files against the same bytes under four names each (are the hard links of session A's corpus what made 64 readers slow?)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -3
{
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -q -x 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_filegrep.py -m gpu -q -x -k "read_ahead or returned_once" 2>&1 | tail -4
} | tee gpurun_out/b_pytest.txt
SW=$R/grab_amd/bin/gscan_sweep
{
for rep in 1 2; do
for lib in lib lib_ab; do
  echo "== $lib (lib: GSCAN_VM_WAVES=4, lib_ab: 1)"
  for p in '(\w)\1{3,}x|foobardoes(?=not)' 'a+b+c' '[a-z]+\([a-z0-9, ]*\);'; do
    LD_LIBRARY_PATH=$R/grab_amd/$lib timeout 300 $SW --gib 8 --iters 3 --variants 38 --bpc 0 --pattern "$p" | tail -1
  done
done
done
} 2>&1 | tee gpurun_out/b_vm_sweep.txt
{
timeout 300 grab_amd/bin/pipe_probe --mib 8
timeout 300 grab_amd/bin/pipe_probe --mib 32 --modes 0,2,4,6,7
} 2>&1 | tee gpurun_out/b_pipe_probe.txt
python - <<'PY'
import os, sys
sys.path.insert(0, ".")
import torch, bench
from grab_amd import synth
bench.interleave_page_placement()
dev = torch.device("cuda", 0)
synth.torch_text(256 << 20, 0, dev).cpu().numpy().tofile("/dev/shm/one256.txt")
for i in range(256):
    sub = "/dev/shm/c16/p%d/d%02d" % (i % 8, i % 16)
    os.makedirs(sub, exist_ok=True)
    synth.torch_text(64 << 20, i, dev).cpu().numpy().tofile(sub + "/f%04d.txt" % i)
# the same 16 GiB as 64 distinct files under four names each
for i in range(256):
    sub = "/dev/shm/c16alias/p%d/d%02d" % (i % 8, i % 16)
    os.makedirs(sub, exist_ok=True)
    src = "/dev/shm/c16/p%d/d%02d/f%04d.txt" % ((i % 64) % 8, (i % 64) % 16, i % 64)
    os.link(src, sub + "/f%04d.txt" % i)
PY
G=grab_amd/bin/grab
{
echo "--- cfg1: one 256 MiB file, read ahead during hipInit vs not"
python scripts/ab_run.py --sleep 0.5 --reps 5 --bytes $((256 << 20)) --interleave --env "" --env "GRAB_NO_READ_AHEAD=1" --env "GSCAN_COPY_STREAMS=1" -- $G foobardoesnotexist /dev/shm/one256.txt
GSCAN_TRACE=1 GRAB_TIMING=1 $G foobardoesnotexist /dev/shm/one256.txt 2>&1 >/dev/null | grep -v "task of\|block in hand\|bytes read" | head -60
} 2>&1 | tee gpurun_out/b_cfg1.txt
{
echo "--- 16 GiB: one process / eight indices in one process / eight processes (one per eighth)"
python - <<'PY'
import os, subprocess, time, json, sys
sys.path.insert(0, "scripts")
import n8_model
G = "grab_amd/bin/grab"
def one(env=None, n="8"):
    time.sleep(0.5); t0 = time.perf_counter()
    subprocess.run([G, "-n", n, "-r", "foobardoesnotexist", "/dev/shm/c16"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env)
    return time.perf_counter() - t0
def eight_procs(n="4", env=None):
    time.sleep(0.5); t0 = time.perf_counter()
    ps = [subprocess.Popen([G, "-n", n, "-r", "foobardoesnotexist", "/dev/shm/c16/p%d" % k], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env) for k in range(8)]
    for p in ps: p.wait()
    return time.perf_counter() - t0
env8 = dict(os.environ, GSCAN_VIRTUAL_DEVICES="8", GSCAN_SYSFS_PCI=n8_model.fake_pci_tree("/tmp/fakepci"))
for rep in range(3):
    print(json.dumps({"one_process_s": round(one(), 4), "eight_indices_s": round(one(env8, "32"), 4), "eight_processes_s": round(eight_procs(), 4)}), flush=True)
e2 = dict(os.environ, GSCAN_READERS="2")
print(json.dumps({"eight_processes_2_readers_each_s": round(eight_procs(env=e2), 4), "eight_indices_2_readers_each_s": round(one(dict(env8, GSCAN_READERS="2"), "32"), 4),
                  "eight_indices_1_stream_each_s": round(one(dict(env8, GSCAN_COPY_STREAMS="1"), "32"), 4)}))
PY
echo "--- host copy ceiling (GSCAN_DIAG=1): 256 distinct files vs 64 files under four names each"
python - <<'PY'
import json, sys
sys.path.insert(0, "scripts")
import n8_model
from grab_amd import bin_path
for d in ("/dev/shm/c16", "/dev/shm/c16alias"):
    m = n8_model.measure(bin_path(), d, 256 * (64 << 20), reps=1)
    print(d, json.dumps({k: m.get(k) for k in ("host_copy_GBps_by_readers", "F8_minus_F1_measured_s")}))
PY
} 2>&1 | tee gpurun_out/b_procs.txt
rm -rf /dev/shm/c16 /dev/shm/c16alias /dev/shm/one256.txt
