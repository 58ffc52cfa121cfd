import sys, json
sys.path.insert(0, '/root/repo')
import bench
pats = [bench.CONFIGS[k][0] for k in sorted(bench.CONFIGS)]
import subprocess, os, shutil, tempfile, glob
sweep = os.path.join(os.path.dirname(bench.bin_path()), "gscan_sweep")
rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
base = tempfile.mkdtemp(prefix="dbg_", dir="/tmp")
argv = [rocprof, "--pmc", "FETCH_SIZE", "--output-format", "csv", "-d", base, "--", sweep, "--gib", "4", "--iters", "1", "--variants", "-1", "--bpc", "0"]
for p in pats: argv += ["--pattern", p]
r = subprocess.run(argv, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, timeout=200)
print("rc", r.returncode); print(r.stdout[-1500:].decode()); print(r.stderr[-1500:].decode())
files = glob.glob(os.path.join(base, "**", "*counter_collection.csv"), recursive=True)
print(files)
import csv
if files:
    names = {}
    for row in csv.DictReader(open(files[0])):
        names.setdefault(row["Kernel_Name"][:90], 0); names[row["Kernel_Name"][:90]] += 1
    print(names)
print("live_traffic", bench.live_traffic(pats, 4))
print("live_sq", bench.live_sq(pats, 4))
