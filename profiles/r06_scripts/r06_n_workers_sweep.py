"""Round 6, session N: the dense patterns over 16 GiB, output to /dev/null, `-n` 8 / 12 / 16 / 24 / 32 -- how many workers the
host's walk wants once the device settles the matches (min of 3, half a second of quiet before every run)."""
import os, subprocess, sys, time, json
import torch
sys.path.insert(0, os.getcwd())
from grab_amd import synth, bin_path
D = "/dev/shm/r06n"; os.makedirs(D, exist_ok=True)
dev = torch.device("cuda", 0)
for i in range(256):
    synth.torch_text(64 << 20, i, dev).cpu().numpy().tofile("%s/f%04d.txt" % (D, i))
del dev
torch.cuda.empty_cache()
G = bin_path()
out = {}
for p in [r"\b[A-Za-z_]\w*\s*\(", r"\b[a-z]{3,}\b", r"\b[A-Z][a-z]+\b", r"\([^()]*\)", r"(?<=\$)\d+", synth.IDENT_RE]:
    subprocess.run([G, "-n", "8", "-r", "-O", "-l", p, D], stdout=subprocess.DEVNULL)
    row = {}
    for w in (8, 12, 16, 24, 32):
        best = None
        for k in range(3):
            time.sleep(0.5)
            t0 = time.perf_counter(); subprocess.run([G, "-n", str(w), "-r", "-O", "-l", p, D], stdout=subprocess.DEVNULL); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        row[w] = round(best, 3)
    out[p] = row
    print("%-28s" % p, "  ".join("-n %d: %.3f s %5.1f GB/s" % (w, t, (16 << 30) / t / 1e9) for w, t in row.items()), flush=True)
import shutil; shutil.rmtree(D)
