#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) rate of the drop-in `grab` binary next to the reference binary, on the GPU box.

Builds a corpus of synthetic text files (SURVEY.md 8d alphabet) in /dev/shm -- `--files N --file-kib K`, optionally
spread over a directory tree -- with the needle planted once per file, then times
    grab_amd/bin/grab -n W -r PATTERN DIR        for W in --workers
    oracle/_ref/grab_jit -n C -r PATTERN DIR     (C = host cores the reference can pin, capped at --ref-cores)
(warm page cache: one untimed pass first; min of --reps).  Output is compared (sorted) between the two.
This is NOT bench.py's `value` (that one is HBM-resident); it is the number DESIGN.md quotes as the PCIe-inclusive rate.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grab_amd import bin_path, synth  # noqa: E402


def build_tree(base, files, file_bytes, fanout):
    import torch

    dev = torch.device("cuda", 0)
    nd = np.frombuffer(synth.NEEDLE, np.uint8)
    t0 = time.perf_counter()
    for i in range(files):
        d = os.path.join(base, "d%03d" % (i % fanout), "s%03d" % ((i // fanout) % fanout)) if fanout > 1 else base
        if i < fanout * fanout or fanout <= 1:
            os.makedirs(d, exist_ok=True)
        buf = synth.torch_text(file_bytes, i, dev).cpu().numpy()
        at = (i * 7919) % max(1, file_bytes - 64)
        buf[at:at + nd.size] = nd
        buf.tofile(os.path.join(d, "f%06d.txt" % i))
    return time.perf_counter() - t0


def timed(argv, reps):
    best, out = None, b""
    path = "/dev/shm/grab_e2e_cli_out_%d.txt" % os.getpid()  # (a file, not a pipe: reading 10^7 lines through a pipe is this script's time)
    for it in range(reps + 1):  # pass 0 warms the page cache
        with open(path, "wb") as o:
            t0 = time.perf_counter()
            r = subprocess.run(argv, stdout=o, stderr=subprocess.PIPE)
            dt = time.perf_counter() - t0
        with open(path, "rb") as f:
            stdout = f.read()
        os.unlink(path)
        if r.returncode != 0:
            return None, r.stderr[-300:]
        out = stdout
        if it > 0:
            best = dt if best is None else min(best, dt)
    return best, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=256)
    ap.add_argument("--file-kib", type=int, default=65536)
    ap.add_argument("--fanout", type=int, default=1)
    ap.add_argument("--pattern", default=synth.NEEDLE.decode())
    ap.add_argument("--flags", default="-O -l")
    ap.add_argument("--workers", default="1,4,16,32")
    ap.add_argument("--ref-cores", type=int, default=64)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    base = "/dev/shm/grab_e2e_%d" % os.getpid()
    os.makedirs(base)
    try:
        file_bytes = a.file_kib << 10
        gen_s = build_tree(base, a.files, file_bytes, a.fanout)
        nbytes = a.files * file_bytes
        flags = a.flags.split()
        res = {"tag": a.tag, "files": a.files, "file_kib": a.file_kib, "bytes": nbytes, "pattern": a.pattern, "flags": a.flags,
               "gen_s": round(gen_s, 1), "grab": {}, "reference": None}
        ref = os.path.join(ROOT, "oracle", "_ref", "grab_jit")
        ref_out = None
        if os.path.exists(ref):
            allowed = sorted(os.sched_getaffinity(0))
            cores = 0
            while cores < len(allowed) and allowed[cores] == cores:
                cores += 1
            cores = max(2, min(cores, a.ref_cores))
            dt, ref_out = timed([ref, "-n", str(cores), "-r"] + flags + [a.pattern, base], a.reps)
            res["reference"] = {"cores": cores, "s": dt and round(dt, 3), "GBps": dt and round(nbytes / dt / 1e9, 2)}
        for w in [int(x) for x in a.workers.split(",")]:
            argv = [bin_path()] + (["-n", str(w)] if w > 1 else []) + ["-r"] + flags + [a.pattern, base]
            dt, out = timed(argv, a.reps)
            same = None
            if dt is not None and ref_out is not None:
                same = sorted(out.splitlines()) == sorted(ref_out.splitlines())
            tr = subprocess.run(argv, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, GRAB_TIMING="1", GSCAN_TIMING="1", GRAB_CLOSE="1"))
            lines = tr.stderr.decode("latin-1").splitlines()
            timing = [ln[14:] for ln in lines if ln.startswith("[grab timing]")][:2] + [ln[15:] for ln in lines if ln.startswith("[gscan timing]")][-1:]
            res["grab"][str(w)] = {"timing": timing, "s": dt and round(dt, 3), "GBps": dt and round(nbytes / dt / 1e9, 2), "lines": out.count(b"\n") if dt else out.decode("latin-1"),
                                   "same_as_reference": same}
        print(json.dumps(res), flush=True)
    finally:
        shutil.rmtree(base, ignore_errors=True)


if __name__ == "__main__":
    main()
