#!/usr/bin/env python3
"""Parity at BASELINE.json's full sizes, on the GPU box: the drop-in `grab` against the reference binary
(oracle/_ref/grab_jit) on the same files in /dev/shm, outputs compared (byte-exact for single-threaded runs,
sorted for -n runs: the reference's own criterion, README.md:206-216).  Also records wall-clock of both.

    cfg2  1024 x 64 MiB, literal needle planted 64x per file, -n -r -O -l
    cfg3  the first --cfg3-files of the same corpus, identifier regex, -n -r -O -l   (full size would print ~4 GB per run)
    cfg4  131072 x 512 KiB in a 64x64x32 tree, one needle per file, -n -r -O -l
    cfg5  one 32 GiB file, 1e6 seeded needles + plants in every overlap window / across every chunk end / ending at a
          chunk end / at every chunk start / in the last 18 bytes; -O -l at 1 GiB chunks and at 32 MiB chunks (-L x5)
"""
import argparse
import hashlib
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

REF = os.path.join(ROOT, "oracle", "_ref", "grab_jit")
NEEDLE = synth.NEEDLE


def run(argv, sort=False):
    t0 = time.perf_counter()
    if sort:  # stream through sort(1): outputs of the threaded modes are big
        p1 = subprocess.Popen(argv, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        p2 = subprocess.Popen(["sort", "--parallel=32", "-S", "8G"], stdin=p1.stdout, stdout=subprocess.PIPE, env=dict(os.environ, LC_ALL="C"))
        p1.stdout.close()
        h, n = hashlib.md5(), 0
        first = True
        for blk in iter(lambda: p2.stdout.read(1 << 24), b""):
            if first:
                dt = time.perf_counter() - t0  # ~ time until sort starts to emit = producer finished
                first = False
            h.update(blk)
            n += blk.count(b"\n")
        p1.wait()
        p2.wait()
        if first:
            dt = time.perf_counter() - t0
        return p1.returncode, h.hexdigest(), n, dt
    r = subprocess.run(argv, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    dt = time.perf_counter() - t0
    return r.returncode, hashlib.md5(r.stdout).hexdigest(), r.stdout.count(b"\n"), dt


def timed_only(argv):
    t0 = time.perf_counter()
    r = subprocess.run(argv, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return r.returncode, time.perf_counter() - t0


def compare(name, ours, ref, sort, nbytes, res):
    rc1, h1, n1, _ = run(ours, sort)
    rc2, h2, n2, _ = run(ref, sort)
    # clean timings with the output discarded (the reference's numbers are taken the same way: BASELINE.md section 3)
    t1 = min(timed_only(ours)[1] for _ in range(2))
    t2 = min(timed_only(ref)[1] for _ in range(2))
    res[name] = {"lines": n1, "ref_lines": n2, "same": (rc1, h1, n1) == (rc2, h2, n2) and rc1 == 0, "md5": h1,
                 "grab_s": round(t1, 3), "ref_s": round(t2, 3), "grab_GBps": round(nbytes / t1 / 1e9, 2), "ref_GBps": round(nbytes / t2 / 1e9, 2),
                 "cmd": " ".join(os.path.basename(a) if a.startswith("/") and os.path.isfile(a) else a for a in ours)}
    print(name, json.dumps(res[name]), flush=True)


def gen_files(base, files, file_bytes, needles, tree=None):
    import torch

    dev = torch.device("cuda", 0)
    nd = np.frombuffer(NEEDLE, np.uint8)
    for i in range(files):
        if tree:
            a, b, c = tree
            d = os.path.join(base, "a%02d" % (i % a), "b%02d" % ((i // a) % b))
            if i < a * b:
                os.makedirs(d, exist_ok=True)
        else:
            d = base
        buf = synth.torch_text(file_bytes, i, dev).cpu().numpy()
        if needles:
            synth.plant(buf, NEEDLE, needles, i)
        buf.tofile(os.path.join(d, "f%06d.txt" % i))


def gen_big(path, size, chunk, n_plants):
    """cfg5 file: synthetic stream + 1e6 seeded plants + the boundary plants of SURVEY.md 8d."""
    import torch

    dev = torch.device("cuda", 0)
    L = len(NEEDLE)
    rng = np.random.default_rng(0xC0FFEE)
    plants = np.sort(rng.integers(0, size - L, n_plants))
    stride = chunk - 4096
    extra = []
    off = stride
    while off < size:  # chunk k starts at k*stride, the previous one ends at (k-1)*stride + chunk = off + 4096
        end_prev = off + 4096
        extra += [off, off + 1000, off + 4096 - L, end_prev - 7, end_prev - L]  # chunk start / inside the overlap / ends at the overlap end / straddles the chunk end / ends exactly at the chunk end
        off += stride
    extra += [size - L, size - L - 3]
    plants = np.unique(np.concatenate([plants, np.array([e for e in extra if 0 <= e <= size - L], np.int64)]))
    nd = np.frombuffer(NEEDLE, np.uint8)
    blk = 1 << 30
    with open(path, "wb") as f:
        for k, base in enumerate(range(0, size, blk)):
            n = min(blk, size - base)
            buf = synth.torch_text(n, 5000 + k, dev).cpu().numpy()
            lo = np.searchsorted(plants, base - L + 1)
            hi = np.searchsorted(plants, base + n)
            for p in plants[lo:hi]:
                a, b = max(p, base), min(p + L, base + n)
                buf[a - base:b - base] = nd[a - p:b - p]
            buf.tofile(f)
    return len(plants)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0, help="shrink every size by this factor (smoke runs)")
    ap.add_argument("--cfg3-files", type=int, default=128)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--only", default="cfg2,cfg3,cfg4,cfg5")
    a = ap.parse_args()
    cores = 0
    allowed = sorted(os.sched_getaffinity(0))
    while cores < len(allowed) and allowed[cores] == cores:
        cores += 1
    cores = max(2, min(cores, 64))
    res = {"ref_cores": cores, "workers": a.workers}
    base = "/dev/shm/grab_full_%d" % os.getpid()
    W = ["-n", str(a.workers)]
    try:
        if "cfg2" in a.only or "cfg3" in a.only:
            d = os.path.join(base, "c2")
            os.makedirs(d)
            files = max(8, int(1024 * a.scale))
            t0 = time.perf_counter()
            gen_files(d, files, 64 << 20, 64)
            res["cfg2_gen_s"] = round(time.perf_counter() - t0, 1)
            nbytes = files * (64 << 20)
            if "cfg2" in a.only:
                compare("cfg2_literal_%dx64MiB" % files, [bin_path()] + W + ["-r", "-O", "-l", NEEDLE.decode(), d],
                        [REF, "-n", str(cores), "-r", "-O", "-l", NEEDLE.decode(), d], True, nbytes, res)
            if "cfg3" in a.only:
                d3 = os.path.join(base, "c3")
                os.makedirs(d3)
                n3 = min(files, a.cfg3_files)
                for i in range(n3):
                    os.link(os.path.join(d, "f%06d.txt" % i), os.path.join(d3, "f%06d.txt" % i))
                compare("cfg3_ident_%dx64MiB" % n3, [bin_path()] + W + ["-r", "-O", "-l", synth.IDENT_RE, d3],
                        [REF, "-n", str(cores), "-r", "-O", "-l", synth.IDENT_RE, d3], True, n3 * (64 << 20), res)
            shutil.rmtree(base, ignore_errors=True)
        if "cfg4" in a.only:
            d = os.path.join(base, "c4")
            os.makedirs(d)
            files = max(64, int(131072 * a.scale))
            t0 = time.perf_counter()
            gen_files(d, files, 512 << 10, 1, tree=(64, 64, 32))
            res["cfg4_gen_s"] = round(time.perf_counter() - t0, 1)
            compare("cfg4_tree_%dx512KiB" % files, [bin_path()] + W + ["-r", "-O", "-l", NEEDLE.decode(), d],
                    [REF, "-n", str(cores), "-r", "-O", "-l", NEEDLE.decode(), d], True, files * (512 << 10), res)
            shutil.rmtree(base, ignore_errors=True)
        if "cfg5" in a.only:
            os.makedirs(base, exist_ok=True)
            size = int((32 << 30) * a.scale) if a.scale >= 1 else max(int((32 << 30) * a.scale), (3 << 30) + 12345)
            p = os.path.join(base, "big.bin")
            t0 = time.perf_counter()
            res["cfg5_plants"] = gen_big(p, size, 1 << 30, int(1_000_000 * min(a.scale, 1.0)))
            res["cfg5_gen_s"] = round(time.perf_counter() - t0, 1)
            compare("cfg5_%dGiB_1GiB_chunks" % (size >> 30), [bin_path(), "-O", "-l", NEEDLE.decode(), p], [REF, "-O", "-l", NEEDLE.decode(), p], False, size, res)
            L5 = ["-L"] * 5
            compare("cfg5_%dGiB_32MiB_chunks" % (size >> 30), [bin_path()] + L5 + ["-O", "-l", NEEDLE.decode(), p], [REF] + L5 + ["-O", "-l", NEEDLE.decode(), p], False, size, res)
            compare("cfg5_%dGiB_lines" % (size >> 30), [bin_path(), "-O", NEEDLE.decode(), p], [REF, "-O", NEEDLE.decode(), p], False, size, res)
    finally:
        shutil.rmtree(base, ignore_errors=True)
    res["all_same"] = all(v["same"] for v in res.values() if isinstance(v, dict))
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
