#!/usr/bin/env python3
"""End-to-end sweep of the host -> HBM ingest knobs on the GPU box (VERDICT r1 task 4: lift the 8-MiB-piece DMA cap).

One corpus per shape, written to /dev/shm once:
    cfg2   --gib G of 64 MiB files, one needle per file             grab -n 8 -r NEEDLE
    cfg4   --small-gib S of 512 KiB files in a 32 x 32 tree          grab -n 8 -r NEEDLE   (batches, the walk)
    cfg5   one file of --single-gib GiB, needles every ~32 KiB       grab -O -l NEEDLE     (one worker, default 1 GiB windows)
For each: the reference binary (`oracle/_ref/grab_jit -n <cores> -r`, 1 core for cfg5) and `grab` under every combination
of GSCAN_BLOCK_MIB x GSCAN_READERS x GSCAN_COPY_STREAMS asked for; wall clock of the whole process, page cache warm, min of
--reps; output compared with the reference's (sorted).  One JSON line per measurement on stdout.
"""
import argparse
import itertools
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
CHILD_MASK = None


def unpin():
    if CHILD_MASK:
        os.sched_setaffinity(0, CHILD_MASK)


def timed(argv, reps, env=None):
    best, out, err = None, b"", b""
    path = "/dev/shm/grab_sweep_out_%d.txt" % os.getpid()  # (a file, not a pipe: reading 10^8 lines from a pipe is this script's time, not the program's)
    for it in range(reps + 1):  # pass 0 warms the page cache
        with open(path, "wb") as o:
            t0 = time.perf_counter()
            r = subprocess.run(argv, stdout=o, stderr=subprocess.PIPE, env=env, preexec_fn=unpin)
            dt = time.perf_counter() - t0
        with open(path, "rb") as f:
            stdout = f.read()
        os.unlink(path)
        if r.returncode != 0:
            return None, stdout, r.stderr
        if it > 0 and (best is None or dt < best):
            best, out, err = dt, stdout, r.stderr
    return best, out, err


def gen_files(base, files, file_bytes, fan, needles_every=0):
    import torch

    dev = torch.device("cuda", 0)
    nd = np.frombuffer(synth.NEEDLE, np.uint8)
    block_files = max(1, (64 << 20) // file_bytes)
    t0 = time.perf_counter()
    for lo in range(0, files, block_files):
        n = min(block_files, files - lo)
        big = synth.torch_text(n * file_bytes, lo, dev).cpu().numpy()
        for j in range(n):
            i = lo + j
            d = os.path.join(base, "d%02d" % (i % fan), "s%02d" % ((i // fan) % fan)) if fan > 1 else base
            if i < fan * fan or fan <= 1:
                os.makedirs(d, exist_ok=True)
            buf = big[j * file_bytes:(j + 1) * file_bytes]
            if needles_every:
                for at in range(1000, file_bytes - 64, needles_every):
                    buf[at:at + nd.size] = nd
            else:
                at = (i * 7919) % max(1, file_bytes - 64)
                buf[at:at + nd.size] = nd
            buf.tofile(os.path.join(d, "f%06d.txt" % i))
    return time.perf_counter() - t0


def ref_cores():
    allowed = sorted(os.sched_getaffinity(0))
    cores = 0
    while cores < len(allowed) and allowed[cores] == cores:
        cores += 1
    return max(2, min(cores, 64))


def sweep(tag, base, nbytes, grab_argv, ref_argv, combos, reps, serial_ref=False):
    ref_out = None
    if os.path.exists(REF):
        dt, ref_out, _ = timed(ref_argv, 1 if serial_ref else reps)
        print(json.dumps({"shape": tag, "who": "reference", "argv": " ".join(ref_argv[1:-1]), "s": dt and round(dt, 3), "GBps": dt and round(nbytes / dt / 1e9, 2),
                          "lines": ref_out.count(b"\n")}), flush=True)
    for blk, rd, cs, extra in combos:
        env = dict(os.environ, GSCAN_BLOCK_MIB=str(blk), GSCAN_READERS=str(rd), GSCAN_COPY_STREAMS=str(cs))
        env.update(extra)
        dt, out, err = timed(grab_argv, reps, env)
        same = None
        if dt is not None and ref_out is not None:
            same = sorted(out.splitlines()) == sorted(ref_out.splitlines())
        tr = subprocess.run(grab_argv, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(env, GRAB_TIMING="1", GSCAN_TIMING="1", GRAB_CLOSE="1"), preexec_fn=unpin)
        lines = tr.stderr.decode("latin-1").splitlines()
        timing = [ln for ln in lines if ln.startswith("[gscan timing] device")][-1:] + [ln for ln in lines if ln.startswith("[grab timing] +")]
        print(json.dumps({"shape": tag, "who": "grab", "block_mib": blk, "readers": rd, "copy_streams": cs, "env": extra, "s": dt and round(dt, 3),
                          "GBps": dt and round(nbytes / dt / 1e9, 2), "same_as_reference": same, "lines": out.count(b"\n") if dt else err.decode("latin-1")[-300:],
                          "timing": timing}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=int, default=64)
    ap.add_argument("--small-gib", type=int, default=16)
    ap.add_argument("--single-gib", type=int, default=8)
    ap.add_argument("--blocks", default="8,16,32")
    ap.add_argument("--readers", default="8,16")
    ap.add_argument("--streams", default="1,2")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--corpus-cpus", default="", help="'local' / 'remote' / a cpulist: where the corpus writer runs, i.e. which NUMA node first-touches the tmpfs pages")
    ap.add_argument("--extra-env", default="", help="';'-separated sets of K=V,K=V run at the first combination in addition (cfg2 only)")
    a = ap.parse_args()
    if a.corpus_cpus:
        from grab_amd import engine
        local = set(engine.parse_cpulist(engine.device_cpulist(0) or ""))
        allowed = os.sched_getaffinity(0)
        want = local & allowed if a.corpus_cpus == "local" else allowed - local if a.corpus_cpus == "remote" else set(engine.parse_cpulist(a.corpus_cpus)) & allowed
        print(json.dumps({"corpus_cpus": a.corpus_cpus, "n": len(want), "local": engine.device_cpulist(0)}), flush=True)
        if want:
            os.sched_setaffinity(0, want)  # (the children are started with the full mask again: see timed())
            global CHILD_MASK
            CHILD_MASK = allowed
    combos = [(b, r, c, {}) for b, r, c in itertools.product([int(x) for x in a.blocks.split(",")], [int(x) for x in a.readers.split(",")], [int(x) for x in a.streams.split(",")])]
    cores = ref_cores()
    needle = synth.NEEDLE.decode()
    base = "/dev/shm/grab_sweep_%d" % os.getpid()
    os.makedirs(base)
    try:
        if a.gib:
            d = os.path.join(base, "cfg2")
            files = a.gib * 16
            g = gen_files(d, files, 64 << 20, 1)
            print(json.dumps({"shape": "cfg2", "files": files, "gen_s": round(g, 1)}), flush=True)
            sweep("cfg2", d, files * (64 << 20), [bin_path(), "-n", str(a.workers), "-r", needle, d], [REF, "-n", str(cores), "-r", needle, d], combos, a.reps)
            # NUMA placement and pinning variants at the middle combination
            mid = combos[len(combos) // 2][:3]
            sweep("cfg2", d, files * (64 << 20), [bin_path(), "-n", str(a.workers), "-r", needle, d], [REF, "-n", str(cores), "-r", needle, d],
                  [mid + ({"GSCAN_NUMA": "0", "GRAB_PIN": "cpu"},), mid + ({"GRAB_PIN": "none", "GSCAN_NUMA": "0"},), mid + ({"GRAB_WALKERS": "1"},)], a.reps)
            for spec in [x for x in a.extra_env.split(";") if x]:
                extra = dict(kv.split("=", 1) for kv in spec.split(","))
                sweep("cfg2", d, files * (64 << 20), [bin_path(), "-n", str(a.workers), "-r", needle, d], [REF, "-n", str(cores), "-r", needle, d], [combos[0][:3] + (extra,)], a.reps)
            sweep("cfg2-n16", d, files * (64 << 20), [bin_path(), "-n", "16", "-r", needle, d], [REF, "-n", str(cores), "-r", needle, d], [mid + ({},)], a.reps)
            shutil.rmtree(d, ignore_errors=True)
        if a.small_gib:
            d = os.path.join(base, "cfg4")
            files = a.small_gib * 2048
            g = gen_files(d, files, 512 << 10, 32)
            print(json.dumps({"shape": "cfg4", "files": files, "gen_s": round(g, 1)}), flush=True)
            few = [c for c in combos if c[1] == combos[0][1]]
            sweep("cfg4", d, files * (512 << 10), [bin_path(), "-n", str(a.workers), "-r", needle, d], [REF, "-n", str(cores), "-r", needle, d],
                  few + [few[0][:3] + ({"GRAB_WALKERS": "1"},), few[0][:3] + ({"GRAB_WALKERS": "8"},)], a.reps)
            sweep("cfg4-n16", d, files * (512 << 10), [bin_path(), "-n", "16", "-r", needle, d], [REF, "-n", str(cores), "-r", needle, d], few[:1], a.reps)
            shutil.rmtree(d, ignore_errors=True)
        if a.single_gib:
            d = os.path.join(base, "cfg5")
            g = gen_files(d, 1, a.single_gib << 30, 1, needles_every=32768 + 77)
            print(json.dumps({"shape": "cfg5", "gen_s": round(g, 1)}), flush=True)
            f = os.path.join(d, "f000000.txt")
            sweep("cfg5", d, a.single_gib << 30, [bin_path(), "-O", "-l", needle, f], [REF, "-O", "-l", needle, f], combos, a.reps, serial_ref=True)
            sweep("cfg5-3ctx", d, a.single_gib << 30, [bin_path(), "-O", "-l", needle, f], [REF, "-O", "-l", needle, f], [combos[len(combos) // 2][:3] + ({"GRAB_DEVICES": "3"},)], a.reps, serial_ref=True)
    finally:
        shutil.rmtree(base, ignore_errors=True)


if __name__ == "__main__":
    main()
