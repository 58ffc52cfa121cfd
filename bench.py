#!/usr/bin/env python3
"""bench.py -- GB/s scanned + match offsets/s of the gfx950 scan engine on the synthetic corpus
BASELINE.json names, with the inputs resident in HBM.

    python bench.py [--gpus N --steps K --warmup W]            (N=1: plain python; N>1: torch.distributed.run)

Workload (default = BASELINE.json configs[1]): 1024 files x 64 MiB of synthetic text per GPU
(SURVEY.md 8d alphabet, generated on the device), literal needle 'foobardoesnotexist' planted
64x per file; one "step" = one pass of the scan kernel over the whole 64 GiB arena (one launch,
1024 segments, candidate offsets compacted into HBM).  `--config cfg3` switches to the
identifier regex (class-run kernel), `--config alt` to a 3-way alternation (bucket-filter
kernel; not a BASELINE config).  Multi-GPU: every rank scans its own corpus (files are
independent units; no data-path collective) -> "scaling": "weak".

One JSON line on rank 0: the driver's contract fields + "roofline" (algorithmic bytes per launch
/ mean kernel time from HIP events on the launch stream, vs the 8 TB/s HBM peak) +
"cpu_baseline" (the reference binary oracle/_ref/grab_jit, or the oracle port, timed on this
box's host cores on a bounded sample of the same corpus; N=1, rank 0 only).
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from grab_amd import engine, synth  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy ceiling)
REC_BYTES = 4           # one u32 candidate start per record

CONFIGS = {
    # name: (pattern, needles planted per file, record capacity per GiB of text)
    "cfg2": (synth.NEEDLE.decode(), 64, 1 << 14),          # BASELINE configs[1]: literal needle (K1)
    "cfg3": (synth.IDENT_RE, 0, 12 << 20),                 # BASELINE configs[2]: identifier regex (K2)
    "alt": (synth.NEEDLE.decode() + "|Linus|555-1234", 64, 1 << 14),  # not a BASELINE config: an alternation (K3), for its roofline line
}


def shard(n_items, rank, world):
    """Items (files) of a shared work list owned by `rank`: round-robin, like the reference's
    thread striping (main.cc:94).  Used when one corpus is split; the default weak-scaling run
    gives every rank its own full corpus instead."""
    return list(range(rank, n_items, world))


def dist_setup(n_gpus):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend, rank=rank, world_size=world)
    if n_gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (n_gpus, world, n_gpus))
    return rank, world, local


def barrier(world, device):
    if world > 1:
        import torch.distributed as dist

        dist.barrier(device_ids=[device.index] if device.type == "cuda" else None)


def reduce_max(value, world, device):
    """MAX over ranks of a python float (the slowest rank defines the step time)."""
    if world == 1:
        return value
    import torch.distributed as dist

    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_sum(value, world, device):
    if world == 1:
        return value
    import torch.distributed as dist

    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def build_corpus(files, file_bytes, needles, rank, device):
    """files x file_bytes of synthetic text in one HBM arena + the planted needle offsets per file."""
    arena = torch.empty(files * file_bytes + 4096, dtype=torch.uint8, device=device)
    arena[files * file_bytes:] = 0
    nd = torch.frombuffer(bytearray(synth.NEEDLE), dtype=torch.uint8).to(device)
    L = nd.numel()
    plants = []
    for i in range(files):
        k = rank * files + i
        view = arena[i * file_bytes:(i + 1) * file_bytes]
        view.copy_(synth.torch_text(file_bytes, k, device))
        if needles:
            rng = np.random.default_rng((synth.SEED0 + k) ^ 0x5EED)
            gap = 600
            slot = (file_bytes - 2 * gap) // needles
            offs = np.array([gap + j * slot + int(rng.integers(0, slot - L - gap)) for j in range(needles)], np.int64)
            idx = (torch.from_numpy(offs).to(device)[:, None] + torch.arange(L, device=device)[None, :]).reshape(-1)
            view[idx] = nd.repeat(needles)
            plants.append(offs)
    torch.cuda.synchronize(device)
    return arena, plants


def measured_traffic(config, nbytes):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
    (profiles/*pmc_traffic.json; PMC cannot be collected from inside the run).  None if the workload
    differs from the profiled one."""
    best = None
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles"))) if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
        if name.endswith("pmc_traffic.json"):
            try:
                rec = json.load(open(os.path.join(ROOT, "profiles", name))).get(config)
            except (OSError, ValueError):
                continue
            if rec and rec.get("workload_bytes") == nbytes:
                best = int(rec["traffic_bytes"])
    return best


def cpu_baseline(arena, files, file_bytes, pattern, flags, want_gib=8):
    """Time the reference (or the oracle port) on this box's host cores over a bounded sample."""
    ref = os.path.join(ROOT, "oracle", "_ref", "grab_jit")
    port = os.path.join(ROOT, "oracle", "grab_oracle")
    allowed = sorted(os.sched_getaffinity(0))
    cores = 0
    while cores < len(allowed) and allowed[cores] == cores:  # the reference pins thread i to CPU i (main.cc:200-215)
        cores += 1
    cores = max(1, min(cores, 64))
    if os.path.exists(ref):
        kind, binary = "reference", ref
    elif os.path.exists(port):
        kind, binary, cores = "port", port, 1
    else:
        return None
    per_core_files = max(1, (1 << 30) // file_bytes)
    nfiles = min(files, per_core_files * cores, max(1, (want_gib << 30) // file_bytes))
    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > nfiles * file_bytes * 1.2 else "/tmp"
    d = os.path.join(base, "grab_bench_%d" % os.getpid())
    os.makedirs(d, exist_ok=True)
    try:
        for i in range(nfiles):
            arena[i * file_bytes:(i + 1) * file_bytes].cpu().numpy().tofile(os.path.join(d, "f%04d.txt" % i))
        argv = [binary] + (["-n", str(cores)] if cores > 1 else []) + ["-r"] + flags + [pattern, d]
        best, lines = None, 0
        for it in range(3):  # first pass warms the page cache, min of the next two
            t0 = time.perf_counter()
            r = subprocess.run(argv, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
            dt = time.perf_counter() - t0
            if r.returncode != 0:
                return None
            lines = r.stdout.count(b"\n")
            if it > 0:
                best = dt if best is None else min(best, dt)
        nbytes = nfiles * file_bytes
        return {"value": round(nbytes / best / 1e9, 3), "unit": "GB/s", "cores": cores, "kind": kind,
                "sample": "%d x %d MiB files of the same corpus in %s, '%s', warm cache, min of 2" % (nfiles, file_bytes >> 20, base, " ".join(os.path.basename(a) if a == binary else a for a in argv[:-1])),
                "matches_per_s": round(lines / best, 1), "engine": "libpcre 8.39 JIT (pcre_exec); the -H hyperscan path does not exist in the mounted reference"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--files", type=int, default=1024)
    ap.add_argument("--file-mib", type=int, default=64)
    ap.add_argument("--variant", type=int, default=None, help="kernel variant (gscan_set_option)")
    ap.add_argument("--blocks-per-cu", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the scan engine has no CPU path")
    rank, world, local = dist_setup(a.gpus)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    pattern, needles, cap_per_gib = CONFIGS[a.config]
    file_bytes = a.file_mib << 20
    arena, plants = build_corpus(a.files, file_bytes, needles, rank, device)
    nbytes = a.files * file_bytes

    ctx = engine.Context(local, 1 << 30)
    if a.variant is not None:
        ctx.set_option("variant", a.variant)
    if a.blocks_per_cu is not None:
        ctx.set_option("blocks_per_cu", a.blocks_per_cu)
    ctx.set_capacity(max(1 << 16, int(cap_per_gib * nbytes / (1 << 30))))
    db = engine.Database(pattern)
    segs = engine.Context.make_segs([(i * file_bytes, file_bytes) for i in range(a.files)])
    stream = torch.cuda.current_stream(device).cuda_stream

    def step():
        return ctx.scan_device(db, arena.data_ptr(), segs, stream)

    for _ in range(a.warmup):
        res = step()
    total, overflow = ctx.dev_sync(res) if a.warmup else (0, False)
    ctx.kernel_time(reset=True)

    barrier(world, device)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        res = step()
    torch.cuda.synchronize(device)
    barrier(world, device)
    elapsed = time.perf_counter() - t0

    total, overflow = ctx.dev_sync(res)
    kern_ms, launches = ctx.kernel_time(reset=True)
    elapsed = reduce_max(elapsed, world, device)
    matches_all = reduce_sum(float(total), world, device)

    # sanity: the planted needles are exactly what came back (first and last file of this rank)
    check = "ok"
    if overflow:
        check = "record buffer overflow"
    elif needles:
        if total != needles * a.files:
            check = "expected %d matches, got %d" % (needles * a.files, total)
        for i in (0, a.files - 1):
            if not np.array_equal(ctx.dev_fetch(res, i).astype(np.int64), plants[i]):
                check = "planted offsets differ in file %d" % i

    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        alg_bytes = nbytes + REC_BYTES * total  # per launch: every input byte once + one u32 per candidate
        kern_avg_ms = kern_ms / max(launches, 1)
        achieved = alg_bytes / (kern_avg_ms * 1e-3) / 1e9
        line = {
            "metric": "GB/s scanned, 64 GiB synthetic corpus resident in HBM (match offsets/s in matches_per_s)",
            "value": round(world * nbytes / (elapsed / a.steps) / 1e9, 2),
            "unit": "GB/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: %d x %d MiB files per GPU, pattern '%s'%s, one launch over all segments, offsets compacted in HBM" % (
                a.config, a.files, a.file_mib, pattern, (", %d needles planted per file" % needles) if needles else ""),
                "bytes_per_gpu": nbytes, "kernel": {engine.TIER_LITERAL: "K1 anchor scan", engine.TIER_CLASSRUN: "K2 class-run scan"}.get(db.info.tier, "K3 bucket filter"),
                "parallelism": "files sharded per GPU, no collective"},
            "matches_per_step": int(matches_all),
            "matches_per_s": round(matches_all / (elapsed / a.steps), 1),
            "check": check,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": measured_traffic(a.config, nbytes),
                         "kernel_ms": round(kern_avg_ms, 4), "launches": int(launches),
                         "algorithmic_bytes_per_launch": int(alg_bytes)},
        }
        if world == 1 and not a.no_cpu_baseline:
            flags = ["-O", "-l"] if a.config == "cfg3" else []
            line["cpu_baseline"] = cpu_baseline(arena, a.files, file_bytes, pattern, flags)
        print(json.dumps(line), flush=True)

    ctx.close()
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
