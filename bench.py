#!/usr/bin/env python3
"""bench.py -- GB/s scanned + match offsets/s of the gfx950 scan engine on the synthetic corpus
BASELINE.json names, with the inputs resident in HBM.

    python bench.py [--gpus N --steps K --warmup W]            (N=1: plain python; N>1: torch.distributed.run)
    python bench.py --mode e2e [--gpus N]                       (the work queue end to end; plain python, any N)

Workload (default = BASELINE.json configs[1]): 1024 files x 64 MiB of synthetic text per GPU
(SURVEY.md 8d alphabet, generated on the device), literal needle 'foobardoesnotexist' planted
64x per file; one "step" = one pass of the scan kernel over the whole 64 GiB arena (one launch,
1024 segments, candidate offsets compacted into HBM).  `--config cfg3` switches the headline to the
identifier regex (class-run kernel), `--config alt` to a 3-way alternation (bucket-filter kernel;
not a BASELINE config).  Multi-GPU: every rank scans its own corpus (files are independent units;
no data-path collective) -> "scaling": "weak".

One JSON line on rank 0: the driver's contract fields, plus
  "roofline"      the headline kernel: algorithmic bytes per launch / mean kernel time from HIP events on the
                  launch stream, vs the 8 TB/s HBM peak
  "kernels"       the same block for the OTHER two kernels (cfg3 = BASELINE configs[2], alt), timed in the same
                  process on the same arena, outside the headline's timed region
  "e2e"           the drop-in binary end to end (PCIe-inclusive): the same corpus written to /dev/shm once,
                  `grab -n max(8, N) -r` over the first N devices -- one walk, one queue, files sharded over the
                  GPUs -- wall clock of the whole process, output line count checked, fraction of the
                  63 GB/s-per-GPU PCIe Gen5 x16 roofline, bytes each device was handed.  "scaling": "strong"
  "cpu_baseline"  the reference binary (oracle/_ref/grab_jit), or the oracle port, on this box's host cores over
                  that same on-disk corpus (N=1, rank 0 only)
`--mode e2e` prints only the e2e measurement as the line's value (metric "GB/s end to end").
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from grab_amd import bin_path, engine, synth  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy ceiling)
PCIE_PEAK_GBPS = 63.0   # same guide: PCIe Gen5 x16 per GPU
REC_BYTES = 4           # one u32 candidate start per record

CONFIGS = {
    # name: (pattern, record capacity per GiB of text)
    "cfg2": (synth.NEEDLE.decode(), 1 << 14),                          # BASELINE configs[1]: literal needle (K1)
    "cfg3": (synth.IDENT_RE, 12 << 20),                                # BASELINE configs[2]: identifier regex (K2)
    "alt": (synth.NEEDLE.decode() + "|Linus|555-1234", 1 << 14),       # not a BASELINE config: an alternation (K3), for its roofline line
}
NEEDLES_PER_FILE = 64
KERNEL_NAMES = {engine.TIER_LITERAL: "K1 anchor scan", engine.TIER_CLASSRUN: "K2 class-run scan", engine.TIER_BUCKET: "K3 bucket filter"}


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


def plant_offsets(k, file_bytes, needles):
    """The seeded, well separated needle offsets of corpus file k."""
    L = len(synth.NEEDLE)
    rng = np.random.default_rng((synth.SEED0 + k) ^ 0x5EED)
    gap = 600
    slot = (file_bytes - 2 * gap) // needles
    return np.array([gap + j * slot + int(rng.integers(0, slot - L - gap)) for j in range(needles)], np.int64)


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
            offs = plant_offsets(k, file_bytes, needles)
            idx = (torch.from_numpy(offs).to(device)[:, None] + torch.arange(L, device=device)[None, :]).reshape(-1)
            view[idx] = nd.repeat(needles)
            plants.append(offs)
    torch.cuda.synchronize(device)
    return arena, plants


def measured_traffic(config, nbytes):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (profiles/*pmc_traffic.json; PMC
    counters cannot be collected from inside the run).  (None, None) if no profile of this workload is committed."""
    best = (None, None)
    pdir = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if name.endswith("pmc_traffic.json"):
            try:
                rec = json.load(open(os.path.join(pdir, name))).get(config)
            except (OSError, ValueError):
                continue
            if rec and rec.get("workload_bytes") == nbytes:
                best = (int(rec["traffic_bytes"]), "profiles/%s (separate rocprofv3 --pmc passes of this command; not collected in this run)" % name)
    return best


def time_kernel(ctx, db, arena, segs, stream, nbytes, cap_per_gib, steps, warmup, device):
    """W untimed + K timed launches of one pattern over the arena; (wall seconds for K steps, records, overflow, kernel ms sum, launches, last result)."""
    ctx.set_capacity(max(1 << 16, int(cap_per_gib * nbytes / (1 << 30))))
    res = None
    for _ in range(warmup):
        res = ctx.scan_device(db, arena.data_ptr(), segs, stream)
    if res is not None:
        ctx.dev_sync(res)
    ctx.kernel_time(reset=True)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        res = ctx.scan_device(db, arena.data_ptr(), segs, stream)
    torch.cuda.synchronize(device)
    wall = time.perf_counter() - t0
    total, overflow = ctx.dev_sync(res)
    kern_ms, launches = ctx.kernel_time(reset=True)
    return wall, total, overflow, kern_ms, launches, res


def roofline_block(config, nbytes, total, kern_ms, launches):
    alg_bytes = nbytes + REC_BYTES * total  # per launch: every input byte once + one u32 per candidate
    kern_avg_ms = kern_ms / max(launches, 1)
    achieved = alg_bytes / (kern_avg_ms * 1e-3) / 1e9
    traffic, source = measured_traffic(config, nbytes)
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": source,
            "kernel_ms": round(kern_avg_ms, 4), "launches": int(launches), "algorithmic_bytes_per_launch": int(alg_bytes)}


# ---------------------------------------------------------------------------------------------------------------------
# end to end: the corpus on disk (page cache), the drop-in binary, the reference binary
# ---------------------------------------------------------------------------------------------------------------------
def e2e_dir_for(nbytes_wanted):
    """(directory, bytes to use): /dev/shm if it holds the corpus (64 GiB wanted; the largest power of two that fits
    otherwise, at least 4 GiB); failing that, 8 GiB under /tmp (then the first pass reads the disk, the timed ones the
    page cache); else None."""
    for base, cap in (("/dev/shm", nbytes_wanted), ("/tmp", min(nbytes_wanted, 8 << 30))):
        if not os.path.isdir(base):
            continue
        free = shutil.disk_usage(base).free
        use = cap
        while use > (4 << 30) and use * 1.2 > free:
            use >>= 1
        if use * 1.2 <= free:
            return os.path.join(base, "grab_bench_%d" % os.getpid()), use
    return None, 0


def write_corpus(arena, d, nfiles, file_bytes):
    """Files 0..nfiles-1 of the HBM arena as d/xx/fNNNN.txt (16 sub-directories: something for the walkers to share)."""
    t0 = time.perf_counter()
    for i in range(nfiles):
        sub = os.path.join(d, "%02d" % (i % 16))
        if i < 16:
            os.makedirs(sub, exist_ok=True)
        arena[i * file_bytes:(i + 1) * file_bytes].cpu().numpy().tofile(os.path.join(sub, "f%04d.txt" % i))
    return time.perf_counter() - t0


def run_timed(argv, env, reps):
    """One untimed pass (warms the page cache), then the min of `reps`; (seconds, stdout, stderr of the best).

    stdout goes to a file in /dev/shm, not to a pipe: with 10^8 output lines this process's own reading (and joining) of a
    pipe is a good part of a second that has nothing to do with the program under test."""
    best = None
    out_path = "/dev/shm/grab_bench_out_%d.txt" % os.getpid()
    try:
        for it in range(reps + 1):
            with open(out_path, "wb") as out:
                t0 = time.perf_counter()
                r = subprocess.run(argv, stdout=out, stderr=subprocess.PIPE, env=env)
                dt = time.perf_counter() - t0
            with open(out_path, "rb") as f:
                stdout = f.read()
            if r.returncode != 0:
                return None, stdout, r.stderr
            if it > 0 and (best is None or dt < best[0]):
                best = (dt, stdout, r.stderr)
    finally:
        if os.path.exists(out_path):
            os.unlink(out_path)
    return best


def e2e_measure(d, nfiles, file_bytes, pattern, flags, n_gpus, want_lines, reps=2):
    """`grab -n max(8, N) -r` over the corpus directory on the first N devices."""
    workers = max(8, n_gpus)
    allowed = len(os.sched_getaffinity(0))
    workers = max(1, min(workers, allowed))
    env = dict(os.environ, GRAB_TIMING="1")
    vis = os.environ.get("HIP_VISIBLE_DEVICES")
    devs = [x for x in vis.split(",") if x] if vis else [str(i) for i in range(torch.cuda.device_count())]
    env["HIP_VISIBLE_DEVICES"] = ",".join(devs[:n_gpus])
    argv = [bin_path(), "-n", str(workers), "-r"] + flags + [pattern, d]
    got = run_timed(argv, env, reps)
    if got is None or got[0] is None:
        return {"error": (got[2] if got else b"")[-300:].decode("latin-1")}
    # The command line runs the scan in a child that hands back its status and leaves the GPU teardown (0.1 - 0.2 s) behind
    # the caller's back (grab_cli.cc, GRAB_DETACH); the same run as ONE process is timed next to it.
    one = run_timed(argv, dict(env, GRAB_DETACH="0"), reps)
    one_s = one[0] if one and one[0] else None
    dt, out, err = got
    nbytes = nfiles * file_bytes
    lines = out.count(b"\n")
    per_dev = {}
    for m in re.finditer(rb"\[grab bytes\] device (\d+): (\d+)", err):
        per_dev[int(m.group(1))] = per_dev.get(int(m.group(1)), 0) + int(m.group(2))
    marks = dict((m.group(2).decode(), float(m.group(1))) for m in re.finditer(rb"\[grab timing\] \+([0-9.]+) s ([^\n]+)", err))
    rate = nbytes / dt / 1e9
    # the part of the run the ingest pipeline is responsible for: from "HIP runtime up" to "every window retired and printed"
    # (what is left of wall_s is exec, hipInit and the kernel tearing the process down: fixed, ~0.3 s)
    t_up, t_done = marks.get("runtime up"), marks.get("workers joined", marks.get("scan done"))
    scan_s = t_done - t_up if t_up is not None and t_done is not None and t_done > t_up else None
    return {"value": round(rate, 2), "unit": "GB/s", "scaling": "strong", "n_gpus": n_gpus, "workers": workers,
            "bytes": nbytes, "wall_s": round(dt, 4), "startup_s": marks.get("runtime up"),
            "one_process_wall_s": one_s and round(one_s, 4), "one_process_GBps": one_s and round(nbytes / one_s / 1e9, 2),
            "one_process_frac": one_s and round(nbytes / one_s / 1e9 / (PCIE_PEAK_GBPS * n_gpus), 4),
            "scan_phase_s": scan_s and round(scan_s, 4), "scan_phase_GBps": scan_s and round(nbytes / scan_s / 1e9, 2),
            "scan_phase_frac": scan_s and round(nbytes / scan_s / 1e9 / (PCIE_PEAK_GBPS * n_gpus), 4),
            "pcie_peak": PCIE_PEAK_GBPS * n_gpus, "frac": round(rate / (PCIE_PEAK_GBPS * n_gpus), 4),
            "lines": lines, "lines_expected": want_lines, "lines_ok": lines == want_lines,
            "matches_per_s": round(lines / dt, 1),
            "per_device_bytes": {str(k): v for k, v in sorted(per_dev.items())},
            "ingest": engine.ingest_info(),
            "command": " ".join([os.path.basename(argv[0])] + argv[1:-1]) + " <dir>, wall clock of the whole process, page cache warm, min of %d" % reps}


def cpu_baseline(d, nfiles, file_bytes, pattern, flags):
    """The reference (or the oracle port) on this box's host cores over the same on-disk corpus."""
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
    argv = [binary] + (["-n", str(cores)] if cores > 1 else []) + ["-r"] + flags + [pattern, d]
    got = run_timed(argv, None, 2)
    if got is None or got[0] is None:
        return None
    dt, out, _ = got
    nbytes = nfiles * file_bytes
    return {"value": round(nbytes / dt / 1e9, 3), "unit": "GB/s", "cores": cores, "kind": kind,
            "sample": "%d x %d MiB files of the same corpus under %s (%.0f GiB), '%s', warm cache, min of 2" % (
                nfiles, file_bytes >> 20, os.path.dirname(d), nbytes / (1 << 30), " ".join(os.path.basename(a) if a == binary else a for a in argv[:-1])),
            "lines": out.count(b"\n"), "matches_per_s": round(out.count(b"\n") / dt, 1),
            "engine": "libpcre 8.39 JIT (pcre_exec); the -H hyperscan path does not exist in the mounted reference"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", default="kernel", choices=["kernel", "e2e"])
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--files", type=int, default=1024)
    ap.add_argument("--file-mib", type=int, default=64)
    ap.add_argument("--variant", type=int, default=None, help="kernel variant (gscan_set_option)")
    ap.add_argument("--blocks-per-cu", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-kernels", action="store_true", help="skip the other two kernels' roofline blocks")
    ap.add_argument("--e2e-gib", type=int, default=64, help="corpus written to /dev/shm for the end-to-end block")
    a = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the scan engine has no CPU path")
    if a.mode == "e2e" and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        rank, world, local = 0, 1, 0  # plain python: one process drives `grab` over the first --gpus devices
    else:
        rank, world, local = dist_setup(a.gpus)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    pattern, cap_per_gib = CONFIGS[a.config]
    file_bytes = a.file_mib << 20
    nbytes = a.files * file_bytes
    e2e_files = max(1, min(a.files, (a.e2e_gib << 30) // file_bytes))

    if a.mode == "e2e":
        # the corpus only has to exist on disk: build it piecewise in HBM (generation is the device's), never 64 GiB at once
        line = None
        if rank == 0:
            d, use = e2e_dir_for(e2e_files * file_bytes)
            if not d:
                raise SystemExit("no room in /dev/shm for an end-to-end corpus")
            nfiles = use // file_bytes
            try:
                t0 = time.perf_counter()
                step = 64
                for lo in range(0, nfiles, step):
                    n = min(step, nfiles - lo)
                    part = torch.empty(n * file_bytes, dtype=torch.uint8, device=device)
                    nd = torch.frombuffer(bytearray(synth.NEEDLE), dtype=torch.uint8).to(device)
                    for i in range(n):
                        view = part[i * file_bytes:(i + 1) * file_bytes]
                        view.copy_(synth.torch_text(file_bytes, lo + i, device))
                        offs = plant_offsets(lo + i, file_bytes, NEEDLES_PER_FILE)
                        idx = (torch.from_numpy(offs).to(device)[:, None] + torch.arange(nd.numel(), device=device)[None, :]).reshape(-1)
                        view[idx] = nd.repeat(NEEDLES_PER_FILE)
                        sub = os.path.join(d, "%02d" % ((lo + i) % 16))
                        os.makedirs(sub, exist_ok=True)
                        view.cpu().numpy().tofile(os.path.join(sub, "f%04d.txt" % (lo + i)))
                    del part
                gen_s = time.perf_counter() - t0
                flags = ["-O", "-l"] if a.config == "cfg3" else []
                want = NEEDLES_PER_FILE * nfiles if a.config != "cfg3" else None
                e = e2e_measure(d, nfiles, file_bytes, pattern, flags, a.gpus, want)
                if want is None and "lines" in e:
                    e["lines_ok"] = None
                line = {"metric": "GB/s end to end (PCIe-inclusive), synthetic corpus in the page cache, grab -n over the work queue",
                        "value": e.get("value"), "unit": "GB/s", "n_gpus": a.gpus, "steps": 1, "warmup": 1,
                        "ms_per_step": round(e.get("wall_s", 0) * 1e3, 2), "higher_is_better": True, "scaling": "strong",
                        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                        "config": {"workload": "%s: %d x %d MiB files in /dev/shm, pattern '%s', one walk -> one queue -> %d device(s)" % (
                            a.config, nfiles, a.file_mib, pattern, a.gpus), "corpus_write_s": round(gen_s, 1)},
                        "e2e": e}
                if not a.no_cpu_baseline:
                    line["cpu_baseline"] = cpu_baseline(d, nfiles, file_bytes, pattern, flags)
            finally:
                shutil.rmtree(d, ignore_errors=True)
        barrier(world, device)
        if rank == 0:
            print(json.dumps(line), flush=True)
        if world > 1:
            import torch.distributed as dist

            dist.destroy_process_group()
        return

    arena, plants = build_corpus(a.files, file_bytes, NEEDLES_PER_FILE, rank, device)

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
    elif a.config != "cfg3":
        if total != NEEDLES_PER_FILE * a.files:
            check = "expected %d matches, got %d" % (NEEDLES_PER_FILE * a.files, total)
        for i in (0, a.files - 1):
            if not np.array_equal(ctx.dev_fetch(res, i).astype(np.int64), plants[i]):
                check = "planted offsets differ in file %d" % i

    line = None
    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        line = {
            "metric": "GB/s scanned, 64 GiB synthetic corpus resident in HBM (match offsets/s in matches_per_s)",
            "value": round(world * nbytes / (elapsed / a.steps) / 1e9, 2),
            "unit": "GB/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%s: %d x %d MiB files per GPU, pattern '%s', %d needles planted per file, one launch over all segments, offsets compacted in HBM" % (
                a.config, a.files, a.file_mib, pattern, NEEDLES_PER_FILE),
                "bytes_per_gpu": nbytes, "kernel": KERNEL_NAMES.get(db.info.tier, "?"),
                "parallelism": "files sharded per GPU, no collective"},
            "matches_per_step": int(matches_all),
            "matches_per_s": round(matches_all / (elapsed / a.steps), 1),
            "check": check,
            "roofline": roofline_block(a.config, nbytes, total, kern_ms, launches),
        }

    # the other two kernels on the same arena, same process (outside the timed region above): every rank runs them so that
    # the ranks stay in step, rank 0 reports its own
    if not a.no_kernels:
        others = {}
        for name in sorted(CONFIGS):
            if name == a.config:
                continue
            pat2, cap2 = CONFIGS[name]
            db2 = engine.Database(pat2)
            wall, tot2, ovf2, kms2, nl2, _ = time_kernel(ctx, db2, arena, segs, stream, nbytes, cap2, max(3, a.steps // 2), 1, device)
            blk = roofline_block(name, nbytes, tot2, kms2, nl2)
            blk.update({"pattern": pat2, "kernel": KERNEL_NAMES.get(db2.info.tier, "?"), "records_per_launch": int(tot2), "overflow": bool(ovf2),
                        "value": round(nbytes / (wall / max(3, a.steps // 2)) / 1e9, 2)})
            others[name] = blk
        if rank == 0:
            line["kernels"] = others
    ctx.close()

    # end to end on the same corpus: rank 0 writes it out and drives `grab` over the first `world` devices while the other
    # ranks wait at the barrier (their arenas stay allocated; nothing of theirs runs)
    if not a.no_e2e:
        if rank == 0:
            d, use = e2e_dir_for(e2e_files * file_bytes)
            if d:
                nfiles = use // file_bytes
                try:
                    wsec = write_corpus(arena, d, nfiles, file_bytes)
                    del arena
                    torch.cuda.empty_cache()
                    flags = ["-O", "-l"] if a.config == "cfg3" else []
                    want = NEEDLES_PER_FILE * nfiles if a.config != "cfg3" else None
                    e = e2e_measure(d, nfiles, file_bytes, pattern, flags, world, want)
                    e["corpus_write_s"] = round(wsec, 1)
                    line["e2e"] = e
                    if world == 1 and not a.no_cpu_baseline:
                        line["cpu_baseline"] = cpu_baseline(d, nfiles, file_bytes, pattern, flags)
                        if line["cpu_baseline"] and "value" in e:
                            e["vs_cpu_baseline"] = round(e["value"] / line["cpu_baseline"]["value"], 3)
                except Exception as ex:  # (the kernel line above is the contract: whatever goes wrong out here must not lose it)
                    line.setdefault("e2e", {})["error"] = "%s: %s" % (type(ex).__name__, str(ex)[:300])
                finally:
                    shutil.rmtree(d, ignore_errors=True)
            else:
                line["e2e"] = {"error": "no room in /dev/shm or /tmp"}
        barrier(world, device)
    if rank == 0:
        print(json.dumps(line), flush=True)

    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
