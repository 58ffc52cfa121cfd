"""Driver of tests/test_gpu_pool.py (run as a subprocess: the ingest configuration is read once per process).

Opens `--per-device` contexts on each of GSCAN_VIRTUAL_DEVICES device indices, one thread per context, and pushes `--mib`
MiB through every one of them as a seeded mix of file windows (gscan_submit_fd: random offsets and lengths, up to three in
flight) and batches of small files (gscan_submit_files), with the staging pool the environment asks for (GSCAN_BLOCK_MIB,
GSCAN_READERS, GSCAN_POOL_CAP, GSCAN_FAIL_ALLOC_AFTER, GSCAN_NT_COPY).  Every chunk's list is compared with libpcre's
candidate set of the same bytes (liboracle.oracle_all_starts over the base text, computed once).  Prints one JSON line:
what was pushed, the mismatches (must be 0) and each device's pool statistics (gscan_pool_stats)."""
import argparse
import ctypes as C
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import scan_oracle as so  # noqa: E402  (the checker)
from grab_amd import engine, synth  # noqa: E402


def pcre_starts(L, pattern, buf):
    cap = buf.size + 1
    out = np.zeros(cap, np.uint32)
    n = L.oracle_all_starts(pattern.encode("latin-1"), buf.ctypes.data, buf.size, out.ctypes.data, None, cap)
    assert 0 <= n <= cap
    return out[:n].astype(np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", required=True)
    ap.add_argument("--mib", type=int, default=256, help="MiB pushed through every context")
    ap.add_argument("--per-device", type=int, default=2)
    ap.add_argument("--base-mib", type=int, default=32)
    a = ap.parse_args()
    L = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    L.oracle_all_starts.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
    L.oracle_all_starts.restype = C.c_long

    n = a.base_mib << 20
    base = synth.text(n, 77)
    synth.plant(base, synth.NEEDLE, 4000, 77, gap=100)
    big = os.path.join(a.dir, "big.bin")
    base.tofile(big)
    patterns = [(synth.NEEDLE.decode(), len(synth.NEEDLE)), (synth.IDENT_RE, 16)]
    want = [pcre_starts(L, p, base) for p, _ in patterns]
    # small files: slices of the base text (their candidates are the base's, shifted), sizes from empty to just below a block
    rng = np.random.default_rng(5)
    blk = engine.ingest_info()["block_bytes"]
    small = []
    for i in range(96):
        ln = int(rng.choice([0, 1, 15, 17, 18, 400, 5000, 70_000, 300_000, blk - 1, blk]))
        at = int(rng.integers(0, n - ln))
        path = os.path.join(a.dir, "s%03d.bin" % i)
        base[at:at + ln].tofile(path)
        small.append((path, at, ln))

    def expect(pi, at, ln):
        w = want[pi]
        lo, hi = np.searchsorted(w, at), np.searchsorted(w, at + ln - patterns[pi][1], side="right")
        return w[lo:max(lo, hi)] - at

    ndev = engine.device_count()
    bad, pushed, lock = [], [0], threading.Lock()
    stats = {}

    def worker(dev, k):
        try:
            work(dev, k)
        except Exception as e:  # (a thread's exception would otherwise be lost)
            bad.append((dev, k, "exception", repr(e)))

    def work(dev, k):
        pi = (dev + k) & 1
        db = engine.Database(patterns[pi][0])
        ctx = engine.Context(dev, 64 << 20)
        r = np.random.default_rng(1000 + 16 * dev + k)
        fd = os.open(big, os.O_RDONLY)
        flight = []  # what each chunk in flight must come to: ("fd", at, ln) | ("files", [(at, ln), ...])
        todo = a.mib << 20

        def retire():
            job = flight.pop(0)
            _, segs, _ = ctx.wait_segs()
            parts = [(job[1], job[2])] if job[0] == "fd" else job[1]
            if len(segs) != len(parts):
                bad.append((dev, k, "segments", len(segs), len(parts)))
                return
            for got, (at, ln) in zip(segs, parts):
                if not so.check_reported(got, expect(pi, at, ln)):
                    bad.append((dev, k, job[0], at, ln))

        while todo > 0:
            if len(flight) == engine.SLOTS:
                retire()
            if r.integers(0, 4) == 0:  # a batch of small files (<= 32 MiB in all: the context's max_chunk is 64 MiB)
                pick, tot = [], 0
                for j in r.permutation(len(small))[: int(r.integers(1, 40))]:
                    if tot + small[j][2] + 16 > (32 << 20):
                        break
                    pick.append(small[j])
                    tot += small[j][2] + 16
                ctx.submit_files(db, [(p, ln) for p, _, ln in pick])
                flight.append(("files", [(at, ln) for _, at, ln in pick]))
                todo -= tot
            else:  # a window of the big file: any offset, any length up to 24 MiB, the last bytes of the file now and then
                ln = int(r.integers(0, 24 << 20)) if r.integers(0, 8) else int(r.integers(0, 40))
                at = int(r.integers(0, n - ln)) if r.integers(0, 6) else n - ln
                ctx.submit_fd(db, fd, at, ln)
                flight.append(("fd", at, ln))
                todo -= ln
        while flight:
            retire()
        with lock:
            pushed[0] += (a.mib << 20) - todo
            stats[dev] = ctx.pool_stats()
        os.close(fd)
        ctx.close()
        db.close()

    ts = [threading.Thread(target=worker, args=(d, k)) for d in range(ndev) for k in range(a.per_device)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    print(json.dumps({"devices": ndev, "contexts": len(ts), "bytes": pushed[0], "mismatches": len(bad), "first": [list(map(str, b)) for b in bad[:5]],
                      "pool": {str(d): stats[d] for d in sorted(stats)}}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
