#!/usr/bin/env python3
"""The host's share of the work, one core, no GPU needed: VERDICT r5's table ("What's weak" 2) with the list the device hands over
NOW.  For every pattern, over --mib MiB of the SURVEY.md 8d corpus, `-O -l`:

  product   grab_report_chunk (filegrep.report_chunk) fed with what the kernels are specified to list -- tests/inputs.py:
            resolved_list (k_resolve's output: match starts + ends, the VM program run on the host, same source) for a
            database with info.resolve, the ends k_ends measures for one with info.ends_ok, else engine_list.  Making the
            list is the DEVICE's work and is not timed; walking it is the host's and is.
  reference liboracle.oracle_scan_chunk: libpcre (8.39, JIT) under the reference's loop (grab.cc:171-213) on the same core.

Outputs are compared byte for byte.  One JSON line per pattern: listed -> printed, ns of host walk per listed record and per printed
line, MB/s both ways, ratio.  `oracle/` is used as the checker and the baseline only."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from grab_amd import engine, filegrep, synth  # noqa: E402
import inputs  # noqa: E402  (tests/inputs.py: the host-side restatement of what the kernels list)

PATTERNS = [  # VERDICT r5's eight rows, then everyday ones of other shapes
    r"\b[A-Za-z_]\w*\s*\(", r"(?<=\$)\d+", r"\s\w{8,}\s", r"\b[A-Z][a-z]+\b", r"\([^()]*\)", r"\b[a-z]{3,}\b", r"\w+(?=\()", synth.IDENT_RE,
    r"\bif\s*\(", r"[-]?\d+\.\d+(?:[eE][+-]?\d+)?", r"\b(?:[a-z]+_)+[a-z]+\b", r"(?m)^\s*[a-z_]+\s*=", r"error|warning|fatal|critical", r"[{][^{}]*[}]",
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=8)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    L = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    L.oracle_scan_chunk.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_longlong, C.c_uint, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.oracle_free.argtypes = [C.c_void_p]
    data = synth.text(a.mib << 20, 0)
    flags = filegrep.OFFSETS | filegrep.NOLINE
    for pat in PATTERNS:
        db = engine.Database(pat)
        info = db.info
        if info.resolve:
            starts, ends = inputs.resolved_list(db, data)
        else:
            starts = inputs.engine_list(db, data)
            ends = db.match_ends(data, starts) if getattr(info, "ends_ok", 0) and hasattr(db, "match_ends") else None
        best = None
        for _ in range(a.reps):
            t0 = time.perf_counter()
            got = filegrep.report_chunk(db, flags, b"", data, 0, starts, ends=ends)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        rbest = None
        for _ in range(a.reps):
            out, n = C.c_void_p(), C.c_size_t()
            t0 = time.perf_counter()
            rc = L.oracle_scan_chunk(pat.encode(), b"", data.ctypes.data, data.size, 0, 3, C.byref(out), C.byref(n))
            dt = time.perf_counter() - t0
            assert rc == 0
            want = C.string_at(out, n.value)
            L.oracle_free(out)
            rbest = dt if rbest is None else min(rbest, dt)
        printed = got.count(b"\n")
        print(json.dumps({"pattern": pat, "tier": info.tier, "resolve": info.resolve, "reach": info.reach, "exact": info.exact, "ends_from_device": ends is not None,
                          "listed": int(len(starts)), "printed": printed, "identical": got == want,
                          "host_walk_MBps": round(data.size / best / 1e6, 1), "ns_per_listed": round(best * 1e9 / max(1, len(starts)), 1),
                          "ns_per_printed": round(best * 1e9 / max(1, printed), 1),
                          "libpcre_MBps": round(data.size / rbest / 1e6, 1), "ratio": round(rbest / best, 2)}), flush=True)


if __name__ == "__main__":
    main()
