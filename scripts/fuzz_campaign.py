#!/usr/bin/env python3
"""CPU-side differential campaign: tests/test_fuzz.py's generators and check() with fresh seeds, differences collected instead
of asserted.  (The product's walk over what the kernels are specified to list, against libpcre under the reference's loop.)

    python scripts/fuzz_campaign.py --seed0 900000 --procs 8 --draws 20000 [--grammar calls|binary|plain]
"""
import argparse
import json
import multiprocessing as mp
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def work(job):
    seed, draws, grammar = job
    import ctypes as C

    import test_fuzz as tf
    lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    lib.oracle_minlen.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
    lib.oracle_scan_chunk.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_longlong, C.c_uint, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.oracle_free.argtypes = [C.c_void_p]
    lib.oracle_free.restype = None
    lib.oracle_resource_errors.restype = C.c_long
    rng = random.Random(seed)
    texts = tf.make_texts(seed)
    if grammar == "binary":
        import numpy as np

        nrng = np.random.default_rng(seed)
        alpha = np.frombuffer(b"ab Z0_\n\r\x0b\x0c\x85\xa0\xff\x00\xe9.", np.uint8)
        texts = [alpha[nrng.integers(0, alpha.size, int(nrng.integers(1, 100)))].tobytes() for _ in range(12)]
    compared, bad = 0, []
    for _ in range(draws):
        pat = tf.gen_calls_and_conditions(rng) if grammar == "calls" else tf.gen(rng, tf.BIN_ATOMS if grammar == "binary" else None)
        try:
            r = tf.check(lib, pat, texts)
        except AssertionError as e:
            bad.append({"pattern": pat, "detail": repr(e.args[0])[:400] if e.args else ""})
            continue
        except Exception as e:  # noqa: BLE001
            bad.append({"pattern": pat, "error": repr(e)[:300]})
            continue
        compared += r is not None
    return seed, compared, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed0", type=int, default=900000)
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--draws", type=int, default=20000)
    ap.add_argument("--grammar", default="plain", choices=["plain", "calls", "binary"])
    a = ap.parse_args()
    jobs = [(a.seed0 + i, a.draws, a.grammar) for i in range(a.procs)]
    with mp.Pool(a.procs) as pool:
        res = pool.map(work, jobs)
    out = {"grammar": a.grammar, "draws": a.procs * a.draws, "compared": sum(r[1] for r in res), "differences": [b for r in res for b in r[2]]}
    print(json.dumps(out, indent=1)[:20000])
    return 1 if out["differences"] else 0


if __name__ == "__main__":
    sys.exit(main())
