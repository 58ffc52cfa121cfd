#!/usr/bin/env python3
"""What pcre_compile rejects, gscan_compile has to reject too: random patterns from tests/test_fuzz.py's three grammars (and, mutated:
one character dropped or doubled, which is where syntax errors come from), every pattern libpcre refuses put to the product's compiler.
(FileGrep::prepare asks libpcre first, so the command line never gets this far; a binding of gscan.h alone does.)  Prints the
patterns the product accepts all the same.

    python scripts/reject_parity.py --seed 1 --draws 30000
"""
import argparse
import ctypes as C
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from grab_amd import engine  # noqa: E402
import test_fuzz as tf  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--draws", type=int, default=30000)
    a = ap.parse_args()
    L = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    L.oracle_minlen.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
    rng = random.Random(a.seed)
    rejected = accepted_anyway = 0
    seen, bad = set(), []
    for k in range(a.draws):
        g = k % 3
        pat = tf.gen_calls_and_conditions(rng) if g == 1 else tf.gen(rng, tf.BIN_ATOMS if g == 2 else None)
        if rng.random() < 0.6 and len(pat) > 1:  # mutate
            at = rng.randrange(len(pat))
            pat = pat[:at] + (pat[at + 1:] if rng.random() < 0.5 else pat[at] + pat[at:])
        if pat in seen:
            continue
        seen.add(pat)
        ml = C.c_int(-9)
        try:
            pb = pat.encode("latin-1")
        except UnicodeEncodeError:
            continue
        if b"\0" in pb or L.oracle_minlen(pb, C.byref(ml)) == 0:
            continue
        rejected += 1
        try:
            engine.Database(pat)
        except ValueError:
            continue
        accepted_anyway += 1
        if len(bad) < 60:
            bad.append(pat)
    print(json.dumps({"seed": a.seed, "distinct_patterns": len(seen), "rejected_by_pcre": rejected, "accepted_by_the_product_all_the_same": accepted_anyway, "examples": bad}, indent=1))


if __name__ == "__main__":
    main()
