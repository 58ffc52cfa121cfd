#!/usr/bin/env python3
"""One-off differential campaign on the GPU box: random patterns of the supported grammar (tests/test_fuzz.py's generator)
through `grab` against the oracle (libpcre under the reference's loop) on one file -- tests/test_gpu_filegrep.py's
test_random_patterns_cli_vs_oracle with other seeds and a time budget instead of 45 patterns.

    python scripts/gpu_random_campaign.py --seed 31337 --seconds 170 [--lead-repeat] [--tree]

--lead-repeat: every pattern gets a leading unbounded repeat in front of it (the shapes whose alternatives share one device
window and have been K1's / K2's since round 3).  --tree (round 4): the text is cut into 40 files of ragged sizes in a small
tree and searched with `-r` (byte-exact: the serial walk is nftw's order) and `-n 3 -r` (sorted) alternately -- every pattern
goes through the small-file path (names queued, the device's readers open and read them, one launch per batch).
Prints one JSON line; exit status 1 on any difference."""
import argparse
import json
import os
import random
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(exe, args, cwd):
    r = subprocess.run([exe] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, GRAB_DIAG="1"))
    return r.returncode, r.stdout, r.stderr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=31337)
    ap.add_argument("--seconds", type=float, default=170)
    ap.add_argument("--lead-repeat", action="store_true")
    ap.add_argument("--tree", action="store_true")
    a = ap.parse_args()
    from grab_amd import engine
    from grab_amd.build import bin_path
    from test_fuzz import gen

    oracle = os.path.join(ROOT, "oracle", "grab_oracle")
    rng = random.Random(a.seed)
    nrng = np.random.default_rng(a.seed)
    alpha = np.frombuffer(b"abcxA01 .\n\nab  ", np.uint8)
    data = alpha[nrng.integers(0, alpha.size, 300_000)]
    data[1000:1003] = np.frombuffer(b"abc", np.uint8)
    tiers, done, skipped, bad = {}, 0, 0, []
    with tempfile.TemporaryDirectory() as d:
        data.tofile(os.path.join(d, "f"))
        if a.tree:
            os.remove(os.path.join(d, "f"))
            cuts = sorted(set([0, data.size] + [int(x) for x in nrng.integers(0, data.size, 37)] + [1000, 1001]))  # (1000..1001: a one-byte file)
            for i, (lo, hi) in enumerate(zip(cuts[:-1], cuts[1:])):
                sub = os.path.join(d, "f", "d%d" % (i % 4), "s%d" % (i % 3))
                os.makedirs(sub, exist_ok=True)
                data[lo:hi].tofile(os.path.join(sub, "p%02d" % i))
            open(os.path.join(d, "f", "empty"), "wb").close()
        t0 = time.time()
        while time.time() - t0 < a.seconds:
            pat = gen(rng)
            if a.lead_repeat:
                pat = rng.choice(["a+", "[ab]+", r"\w+", "[a-c0-9]+", "x*", r"\d+", ".+"]) + pat
            try:
                db = engine.Database(pat)
            except ValueError:
                skipped += 1
                continue
            if db.minlen < 0:
                skipped += 1
                continue
            flags = [["-O", "-l"], ["-O"], []][done % 3]
            threaded = a.tree and done % 2 == 1
            if a.tree:
                flags = (["-n", "3"] if threaded else []) + ["-r"] + flags
            orc, oout, oerr = run(oracle, flags + [pat, "f"], d)
            if orc != 0:
                skipped += 1
                continue
            rc, out, err = run(bin_path(), flags + [pat, "f"], d)
            if b"gave up" in oerr or b"abandoned" in err:
                skipped += 1
                continue
            key = "tier%d%s" % (db.info.tier, "+resolve" if db.info.resolve else "+vm" if db.info.vm else "")
            tiers[key] = tiers.get(key, 0) + 1
            if threaded and "-O" in flags and "-l" not in flags:
                # offset line + text belong together, and the text may hold newlines of its own (a match of [^a] or \s):
                # a file's output is written in one piece (one chunk per file here), so cut the stream where a record of
                # ANOTHER file begins and compare file by file
                def by_file(b):
                    import re
                    out_, cur_ = {}, None
                    for ln in (b[:-1] if b.endswith(b"\n") else b).split(b"\n"):
                        m_ = re.match(rb"^(f/[^:]+):Match at offset \d+$", ln)
                        if m_:
                            cur_ = m_.group(1)
                        out_.setdefault(cur_, []).append(ln)
                    return out_
                same = len(out) == len(oout) and by_file(out) == by_file(oout)
            elif threaded:
                same = sorted(out.splitlines()) == sorted(oout.splitlines())
            else:
                same = out == oout
            if rc != 0 or not same:
                bad.append({"pattern": pat, "flags": flags, "rc": rc, "lines": out.count(b"\n"), "oracle_lines": oout.count(b"\n"), "err": err[-200:].decode("latin-1")})
            done += 1
    print(json.dumps({"seed": a.seed, "lead_repeat": a.lead_repeat, "tree": a.tree, "compared": done, "skipped": skipped, "by_tier": tiers, "differences": bad}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
