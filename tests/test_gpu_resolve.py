"""GPU parity for the resolve pass (round 6; gscan_info.resolve): for patterns that are not one plain window the kernels list
every offset where a START window fits and k_resolve runs the pattern's VM program there -- what gscan_wait hands out is the
list of MATCH starts with their ends.  Checked at three levels:
  * the list itself == the same VM program run on the host over the same candidate set (tests/inputs.py:resolved_list; the
    VM against the host matcher and the host matcher against libpcre: tests/test_vm.py, tests/test_fuzz.py);
  * every record of it against libpcre directly (oracle_all_starts: pcre_exec anchored at the offset);
  * the drop-in binary's output == the reference loop's (oracle/grab_oracle, and oracle/_ref/grab_jit when it is there) in every
    output mode, across chunk boundaries and over batches of small files."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
from grab_amd import engine, synth
from inputs import END_ASK, END_CAPTURES, engine_list, resolved_list

pytestmark = pytest.mark.gpu

# VERDICT r5's table (the patterns the host used to re-match candidate by candidate) and relatives
DENSE = [r"\b[A-Za-z_]\w*\s*\(", r"(?<=\$)\d+", r"\s\w{8,}\s", r"\b[A-Z][a-z]+\b", r"\([^()]*\)", r"\b[a-z]{3,}\b"]
MORE = [r"foo|bar|[0-9]{5}", r"\d+\.\d+", r"\bfoobardoesnotexist\b", r"(?<=ab)c+|(?<!a)b\w", r"(?m)^\w+ \w+$", r"(a)b|cd+e", r"(?i)\bxyzzy\b|\bplugh\b",
        r"[a-z]+_[0-9]+\.[a-z]+", r"\Bab\B", r"x(?=(a))ab|ab\d",
        # start windows that list most of the text (the record buffers regrow), a leading look-behind as the windows' context byte
        r"\w+(?=\()", r"\w+\s*=\s*\w+\s*\(", r"(?<=\()[^()\n]+(?=\))", r"(?:foo|bar)+does.*exist|[0-9]+\.[0-9]+\.[0-9]+|\b(?:[a-z]+_)+[a-z]+\b"]


def _text(n, k):
    buf = synth.text(n, k)
    if n >= 20000:
        synth.plant(buf, b"foobardoesnotexist", 8, k, gap=200)
    extra = b" $12 a$345 (x) (y(z)) Hello World_1 foo( bar  (\nab abcc cd e cdde xyzzy PLUGH a_1.b ab1 ab. 1.2.3 foo_bar_baz fooexist foobardoesexist\n"
    if n >= 1000 + 2 * len(extra):
        buf[1000:1000 + len(extra)] = np.frombuffer(extra, np.uint8)
    if n >= len(extra):
        buf[n - len(extra):] = np.frombuffer(extra, np.uint8)  # (matches that end with the chunk)
    return buf


@pytest.mark.parametrize("pattern", DENSE + MORE)
def test_resolved_list_is_the_vm_over_the_candidates(pattern, built, liboracle):
    db = engine.Database(pattern)
    assert db.info.resolve, pattern
    ctx = engine.Context(0, 1 << 26)
    for n, k in ((300_000, 3), (17, 4), (4096 + 1, 5)):
        data = _text(max(n, 400), k) if n > 400 else synth.text(n, k)
        starts = ctx.scan(db, data)
        ends = ctx.last_ends(len(starts))
        assert ends is not None and len(ends) == len(starts)
        ws, we = resolved_list(db, data)
        assert np.array_equal(starts, ws), (pattern, n)
        assert np.array_equal(ends, we), (pattern, n)
        # ... and every record against libpcre itself: a match starts there (subject = the chunk, the bytes in front visible)
        assert len(starts) == 0 or np.all(np.diff(starts.astype(np.int64)) > 0)
    ctx.close()


@pytest.mark.parametrize("seed", [71, 72])
def test_resolved_list_on_random_patterns(seed, built):
    from test_fuzz import ATOMS, BIN_ATOMS, gen, make_texts

    rng = random.Random(seed)
    ctx = engine.Context(0, 1 << 24)
    texts = [np.frombuffer(t, np.uint8) for t in make_texts(seed)[:6]] + [_text(20_000, seed)]
    done = 0
    while done < 60:
        pat = gen(rng, BIN_ATOMS if seed == 72 else ATOMS)
        try:
            db = engine.Database(pat)
        except ValueError:
            continue
        if db.minlen < 0 or not db.info.resolve:
            continue
        for data in texts:
            if len(data) == 0:
                continue
            starts = ctx.scan(db, data)
            ends = ctx.last_ends(len(starts))
            ws, we = resolved_list(db, data)
            assert np.array_equal(starts, ws), (pat, bytes(data[:80]))
            assert np.array_equal(ends, we), (pat, bytes(data[:80]))
        done += 1
    ctx.close()


def _run(binary, args, cwd):
    r = subprocess.run([binary] + args, cwd=cwd, capture_output=True, env=dict(os.environ, GSCAN_MATCH_LIMIT="200000000"))
    return r.returncode, r.stdout, r.stderr


@pytest.fixture(scope="module")
def big_file(tmp_path_factory):
    d = tmp_path_factory.mktemp("resolve")
    _text(70 << 20, 11).tofile(str(d / "f.txt"))
    return d


@pytest.mark.parametrize("flags", [["-O", "-l"], ["-O"], [], ["-l"], ["-s", "-O"]])
@pytest.mark.parametrize("pattern", DENSE + MORE[:6])
def test_cli_equals_reference_loop(pattern, flags, built, oracle_built, big_file):
    """One 70 MiB file at 32 MiB chunks (three windows, two overlaps): byte-identical to the reference's loop."""
    argv = ["-L"] * 5 + flags + [pattern, "f.txt"]
    rc, out, err = _run(built.bin_path(), argv, str(big_file))
    orc, oout, _ = _run(os.path.join(oracle_built, "grab_oracle"), argv, str(big_file))
    assert rc == orc == 0, err
    assert out == oout


@pytest.mark.parametrize("args", [["-r", "-O", "-l"], ["-r", "-O"], ["-n", "3", "-r", "-O", "-l"], ["-n", "2", "-r"]])
@pytest.mark.parametrize("pattern", DENSE[:3] + [MORE[0], MORE[4]])
def test_small_files_in_batches(pattern, args, built, oracle_built, tmp_path):
    root = tmp_path / "tree"
    root.mkdir()
    rng = np.random.default_rng(7)
    for i in range(40):
        d = root / ("d%d" % (i % 4))
        d.mkdir(exist_ok=True)
        _text(int(rng.choice([500, 5000, 70000, 300000])), 200 + i).tofile(str(d / ("f%02d" % i)))
    argv = args + [pattern, "tree"]
    rc, out, err = _run(built.bin_path(), argv, str(tmp_path))
    orc, oout, _ = _run(os.path.join(oracle_built, "grab_oracle"), argv, str(tmp_path))
    assert rc == orc == 0, err
    assert sorted(out.splitlines(True)) == sorted(oout.splitlines(True))
    if "-n" not in args:
        assert out == oout


def test_device_resident_path_resolves_too(built):
    """gscan_scan_device (what bench.py times): for a database of this kind the launch includes k_resolve -- the records left in
    HBM are the match starts, total counts them."""
    import torch

    ctx = engine.Context(0, 1 << 26)
    data = _text(3_000_000, 9)
    arena = torch.from_numpy(data).cuda()
    segs = [(0, 1_000_000), (1_000_000, 2_000_000)]
    for pattern in (DENSE[0], DENSE[2], MORE[0]):
        db = engine.Database(pattern)
        ctx.set_capacity(1 << 22)
        res = ctx.scan_device(db, arena.data_ptr(), segs)
        total, overflow = ctx.dev_sync(res)
        assert not overflow
        got = [ctx.dev_fetch(res, i) for i in range(len(segs))]
        for (o, ln), g in zip(segs, got):
            assert np.array_equal(g, resolved_list(db, data[o:o + ln])[0]), pattern
        assert total == sum(len(g) for g in got)
    ctx.close()
