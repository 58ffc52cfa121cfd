"""Host-side printing rule (grab_report_chunk, grab_amd/csrc/filegrep.cc) on the CPU: with the
candidate list supplied by the oracle instead of the GPU, the product's chunk walk must
reproduce the reference binary's output byte for byte (golden.json), including chunk-overlap
duplicates, the strict loop bound, the 511-byte context caps and -s / -l rules."""
import hashlib
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, golden_ids, split_args
from grab_amd import engine, filegrep
from inputs import build, db_candidates, engine_list, resolved_list

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import scan_oracle as so  # noqa: E402


def _supported(case):
    flags, pattern, paths = split_args(case["args"])
    if len(paths) != 1 or paths[0] not in case["inputs"] or "-r" in flags or "-n" in flags or case["rc"] != 0 or not case["inputs"]:
        return False
    try:
        engine.Database(pattern)
    except ValueError:
        return False
    return True


def host_find(db, data, flags, chunk, path=b"", minimal=True):
    """FileGrep::find's control flow with oracle-made candidates in place of gscan_wait.
    minimal=True hands over only what the engine must report (group starts); False hands over every
    candidate (the engine may report any superset of the group starts)."""
    minlen = db.minlen
    if minlen < 0 or minlen > len(data):
        return b""
    out = []
    for off, clen in so.chunks(len(data), chunk):
        part = data[off:off + clen]
        starts = db_candidates(db, part)
        ends = None
        if db.info.resolve:  # the device settles the matches: the list of match starts and their ends (minimal) / every offset a start window fits, for the host matcher (not minimal)
            starts, ends = resolved_list(db, part) if minimal else (engine_list(db, part), None)
        elif db.info.vm:
            starts = engine_list(db, part)  # the device confirms the candidates itself: every hit it keeps, nothing else
        elif minimal:
            starts = so.group_starts(starts)
        text = filegrep.report_chunk(db, flags, path, part, off, starts.astype(np.uint32), ends=ends)
        if text:
            out.append(text)
            if flags & filegrep.SINGLE:
                break
    return b"".join(out)


# (the 72 MB fixtures with 80+ byte windows cost 40-110 s each in the numpy stand-in for the kernels; the GPU suite runs them through the CLI)
# (round 6: three more of the 72 MB cases -- 27-66 s each here, a third of the CPU suite's wall clock between them; like the others they
# run through the CLI on the GPU, tests/test_gpu_filegrep.py::test_cli_matches_reference, and through the oracle in tests/test_oracle.py)
_CASES = [c for c in GOLDEN if not c["name"].startswith("syn256") and c["name"] not in ("big_lines_L5", "big_alt_Ol_L5", "cap_big_L5", "big_caret_L5", "big_inx_L5", "big_gap_L5", "big2_gap_L5")]


@pytest.mark.parametrize("case", _CASES, ids=golden_ids(_CASES))
def test_report_matches_reference(case, built):
    if not _supported(case):
        pytest.skip("pattern or mode outside this test (unsupported pattern cases are covered by test_pattern)")
    flags, pattern, paths = split_args(case["args"])
    f = (filegrep.OFFSETS if "-O" in flags else 0) | (filegrep.NOLINE if "-l" in flags else 0) | (filegrep.SINGLE if "-s" in flags else 0)
    db = engine.Database(pattern)
    data = build(case["inputs"][paths[0]])
    for minimal in (True, False):
        out = host_find(db, data, f, so.chunk_size(flags.count("-L")), minimal=minimal)
        assert len(out) == case["stdout_len"]
        assert hashlib.md5(out).hexdigest() == case["stdout_md5"]


def test_report_prefix_and_colour(built):
    db = engine.Database("foo")
    data = np.frombuffer(b"a foo b\nfoo\n", np.uint8)
    starts = np.array([2, 8], np.uint32)
    out = filegrep.report_chunk(db, filegrep.OFFSETS | filegrep.PREFIX | filegrep.COLOR, "dir/f", data, 1000, starts)
    assert out == b"dir/f:Match at offset 1002\na \x1b[7mfoo\x1b[27m b\ndir/f:Match at offset 1008\n\x1b[7mfoo\x1b[27m\n"


def test_report_matches_python_oracle_random(built):
    """Random texts x patterns x flags: product walk == scan_oracle.grab_file."""
    rng = np.random.default_rng(11)
    alphabet = np.frombuffer(b"abcdeffoo0123456789_AZ \n\n", np.uint8)
    for pattern in ["foo", "ff", "f", "[a-z]{2,5}", "abc[0-9]*", "e+", "[A-Za-z_][A-Za-z0-9_]{3,}", r"\d\d", "[^\\n]{4}", "[a-f]{3}",
                    "foo|ab", "a|ab", "ab|a", "fo?o", "(?:f|e){1,3}0", "(?i)Az|f+", "[a-f]{1,2}[0-9]", "(?:ab|cd)?e", "f{2,4}?o", "0|1|2|[3-9]+",
                    "(a|b)0|af", "(f)?o", "f(o){0,2}0", "(?P<w>ab)?c|e",
                    r"\bfoo", r"\Bf", r"o\b", "(?m)^f", "(?m)o$", "^f|e$", r"\b\w+\b", r"\b[a-f_]{2}\B"]:
        db = engine.Database(pattern)
        for trial in range(6):
            n = int(rng.integers(0, 3000))
            data = alphabet[rng.integers(0, alphabet.size, n)]
            for f in (0, 1, 3, 2, 4, 5, 7):
                want = so.grab_file(pattern, data.tobytes(), f, 1 << 30)
                assert host_find(db, data, f, 1 << 30) == want, (pattern, n, f)
                assert host_find(db, data, f, 1 << 30, minimal=False) == want, (pattern, n, f)


def test_next_match_any_restart_sequence(built):
    """gscan_next_match keeps per-alternative answers in its cursor from call to call.  Whatever increasing sequence of
    restart positions it is asked for -- not only the ones the reference's loop would produce -- its answer must be the
    leftmost offset >= s at which the matcher reports a match with the subject starting at s."""
    rng = np.random.default_rng(77)
    alphabet = np.frombuffer(b"abcfoo01_ AZ.\n\n", np.uint8)
    patterns = ["foo", "oo", "ofo", "abab", "[ab][ab]", "[ab]0[ab]", "a|ab", r"\bfoo", r"o\b", "(?m)^f", "e$|o$", "a+b", r"[a-z]+\b", r"\w+@?\w+\.[a-c]+", "(?:fo|ab)+c", "a.*b.*c",
                "(?<=o)o|b(?=c)", r"(a|b)\1|c", r"fo\Ko|ab", r"(?>a+)b|o", "[a-c]{1,20}0", "(?:a|b|c|f|o|0|1|_){3}"]
    for pattern in patterns:
        db = engine.Database(pattern)
        for trial in range(4):
            n = int(rng.integers(1, 600))
            data = alphabet[rng.integers(0, alphabet.size, n)]
            starts = np.zeros(0, np.uint32) if db.info.tier == engine.TIER_ANCHORED else engine_list(db, data)
            cur = engine.Cursor()
            s = 0
            while s < n:
                rc, m0, m1 = db.next_match(data, starts, cur, s)
                want = (0, 0, 0)
                for p in range(s, n):
                    kind, end = db.match_info(data, p, s)
                    if kind:
                        want = (kind, p, end)
                        break
                got = (rc, m0, m1) if rc else (0, 0, 0)
                if r"\K" not in pattern:  # (with \K the reported start is not the offset the match was found at)
                    assert got == want, (pattern, data.tobytes(), s, got, want)
                else:
                    assert (got[0], got[2]) == (want[0], want[2]), (pattern, data.tobytes(), s, got, want)
                s += int(rng.integers(1, 40))


def test_listed_literals_are_walked_without_reading_the_text(built):
    """A plain window that cannot match at two adjacent offsets (here: literals with two different neighbouring bytes) has
    every candidate in the engine's list, so the walk takes "the first listed start >= s" and never reads the chunk -- with
    offsets-only output a fresh mapping is then not faulted in at all (cfg5: a page fault per match was most of its time).
    Stated as: the answers do not change when the text handed to the walk is blanked out.  A window that CAN match at
    neighbouring offsets ("oo", a class pair) is listed by groups only and does read the text."""
    rng = np.random.default_rng(5)
    alphabet = np.frombuffer(b"fo ab\n", np.uint8)
    data = alphabet[rng.integers(0, alphabet.size, 40000)]
    blank = np.zeros_like(data)
    for pattern, reads_text in (("foo", False), ("ab", False), ("abab", False), ("oo", True), ("[ab][ab]", True)):
        db = engine.Database(pattern)
        starts = engine_list(db, data)
        answers = []
        for text in (data, blank):
            cur, s, got = engine.Cursor(), 0, []
            while s < data.size:
                rc, m0, m1 = db.next_match(text, starts, cur, s)
                if rc != 1:
                    break
                got.append((m0, m1))
                s = m1
            answers.append(got)
        assert len(answers[0]) >= 3, pattern
        assert (answers[0] == answers[1]) == (not reads_text), pattern
        assert bool(db.info.textfree) == (not reads_text), pattern  # what FileGrep goes by when it decides whether a small file must be mapped for the report
    assert not engine.Database("foo[a-z]*").info.textfree and not engine.Database(r"\bfoo").info.textfree and not engine.Database("foo|bar").info.textfree


def test_long_lines_all_modes_match_libpcre(built, liboracle):
    """Lines of ~800 bytes (the 511-byte context caps of grab.cc:173,190-197 bite), every output mode incl. -s, patterns of every
    kind the host matcher serves: the product's chunk walk == libpcre under the reference's loop (oracle_scan_chunk)."""
    import ctypes as C

    rng = np.random.default_rng(5)
    alpha = np.frombuffer(b"abcxyz01 ._@()=;" + b"\n", np.uint8)
    w = np.ones(alpha.size)
    w[-1] = 0.02
    w /= w.sum()
    patterns = [r"\w+@\w+\.\w+", r"(?:ab|xy)+z", r"a.*b.*c", r"(\w)\1\1", r"ab\Kc+", r"(?<=\()\w+(?=\))", r"[a-c]+\([a-z0-9, ]*\);", r"(?>a+)b",
                r"x{2,}y{2,}z*0", r"(?m)^\w+ = \w+;$", r"\b(?:[a-c]+_)+[a-c]+\b", "ab", "[a-c]{3,}"]
    for trial in range(5):
        text = alpha[rng.choice(alpha.size, 20000, p=w)].tobytes()
        data = np.frombuffer(text, np.uint8)
        for pattern in patterns:
            db = engine.Database(pattern)
            starts = engine_list(db, data)
            for f in (0, 1, 3, 2, 4, 5):
                out = C.c_void_p()
                n = C.c_size_t()
                assert liboracle.oracle_scan_chunk(pattern.encode(), b"", text, len(text), 0, f, C.byref(out), C.byref(n)) == 0
                want = C.string_at(out, n.value) if n.value else b""
                liboracle.oracle_free(out)
                assert filegrep.report_chunk(db, f, b"", data, 0, starts) == want, (pattern, trial, f)



ENDS_OK = ["[A-Za-z_][A-Za-z0-9_]{3,}", "e+", r"foo\w*", "[a-f]{2}[a-f0-9]*", r"\d\d+", "ab[a-z_]*"]
ENDS_NOT = ["foo", "[a-f]{3}", "abc[0-9]*", "[a-z]{2,5}", "(f)o+", r"\bfoo\w*", "0|1+", r"fo+\b"]


def test_ends_ok_flag(built):
    """gscan_info.ends_ok: one plain alternative, unbounded greedy tail whose class contains the window's first class --
    the byte that stops the tail cannot begin a match."""
    for p in ENDS_OK:
        assert engine.Database(p).info.ends_ok, p
    for p in ENDS_NOT:
        assert not engine.Database(p).info.ends_ok, p


def test_offsets_walk_without_the_text(built):
    """-O -l with the match ends in the list (the device's k_ends pass; here gscan_match_end stands in for it): the walk's
    output equals the reference loop's (scan_oracle.grab_file over libpcre-free `re`) WITHOUT being handed the chunk, for
    the minimal list (group starts) and for the full candidate list, with and without a path prefix, single-match mode
    included.  An end the device left open (0) sends that one step to gscan_next_match, which does need the text."""
    rng = np.random.default_rng(23)
    alphabet = np.frombuffer(b"abcdeffoo0123456789_AZ \n\n", np.uint8)
    for pattern in ENDS_OK:
        db = engine.Database(pattern)
        for trial in range(8):
            n = int(rng.integers(0, 4000))
            data = alphabet[rng.integers(0, alphabet.size, n)]
            cands = db_candidates(db, data)
            for starts in (so.group_starts(cands).astype(np.uint32), cands.astype(np.uint32)):
                ends = np.array([db.match_end(data, int(p)) for p in starts], np.uint32)
                assert (ends > starts).all()
                for f in (filegrep.OFFSETS | filegrep.NOLINE, filegrep.OFFSETS | filegrep.NOLINE | filegrep.PREFIX, filegrep.OFFSETS | filegrep.NOLINE | filegrep.SINGLE):
                    want = so.grab_file(pattern, data.tobytes(), f, 1 << 30, path=b"d/f")
                    got = filegrep.report_chunk(db, f, "d/f", None, 0, starts, ends=ends, clen=n)
                    assert got == want, (pattern, n, f)
                    if len(ends):  # some ends left to the host: same output, with the text at hand
                        holes = ends.copy()
                        holes[rng.integers(0, len(ends), max(1, len(ends) // 3))] = 0
                        assert filegrep.report_chunk(db, f, "d/f", data, 0, starts, ends=holes) == want, (pattern, n, f)
                # the line-printing modes ignore the ends
                assert filegrep.report_chunk(db, filegrep.OFFSETS, "d/f", data, 0, starts, ends=ends) == so.grab_file(pattern, data.tobytes(), 1, 1 << 30)


def _lines_pass(db, data, starts, rng, ask_rate):
    """What the device's line pass (k_lines) hands over for one chunk, restated in Python: per listed start {m1, lb, le,
    goff} + the gathered text.  m1 == 0: not printed (an earlier listed start in the same line); lb == 0xffffffff: ask the
    host (the line runs on past the printed context, a line start or tail out of reach -- here also at random, ask_rate:
    the kernel may ask whenever it likes); else the printed line [lb, le), whose text sits at gather[goff:]."""
    ASK = 0xFFFFFFFF
    content = data.tobytes()
    n = len(content)
    ext = np.zeros((len(starts), 4), np.uint32)
    gathered = bytearray()
    for i, p in enumerate(starts.tolist()):
        nl = content.rfind(b"\n", 0, p)
        ls = nl + 1
        if i > 0 and starts[i - 1] >= ls:
            continue  # not printed
        m1 = db.match_end(data, p)
        end = content.find(b"\n", m1, min(n, m1 + 511))
        le = end if end >= 0 else min(n, m1 + 511)
        runs_on = le < n and content[le] != 0x0A
        if p - ls > 4096 or m1 - p > 4096 or runs_on or rng.random() < ask_rate:
            ext[i] = (1, ASK, 0, 0)
            continue
        lb = max(ls, p - 511)
        goff = len(gathered) if rng.random() > 0.1 else ASK  # (a few lines the device could not gather: taken from the window)
        if goff != ASK:
            gathered += content[lb:le]
        ext[i] = (m1, lb, le, goff)
    return ext, bytes(gathered)


def test_line_pass_with_resync(built):
    """The line-printing modes driven by the device's line pass: printed lines come from the gathered text, "ask the host"
    records go through the reference's loop, and the pass takes over again once a printed line has ended at its newline --
    byte-identical to the reference's loop (scan_oracle.grab_file) on texts with lines far longer than the 511 bytes of
    printed context and the 4 KiB the device searches, with random extra asks, with and without gathered text, -s included."""
    rng = np.random.default_rng(31)
    words = [b"foo", b"foobar_identifier_0123456789", b"x1", b"abc0123456789", b"e", b" ", b"_", b"9", b"zz zz", b"\n", b"\n"]
    for pattern in ["[A-Za-z_][A-Za-z0-9_]{15,}", "foo", "e+", "[0-9]{3}[0-9a-f]*", "[a-f]{2}[a-f0-9]*"]:
        db = engine.Database(pattern)
        assert db.info.lines_ok, pattern
        for trial in range(10):
            parts = []
            for _ in range(int(rng.integers(1, 400))):
                w = words[int(rng.integers(0, len(words)))]
                parts.append(w)
                if rng.random() < 0.03:  # a long run without a newline: 600 .. 6000 bytes
                    parts.append(bytes(rng.choice(np.frombuffer(b"abcdefoo_0189 ", np.uint8), int(rng.integers(600, 6000))).tolist()))
            data = np.frombuffer(b"".join(parts), np.uint8)
            starts = so.group_starts(db_candidates(db, data)).astype(np.uint32)
            for ask_rate in (0.0, 0.3):
                ext, gathered = _lines_pass(db, data, starts, rng, ask_rate)
                for f in (filegrep.OFFSETS, 0, filegrep.OFFSETS | filegrep.PREFIX, filegrep.SINGLE, filegrep.OFFSETS | filegrep.COLOR):
                    want = so.grab_file(pattern, data.tobytes(), f, 1 << 30, path=b"d/f")
                    assert filegrep.report_chunk_ext(db, f, "d/f", data, 0, starts, ext, gathered) == want, (pattern, trial, ask_rate, f)
                    noga = ext.copy()
                    noga[:, 3] = 0xFFFFFFFF
                    assert filegrep.report_chunk_ext(db, f, "d/f", data, 0, starts, noga, None) == want, (pattern, trial, ask_rate, f)


def test_offsets_formatter_digits(built):
    """The -O -l offsets loop (match ends from the device) writes its numbers two digits at a time: every digit count from 1 to
    19, with and without a path prefix, more lines than its 64 KiB buffer holds -- against Python's own formatting."""
    db = engine.Database(r"foo\w*")
    assert db.info.ends_ok
    data = np.frombuffer(b"foo1 " * 5000, np.uint8)
    starts = np.arange(0, data.size, 5, dtype=np.uint32)
    ends = starts + 4
    for off in [0, 3, 7, 96, 995, 9_996, 99_997, 999_998, 10 ** 7 - 1, 10 ** 8, 10 ** 9 + 1, 2 ** 32 - 2, 2 ** 32 + 5, 10 ** 12, 10 ** 15 + 3, 2 ** 62]:
        for flags, prefix in ((filegrep.OFFSETS | filegrep.NOLINE, b""), (filegrep.OFFSETS | filegrep.NOLINE | filegrep.PREFIX, b"some/dir/file.txt:")):
            got = filegrep.report_chunk(db, flags, "some/dir/file.txt", None, off, starts, ends=ends, clen=data.size)
            want = b"".join(prefix + b"Match at offset %d\n" % (off + int(p)) for p in starts)
            assert got == want, (off, flags)
