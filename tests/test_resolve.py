"""The resolve pass on the CPU (round 6; gscan_info.resolve): for every pattern that is not one plain window the
device settles "which offsets are matches and where they end" itself -- the kernels list every offset where a START window fits
and k_resolve runs the pattern's VM program there with the chunk's real bytes in front.  Three things make that exact, each
checked here without a GPU (tests/test_gpu_resolve.py does the kernels):
  * the VM's full answer -- verdict, match end, "the path closed a capturing group" -- equals the host matcher's (which
    tests/test_fuzz.py pins against libpcre) offset by offset, with the subject starting at 0 and at the offset itself;
  * the list k_resolve leaves (tests/inputs.py:resolved_list -- the same VM source run on the host over the start windows'
    hits) walked by gscan_next_resolved prints what libpcre prints under the reference's loop (src/grab.cc:175-213), in every
    output mode -- and so does the UNRESOLVED list (every start-window hit, each put to the host matcher);
  * what the host still asks its own matcher about: the `reach` offsets behind a restart position (\\b ^ look-behind see
    nothing in front of it), unless the byte in front of it is to the pattern what the subject start is."""
import ctypes as C
import random

import numpy as np
import pytest

from grab_amd import engine, filegrep
from inputs import END_ASK, END_CAPTURES, engine_list, resolved_list
from test_fuzz import ATOMS, BIN_ATOMS, gen, gen_calls_and_conditions, make_texts, ref_chunk
from test_vm import TARGETS

# VERDICT r5's table + relatives: (pattern, resolve, reach)
KINDS = [
    (r"\b[A-Za-z_]\w*\s*\(", 1, 1), (r"(?<=\$)\d+", 1, 1), (r"\s\w{8,}\s", 1, 0), (r"\b[A-Z][a-z]+\b", 1, 1), (r"\([^()]*\)", 1, 0), (r"\b[a-z]{3,}\b", 1, 1),
    (r"\w+(?=\()", 1, 0), ("foo|bar", 1, 0), (r"\bfoo\b", 1, 1), (r"(?<=abc)d+", 1, 3), (r"(?<=a|bc)d", 1, 2), (r"x(?<=\bx)y", 1, 2), (r"(?m)^\w+", 1, 1),
    (r"\d+\.\d+", 1, 0), ("a+b", 1, 0), (r"^foo|bar", 1, 1),
    # one plain window: the kernels and the per-record passes of rounds 1-3 settle these on their own
    ("foobardoesnotexist", 0, 0), ("[A-Za-z_][A-Za-z0-9_]{15,}", 0, 0), ("[0-9]{16}", 0, 0),
    # \K: where a match is REPORTED to start is not something the VM tracks
    (r"foo\Kbar|baz", 0, 0),
    # start windows that would list every byte of a line where the hit windows list next to nothing
    (r".*foobar", 0, 0),
]


@pytest.mark.parametrize("pattern,resolve,reach", KINDS)
def test_which_patterns_the_device_resolves(pattern, resolve, reach, built):
    info = engine.Database(pattern).info
    assert (info.resolve, info.reach if resolve else 0) == (resolve, reach), pattern
    if resolve:
        assert not info.vm and 1 <= info.n_windows <= 64


def _lo(liboracle):
    liboracle.oracle_resource_errors.restype = C.c_uint64
    return liboracle


EXTRA = [p for p, r, _ in KINDS if r] + [r"(?<!a)b\w", r"\Bab\B", r"(a)b|cd", r"(?i)\bxyzzy\b|\bplugh\b", r"(?<=\n)a|b\b", r"\b\w+\b \b", r"(?m)^a|(?<=b)c"]
TEXTS = [b"xx aaaax bbbbbx foobardoesnot foo(a, b); foobarbaz a@b.com abaz $12 a$3 cd abcc bcd abcd bcdd 1.5 22.75x",
         b"ab ab ab abc\nabab c abcabc\nfoo bar (x) (y(z)) Hello World_1 helloworld9 \nfoo\nbar foo", b"a\nb\nab\n\nba b", b"xyzzy Plugh XYZZY_ plugh\n"]


@pytest.mark.parametrize("seed", [81, 82, 83])
def test_vm_answer_equals_the_host_matcher(seed, built):
    rng = random.Random(seed)
    texts = make_texts(seed)[:6] + TEXTS[:2]
    pats = list(TARGETS) + EXTRA if seed == 81 else []
    programs = checked = unknown = 0
    for _ in range(260):
        pat = pats.pop() if pats else (gen_calls_and_conditions(rng) if seed == 82 else gen(rng, BIN_ATOMS if seed == 83 else ATOMS))
        try:
            db = engine.Database(pat)
        except ValueError:
            continue
        if db.minlen < 0 or db.vm_verdict(np.zeros(1, np.uint8), 0, 0) < 0:
            continue
        programs += 1
        for t in texts:
            data = np.frombuffer(t, np.uint8)
            for p in range(len(t)):
                for s0 in {0, p}:
                    e0 = engine.resource_errors()
                    k, kend = db.match_info(data, p, s0)
                    if engine.resource_errors() != e0:
                        continue
                    v, vend, vcap = db.vm_match(data, p, s0)
                    checked += 1
                    if v == 2:
                        unknown += 1
                        continue
                    assert (v == 1) == (k != 0), (pat, t, p, s0)
                    if v == 1:
                        assert vend == kend and (vcap != 0) == (k == 2), (pat, t, p, s0, vend, kend, vcap, k)
    assert programs > (60 if seed == 82 else 100) and checked > 30_000 and unknown < 0.01 * checked


@pytest.mark.parametrize("seed", [91, 92, 93])
def test_resolved_walk_prints_what_pcre_prints(seed, built, liboracle):
    lo = _lo(liboracle)
    rng = random.Random(seed)
    texts = make_texts(seed) + TEXTS
    pats = list(TARGETS) + EXTRA if seed == 91 else []
    done = asks = 0
    while done < (len(TARGETS) + len(EXTRA) if seed == 91 else 90):
        pat = pats.pop() if pats else (gen_calls_and_conditions(rng) if seed == 92 else gen(rng, BIN_ATOMS if seed == 93 else ATOMS))
        pb = pat.encode("latin-1")
        ml = C.c_int(-9)
        if lo.oracle_minlen(pb, C.byref(ml)) != 0:
            continue
        try:
            db = engine.Database(pat)
        except ValueError:
            continue
        if db.minlen < 0 or not db.info.resolve:
            if not pats and seed == 91:
                done += 1
            continue
        for text in texts:
            data = np.frombuffer(text, np.uint8)
            raw = engine_list(db, data)
            rs, re_ = resolved_list(db, data)
            assert np.all(np.isin(rs, raw)) and len(rs) == len(re_)
            asks += int(np.count_nonzero(re_ == END_ASK))
            for f in (1 | 2, 1, 0, 2, 1 | 2 | 4, 4):
                e0, g0 = lo.oracle_resource_errors(), engine.resource_errors()
                want = ref_chunk(lo, pb, text, f) if ml.value <= len(text) else b""
                got_raw = filegrep.report_chunk(db, f, b"", data, 0, raw) if db.minlen <= len(text) else b""
                got = filegrep.report_chunk(db, f, b"", data, 0, rs, ends=re_) if db.minlen <= len(text) else b""
                if lo.oracle_resource_errors() != e0 or engine.resource_errors() != g0:
                    continue
                assert got == want, (pat, text, f, "resolved list")
                assert got_raw == want, (pat, text, f, "every start-window hit, put to the host matcher")
        done += 1


def test_a_match_that_sets_a_group_ends_the_chunk(built):
    """ovector[3] (src/grab.cc:171,179): the device marks such a record END_CAPTURES and the loop stops there."""
    db = engine.Database("(a)b|cd")
    data = np.frombuffer(b"cd cd ab cd", np.uint8)
    rs, re_ = resolved_list(db, data)
    assert rs.tolist() == [0, 3, 6, 9] and re_.tolist() == [2, 5, END_CAPTURES, 11]
    assert filegrep.report_chunk(db, 3, b"", data, 0, rs, ends=re_) == b"Match at offset 0\nMatch at offset 3\n"


def test_the_host_asks_only_behind_a_restart_position(built):
    """\\bfoo on "foofoo foo_": the device's list (bytes in front visible; offset 0 has none and is never listed) holds offset 7 only; the reference, which restarts
    pcre_exec AT the end of a match, also finds the second foo -- the host's own test at the restart position."""
    db = engine.Database(r"\bfoo")
    data = np.frombuffer(b"foofoo foo_", np.uint8)
    rs, re_ = resolved_list(db, data)
    assert rs.tolist() == [7] and re_.tolist() == [10]  # (offset 0 has no byte in front of it: never listed, the host's)
    assert filegrep.report_chunk(db, 3, b"", data, 0, rs, ends=re_) == b"Match at offset 0\nMatch at offset 3\nMatch at offset 7\n"
    # a byte in front of the restart position that is to the pattern what the subject start is: nothing to ask
    first = (C.c_uint8 * 256)()
    assert engine.lib().gscan_db_first(db._h, first) == 1
    assert first[ord("f")] & 1 and not first[ord("o")] & 1 and first[ord(" ")] & 2 and not first[ord("o")] & 2 and first[ord("(")] & 2
