"""The device's backtracking VM (grab_amd/csrc/vm.h), run on the host: the same source the K3 kernel calls for every
filter hit of an inexact pattern.  It is a FILTER -- only its verdict 0 ("no match starts here") drops a candidate -- so
the property that must hold is one-sided: never 0 where the host matcher (TreeMatch, itself pinned against libpcre by
tests/test_fuzz.py) finds a match.  Checked offset by offset on random patterns, with the subject starting at 0 and at
the offset itself (look-behind, \\b and ^ see the difference); how often the VM agrees exactly is measured too -- a VM
that keeps everything would be sound and useless.  Then the whole host side of the new contract: the list "every device
hit the VM keeps" fed to the reference's loop prints what libpcre prints."""
import ctypes as C
import os
import random

import numpy as np
import pytest

from grab_amd import engine, filegrep
from inputs import db_candidates, engine_list
from test_fuzz import ATOMS, BIN_ATOMS, gen, gen_calls_and_conditions, make_texts, ref_chunk

TARGETS = [r"(\w)\1{3,}x|foobardoes(?=not)", r"[a-z]+\([a-z0-9, ]*\);", r"[a-z]+_[0-9]+\.[a-z]+", r"(?:foo|bar|ab)+baz", r"\w+@\w+\.com",
           r"a.*b.*c", r"(?:ab|cd)+?e", r"e++f", r"(?>a+)b|(?>a+)a", r"(a|b\1)+c|z", r"(?i)(ab)\1+", r"foo(?!bar)\w+", r"x(?=(a))ab|ab",
           r"(?:a|b)*?c{2,3}d", r"[0-9]{1,40}x", r"(?s)a.{2,}?b", r"(a)(b)?\2c|abc", r"(?:(?:ab)+c)+d", r"a{2,}+b", r"(?:\s|x)+y$",
           r"(a)(?:\1b|c)d*e*",  # (a back reference at the head of an alternative: the split's first-byte prediction must not rule it out)
           # conditional groups: compiled into two guarded branches (vm_compile.cc)
           r"(a)?(?(1)b|c)d*e", r"x(a)?(?(1)b|c)", r"(?(?=a)ab|cd)e+f+", r"(?(?!a)b|ab)c+d+", r"(?:a|(b))(?(1)c|d)e?f", r"(?<n>a)?(?(<n>)b)c+d", r"a(?(?!(b))c)d*e|ab"]


def verdicts(db, text):
    data = np.frombuffer(text, np.uint8)
    out = []
    for p in range(len(text)):
        for s0 in {0, p}:
            v = db.vm_verdict(data, p, s0)
            e0 = engine.resource_errors()
            k = db.match_info(data, p, s0)[0]
            if engine.resource_errors() != e0:
                continue  # the host matcher gave up at its own limit: nothing to compare
            out.append((p, s0, v, k))
    return out


@pytest.mark.parametrize("pattern", TARGETS)
def test_vm_on_named_patterns(pattern, built):
    db = engine.Database(pattern)
    texts = [b"xx aaaax bbbbbx foobardoesnot foo(a, b); foobarbaz a@b.com abaz", b"abcde ababe cde e eef aab aaa abab ABab abAB 11ax x9x",
             b"foobar foobaz food(x); f_1.a a_12.bc  \tx y\n", b"abc abbc ab c accd bccd aaccd ccd", b"aXXb a\nb a12b ab", b"ababcabcd abd ababd", b" x y\n  y",
             b"abde ce abeef cdef bcd abccdd bcef adf abc cd acde ade abd aab ac aabde"] + make_texts(3)
    hits = agree = 0
    for t in texts:
        for p, s0, v, k in verdicts(db, t):
            assert v in (0, 1, 2)
            assert not (v == 0 and k != 0), (pattern, t, p, s0, "the VM drops an offset at which a match starts")
            hits += k != 0
            agree += (v == 1) == (k != 0)
    assert hits > 0, "the texts hold matches of every pattern"


@pytest.mark.parametrize("seed", [41, 42, 43, 44])
def test_vm_never_drops_a_match(seed, built):
    rng = random.Random(seed)
    texts = make_texts(seed)[:10]
    total = exact = unknown = programs = 0
    for _ in range(500):
        pat = gen_calls_and_conditions(rng) if seed == 43 else gen(rng, BIN_ATOMS if seed == 44 else ATOMS)
        try:
            db = engine.Database(pat)
        except ValueError:
            continue
        if db.minlen < 0 or db.vm_verdict(np.zeros(1, np.uint8), 0, 0) < 0:
            continue  # matches "" (every file is skipped, Q2) / no VM program (too big for the VM's limits)
        programs += 1
        for t in texts:
            for p, s0, v, k in verdicts(db, t):
                assert not (v == 0 and k != 0), (pat, t, p, s0)
                total += 1
                exact += (v == 1) == (k != 0) and v != 2
                unknown += v == 2
    assert programs > (100 if seed == 43 else 200)
    assert exact > 0.995 * total, (exact, total, unknown)  # a filter that keeps everything would be sound and useless


@pytest.mark.parametrize("seed", [61, 62, 63])
def test_prefix_probe_is_one_sided(seed, built):
    """The two-byte table in front of the VM (DevProgram::vm_pair) is filled by running the host matcher on two-byte prefixes in a
    probe mode (tree_prefix_viable) that may only answer "no" when the failure is decided by those bytes alone.  Checked the
    other way round: wherever the matcher finds a match at offset p of a text, the probe on text[p:p+1] and text[p:p+2] must
    say "viable", and so must the table of a database that has one; how often the probe does say no is measured too."""
    rng = random.Random(seed)
    texts = make_texts(seed)[:10] + [b"xx aaaax bbbbbx foobardoesnot foo(a, b); a@b.com"]
    checked = said_no = tables = 0
    pats = TARGETS if seed == 61 else [gen(rng, BIN_ATOMS if seed == 63 else ATOMS) for _ in range(400)]
    for pat in pats:
        # (the table belongs to K3's VM form, which since round 6 only the patterns outside the resolve pass use: compiled with that
        # pass switched off, every inexact pattern that never looks behind gets one, as in rounds 2-5)
        os.environ["GSCAN_NO_RESOLVE"] = "1"
        try:
            db = engine.Database(pat)
        except ValueError:
            continue
        finally:
            del os.environ["GSCAN_NO_RESOLVE"]
        if db.minlen < 0:
            continue
        # (the probe reads nothing in front of the prefix: only patterns the device may judge by offset alone get a table)
        has_table = db.vm_pair(0, 0) >= 0
        tables += has_table
        if not has_table:
            continue
        for t in texts:
            data = np.frombuffer(t, np.uint8)
            for p in range(len(t)):
                e0 = engine.resource_errors()
                k = db.match_info(data, p, p)[0]
                if engine.resource_errors() != e0:
                    continue
                one, two = db.prefix_viable(t[p:p + 1]), db.prefix_viable(t[p:p + 2])
                if k:
                    assert one and two, (pat, t, p, "the probe rules out a prefix a match begins with")
                    if p + 1 < len(t):
                        assert db.vm_pair(t[p], t[p + 1]) == 1, (pat, t, p)
                checked += 1
                said_no += not two
    assert tables > (10 if seed == 61 else 20) and checked > 1000
    assert said_no > 0.2 * checked  # (a probe that never says no would be sound and useless)


@pytest.mark.parametrize("seed", [51, 52])
def test_vm_filtered_list_prints_what_pcre_prints(seed, built, liboracle):
    """For patterns whose candidates the device confirms (info.vm): the list the kernel produces -- every device hit its VM
    keeps, no group-start compression -- under gscan_next_match / the reference's loop, against libpcre under the same loop."""
    rng = random.Random(seed)
    texts = make_texts(seed) + [b"xx aaaax bbbbbx foobardoesnot foo(a, b); foobarbaz a@b.com abaz", b"ab ab ab abc\nabab c abcabc"]
    done = 0
    pats = list(TARGETS) if seed == 51 else []
    while done < 120:
        pat = pats.pop() if pats else gen(rng)
        pb = pat.encode("latin-1")
        ml = C.c_int(-9)
        if liboracle.oracle_minlen(pb, C.byref(ml)) != 0:
            continue
        try:
            db = engine.Database(pat)
        except ValueError:
            continue
        if not db.info.vm or db.info.resolve or db.minlen < 0:
            continue
        assert db.info.tier == engine.TIER_BUCKET and not db.info.exact
        for text in texts:
            data = np.frombuffer(text, np.uint8)
            starts = engine_list(db, data)
            assert np.all(np.isin(starts, db_candidates(db, data)))
            for f in (1 | 2, 1, 0):
                e0, g0 = liboracle.oracle_resource_errors(), engine.resource_errors()
                want = ref_chunk(liboracle, pb, text, f) if ml.value <= len(text) else b""
                got = filegrep.report_chunk(db, f, b"", data, 0, starts) if db.minlen <= len(text) else b""
                if liboracle.oracle_resource_errors() != e0 or engine.resource_errors() != g0:
                    continue
                assert got == want, (pat, text, f)
        done += 1
