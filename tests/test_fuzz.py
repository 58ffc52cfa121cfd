"""Grammar-based differential fuzz of the pattern compiler + host match rule against libpcre.

Random patterns are drawn from the grammar the engine claims to take (byte classes, greedy / lazy / possessive repeats,
alternation, plain / capturing / option-scoped / atomic groups, back references, look-ahead and look-behind, ^ $ \\b \\B \\A \\z \\Z, (?i) (?m) (?s)).  For every pattern PCRE
accepts and the engine does not refuse, the product's chunk walk (grab_report_chunk -> gscan_next_match) -- fed with
exactly what the kernels are specified to report for the text (device windows: tests/inputs.py:db_candidates) -- must
print what the reference's loop prints; the oracle runs pcre_exec the way /root/reference/src/grab.cc:175-213 does, with
ovector[3] and the subject restarted at every match.  minlen must equal PCRE_INFO_MINLENGTH.

Seeds are fixed; the campaign that found the bugs pinned in REGRESSIONS ran a few hundred thousand patterns."""
import ctypes as C
import os
import random
import sys

import numpy as np
import pytest

from conftest import ROOT
from grab_amd import engine, filegrep
from inputs import db_candidates, engine_list

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import scan_oracle as so  # noqa: E402

# a second vocabulary for texts with control characters and high bytes: the escapes whose C-locale meaning is easy to get wrong
BIN_ATOMS = ["\\S", "\\D", "\\s", "\\h", "\\x00", "\\xff", "[\\x80-\\xff]", "\\v", "\\N", "[[:alpha:]]", "\\H", "[[:^space:]]", "\\V", "[[:punct:]]", "\\x85",
             "a", "b", "Z", " ", "\\n", ".", "0", "[ab]", "[^a]", "\\w", "\\W", "_",
             "\\p{L}", "\\P{L}", "\\p{Lu}", "\\pN", "\\p{Xwd}", "\\p{Xsp}", "[\\p{Ll}0]", "\\p{Latin}", "\\R", "\\X", "\\o{12}"]
ATOMS = ["a", "b", "c", "x", " ", "\\n", ".", "0", "1", "[ab]", "[^a]", "[a-c]", "\\w", "\\d", "\\s", "\\W", "[b0 ]", "\\.", "[^\\n]", "A", "[x.]"]
QUANTS = ["?", "*", "+", "{2}", "{1,2}", "{0,2}", "{2,}", "{1,3}", "??", "+?", "*?", "{1,2}?", "?+", "?", "+ ?", "* +", "{1,2} ?"]


def gen(rng, atoms=None):
    atoms = atoms or ATOMS

    def atom(d):
        r = rng.random()
        if r < 0.04:  # back references (most draws refer to a group that does not exist: PCRE rejects those)
            return rng.choice(["\\1", "\\2", "\\1", "\\g{-1}", "\\3"])
        if r < 0.70 or d > 2:
            return rng.choice(atoms)
        if r < 0.85:
            return "(?:" + alt(d + 1) + ")"
        if r < 0.91:
            return "(" + alt(d + 1) + ")"
        if r < 0.94:
            return "(?i:" + alt(d + 1) + ")"
        if r < 0.955:
            return rng.choice(["(?=", "(?!"]) + alt(d + 1) + ")"
        if r < 0.975:  # look-behind: fixed-length alternatives
            return rng.choice(["(?<=", "(?<!"]) + "|".join("".join(rng.choice(atoms) for _ in range(rng.choice([1, 1, 2, 3]))) for _ in range(rng.choice([1, 1, 2]))) + ")"
        return "(?>" + alt(d + 1) + ")"

    def piece(d):
        a = atom(d)
        return a if rng.random() < 0.55 else a + rng.choice(QUANTS)

    def seq(d):
        parts = []
        for _ in range(rng.choice([1, 1, 2, 2, 3, 4])):
            if rng.random() < 0.12:
                parts.append(rng.choice(["\\b", "\\B", "^", "$", "\\A", "\\z", "\\Z", "\\K"]))
            parts.append(piece(d))
        if rng.random() < 0.10:
            parts.append(rng.choice(["\\b", "$", "\\B", "\\z"]))
        return "".join(parts)

    def alt(d):
        return "|".join(seq(d) for _ in range(rng.choice([1, 1, 1, 2, 2, 3])))

    p = alt(0)
    if rng.random() < 0.25:
        p = rng.choice(["(?i)", "(?m)", "(?s)", "(?im)", "(?ms)", "(?x)", "(?xi)", "(?x) # c\n"]) + p
    return p


def ref_chunk(liboracle, pat, text, flags):
    out = C.c_void_p()
    n = C.c_size_t()
    assert liboracle.oracle_scan_chunk(pat, b"", text, len(text), 0, flags, C.byref(out), C.byref(n)) == 0
    r = C.string_at(out, n.value) if n.value else b""
    liboracle.oracle_free(out)
    return r


def check(liboracle, pat, texts):
    """None if the pattern is outside what is compared, else the number of texts compared (asserts on any difference)."""
    pb = pat.encode("latin-1")
    ml = C.c_int(-9)
    if liboracle.oracle_minlen(pb, C.byref(ml)) != 0:
        return None  # PCRE rejects it (FileGrep::prepare asks PCRE first and reports its error)
    try:
        db = engine.Database(pat, pcre_checked=True)  # (FileGrep::prepare's sequence: pcre_compile first, then the engine's compiler)
    except engine.Unsupported:
        return None
    assert db.minlen == ml.value, (pat, db.minlen, ml.value)
    if db.minlen < 0:
        return None  # can match "": every file is skipped (Q2)
    for text in texts:
        data = np.frombuffer(text, np.uint8)
        starts = np.zeros(0, np.uint32) if db.info.tier == engine.TIER_ANCHORED else engine_list(db, data)
        for f in (1 | 2, 1, 0):
            e0, g0 = liboracle.oracle_resource_errors(), engine.resource_errors()
            want = ref_chunk(liboracle, pb, text, f) if ml.value <= len(text) else b""
            got = filegrep.report_chunk(db, f, b"", data, 0, starts) if db.minlen <= len(text) else b""
            if liboracle.oracle_resource_errors() != e0 or engine.resource_errors() != g0:
                continue  # PCRE_ERROR_MATCHLIMIT / the host matcher's own limit: where an engine gives up is its own business
            assert got == want, (pat, text, f)
    return len(texts)


def make_texts(seed):
    nrng = np.random.default_rng(seed)
    alpha = np.frombuffer(b"abcxA01 .\n\nab  ", np.uint8)
    return [alpha[nrng.integers(0, alpha.size, int(nrng.integers(1, 120)))].tobytes() for _ in range(14)] + \
           [b"a", b"ab", b"\n", b"abcabc abc\nabc", b"aaaa", b"a.b a1b ab\n"]


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
def test_random_patterns_match_pcre(seed, built, liboracle):
    rng = random.Random(seed)
    texts = make_texts(seed)
    tested = 0
    for _ in range(1000):
        tested += check(liboracle, gen(rng), texts) is not None
    assert tested > 450  # (the rest: rejected by PCRE -- references to groups that do not exist --, patterns that can match "" -- every file is skipped, Q2 -- and a few per cent refused)


@pytest.mark.parametrize("seed", [21, 22])
def test_random_patterns_on_binary_text(seed, built, liboracle):
    """The same comparison over texts with NUL, VT, FF, CR, NEL (0x85), NBSP (0xa0) and other high bytes, patterns drawn from
    the escapes whose C-locale sets are easy to get wrong (\\s \\S \\h \\v \\N [[:classes:]] \\xhh)."""
    rng = random.Random(seed)
    nrng = np.random.default_rng(seed)
    alpha = np.frombuffer(b"abZ01 .\n\nab  \x00\xff\x85\x0b\r\t\x0c\xa0\xe9_", np.uint8)
    texts = [alpha[nrng.integers(0, alpha.size, int(nrng.integers(1, 120)))].tobytes() for _ in range(14)] + [b"\x85", b"\xa0 \x0b", b"a\r\n\x00b"]
    tested = 0
    for _ in range(700):
        tested += check(liboracle, gen(rng, BIN_ATOMS), texts) is not None
    assert tested > 250


@pytest.mark.parametrize("seed", [31, 32])
def test_backrefs_to_groups_closed_inside_assertions(seed, built, liboracle):
    """Capturing groups inside positive / negative look-arounds with a back reference to them behind the assertion
    (the unfolder had dropped such paths as dead)."""
    rng = random.Random(seed)
    texts = make_texts(seed) + [b"xx xa", b"abab aab", b"a1a1 0a0a"]
    tested = 0
    for _ in range(400):
        body = "|".join(rng.choice(ATOMS) + rng.choice(["", "", "?", "+"]) for _ in range(rng.choice([1, 1, 2])))
        look = rng.choice(["(?=", "(?=", "(?!", "(?<="]) + ("(" + rng.choice(ATOMS) + ")" if rng.random() < 0.3 else "(" + body + ")") + ")"
        if look.startswith("(?<="):
            look = "(?<=(" + rng.choice(ATOMS) + "))"
        tail = rng.choice(["\\1", "\\1x", "\\1" + rng.choice(QUANTS), rng.choice(ATOMS) + "\\1", "\\1|" + rng.choice(ATOMS) + rng.choice(ATOMS)])
        pat = rng.choice(["", "", rng.choice(ATOMS), rng.choice(ATOMS) + "|"]) + look + tail + rng.choice(["", "| xa", "|" + rng.choice(ATOMS) + "{2}"])
        tested += check(liboracle, pat, texts) is not None
    assert tested > 60


def gen_calls_and_conditions(rng):
    """The round-2 grammar: conditional groups (on a group, on (R) / (Rn), on an assertion, DEFINE), subroutine calls and
    recursion ((?R) (?1) (?-1) (?+1) \\g<1> (?&n)), (*FAIL), callouts, (?U) (?J), branch-reset groups, next to the constructs of gen()."""
    atoms = ATOMS

    def atom(d):
        r = rng.random()
        if r < 0.04:
            return rng.choice(["\\1", "\\2", "\\g{-1}"])
        if r < 0.10:
            return rng.choice(["(?1)", "(?2)", "(?R)", "(?-1)", "(?+1)", "\\g<1>", "(?&n)"])
        if r < 0.62 or d > 2:
            return rng.choice(atoms)
        if r < 0.69:
            return "(?:" + alt(d + 1) + ")"
        if r < 0.72:
            return "(?|" + alt(d + 1) + ")"
        if r < 0.80:
            return "(" + alt(d + 1) + ")"
        if r < 0.83:
            return "(?<n>" + alt(d + 1) + ")"
        if r < 0.93:
            c = rng.choice(["1", "2", "R", "R1", "<n>", "?=" + seq(d + 1), "?!" + seq(d + 1), "?<=" + rng.choice(atoms), "DEFINE", "-1", "+1"])
            body = seq(d + 1) if rng.random() < 0.35 or c == "DEFINE" else seq(d + 1) + "|" + seq(d + 1)
            return "(?(" + c + ")" + body + ")"
        if r < 0.95:
            return rng.choice(["(*F)", "(*FAIL)", "(?C)", "(?C7)"])
        if r < 0.975:
            return rng.choice(["(?=", "(?!"]) + alt(d + 1) + ")"
        return "(?>" + alt(d + 1) + ")"

    def piece(d):
        a = atom(d)
        return a if rng.random() < 0.6 else a + rng.choice(QUANTS)

    def seq(d):
        parts = []
        for _ in range(rng.choice([1, 1, 2, 2, 3])):
            if rng.random() < 0.08:
                parts.append(rng.choice(["\\b", "\\B", "^", "$"]))
            parts.append(piece(d))
        return "".join(parts)

    def alt(d):
        return "|".join(seq(d) for _ in range(rng.choice([1, 1, 1, 2, 2, 3])))

    p = alt(0)
    if rng.random() < 0.2:
        p = rng.choice(["(?i)", "(?U)", "(?s)", "(?Ui)", "(?x)", "(?J)"]) + p
    return p


@pytest.mark.parametrize("seed", [41, 42, 43, 44])
def test_random_conditionals_and_subroutine_calls_match_pcre(seed, built, liboracle):
    rng = random.Random(seed)
    texts = make_texts(seed)
    tested = 0
    for _ in range(900):
        tested += check(liboracle, gen_calls_and_conditions(rng), texts) is not None
    assert tested > 200  # (the rest: rejected by PCRE -- calls of groups that do not exist --, patterns that can match "", and the refused quirk shapes)


# found by the campaign (each one printed something else than the reference before its fix)
REGRESSIONS = [
    (r"\W*?0|\b\n", b"  \n0a c A0x0c\n 0\nbaAacA11 c  a"),                      # a group of hits starting AT the restart position (list cursor)
    (r"(?ms)1c{2}|b{2,}[ab]|$[^a]{2,}\n.", b"bb.bA1a.  .1a  a\n\nc  \nAA\nc...a.\n b bac x"),  # (?m)$ in front of a gapped path
    (r"^\W*?\w[^a]", b"\nc \n\n ab1cca\na \n0aA c."),                            # a device window with context cannot report offset 0
    (r"\Bc{0,2} *?[^a]\W", b" 1\nA\n1\n\n  b.b.bbc c..b.x  \naaxa"),
    (r"\b0{1,2}|[b0 ]*0", b"b0\nxxcb\n0caAa01 \nxb0a0.aabbax  c00x\n"),
    (r"(?m)\z |\d[^a]", b"1b 2  \n"),                                           # PCRE_INFO_MINLENGTH counts branches that can never match
    (r"(?m)[a-c]{2,}.|x*?0{1,2}.", b"bb1ax01Axb\n\n A\nbx1b 0bbAc.0xbbxAa\n"),
    (r"(?i:0?? *?\b[b0 ]?[^a])|x|0a", b".bx0a \nac 0ac0ac0  a\n 1ca.."),
    (r"\A[^a]\n|x\d", b"c\n. cba .  x1 c \nb1.ab1c 1b \n"),
    (r"a(?(?!(b)))", b"xab a"),              # what a NEGATIVE condition's assertion captured stays captured (JIT): rc 0 at offset 1
    (r"((R))?a(?(?!()))", b"xab a"),
    (r"()(](?2)){2}", b"]] ]]] ]"),          # PCRE_INFO_MINLENGTH of a counted repeat of a group that calls itself (3, not 2)
    (r"[\Qa\E-z]x", b"-x mx ax zx"),          # a quoted byte may begin a class range
]

# libpcre quirks around conditional groups and subroutine calls that are refused rather than imitated (pattern.cc); should
# one of them be taken after all, the output must still be the reference's
CALL_AND_CONDITION_QUIRKS = [(r"(?=.*)[a-c]", b" b\nb x b"), (r"(?=.*? [x.]* +)[a-c]+", b"a x \n b ."), (r"^(?=.*1)(?=.*a)\w{3,}", b"a1b\nab1 zz\n1a"),
                             (r"(?>(?=\xff){2}[0]?\xff(N)?)", b"1\xff \xff\xff"), (r"(?>(?=\b\xff){2,}[\p{Ll}0]?\xff(?<!\pN)+ ?)", b"1\xff0\n a\xff \xff"),
                             (r"(a++){2}|(?1)", b"xa aa"),
                             (r"(?= )c* ", b". "), (r"(?=1)c*1", b"1 11 c1"), (r"(?Ui)(?= + ?)\Bc* +", b"a ba\n0.cb1 a\n. A \n "),
                             (r"(?<n>( ?\1?)\2(()( )|[^a]()))|.", b"\na b"), (r"(( ?\3?)\2[^a])()|.", b"\na  b"),  # (a dead path shadowed a live one with the same window)
                             (r"(?=1[a-c]| 1{1,3})1", b"1a 1a1  1b"), (r"(?)\B(a*?\s{1})]|\g<1>", b" 0 a ]"), (r"\g<1>(?(DEFINE)(()(c{1})))* ", b" c  cc "),
                             (r"(?|a{1}((?+1))?[c])|\b(1*[a])", b"abac 1a ac"), (r"(?)(?(2)(](a?))?\1{|(?2)\s{2})", b"0\n  ]a{"),
                             (r"(?(DEFINE)^)\w", b"ab cd"), (r"(?:(?(?=^))[.])", b"a. ."), (r"(?1)* (\s)?", b"\n AAab\n  \n"), (r"\g<1>?(?:b([\n]([b])){0})", b"bb\naa c.b0c bxb\nbb A"),
                             (r"b(?1)c|(A*)x", b"bx bAx bAAx x"), (r"b(?1)c|([^x]*)x", b"bAAx x"), (r"(?<n>([b](?+1)){(?(<n>)))|[\n]|(?:(A*)()x)", b" Ab.bbbx \n\n  acx")]


@pytest.mark.parametrize("pattern,text", CALL_AND_CONDITION_QUIRKS)
def test_fuzz_call_and_condition_quirks(pattern, text, built, liboracle):
    check(liboracle, pattern, [text] + make_texts(7))  # (None = refused: fine; a difference asserts)

# a group closed inside a positive assertion keeps its capture, so a back reference behind it is alive: the unfolder had dropped
# such a path as dead and printed matches the reference does not.  Now the path survives; with nothing fixed in
# front of the reference the pattern is refused loudly.  Either way: never a different output.
ASSERTION_CAPTURES = [(r"(?=(x))\1x| xa", b"xx xa"), (r"(?=(x))\1x", b"xx xa xx"), (r"(?<=(a))\1b|c", b"aab c ab"), (r"a(?=(x))\1x| xa", b"axx xa"),
                      (r"x(?=(x))\1| xa", b"xxx xa xx")]


@pytest.mark.parametrize("pattern,text", ASSERTION_CAPTURES)
def test_fuzz_assertion_captures(pattern, text, built, liboracle):
    check(liboracle, pattern, [text] + make_texts(7))  # (None = refused: fine; a difference asserts)


# ... and patterns that were accepted wrongly: an assertion that always holds where a greedy repeat stops (\w+\b, (?m).*$)
# had been dropped although more pattern followed the repeat (PCRE then backtracks into it and the assertion decides).
# Such a path now stops in front of the repeat and the host matcher confirms every candidate (gscan_info.exact == 0).
BACKTRACK_INTO_SETTLED = [r"(?im)[^\n]+$[^\n]{2}", r"\B\W\.?\w{2,}\b\d|\A[^a]\n", r"(?i:\w+\b|  )1|A{2,}\d\z", r"\w+\bx", r"(?m)a.*$\nb"]


@pytest.mark.parametrize("pattern", BACKTRACK_INTO_SETTLED)
def test_fuzz_settled_repeat_with_more_behind(pattern, built, liboracle):
    assert not engine.Database(pattern).info.exact
    texts = [b"ab cd\nxy\n", b"a.bc1 x\n", b"  1 ab1 AA1", b"abx a bx\n", b"a..\nb a\nb"]
    assert check(liboracle, pattern, texts + make_texts(5)) is not None



@pytest.mark.parametrize("pattern,text", REGRESSIONS)
def test_fuzz_regressions(pattern, text, built, liboracle):
    assert check(liboracle, pattern, [text] + make_texts(7)) is not None
