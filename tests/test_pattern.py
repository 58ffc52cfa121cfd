"""Pattern compiler (grab_amd/csrc/pattern.cc through the C ABI): tiering, PCRE-equal minlen,
class tables and greedy match ends, checked against libpcre (liboracle) and Python's re."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from grab_amd import engine

SUPPORTED = [
    # pattern, tier, minlen
    ("foo", engine.TIER_LITERAL, 3),
    ("foobardoesnotexist", engine.TIER_LITERAL, 18),
    ("[A-Za-z_][A-Za-z0-9_]{15,}", engine.TIER_CLASSRUN, 16),
    ("abc[0-9]*", engine.TIER_LITERAL, 3),
    ("[a-z]{2,5}", engine.TIER_CLASSRUN, 2),
    ("[a-z]+", engine.TIER_CLASSRUN, 1),
    (r"\d{3}-\d{4}", engine.TIER_CLASSRUN, 8),
    ("[Ll]inus", engine.TIER_LITERAL, 5),
    ("a.c", engine.TIER_CLASSRUN, 3),
    (r"\w+", engine.TIER_CLASSRUN, 1),
    (r"foo\.bar", engine.TIER_LITERAL, 7),
    (r"\x41\n", engine.TIER_LITERAL, 2),
    ("ab{2}c", engine.TIER_LITERAL, 4),
    ("[^x]{5,}", engine.TIER_CLASSRUN, 5),
    ("[[:alpha:]]{3}", engine.TIER_CLASSRUN, 3),
    (r"\Qa.b\E", engine.TIER_LITERAL, 3),
    ("[0-9a-f]{32}", engine.TIER_CLASSRUN, 32),
    (r"[a-z][0-9][A-Z][_]x", engine.TIER_BUCKET, 5),       # 5 classes > 4 and only a 2-byte literal run -> bucket filter
    (r"[a-z][0-9][A-Z][.,][;:]q", engine.TIER_BUCKET, 6),  # >4 classes, 1-byte literal
    (r"[a-z][0-9][A-Z][.,][;:]", engine.TIER_BUCKET, 5),   # >4 classes, no literal at all
    # several alternatives (priority order) -> bucket filter
    ("foo|bar", engine.TIER_BUCKET, 3),
    ("a|ab", engine.TIER_BUCKET, 1),
    ("(?:foo)", engine.TIER_LITERAL, 3),
    ("colou?r", engine.TIER_BUCKET, 5),
    ("a?b", engine.TIER_BUCKET, 1),
    ("[ab]{1,3}c", engine.TIER_BUCKET, 2),
    ("(?:foo|bar)baz", engine.TIER_BUCKET, 6),
    ("(?:ab|cde){2}", engine.TIER_BUCKET, 4),
    ("(?i)foo", engine.TIER_CLASSRUN, 3),
    ("(?i:foo)|bar", engine.TIER_BUCKET, 3),
    ("(?s)a.c", engine.TIER_CLASSRUN, 3),
    ("a{2,}?", engine.TIER_LITERAL, 2),
    ("a++", engine.TIER_LITERAL, 1),
    ("ab{2,5}?c|x", engine.TIER_BUCKET, 1),
    ("(foo)", engine.TIER_LITERAL, 3),
    ("(foo|bar)", engine.TIER_BUCKET, 3),
    ("(?P<n>a)c|c", engine.TIER_BUCKET, 1),
    ("x(a){0,2}c", engine.TIER_BUCKET, 2),
    # zero-width assertions: one byte of context in front of / behind the window
    (r"\bfoo", engine.TIER_LITERAL, 3),
    (r"foo\b", engine.TIER_LITERAL, 3),
    (r"\Bfoo\B", engine.TIER_LITERAL, 3),
    ("(?m)^foo", engine.TIER_LITERAL, 3),
    ("(?m)foo$", engine.TIER_LITERAL, 3),
    ("^foo", engine.TIER_ANCHORED, 3),
    ("foo$", engine.TIER_ANCHORED, 3),
    (r"\Afoo", engine.TIER_ANCHORED, 3),
    (r"foo\z", engine.TIER_ANCHORED, 3),
    (r"foo\Z", engine.TIER_ANCHORED, 3),
    ("^foo|bar", engine.TIER_LITERAL, 3),     # (start windows: ^foo can only sit at a restart position -- the host's test there finds it; the kernels look for bar)
    (r"\b\w+\b", engine.TIER_CLASSRUN, 1),
    (r"\b[a-z.]o", engine.TIER_BUCKET, 2),   # mixed first class: split into its word / non-word parts
    (r"\bfoo\w*\b", engine.TIER_LITERAL, 3),
    ("(?m)^fo.*$", engine.TIER_LITERAL, 2),
    (r"a\b b", engine.TIER_LITERAL, 3),       # decided on the spot: always true
    # one unbounded repeat in the middle ("gapped"): the kernels look for one repeat byte + the rest.  Two alternatives (with
    # the repeat in front of the window and without) -- over ONE device window where the repeat leads: K1's or K2's then
    # (round 6: such patterns go through the resolve pass and are scanned for by their START windows -- aa | ab, \d\d | \d\.\d)
    ("a+b", engine.TIER_BUCKET, 2),
    (r"\d+\.\d+", engine.TIER_BUCKET, 3),
    ("foo.*bar", engine.TIER_BUCKET, 6),
    (r"[a-z]+\b", engine.TIER_BUCKET, 1),
    ("a*b", engine.TIER_BUCKET, 1),
    ("a{2,}b", engine.TIER_BUCKET, 3),
    ("fo.*$", engine.TIER_ANCHORED, 2),
    (r"(?i)\Qab\E+", engine.TIER_CLASSRUN, 2),
    ("a{3}", engine.TIER_LITERAL, 3),
    ("x{0}abc", engine.TIER_LITERAL, 3),
    ("a{2}[b-c]{2}d?", engine.TIER_CLASSRUN, 4),
    (r"[\d\-x]{4}", engine.TIER_CLASSRUN, 4),
    (r"[]a]{2}", engine.TIER_CLASSRUN, 2),
    (r"\t\e\f\a\cA\0\07\x7", engine.TIER_LITERAL, 8),
    ("{,3}", engine.TIER_LITERAL, 4),  # not a quantifier: literal text
    ("a{1,2}", engine.TIER_LITERAL, 1),
]

# Round 2: conditional groups, subroutine calls / recursion, (*FAIL), callouts, \\o{}, \\Q..\\E in classes, (?U) (?J) (?X).
# Each one against the reference's loop over libpcre on texts made for it (plus random ones): test_calls_and_conditions.
CALLS_AND_CONDITIONS = [
    (r"(a)?(?(1)b|c)", b"ab cb c ac b"), (r"x(a)?(?(1)b|c)", b"xab xc xb xac"), (r"(?<n>a)?(?(<n>)b|c)", b"ab c b"), (r"(?<n>a)?(?('n')b|c)", b"ab c"),
    (r"(?<n>a)?(?(n)b|c)", b"ab c"), (r"(a)?(?(1)b)c", b"abc c bc ac"), (r"(?(?=a)ab|cd)", b"ab cd ad cb"), (r"(?(?!a)b|ab)c", b"bc abc ac"),
    (r"x(?(?<=xa)b|c)", b"xc xab"), (r"(?(R)a|b)", b"a b"), (r"(?(R1)a|b)", b"a b"), (r"(?(DEFINE)(?<d>[0-9]+))x(?&d)", b"x12 x y9 x7"),
    (r"\((?:[^()]|(?R))*\)", b"(a(b)c) ((x)) (() a(b"), (r"\((?:[^()]++|(?R))*\)", b"(a(b)c) ((x)) (() a(b"), (r"(a(?1)?b)", b"ab aabb aab abb"),
    (r"x(?P>n)(?P<n>x1?)", b"xxx xx1x1 xx1x"), (r"(\()?[^()]+(?(1)\))", b"(abc) abc) (abc x"), (r"a(?1)(b|c)", b"abb acc abc ab"),
    (r"(?:a|(b))(?(1)c|d)", b"ad bc ac bd"), (r"\b(?:(\w)(?:(?R)|\w?)\1)\b", b"abba racecar xyzzyx noon ab aa"), (r"(?<A>a|b(?&A)c)x", b"ax bacx bbaccx bax"),
    (r"<(?:[^<>]+|(?R))*>", b"<a<b>c> <<>> <a"), (r"(a)(?(1)b|c)\1", b"aba aca"), (r"(?(1)a|b)(c)", b"bc ac"), (r"((?(R)a|b))(?1)", b"ba bb ab"),
    (r"(x(?(R)a|b))(?1)?c", b"xbc xbxac xac"), (r"(a)\g<1>", b"aa a"), (r"(?<n>a.)\g<n>", b"a1a2 a1b2"), (r"(a|b\g'1'c)d", b"ad bacd bbaccd bd"),
    (r"(ab)(?-1)", b"abab ab"), (r"(?+1)(ab)", b"abab ab"), (r"(?&w) (?<w>[a-c]+)", b"abc cab  ab"), (r"(?2)(a)(b\1?)", b"bab baba ab"),
    (r"()(](?2)){2}", b"]] ]]] ]"), (r"(a(?1)?b){2}", b"abab aabbab"), (r"(?|(?<n>W\2))((?&n)){2}", b"WWW WW WWWW"), (r"(((?&n))((?<n>.\2)))", b"abab aaa"), (r"x(\2(\1>)){2}|>>>", b">>> x>>>"), (r"()(?<p>(.(?&p)(?-1)){2}){2}|q", b"q abcdefghijklmnop q"), (r"(?|x)?ab", None), (r"a(*FAIL)|b", b"a b ab"), (r"a(*F)b|ab", b"ab"),
    (r"ab(?C)c", b"abc ab"), (r"ab(?C12)c", b"abc"), (r"a\o{142}c", b"abc aBc"), (r"[\Qa-z\E]x", b"-x mx ax zx"), (r"[\Qa\E-z]x", b"-x mx ax zx"),
    (r"[\Q]\E]b", b"]b ab"), (r"[^\Qa-\E]x", b"-x mx ax"), (r"(?U)a+b", b"aab ab b"), (r"(?U)a+?b", b"aab"), (r"(?U)[ab]{1,3}c", b"ababc"),
    (r"(?U:a+)b+ ", b"aabb  ab "), (r"a\E{2}b", b"aab ab"), (r"ab\E+c", b"abbc ac"), (r"x(?U)a*(?-U)b*c", b"xaabbc xc"), (r"(?X)ab", b"ab"), (r"(?J)(?<n>a)b", b"ab"),
]


@pytest.mark.parametrize("pattern,text", [c for c in CALLS_AND_CONDITIONS if c[1] is not None])
def test_calls_and_conditions(pattern, text, built, liboracle):
    from test_fuzz import check, make_texts
    assert check(liboracle, pattern, [text, text + b"\n" + text[::-1], b" " + text] + make_texts(3)) is not None


def test_unicode_properties_latin1(built, liboracle):
    """\\p{..} / \\P{..} without UTF: every property the engine has a table for (grab_amd/csrc/ucp_latin1.h, generated from
    Python's unicodedata by scripts/gen_ucp_latin1.py), every byte value, against what libpcre matches."""
    import re
    src = open(os.path.join(ROOT, "grab_amd", "csrc", "ucp_latin1.h")).read()
    names = re.findall(r'\{"([^"]+)", \{', src)
    assert len(names) >= 46 and "Lu" in names and "Latin" in names
    subject = bytes(range(256))
    for name in names:
        for esc in ("\\p{%s}", "\\P{%s}", "\\p{^%s}", "[\\p{%s}]", "(?i)\\p{%s}"):
            pattern = esc % name
            s = np.zeros(300, np.uint32)
            e = np.zeros(300, np.uint32)
            n = liboracle.oracle_all_starts(pattern.encode(), subject, len(subject), s.ctypes.data, e.ctypes.data, 300)
            assert n >= 0, pattern
            want = sorted(s[:n].tolist())
            if not want or len(want) == 0:
                continue  # (an empty set: nothing to compile a window from)
            db = engine.Database(pattern)
            got = [b for b in range(256) if db.class_table(0)[b]]
            assert got == want, (pattern, [hex(b) for b in sorted(set(got) ^ set(want))])
    for short in ("\\pL", "\\PL", "\\pN"):
        assert [b for b in range(256) if engine.Database(short).class_table(0)[b]] == [b for b in range(256) if engine.Database(short[:2] + "{" + short[2] + "}").class_table(0)[b]]


NEWLINE_SEQUENCES = [r"a\Rb", r"\Rb", r"a\R", r"a\R\Rb", r"a\R{2}b", r"a\Xb", r"a\X\Xb", r"a\X{2}b", r"x\R\R\Ry", r"\s\Rb", r"a\R\s", r"a\R\nb", r"(?:a|\R)b", r"a(?=\R)", r"a\R(?!b)",
                    r"a\Rb|a\Xc", r"\X\R", r"\R\X", r"a\X\s", r"[\R]b", r"[\X\B]b"]


@pytest.mark.parametrize("pattern", NEWLINE_SEQUENCES)
def test_newline_sequences(pattern, built, liboracle):
    """\\R = (?>\\r\\n|\\n|\\x0b|\\f|\\r|\\x85), \\X = (?>\\r\\n|any byte) over Latin-1; inside a class both are their letters."""
    from test_fuzz import check, make_texts
    texts = [b"a\r\nb a\nb a\rb a\x0bb a\x0cb a\x85b a\r\n\nb a\n\rb ab Rb Xb Bb", b"x\r\n\r\ny x\n\ny \r\n", b"a\xadb a\r\nb a\n\rb axb a\r\n c a\n\n"]
    assert check(liboracle, pattern, texts + make_texts(4)) is not None


def test_accept_has_no_minimum_length(built, liboracle):
    """(*ACCEPT): pcre_study gives no minimum length, PCRE_INFO_MINLENGTH is -1 and the reference skips every file (Q2)."""
    for pattern in ("a(*ACCEPT)b", "(?:ab(*ACCEPT)|c)d"):
        ml = C.c_int()
        assert liboracle.oracle_minlen(pattern.encode(), C.byref(ml)) == 0 and ml.value == -1
        db = engine.Database(pattern)
        assert db.info.tier == engine.TIER_NULL and db.minlen == -1


NULL_TIER = ["a?", "x*", "", "a{0}", "[a-z]{0,3}", "x{0}", "a|", "(?:foo|b?)", "a??", "(?:ab)?", "x*?y{0}"]

UNSUPPORTED = [r"(a|\1?)b*", r"\p{Greek}", r"a\R?b", r"a+\Rb", r"x\X+", r"(?|(?<n>a)|(?<n>b))",
               "(*UTF8)a", "a(*COMMIT)b", "(*ANYCRLF)a$", r"(?(DEFINE)^)\w", r"(?1)* (\s)?", r"b(?1)c|(A*)x", r"(?<=(?(1)a|b))c(x)", r"x*(?>(?:0)?)x", r"1\n{0,2}(?>a|[b0 ]{0,2}){2}c", r"(?:(?=x))x\B", r"(?:a|(?=x)x)b", r"\S+\h", r"\v*\S{2}", "x|(a)*+b", r"x*(?:ab)?+x", r"b[x.]{0,2}(?:0)?+[x.]{1,3} "]

# "a(" is reported as unsupported (groups) by the engine alone; FileGrep::prepare asks libpcre first and
# gives the reference's "pcre_compile error" for it (tests/test_gpu_filegrep.py, golden case bad_regex)
UNSUPPORTED += [r"(?=a\K)b"]

# assertions that contradict each other: pcre_exec never matches, and neither does the engine (nothing is scanned at all)
NEVER = [r"fo\bo", r"a\Ab", r"x^y|a\zb", r"(?m)a$b"]


@pytest.mark.parametrize("pattern", NEVER)
def test_never_matching(pattern, built, liboracle):
    from test_fuzz import check
    db = engine.Database(pattern)
    assert db.info.tier == engine.TIER_ANCHORED and db.info.n_alts == 0
    assert check(liboracle, pattern, [b"fo o a\nb ab xy x\ny", b"ab", b"a\nb\n"]) == 3


# Patterns whose alternatives stop in front of something the unfolder leaves alone -- a second unbounded repeat, a
# repeated group, a possessive or large bounded repeat or an awkward assertion with more pattern behind it.  The kernels
# look for what every match must begin with; the host's backtracking matcher confirms each offset (gscan_info.exact == 0).
INEXACT = ["a+b+c", "a{1,40}b", "a++b", "(?:ab)+", "(?:a|b)*c", "(?:ab)?+c", "(a)+", r"fo+\bx?", "a.*b.*c", "(?:a+b){2}", r"\b.+x",
           "fo$o", r"(?:a\b){2}", r"\w+@\w+\.com", r"(?:foo|bar)+baz", r"(\w+\s)+x", r"[a-c]+\d+[x.]", r"(?:ab|c)+?d", r"(?i)(?:li|nu)+s",
           r"(?m)^\w+ \w+$", r"a(?:b|c)*+d", r"x.*y.*z", r"(?s)a.+b.+c", r"(?:a|b)+(?:c|d)+", r"((a|b)+c)+d",
           "(?:a|)+b", "(?:a|b|c|d){4}", r"(?:\.?+a)+b", "(?:a*)+b", "a{0,40}b", "(?:|a){2}b", r"a(?:$cb??|a?+[^a]{0,2}\b){2}",
           "(?:ab|cd|li|nu|fo|ob|ar|ba){3}", "|".join("w%03d" % i for i in range(65)) + "|foo|linus",
           # look-around and atomic groups: nothing of them reaches the kernels, the matcher evaluates them
           "foo(?=bar)", "foo(?!bar)", "(?<=x)y", "(?<!a)b", "(?<=ab|c)d", "(?>a+)b", "(?>ab|a)c", r"\b(?=\w{3}\b)[a-z]+", "(?=(a))ab|b",
           r"(?<![a-z])li(?=nus)", "a(?=b)?b", "x(?!y){2}.", "(?<=\n)[a-z]+(?= )", "(?s)a(?=.*z)b",
           # back references: the matcher remembers what the groups captured; with the reference's ovector[3] a match that used
           # one has set a group and ends the chunk (Q5) -- what prints are the matches of the alternatives without groups
           r"(a|b)\1|li", r"(\w)\1+x|foo", r"(?P<q>ab)(?P=q)|nus", r"(?i)(ab)\1|c", r"(a)(b)\2\1|x", r"(?:(a)|b)\1?c", r"(a|b\1)+c|z",
           r"(\w+) \1\b|ab", r"(ab)\g{-1}|(?<n>l)\k<n>|f", r"((\2a|b){2}c){2}|li",
           # \K: the reported start moves (ovector[0]); where the match is FOUND does not
           r"foo\Kbar|li", r"\w+\K\d", r"a\Kb|b\Kc|c", r"(?:a\K)+b|nus",
           # longer than the kernels' windows go: they look for the first 253 bytes
           "[a-z]{300}|linus", "(?:ab){140}c|foo"]

MALFORMED = ["a{2}{3}", "a**", "(?x)a + ? *b", r"\b*a", r"\1", r"(a)\2", "(?P=n)", r"(?<n>a)(?<n>b)", r"(a)(?<=\1)b", "(?<=a+)b", "(?<!ab|c*)d", "[abc", "*a", "+", "?x", "a{3,2}", "[z-a]", "\\", "a)", "[[:nope:]]", "(?:a", "(?i", "(?:a|*b)", "a|+", "(?z)a", "(?i)+a"]


def _fix5(p):
    return p


@pytest.mark.parametrize("pattern,tier,minlen", SUPPORTED)
def test_supported(pattern, tier, minlen, built, liboracle):
    db = engine.Database(pattern)
    assert db.info.tier == tier, pattern
    assert db.minlen == minlen
    ml = C.c_int()
    assert liboracle.oracle_minlen(pattern.encode(), C.byref(ml)) == 0
    assert ml.value == minlen, "engine minlen must equal PCRE_INFO_MINLENGTH"


@pytest.mark.parametrize("pattern", NULL_TIER)
def test_empty_matchable(pattern, built, liboracle):
    db = engine.Database(pattern)
    assert db.info.tier == engine.TIER_NULL and db.minlen == -1
    ml = C.c_int()
    assert liboracle.oracle_minlen(pattern.encode(), C.byref(ml)) == 0 and ml.value == -1


@pytest.mark.parametrize("pattern", UNSUPPORTED)
def test_unsupported(pattern, built):
    with pytest.raises(engine.Unsupported):
        engine.Database(pattern)


@pytest.mark.parametrize("pattern", MALFORMED)
def test_malformed(pattern, built, liboracle):
    with pytest.raises(ValueError) as ei:
        engine.Database(pattern)
    assert not isinstance(ei.value, engine.Unsupported)
    ml = C.c_int()
    assert liboracle.oracle_minlen(pattern.encode(), C.byref(ml)) != 0, "PCRE rejects it too"


def test_literal_flag(built):
    db = engine.Database("a.c|(x", literal=True)
    assert db.info.tier == engine.TIER_LITERAL and db.minlen == 6 and db.info.is_literal
    assert db.class_table(1)[ord(".")] and db.class_table(1).sum() == 1


ATOMS = [("a", "a"), (".", "."), (r"\d", r"\d"), (r"\D", r"\D"), (r"\w", r"\w"), (r"\W", r"\W"), (r"\s", r"\s"),
         (r"\S", r"\S"), ("[a-fA-F0-9]", "[a-fA-F0-9]"), ("[^a-z\\n]", "[^a-z\\n]"), (r"[\w.-]", r"[\w.-]"),
         (r"[\x00-\x1f]", r"[\x00-\x1f]"), (r"[]x]", r"[]x]"), (r"[a\-z]", r"[a\-z]"), (r"\x80", r"\x80"),
         (r"[\x80-\xff]", r"[\x80-\xff]"), (r"\.", r"\."), (r"[a-]", r"[a-]"), (r"[\d-x]", r"[\d\-x]"),
         (r"\h", "[\\t \\xa0]"), (r"\v", "[\\n\\x0b\\f\\r\\x85]"), (r"\N", "[^\\n]"), ("[[:digit:][:upper:]]", "[0-9A-Z]"),
         ("[[:^space:]]", r"\S"), ("[[:punct:]]", "[!-/:-@\\[-`{-~]"), ("[[:xdigit:]]", "[0-9a-fA-F]")]


@pytest.mark.parametrize("atom,pyatom", ATOMS)
def test_class_tables(atom, pyatom, built, liboracle):
    """byte membership == Python re == libpcre, for all 256 byte values."""
    db = engine.Database(atom + "q")  # 'q' keeps a literal in the pattern; position 0 is the atom
    tbl = db.class_table(0)
    rx = re.compile(pyatom.encode("latin-1"))
    want = np.array([rx.fullmatch(bytes([b])) is not None for b in range(256)])
    assert np.array_equal(tbl, want)
    buf = bytes(range(256))
    starts = np.zeros(256, np.uint32)
    n = liboracle.oracle_all_starts(atom.encode("latin-1"), buf, 256, starts.ctypes.data, None, 256)
    got = np.zeros(256, bool)
    got[starts[:n]] = True
    assert np.array_equal(tbl, got)


TAILS = ["abc[0-9]*", "[a-z]{2,5}", "[A-Za-z_][A-Za-z0-9_]{15,}", "e+", "x[^y]{0,7}", "foo", r"\d{3}-\d{4}", "ab?"]


@pytest.mark.parametrize("pattern", TAILS)
def test_match_end(pattern, built, liboracle):
    rng = np.random.default_rng(7)
    data = np.frombuffer(b"abcexy0123456789_Z \n", np.uint8)[rng.integers(0, 20, 5000)]
    data[200:260] = ord("e")
    data[4990:] = ord("e")
    data[300:312] = np.frombuffer(b"abc01234567x", np.uint8)
    data[400:404] = np.frombuffer(b"abcx", np.uint8)
    data[500:512] = np.frombuffer(b"x555-1234567", np.uint8)
    data[600:606] = np.frombuffer(b"foofoo", np.uint8)
    buf = data.tobytes()
    starts = np.zeros(len(buf), np.uint32)
    ends = np.zeros(len(buf), np.uint32)
    n = liboracle.oracle_all_starts(pattern.encode(), buf, len(buf), starts.ctypes.data, ends.ctypes.data, len(buf))
    assert n > 0
    db = engine.Database(pattern)
    for p, e in zip(starts[:n].tolist(), ends[:n].tolist()):
        assert db.match_end(buf, p) == e


ALT_PATTERNS = ["foo|bar", "colou?r", "(?i)linus", "[ab]{1,3}c", "(?:foo|bar)baz", "a{2,5}b?", "(?:a|ab){1,2}", "x(?:a|b|)y", "(?:ab)?c",
                "fo{1,}|bar+", "(?i:foo)|bar", "a(?i)b|c", "a{2,5}?b", "(?:ab|cde){2}", "(?s)a.c", "(?i)[^a]b", "(?i)[a-c]x", "ab+?", "ab++",
                "a??b", "(?m)foo", "a|ab", "ab|a", "(?:a|b)(?:c|d)e*", "(?:ab){1,3}?c", "a(?:b|c){0,2}a", "foo|fo|f", "f|fo|foo",
                "(?:a?b){2}", "(?i)a[b-d]+|X{2}", "(?-i)ab|(?i)cd", r"(?i)\x41b", "(?#c)ab", "a(?#x)b|c", "(?:foo|bar){0}ab", "ab{0}|c",
                "[[:upper:]]b|(?i)[[:upper:]]c", r"\Qa|b\E|c", r"(?i)\Qab\E+", "a{0,3}b{0,2}c", "(?:a|b|c|d|e|f|x|y|0|1)x",
                "(?:ab|ba){2,3}x", "(?i)(?:li|LI)nus|(?-i:Foo)", "a(?:b(?:c|d)|e)f", "(?:a|b){2}(?:c|d){2}",
                "(a|b)c|ad", "(a)?c", "x(a){0,2}c", "(?<n>a)c|c", "((a))c|c", "ba(?:r|(z))|foo", "(?'q'ab){1,2}x|ab",
                "(?i)[[:^upper:]]a", "(?i)[[:^lower:]x]b|[^[:upper:]]c", "(?i)[[:upper:]][[:^xdigit:]]"]


@pytest.mark.parametrize("pattern", ALT_PATTERNS)
def test_alternatives_match_pcre(pattern, built, liboracle):
    """Unfolded alternatives in priority order: for EVERY offset p of a random text, "a match starts at p" and
    its end (ovector[1]) are exactly what pcre_exec(ANCHORED) gives with the subject starting at p; minlen ==
    PCRE_INFO_MINLENGTH.  (pcre_exec is libpcre's, called by the oracle exactly as the reference calls it.)"""
    rng = np.random.default_rng(11)
    alpha = np.frombuffer(b"abcdefoxyABFOLINUSlinusrz01 \n.", np.uint8)
    data = alpha[rng.integers(0, alpha.size, 12000)]
    for w in [b"foobar", b"colour", b"color", b"LiNuS", b"barbaz", b"ababc", b"aaaaab", b"abcde", b"xay", b"xy", b"abab", b"abababc",
              b"cdCD", b"aaabbc", b"abbaabx", b"abdf", b"aef", b"abcd"]:
        for _ in range(12):
            o = int(rng.integers(0, data.size - 10))
            data[o:o + len(w)] = np.frombuffer(w, np.uint8)
    buf = data.tobytes()
    ml = C.c_int()
    assert liboracle.oracle_minlen(pattern.encode("latin-1"), C.byref(ml)) == 0
    db = engine.Database(pattern)
    assert db.minlen == ml.value
    s = np.zeros(len(buf) + 1, np.uint32)
    e = np.zeros(len(buf) + 1, np.uint32)
    n = liboracle.oracle_all_starts(pattern.encode("latin-1"), buf, len(buf), s.ctypes.data, e.ctypes.data, len(buf) + 1)
    want = dict(zip(s[:n].tolist(), e[:n].tolist()))  # pcre_exec with the reference's ovector[3]: rc > 0 only
    info = {p: db.match_info(buf, p) for p in range(len(buf)) if db.match_at(buf, p)}
    got = {p: end for p, (kind, end) in info.items() if kind == 1}  # kind 2 = the match sets a capturing group: rc == 0 there
    assert got == want
    assert all(db.match_end(buf, p) == end for p, (kind, end) in info.items())
    # the device-side view of the same thing: the union of the alternatives' windows is the candidate set
    from inputs import db_candidates, engine_list
    assert db_candidates(db, data).tolist() == sorted(info)


@pytest.mark.parametrize("pattern", INEXACT)
def test_inexact_patterns_match_pcre(pattern, built, liboracle):
    """For EVERY offset p of a random text the host matcher's verdict and match end equal pcre_exec(ANCHORED)'s, and the
    whole chunk walk over what the kernels are specified to report prints what the reference's loop prints."""
    from test_fuzz import check
    rng = np.random.default_rng(23)
    alpha = np.frombuffer(b"abcdfoxyzABLINUSlinusrz019@. \n.", np.uint8)
    data = alpha[rng.integers(0, alpha.size, 6000)]
    for w in [b"foobarbaz", b"aaabbbc", b"ababab", b"ab", b"linus", b"NuLis", b"li nus\nab cd\n", b"x1y2z", b"a12.5", b"abcd", b"x y\nz", b"aab",
              b"ab@cd.com", b"abc12.", b"ccabd", b"fooo x", b"abababcd", b"a\nb c", b"a bb  c", b"acbd"]:
        for _ in range(8):
            o = int(rng.integers(0, data.size - 16))
            data[o:o + len(w)] = np.frombuffer(w, np.uint8)
    buf = data.tobytes()
    ml = C.c_int()
    assert liboracle.oracle_minlen(pattern.encode("latin-1"), C.byref(ml)) == 0
    db = engine.Database(pattern)
    assert db.minlen == ml.value and not db.info.exact
    s = np.zeros(len(buf) + 1, np.uint32)
    e = np.zeros(len(buf) + 1, np.uint32)
    n = liboracle.oracle_all_starts(pattern.encode("latin-1"), buf, len(buf), s.ctypes.data, e.ctypes.data, len(buf) + 1)
    want = dict(zip(s[:n].tolist(), e[:n].tolist()))
    info = {p: db.match_info(buf, p) for p in range(len(buf))}
    got = {p: end for p, (kind, end) in info.items() if kind == 1}
    assert got == want
    texts = [buf[o:o + 400] for o in range(0, len(buf), 400)] + [buf]
    assert check(liboracle, pattern, texts) == len(texts)


EXTENDED = [("(?x) f o o # the needle\n b a r", "foobar"), ("(?x)a +b", "a+b"), (r"(?x)a\ b", "a b"), ("(?x)[ #]a", "[ #]a"), ("(?x:a b)c d", "abc d"),
            ("a(?x) b (?-x) c", "ab c"), ("(?x) (?: fo | ba ) {2} r", "(?:fo|ba){2}r"), ("(?xi) li nus", "(?i)linus"),
            ("(?x)fo + ?o", "fo+?o"), ("(?x)fo {1,2} # twice at most\n +b", "fo{1,2}+b")]


@pytest.mark.parametrize("spaced,plain", EXTENDED)
def test_extended_mode(spaced, plain, built, liboracle):
    """(?x): white space and #-comments between items mean nothing -- the compiled form equals the plain spelling's, and
    libpcre agrees on minlen."""
    a, b = engine.Database(spaced), engine.Database(plain)
    ml = C.c_int()
    assert liboracle.oracle_minlen(spaced.encode(), C.byref(ml)) == 0 and ml.value == a.minlen == b.minlen
    assert a.info.n_alts == b.info.n_alts and a.info.tier == b.info.tier
    for i in range(a.info.n_alts):
        assert a.alt_len(i) == b.alt_len(i)
        assert all((a.class_table(k, i) == b.class_table(k, i)).all() for k in range(a.alt_len(i)))


def test_alternative_order_and_limits(built):
    db = engine.Database("colou?r")  # greedy ?: the longer path is tried first
    assert db.info.n_alts == 2 and [db.alt_len(0), db.alt_len(1)] == [6, 5]
    db = engine.Database("colou??r")  # lazy: the shorter one first
    assert [db.alt_len(0), db.alt_len(1)] == [5, 6]
    db = engine.Database("(?:a|ab){1,2}")  # depth-first: a.a, a.ab, a, ab.a, ab.ab, ab
    assert [db.alt_len(i) for i in range(db.info.n_alts)] == [2, 3, 1, 3, 4, 2]
    db = engine.Database("foo|foo|bar|foo")  # a later duplicate can never be the first to match
    assert db.info.n_alts == 2
    # > 64 alternatives: the kernels look for short prefixes of them and the host matcher confirms
    db = engine.Database("|".join("w%03d" % i for i in range(65)))
    assert db.info.n_alts <= 64 and not db.info.exact
    assert engine.Database("|".join("w%03d" % i for i in range(64))).info.n_alts == 64


CAPTURE = ["(a|b)c|ad", "(a)?c", "x(a){0,2}c", "(?<n>a)c|c", "((a))c|c", "ba(?:r|(z))|foo", "(foo)", "(?:(a)|b)c", "(a)b|ac", "()ab|b", "(?P<w>ab)?c"]


@pytest.mark.parametrize("pattern", CAPTURE)
def test_capture_groups_end_the_chunk(pattern, built, liboracle):
    """The reference calls pcre_exec with int ovector[3] (grab.cc:171): a match that sets a capturing group comes back
    as 0 and the chunk loop ends.  match_info's kind 2 must mark exactly those matches: compared, start by start, with
    pcre_exec(ovecsize 3) run by the oracle the way the reference runs it (oracle_scan_chunk, -O -l -s from each offset)."""
    rng = np.random.default_rng(5)
    alpha = np.frombuffer(b"abcdxzfor \n", np.uint8)
    data = alpha[rng.integers(0, alpha.size, 3000)]
    for w in [b"foo", b"bar", b"baz", b"ad", b"ac", b"bc", b"xaac", b"xc", b"abc", b"ababx"]:
        for _ in range(15):
            o = int(rng.integers(0, data.size - 8))
            data[o:o + len(w)] = np.frombuffer(w, np.uint8)
    buf = data.tobytes()
    db = engine.Database(pattern)
    n_cap = 0
    for p in range(len(buf)):
        kind, end = db.match_info(buf, p)
        if kind == 0:
            continue
        # what the reference prints for the chunk buf[p:] in -O -l -s mode: one offset line if pcre_exec returned > 0 at the
        # leftmost match -- which is p itself here -- and nothing if it returned 0
        out = C.c_void_p()
        outlen = C.c_size_t()
        assert liboracle.oracle_scan_chunk(pattern.encode(), b"", buf[p:], len(buf) - p, p, 1 | 2 | 4, C.byref(out), C.byref(outlen)) == 0
        text = C.string_at(out, outlen.value) if outlen.value else b""
        liboracle.oracle_free(out)
        if p + db.minlen >= len(buf):
            continue  # the strict loop bound (Q3) keeps the reference from looking here at all
        if kind == 2:
            assert text == b"", (pattern, p)
            n_cap += 1
        else:
            assert text == b"Match at offset %d\n" % p, (pattern, p, text)
    assert n_cap > 0


GAP_PATTERNS = ["a+b", "fo.*b", "foo.*bar", r"[a-z]+\b", r"x[a-z]+\b", r"\bfo+\b", "a*b", "a+?b", "a*?b", "ba*b", "a+(?:ab|bc)", "a+b|a+c",
                "a+(?:b|c)", "o+ |x", r"(?m)^\s*foo", "(?i)fo+bar", "a{2,}b", ".+x", "fo.*$", r"\w+ \w", "(a+)b", "a+(b)|a+c", r"\bo+\b",
                r"[0-9]+\.[0-9]+|x\d*y", r"(?m)^[a-z]+ = \d\d?;?$", "f.*?o", r"\w+@[a-z]\.(?:com|org)", r"(?m)^.*foo", "[ab]+b{2}", r"\s+\S", r"1+\b\.|o+$",
                r"\d+\b", r"\Bo+x?\b| +\bf"]

ASSERT_PATTERNS = [r"\bfoo", r"\Bfoo", r"foo\b", r"foo\B", "^foo", "(?m)^foo", "foo$", "(?m)foo$", r"\Afoo", r"foo\z", r"foo\Z", r"\b[a-z.]o",
                   r"o[a-z.]\b", r"a\b b", "^foo|bar", r"[x\n](?m)^y", r"x(?m)$\ny", r"\B[a-z]{2}", r"foo\b|fo", r"\bfo{1,3}\b", "(?m)^[a-z]+",
                   r"\b\w+", r"(?m)^\s?foo", r"foo\b.*", r"\b(?:foo|ba)\b", "(?m)^foo$", r"(?i)\bFOO\b|x$", r"\Bo\B", r"\b.\b", "(?m)^.|o$",
                   r"(a\b)|b", "^(foo)|o", r"\Gfo", r"(?m)$\nf", r"\w+\b", r"\bfoo\w*\b", r"\b\w+\b", "(?m)^fo.*$", r"o\w+\b|x",
                   r"a{1,3}\b", r"\ba{1,3}?\b", r"\bfoo\b[a-z ]{0,3}", "(?m)^foo$.?"]


@pytest.mark.parametrize("pattern", ASSERT_PATTERNS + GAP_PATTERNS)
def test_assertions_match_reference_loop(pattern, built, liboracle):
    """^ $ \\b \\B \\A \\z \\Z (?m): the product's chunk walk (grab_report_chunk) fed with exactly what the kernels are
    specified to report (device windows, tests/inputs.py:db_candidates; nothing for the ANCHORED tier) prints what the
    reference's loop prints -- the oracle runs pcre_exec the way grab.cc:175-213 does, subject restarted at every match
    (quirk Q4) -- on fixed and random texts, in all output modes."""
    from grab_amd import filegrep
    from inputs import db_candidates, engine_list
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import scan_oracle as so

    rng = np.random.default_rng(3)
    alpha = np.frombuffer(b"fooab xy.\n\n  11=;@cm", np.uint8)
    texts = [b"ab aab b aaab abc aabc", b"3.14 1.5.25 x12y .5 5. 10.02", b"foo bar\nfoo x bar bar\nfoobar", b"a@b.com x@y.org z@w.net\n", b"ab = 12;\nb = 1\n c = 3;",
             b"foofoo xfoo foo", b"foofoo\nfoo", b"foo\nfoo\n", b"foo\nfoox\nfoo", b"xo .o ao.o", b"oa o. oab o.a", b"a b ab", b"foobar foo bar",
             b"x\ny xy\n\ny", b"xabcd", b"foox foo fo", b"  foo\nfoo bar\n\tfoo", b"foo", b"xfoo", b"foo\n", b"\n", b"o", b"fo"]
    texts += [alpha[rng.integers(0, alpha.size, int(rng.integers(1, 200)))].tobytes() for _ in range(100)]
    ml = C.c_int()
    assert liboracle.oracle_minlen(pattern.encode("latin-1"), C.byref(ml)) == 0
    db = engine.Database(pattern)
    assert db.minlen == ml.value
    for text in texts:
        data = np.frombuffer(text, np.uint8)
        if db.info.tier == engine.TIER_ANCHORED or db.minlen < 0:
            starts = np.zeros(0, np.uint32)
        else:
            starts = engine_list(db, data)
        for f in (1 | 2, 1, 0, 1 | 2 | 4, 2):
            want = b""
            if 0 <= ml.value <= len(text):
                out = C.c_void_p()
                n = C.c_size_t()
                assert liboracle.oracle_scan_chunk(pattern.encode("latin-1"), b"", text, len(text), 0, f, C.byref(out), C.byref(n)) == 0
                want = C.string_at(out, n.value) if n.value else b""
                liboracle.oracle_free(out)
            got = filegrep.report_chunk(db, f, b"", data, 0, starts) if 0 <= db.minlen <= len(text) else b""
            assert got == want, (pattern, text, f)


def test_deep_group_repeats_and_small_stacks(built, liboracle):
    """Thousands of nested group iterations: the matcher leaves the caller's stack alone (192 KiB budget, then a stack of
    its own), so a thread with a 1 MiB stack survives what would otherwise be megabytes of recursion; up to libpcre-JIT's own
    depth (4095 iterations of (?:ab)+ on its 32 KiB stack) the two agree, beyond it only the matcher still answers."""
    import threading

    results = {}

    def work():
        db = engine.Database(r"(?:ab)+x")
        for n in (10, 369, 371, 4000, 11000, 13000):
            text = np.frombuffer(b"ab" * n + b"x", np.uint8)
            results[n] = db.match_info(text, 0)
        db2 = engine.Database(r"(?:a|b)+c")
        results["long"] = db2.match_info(np.frombuffer(b"a" * 1000000 + b"bc", np.uint8), 0)

    threading.stack_size(1 << 20)
    try:
        t = threading.Thread(target=work)
        t.start()
        t.join()
    finally:
        threading.stack_size(0)
    for n in (10, 369, 371, 4000, 11000):
        assert results[n] == (1, 2 * n + 1), n
    assert results[13000][0] == 0 and results["long"][0] == 0  # beyond 12000 nested iterations: given up, like a pcre_exec error
    s = np.zeros(4, np.uint32)
    e = np.zeros(4, np.uint32)
    text = b"ab" * 4000 + b"x"
    assert liboracle.oracle_all_starts(b"(?:ab)+x", text, len(text), s.ctypes.data, e.ctypes.data, 4) >= 1 and (s[0], e[0]) == (0, 8001)



OCTAL = [  # (pattern, accepted by pcre_compile)
    (r"\101", True), (r"\377", True), (r"\400", False), (r"\777", False), (r"\12", True), (r"\1011", True), (r"x\1231", True), (r"[\101]x", True), (r"\18", True),
    (r"\7", False), (r"\8", True), (r"\9", True), (r"\81", True), (r"\80", True), (r"[\8]", True), (r"\08", True), (r"(a)\8", True), (r"(a)\11", True),
    (r"\N{3}", True), (r"\N{2,}x", True), (r"\N{U+41}", False), (r"\N{a}", False), (r"\N{", False), (r"\N{,3}", False),  # \N{..}: a quantifier or nothing (pcre_compile's error 37)
    (r"(a)(b)(c)(d)(e)(f)(g)(h)\8", True), (r"(a)(b)(c)(d)(e)(f)(g)(h)(i)(j)\10", True), (r"(a)(b)(c)(d)(e)(f)(g)(h)(i)(j)\11", True), (r"\11(a)(b)(c)(d)(e)(f)(g)(h)(i)(j)(k)", True),
]


@pytest.mark.parametrize("pattern,ok", OCTAL)
def test_octal_escapes_and_digits_that_are_no_back_references(pattern, ok, built, liboracle):
    r"""pcre_compile's rule for \N outside a class: a back reference if N < 8 or the pattern has N groups (anywhere), else \8 \9
    are the digits themselves and anything else is an octal escape of up to three digits (> \377: an error in 8-bit mode).
    Until round 6 every such escape was refused ("back reference / octal escape").  minlen and the chunk's output against
    libpcre under the reference's loop."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import inputs
    from grab_amd import filegrep

    pb = pattern.encode()
    ml = C.c_int(-9)
    rc = liboracle.oracle_minlen(pb, C.byref(ml))
    assert (rc == 0) == ok
    if not ok:
        with pytest.raises(ValueError):
            engine.Database(pattern)
        return
    db = engine.Database(pattern)
    assert db.minlen == ml.value
    text = np.frombuffer(b"xxAxx A1 \n yy\t81 9 S \xff z aa\t a\n1 AB 8 80 a8 xS1 \x018 abcdefghijb abcdefghijH abcdefghij\t abcdefgh8 \tabcdefghijk\n".ljust(320, b"."), np.uint8).copy()
    starts, ends = inputs.resolved_list(db, text) if db.info.resolve else (inputs.engine_list(db, text), None)
    for flags in (3, 1, 0):
        out, n = C.c_void_p(), C.c_size_t()
        assert liboracle.oracle_scan_chunk(pb, b"", text.ctypes.data, text.size, 0, flags, C.byref(out), C.byref(n)) == 0
        want = C.string_at(out, n.value)
        liboracle.oracle_free(out)
        assert filegrep.report_chunk(db, flags, b"", text, 0, starts, ends=ends) == want, (pattern, flags)


SYNTAX_CORNERS = [  # accepted by pcre_compile (8.34 and later) or not -- the product's compiler agrees, and where both accept, so do minlen and output
    (r"\x{41}", True), (r"\x{041}", True), (r"\xg{", True), (r"\x{", False), (r"\x{}", False), (r"\x{41", False), (r"\x{4g}", False), (r"[\x{}]", False),
    (r"[a-\d]", False), (r"[a-\w]", False), (r"[\x41-\d]", False), (r"[\d-z]", True), (r"[\w-a]", True), (r"[a-b-\d]", True), (r"[a-[:digit:]]", False), (r"[[:digit:]-z]", True),
    (r"[A-[:x]", True), (r"[+--]", True), (r"[a-z-9]", True), (r"[--a]", True),
    (r"[:alpha:]", False), (r"x[:alpha:]", False), (r"[=a=]", False), (r"[.a.]", False), (r"[::]", False), (r"[:a-z:]", False), (r"[:a\]:]", False), (r"[:]", True), (r"[a:alpha:]", True),
    (r"[^:alpha:]", True), (r"[:ab[:digit:]]", True),
    (r"a\Q\E*b", True), (r"a\Q\E\Q\E{2}", True), (r"a\E\Q\E?b", True), (r"\Q\E*a", False), (r"\Qab\E+c", True),
]


@pytest.mark.parametrize("pattern,ok", SYNTAX_CORNERS)
def test_syntax_corners_follow_pcre_compile(pattern, ok, built, liboracle):
    r"""Corners of pcre_compile's syntax the zoo (scripts/pattern_zoo.py) found handled the way libpcre < 8.34 did, or not at all:
    \x{ must be well formed (error 79), a class escape or POSIX class cannot end a range (error 83), "[:alpha:]" outside a class is
    an error (12), an empty \Q\E in front of a quantifier means nothing."""
    test_octal_escapes_and_digits_that_are_no_back_references(pattern, ok, built, liboracle)


LEFT_RECURSION = [  # pcre_compile's error 40 ("recursive call could loop indefinitely") and its corners, and quantified verbs; True: libpcre accepts
    (r"(?R)?x", False), (r"(?R)", False), (r"(?0)", False), (r"a(?R)", True), (r"(?R)a", False), (r"(a|(?R))", False), (r"a(?R)?b", True),
    (r"(?R)*a", False), (r"(?:(?R))", False), (r"((?1))", False), (r"(a(?1)?)", True), (r"((?1)a)", False), (r"(?1)(a)", True),
    (r"((?2))((?1))", True), (r"(?:a|(?R))b", False), (r"^(?R)", False), (r"(?=(?R))a", False), (r"\b(?R)", False), (r"(a?(?1))", False),
    (r"(a*(?1)b)", False), (r"(?<n>(?&n))", False), (r"(?<n>x(?&n)?)", True), (r"((a)|(?1))", False), (r"(a|b(?1))", True),
    (r"(a|(?2))(b(?1)?)", True), (r"(\1(?1))", False), (r"(x)(\1?(?2))", False), (r"((?(1)a|b)(?1))", True), (r"(?|(?1))", False),
    (r"(?>(?R))", False), (r"(?:a{0}(?R))", False), (r"((?=a)(?1))", False), (r"((?!a)(?1))", False), (r"(?i)((?1))", False), (r"(a(?R)b|c)", True),
    (r"(\((?:[^()]|(?1))*\))", True), (r"\((?:[^()]++|(?R))*\)", True), (r"^(?:a(?R)?b)$", True), (r"(?<p>\((?:[^()]|(?&p))*\))", True),
    (r"(x)(\1(?2))", False), (r"(x?)(\1(?2))", False), (r"(\1a(?1))", True), (r"()(\1(?2))", False), (r"(x)(\1+(?2))", False),
    (r"(x)(\1{2}(?2))", False), (r"(x)((?1)(?2))", True), (r"(x?)((?1)(?2))", False), (r"(x)((?(1)a)(?2))", False), (r"(x)((?(1)a|b)(?2))", True),
    (r"(x)((?(1)|b)(?2))", False), (r"(x)((?(?=a)a)(?2))", False), (r"(x)((?(?=a)a|b)(?2))", True), (r"(x)(y)((?1)?(?2)*(?3))", False),
    (r"(x)(y)((?1)?(?2)(?3))", True), (r"(a|b*)((?1)(?2))", False), (r"((?2)(?1))(a?)", False), (r"((?2)(?1))(a)", False), (r"(a((?1)(?2)))", False),
    (r"(a((?1)?(?2)))", False), (r"(a(b(?1)(?2)))", True), (r"(b(?2)(?1))(a)", True), (r"((?2)b(?1))(a)", False), (r"(?:(?1)(?R))(a)", False),
    (r"(?:(?1)x(?R))(a)", False), (r"((?3)(?1))(x)(a)", False), (r"(?<a>(?&b)(?&a))(?<b>x)", False), (r"((?:(?2))b(?1))(a)", True),
    (r"(?:(?1)|b)x(?R)(a)", True), (r"(x(?2))(a(?1)?)", True), (r"(?:x|(?1)y(?R))(a)", False), (r"((?2)|b(?1))(a)", True),
    (r"(a(?2)b)(c(?1)?d)", True), (r"((?1)?a)", False), (r"(a|(?1)b)", False), (r"(a|(?1)?b)", False), (r"((?:a|(?1))b)", False),
    (r"(?:(a)|b)(?1)", True), (r"(?1)(?:(a)|b)", True), (r"(?:(?1))(a)", True), (r"(?!a|(?R))b", True), (r"(?!(?R))a", False),
    (r"(?(R)a|b)(?R)", True), (r"(?(R)a|)(?R)", False), (r"(?(1)a)(x)(?R)", True), (r"\w{2}(?(1)((?1))+x)", True), (r"(?|(x(?1)?)|((?1)y))", True),
    (r"(*F){1,2}a", False), (r"a(*FAIL)+", False), (r"(*ACCEPT){2}", False),
]


@pytest.mark.parametrize("pattern,ok", LEFT_RECURSION)
def test_left_recursion_is_diagnosed_the_way_pcre_compile_does(pattern, ok, built, liboracle):
    r"""A call of a group from inside it that can be reached without consuming a byte: could_be_empty_branch's view of "may match
    nothing" (assertions, back references, calls of groups that may, conditions without an else), an incomplete call ends the scan of
    its branch, nothing inside a conditional group is checked, nor the later alternatives of an assertion; with (?| the first group
    of a number is the one a call goes to.  With GSCAN_PCRE_CHECKED (what FileGrep::prepare passes once pcre_compile has accepted the
    text) the diagnosis is not made at all."""
    ml = C.c_int(-9)
    assert (liboracle.oracle_minlen(pattern.encode(), C.byref(ml)) == 0) == ok
    try:
        db = engine.Database(pattern)
        assert ok and db.minlen == ml.value, (pattern, db.minlen, ml.value)
    except engine.Unsupported:
        pass
    except ValueError:
        assert not ok, pattern
    if ok:
        try:
            assert engine.Database(pattern, pcre_checked=True).minlen == ml.value
        except engine.Unsupported:
            pass
