#!/usr/bin/env python3
"""Generate tests/golden/golden.json by running the REAL reference (oracle/_ref/grab_jit, built
from /root/reference/src by `make -C oracle ref`) on inputs described by small recipes.

Run in the build container only (the GPU box has no /root/reference and only consumes the
committed JSON).  Every case stores the recipe of its input, the command-line flags and either
the full stdout (small) or its md5 + line count + head (large).  The non-JIT build
(oracle/_ref/grab, PCRE 8.45) is run on every case too and must agree byte for byte.

    python tests/golden/make_golden.py
"""
import base64
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from inputs import materialize  # noqa: E402

REF_JIT = os.path.join(ROOT, "oracle", "_ref", "grab_jit")
REF_NOJIT = os.path.join(ROOT, "oracle", "_ref", "grab")
NEEDLE = "foobardoesnotexist"
IDENT = "[A-Za-z_][A-Za-z0-9_]{15,}"
T1 = "hello foo world\nno match here\nfoo at start and foo again\ntail foo"
L5 = ["-L"] * 5  # 32 MiB chunks


def lit(s):
    return {"kind": "bytes", "b64": base64.b64encode(s.encode("latin-1") if isinstance(s, str) else s).decode()}


CASES = [
    # ---- output formats ----
    dict(name="t1_default", inputs={"t1.txt": lit(T1)}, args=["foo", "t1.txt"]),
    dict(name="t1_O", inputs={"t1.txt": lit(T1)}, args=["-O", "foo", "t1.txt"]),
    dict(name="t1_Ol", inputs={"t1.txt": lit(T1)}, args=["-O", "-l", "foo", "t1.txt"]),
    dict(name="t1_l", inputs={"t1.txt": lit(T1)}, args=["-l", "foo", "t1.txt"]),
    dict(name="t1_s", inputs={"t1.txt": lit(T1)}, args=["-s", "foo", "t1.txt"]),
    dict(name="t1_sOl", inputs={"t1.txt": lit(T1)}, args=["-s", "-O", "-l", "foo", "t1.txt"]),
    dict(name="t1_two_paths", inputs={"t1.txt": lit(T1), "t2.txt": lit("a foo b\n")}, args=["-O", "foo", "t1.txt", "t2.txt"]),
    dict(name="tree_rO", inputs={"d/t1.txt": lit(T1), "d/sub/t2.txt": lit("x foo y\nfoo\n"), "d/sub/none.txt": lit("nothing\n")},
         args=["-r", "-O", "foo", "d"], sort=True),
    dict(name="tree_n2", inputs={"d/t1.txt": lit(T1), "d/sub/t2.txt": lit("x foo y\nfoo\n"), "d/a/b/c.txt": lit("foofoo foo\n")},
         args=["-n", "2", "-r", "-O", "-l", "foo", "d"], sort=True),
    dict(name="nomatch", inputs={"t1.txt": lit(T1)}, args=["zzz", "t1.txt"]),
    dict(name="dir_without_r", inputs={"d/t1.txt": lit(T1)}, args=["foo", "d"]),
    dict(name="missing_file", inputs={}, args=["foo", "nope.txt"]),
    dict(name="bad_regex", inputs={"t1.txt": lit(T1)}, args=["a(", "t1.txt"]),
    dict(name="n2_without_r", inputs={"t1.txt": lit(T1)}, args=["-n", "2", "foo", "t1.txt"]),
    # ---- A.2 quirks ----
    dict(name="q2_empty_matchable", inputs={"t1.txt": lit(T1)}, args=["-O", "q*", "t1.txt"], jit_only=True),  # Q12: the non-JIT lib fails pcre_study here
    dict(name="q3_exact_len", inputs={"f": lit("foo")}, args=["-O", "-l", "foo", "f"]),
    dict(name="q3_xfoofoo", inputs={"f": lit("xfoofoo")}, args=["-O", "-l", "foo", "f"]),
    dict(name="q3_foox", inputs={"f": lit("foox")}, args=["-O", "-l", "foo", "f"]),
    dict(name="q4_wordb", inputs={"f": lit("foofoo")}, args=["-O", "-l", "\\bfoo", "f"]),
    dict(name="q4_caret", inputs={"f": lit("ab\nfoo\nfoo\n")}, args=["-O", "-l", "^foo", "f"]),
    dict(name="q4_caret_m", inputs={"f": lit("ab\nfoo\nfoo\n")}, args=["-O", "-l", "(?m)^foo", "f"]),
    dict(name="q4_dollar", inputs={"f": lit("ab\nfoo\nfoo\n")}, args=["-O", "-l", "foo$", "f"]),
    dict(name="q5_capture", inputs={"t1.txt": lit(T1)}, args=["-O", "(foo)", "t1.txt"]),
    dict(name="q5_noncapture", inputs={"t1.txt": lit(T1)}, args=["-O", "(?:foo)", "t1.txt"]),
    dict(name="q7_context_cap", inputs={"f": lit("A" * 600 + "NEEDLE" + "B" * 600)}, args=["NEEDLE", "f"]),
    dict(name="q7_cap_restart", inputs={"f": lit("A" * 600 + "NEEDLE" + "B" * 520 + "NEEDLE" + "C" * 30 + "\nNEEDLE tail\n")}, args=["-O", "NEEDLE", "f"]),
    dict(name="q21_alt_priority", inputs={"f": lit("xabab_")}, args=["-O", "-l", "a|ab", "f"]),
    # ---- class / tail patterns inside the engine's subset ----
    dict(name="cls_digits_tail", inputs={"f": lit("abc123 abc abc9x\nabc00000000000000000000\nxabc")}, args=["-O", "abc[0-9]*", "f"]),
    dict(name="cls_bounded_tail", inputs={"f": lit("abcdefghijklmnop qr s tuv\n" * 3)}, args=["-O", "-l", "[a-z]{2,5}", "f"]),
    dict(name="cls_phone", inputs={"f": lit("call 555-1234 or 5555-12345, not 55-1234\n555-12345678\n")}, args=["-O", "\\d{3}-\\d{4}", "f"]),
    dict(name="cls_linus", inputs={"f": lit("Linus and linus and LINUS\nxlinusx Linu s\n")}, args=["-O", "[Ll]inus", "f"]),
    dict(name="cls_dot", inputs={"f": lit("a.c abc a\nc axc\n")}, args=["-O", "-l", "a.c", "f"]),
    dict(name="cls_negated_nl", inputs={"f": lit("ab\ncd xxxxx\nefghijk\nx\n")}, args=["-O", "[^x]{5,}", "f"]),
    dict(name="cls_posix", inputs={"f": lit("ab1 abc ABC9 a_c\n")}, args=["-O", "-l", "[[:alpha:]]{3}", "f"]),
    dict(name="cls_escape_meta", inputs={"f": lit("foo.bar fooxbar foo.bar\n")}, args=["-O", "-l", "foo\\.bar", "f"]),
    dict(name="cls_hex", inputs={"f": lit("A\nB A\n\nA\n")}, args=["-O", "-l", "\\x41\\n", "f"]),
    dict(name="cls_quote", inputs={"f": lit("a.b a+b axb\n")}, args=["-O", "-l", "\\Qa.b\\E", "f"]),
    # ---- alternation / optional / bounded repeats / inline options (unfolded into priority-ordered alternatives) ----
    dict(name="alt_foo_tail", inputs={"t1.txt": lit(T1)}, args=["-O", "foo|tail|here", "t1.txt"]),
    dict(name="alt_foo_tail_Ol", inputs={"t1.txt": lit(T1)}, args=["-O", "-l", "foo|tail|here", "t1.txt"]),
    dict(name="alt_prefix_order", inputs={"f": lit("foo fo f foofo ffoo\n")}, args=["-O", "-l", "f|fo|foo", "f"]),
    dict(name="alt_prefix_order2", inputs={"f": lit("foo fo f foofo ffoo\n")}, args=["-O", "-l", "foo|fo|f", "f"]),
    dict(name="alt_optional", inputs={"f": lit("color colour colouur\ncolr colouR\n")}, args=["-O", "colou?r", "f"]),
    dict(name="alt_optional_lazy", inputs={"f": lit("abbbc abc ac abbbbbc\n")}, args=["-O", "-l", "ab{1,3}?c|ab", "f"]),
    dict(name="alt_caseless", inputs={"f": lit("Linus and linus and LINUS\nxlInUsx Linu s\n")}, args=["-O", "(?i)linus", "f"]),
    dict(name="alt_caseless_scoped", inputs={"f": lit("FOO foo Bar bar BAR\n")}, args=["-O", "-l", "(?i:foo)|bar", "f"]),
    dict(name="alt_bounded_mid", inputs={"f": lit("ababc abc c bbbbc aac\n")}, args=["-O", "-l", "[ab]{1,3}c", "f"]),
    dict(name="alt_group_tail", inputs={"f": lit("foobaz barba foobazzzz barbaz\nfooba\n")}, args=["-O", "(?:foo|bar)baz*", "f"]),
    dict(name="alt_group_repeat", inputs={"f": lit("abab abcd cdcdcd ab cdab\n")}, args=["-O", "-l", "(?:ab|cd){2,3}", "f"]),
    dict(name="alt_dfs_order", inputs={"f": lit("aba abab aab\n")}, args=["-O", "-l", "(?:a|ab){1,2}", "f"]),
    dict(name="alt_dotall", inputs={"f": lit("a\nc abc a\n\nc\n")}, args=["-O", "-l", "(?s)a.c", "f"]),
    dict(name="alt_empty_branch", inputs={"t1.txt": lit(T1)}, args=["-O", "foo|", "t1.txt"], jit_only=True),  # can match "": every file skipped (Q2)
    dict(name="alt_end_of_file", inputs={"f": lit("xx foobar")}, args=["-O", "-l", "foobar|foo|bar", "f"]),  # strict loop bound (Q3) with alternatives of different lengths
    dict(name="alt_end_of_file2", inputs={"f": lit("xx foobarx")}, args=["-O", "-l", "foobar|foo|bar", "f"]),
    # capturing groups: ovector[3] holds one pair, a match that sets a group returns 0 and ends the chunk (Q5); one that bypasses it prints
    dict(name="cap_bypass", inputs={"f": lit("ad xx ad ac bc ad\nad\n")}, args=["-O", "-l", "(a|b)c|ad", "f"]),
    dict(name="cap_bypass_lines", inputs={"f": lit("ad xx ad\nad ac bc ad\nad\n")}, args=["-O", "(a|b)c|ad", "f"]),
    dict(name="cap_optional", inputs={"f": lit("xc c zc ac c\n")}, args=["-O", "-l", "(a)?c", "f"]),
    dict(name="cap_counted", inputs={"f": lit("xc xc xac xc\n")}, args=["-O", "-l", "x(a){0,2}c", "f"]),
    dict(name="cap_named", inputs={"f": lit("c c ac c\n")}, args=["-O", "-l", "(?<n>a)c|c", "f"]),
    dict(name="cap_nested_alt", inputs={"f": lit("foo bar baz foobar\n")}, args=["-O", "-l", "ba(?:r|(z))|foo", "f"]),
    dict(name="cap_all_paths", inputs={"t1.txt": lit(T1)}, args=["-O", "(foo|tail)", "t1.txt"]),
    dict(name="cap_big_L5", inputs={"big.txt": {"kind": "big"}}, args=L5 + ["-O", "-l", "NEEDLE|(\\n)\\.{79}\\n\\.{3}N", "big.txt"], big=True,
         jit_only=True),  # the interpreter build keeps its overflow flag from FAILED attempts that closed the group and then returns 0 for the NEEDLE match too; the JIT build (timing build, golden source) does not
    # zero-width assertions (the subject restarts at every match: Q4)
    dict(name="asrt_wordb", inputs={"t1.txt": lit(T1)}, args=["-O", "\\bfoo\\b", "t1.txt"]),
    dict(name="asrt_nonwordb", inputs={"f": lit("foofoo xfoo foo\nfoofoofoo\n")}, args=["-O", "-l", "\\Bfoo", "f"]),
    dict(name="asrt_caret", inputs={"f": lit("foofoo\nfoo\nxfoo foo\n")}, args=["-O", "-l", "^foo", "f"]),
    dict(name="asrt_caret_m", inputs={"f": lit("foofoo\nfoo\nxfoo foo\n")}, args=["-O", "(?m)^foo", "f"]),
    dict(name="asrt_dollar_m", inputs={"f": lit("foo\nfoox\nxfoo\nfoo")}, args=["-O", "-l", "(?m)foo$", "f"]),
    dict(name="asrt_dollar_nl", inputs={"f": lit("foo\nfoo\n")}, args=["-O", "-l", "foo$", "f"]),
    dict(name="asrt_z", inputs={"f": lit("foo\nfoo\n")}, args=["-O", "-l", "foo\\z", "f"]),
    dict(name="asrt_Z", inputs={"f": lit("foo\nfoo\n")}, args=["-O", "-l", "foo\\Z", "f"]),
    dict(name="asrt_words", inputs={"t1.txt": lit(T1)}, args=["-O", "-l", "\\b\\w+\\b", "t1.txt"]),
    dict(name="asrt_alt", inputs={"f": lit("foo bar baz\nbar foo\nxbaz bar")}, args=["-O", "-l", "^foo|bar$|\\bbaz", "f"]),
    dict(name="asrt_line", inputs={"f": lit("foo\n foo\nfoo \nfoo\n")}, args=["-O", "(?m)^foo$", "f"]),
    dict(name="syn8_words", inputs={"syn": {"kind": "synth", "nbytes": 8 << 20, "k": 2}}, args=["-O", "-l", "\\b[a-z]{12,}\\b", "syn"]),
    dict(name="syn8_bol", inputs={"syn": {"kind": "synth", "nbytes": 8 << 20, "k": 2}}, args=["-O", "(?m)^[a-z]{4}\\b|;$", "syn"]),
    # one unbounded repeat in the middle of the pattern ("gapped" alternatives)
    dict(name="gap_plus", inputs={"f": lit("ab aab b aaab abc\naabc xaab\n")}, args=["-O", "-l", "a+b", "f"]),
    dict(name="gap_float", inputs={"f": lit("pi 3.14 v1.5.25 x12y .5 5. 10.02\n")}, args=["-O", "\\d+\\.\\d+", "f"]),
    dict(name="gap_dotstar", inputs={"f": lit("foo bar\nfoo x bar bar\nfoobar barfoo\nfoo\nbar\n")}, args=["-O", "foo.*bar", "f"]),
    dict(name="gap_dotstar_lazy", inputs={"f": lit("foo x bar bar foo bar\n")}, args=["-O", "-l", "foo.*?bar", "f"]),
    dict(name="gap_word_end", inputs={"t1.txt": lit(T1)}, args=["-O", "-l", "[a-z]+\\b", "t1.txt"]),
    dict(name="gap_count_major", inputs={"f": lit("aabc aab aaab abc\n")}, args=["-O", "-l", "a+(?:ab|bc)", "f"]),
    dict(name="gap_mail", inputs={"f": lit("a@b.com x@y.org z@w.net\nfoo@bar.com.\n")}, args=["-O", "\\w+@[a-z]\\.(?:com|org)", "f"]),
    dict(name="gap_assign", inputs={"f": lit("ab = 12;\nb = 1\n c = 3;\nxyz = 77\n")}, args=["-O", "(?m)^[a-z]+ = \\d\\d?;?$", "f"]),
    dict(name="gap_capture", inputs={"f": lit("ac aab ab\n")}, args=["-O", "-l", "a+(b)|a+c", "f"]),
    dict(name="syn8_gap_float", inputs={"syn": {"kind": "synth", "nbytes": 8 << 20, "k": 4}}, args=["-O", "-l", "[0-9]+\\.[0-9]+", "syn"]),
    dict(name="syn8_gap_assign", inputs={"syn": {"kind": "synth", "nbytes": 8 << 20, "k": 4}}, args=["-O", "\\b[a-z_]+ ?= ?[0-9A-F]{1,4};", "syn"]),
    dict(name="syn8_gap_call", inputs={"syn": {"kind": "synth", "nbytes": 8 << 20, "k": 4}}, args=["-O", "-l", "[a-z]{3}\\(.*\\)", "syn"]),
    dict(name="syn8_alt_Ol", inputs={"syn": {"kind": "synth", "nbytes": 8 << 20, "k": 3, "plant": [NEEDLE, 64]}},
         args=["-O", "-l", "foobardoes(?:not)?exist|[0-9A-F]{7}[a-z]?|(?i:xyzzy)", "syn"]),
    dict(name="syn8_alt_O", inputs={"syn": {"kind": "synth", "nbytes": 8 << 20, "k": 3, "plant": [NEEDLE, 64]}},
         args=["-O", "foobardoes(?:not)?exist|[0-9A-F]{7}[a-z]?|(?i:xyzzy)", "syn"]),
    dict(name="syn8_alt_lines", inputs={"syn": {"kind": "synth", "nbytes": 8 << 20, "k": 3, "plant": [NEEDLE, 64]}},
         args=["foobardoes(?:not)?exist|[0-9A-F]{7}[a-z]?|(?i:xyzzy)", "syn"]),
    dict(name="syn8_caseless_hex", inputs={"syn": {"kind": "synth", "nbytes": 8 << 20, "k": 1}}, args=["-O", "-l", "(?i)[a-f]{5}[g-z]_", "syn"]),
    # ---- synthetic text (SURVEY section 8d generator) ----
    dict(name="syn8_ident_Ol", inputs={"syn": {"kind": "synth", "nbytes": 8 << 20, "k": 0}}, args=["-O", "-l", IDENT, "syn"]),
    dict(name="syn8_ident_O", inputs={"syn": {"kind": "synth", "nbytes": 8 << 20, "k": 0}}, args=["-O", IDENT, "syn"]),
    dict(name="syn8_ident_lines", inputs={"syn": {"kind": "synth", "nbytes": 8 << 20, "k": 0}}, args=[IDENT, "syn"]),
    dict(name="syn8_needle_absent", inputs={"syn": {"kind": "synth", "nbytes": 8 << 20, "k": 0}}, args=["-O", NEEDLE, "syn"]),
    dict(name="syn8_needle_planted", inputs={"syn": {"kind": "synth", "nbytes": 8 << 20, "k": 3, "plant": [NEEDLE, 64]}}, args=["-O", "-l", NEEDLE, "syn"]),
    dict(name="syn8_needle_planted_lines", inputs={"syn": {"kind": "synth", "nbytes": 8 << 20, "k": 3, "plant": [NEEDLE, 64]}}, args=["-O", NEEDLE, "syn"]),
    dict(name="syn8_hex4", inputs={"syn": {"kind": "synth", "nbytes": 8 << 20, "k": 1}}, args=["-O", "-l", "[0-9A-F]{6}[a-z]", "syn"]),
    dict(name="syn8_e_run", inputs={"syn": {"kind": "synth", "nbytes": 1 << 20, "k": 2}}, args=["-O", "-l", "e+", "syn"]),
    # SURVEY A.4 known answers (256 MiB, seed k=0)
    dict(name="syn256_needle", inputs={"syn": {"kind": "synth", "nbytes": 256 << 20, "k": 0}}, args=[NEEDLE, "syn"], big=True),
    dict(name="syn256_ident_Ol", inputs={"syn": {"kind": "synth", "nbytes": 256 << 20, "k": 0}}, args=["-O", "-l", IDENT, "syn"], big=True),
    dict(name="syn256_ident_Ol_L5", inputs={"syn": {"kind": "synth", "nbytes": 256 << 20, "k": 0}}, args=L5 + ["-O", "-l", IDENT, "syn"], big=True),
    dict(name="syn256_ident_O", inputs={"syn": {"kind": "synth", "nbytes": 256 << 20, "k": 0}}, args=["-O", IDENT, "syn"], big=True),
    # ---- A.5 chunk-boundary fixtures, 32 MiB chunks ----
    dict(name="big_Ol_L5", inputs={"big.txt": {"kind": "big"}}, args=L5 + ["-O", "-l", "NEEDLE", "big.txt"], big=True),
    dict(name="big_O_L5", inputs={"big.txt": {"kind": "big"}}, args=L5 + ["-O", "NEEDLE", "big.txt"], big=True),
    dict(name="big_Ol_1g", inputs={"big.txt": {"kind": "big"}}, args=["-O", "-l", "NEEDLE", "big.txt"], big=True),
    dict(name="big2_needle_L5", inputs={"big2.txt": {"kind": "big2"}}, args=L5 + ["-O", "-l", "NEEDLE", "big2.txt"], big=True),
    dict(name="big2_arun_L5", inputs={"big2.txt": {"kind": "big2"}}, args=L5 + ["-O", "-l", "a{30,}", "big2.txt"], big=True),
    dict(name="big2_arun_1g", inputs={"big2.txt": {"kind": "big2"}}, args=["-O", "-l", "a{30,}", "big2.txt"], big=True),
    dict(name="big_alt_Ol_L5", inputs={"big.txt": {"kind": "big"}}, args=L5 + ["-O", "-l", "NEE?DLE|DLE|\\n\\.{79}\\n\\.{3}N", "big.txt"], big=True),
    dict(name="big_alt_O_L5", inputs={"big.txt": {"kind": "big"}}, args=L5 + ["-O", "NEE?DLE|DLE", "big.txt"], big=True),
    dict(name="big2_alt_L5", inputs={"big2.txt": {"kind": "big2"}}, args=L5 + ["-O", "-l", "a{30,}|NEEDLE|a{5}\\.", "big2.txt"], big=True),
    dict(name="big_wordb_L5", inputs={"big.txt": {"kind": "big"}}, args=L5 + ["-O", "-l", "\\bNEEDLE\\b", "big.txt"], big=True),
    dict(name="big_dollar_L5", inputs={"big.txt": {"kind": "big"}}, args=L5 + ["-O", "-l", "NEEDLE$", "big.txt"], big=True),  # only where a window ends with the chunk
    dict(name="big_lines_L5", inputs={"big.txt": {"kind": "big"}}, args=L5 + ["-O", "-l", "(?m)^\\.{79}$|NEEDLE", "big.txt"], big=True),
    dict(name="big_caret_L5", inputs={"big.txt": {"kind": "big"}}, args=L5 + ["-O", "-l", "^\\.{10}NEE|^\\.{4090}|DLE\\B", "big.txt"], big=True),
    dict(name="big_gap_L5", inputs={"big.txt": {"kind": "big"}}, args=L5 + ["-O", "-l", "\\.+NEEDLE", "big.txt"], big=True),   # runs of dots reach back across lines? no: up to the newline
    dict(name="big2_gap_L5", inputs={"big2.txt": {"kind": "big2"}}, args=L5 + ["-O", "-l", "a+\\.|\\.+N", "big2.txt"], big=True),
    # ---- patterns the kernels can only pre-filter (gscan_info.exact == 0): the host's backtracking matcher confirms every offset ----
    dict(name="inx_two_repeats", inputs={"f": lit("a@b.com x@y.org zz@ww.com.\nfoo@bar.comx @a.com a@.com\n")}, args=["-O", "\\w+@\\w+\\.com", "f"]),
    dict(name="inx_two_repeats_Ol", inputs={"f": lit("a@b.com x@y.org zz@ww.com.\nfoo@bar.comx @a.com a@.com\n")}, args=["-O", "-l", "\\w+@\\w+\\.com", "f"]),
    dict(name="inx_group_plus", inputs={"f": lit("foobaz foobarbaz barbar baz bazfoo foofoobaz\n")}, args=["-O", "-l", "(?:foo|bar)+baz", "f"]),
    dict(name="inx_group_star_cap", inputs={"f": lit("abab ab c abc x\n")}, args=["-O", "-l", "(ab)*c|x", "f"]),
    dict(name="inx_three_gaps", inputs={"f": lit("a1b2c\nabc\na b\nc a..b..c..a.b.c\n")}, args=["-O", "a.*b.*c", "f"]),
    dict(name="inx_possessive", inputs={"f": lit("aaab aab ab b aaa\n")}, args=["-O", "-l", "a++b|a{1,40}$", "f"]),
    dict(name="inx_lazy_group", inputs={"f": lit("abcd ababd cd abcabd\n")}, args=["-O", "-l", "(?:ab|c)+?d", "f"]),
    dict(name="inx_many_alts", inputs={"f": lit("w001 w064 w065 xw03 w0640\n")}, args=["-O", "-l", "|".join("w%03d" % i for i in range(65)), "f"]),
    dict(name="look_ahead", inputs={"f": lit("foobar foobaz foo\nfoobar")}, args=["-O", "-l", "foo(?=bar)", "f"]),
    dict(name="look_ahead_neg", inputs={"f": lit("foobar foobaz foo\nfoobar")}, args=["-O", "-l", "foo(?!bar)", "f"]),
    dict(name="look_behind", inputs={"f": lit("xfoo foo yfoo\nfoo xfoofoo")}, args=["-O", "-l", "(?<=x)foo", "f"]),
    dict(name="look_behind_neg", inputs={"f": lit("xfoo foo yfoo\nfoo xfoofoo")}, args=["-O", "-l", "(?<!x)foo", "f"]),  # Q4: nothing lies before a restart position
    dict(name="look_behind_alt", inputs={"f": lit("abd cd xd abcd\n")}, args=["-O", "-l", "(?<=ab|c)d", "f"]),
    dict(name="look_atomic", inputs={"f": lit("aaab aaa ab\n")}, args=["-O", "-l", "(?>a+)b|(?>a+)a", "f"]),
    dict(name="look_capture", inputs={"f": lit("xab ab\n")}, args=["-O", "-l", "x(?=(a))ab|ab", "f"]),
    # back references: a match that used one has set a group -> rc == 0 -> the chunk ends there (Q5); before that, the
    # alternatives without groups print
    dict(name="bref_ends_chunk", inputs={"f": lit("li nus li ab aa li\nli\n")}, args=["-O", "-l", "(a|b)\\1|li", "f"]),
    dict(name="bref_never_set", inputs={"f": lit("li nus li ab ba li\nli\n")}, args=["-O", "-l", "(a|b)\\1|li", "f"]),
    dict(name="bref_named_lines", inputs={"f": lit("x nus\nnus abab nus\nnus\n")}, args=["-O", "(?P<q>ab)(?P=q)|nus", "f"]),
    dict(name="bref_self", inputs={"f": lit("z abbc z ac z\n")}, args=["-O", "-l", "(a|b\\1)+c|z", "f"]),
    # \K: ovector[0] moves -- the offset printed, the highlighted part and the "before" context follow it
    dict(name="keep_offsets", inputs={"f": lit("foobar foo bar foobar\nxfoobar")}, args=["-O", "-l", "foo\\Kbar", "f"]),
    dict(name="keep_lines", inputs={"f": lit("foobar foo bar foobar\nxfoobar\nfoo\nbar\n")}, args=["-O", "foo\\K(?:bar)?", "f"]),
    dict(name="syn8_inx_call", inputs={"syn": {"kind": "synth", "nbytes": 8 << 20, "k": 5}}, args=["-O", "[a-z]+\\([a-z0-9, ]*\\);", "syn"]),
    dict(name="syn8_inx_member", inputs={"syn": {"kind": "synth", "nbytes": 8 << 20, "k": 5}}, args=["-O", "-l", "[a-z]+_[0-9]+\\.[a-z]+", "syn"]),
    dict(name="syn8_inx_look", inputs={"syn": {"kind": "synth", "nbytes": 8 << 20, "k": 5}}, args=["-O", "-l", "(?<![a-z_])[a-z]{3}(?=\\()|(?<=;)\\n(?!\\n)", "syn"]),
    dict(name="big_inx_L5", inputs={"big.txt": {"kind": "big"}}, args=L5 + ["-O", "-l", "\\.+N+E+DLE|(?<=\\n)\\.{3}N", "big.txt"], big=True),
    dict(name="big_s_L5", inputs={"big.txt": {"kind": "big"}}, args=L5 + ["-s", "-O", "-l", "NEEDLE", "big.txt"], big=True),
    dict(name="big_l_L5", inputs={"big.txt": {"kind": "big"}}, args=L5 + ["-l", "NEEDLE", "big.txt"], big=True),
]

R2TXT = ("f(a(b)c) ((x)) (() a(b\nab cb c ac xab xc\n12-34 x12 x y9\nabba racecar noon ab\n<a<b>c> <<>>\n"
         "a\r\nb a\rb a\x85b ab\nab\xe9\xc9 12\xb2 x_\xb5\n1-23-4x ad bc\nabc ab aBc aab aaab\n")
R2 = {"r2.txt": lit(R2TXT)}
CASES += [
    # ---- round 2, second half: the rest of PCRE1's syntax (conditionals, calls / recursion, branch reset, \R \X \p, verbs) ----
    dict(name="r2_cond_group", inputs=R2, args=["-O", "-l", r"(a)?(?(1)b|c)", "r2.txt"]),
    dict(name="r2_cond_group_lines", inputs=R2, args=["-O", r"x(a)?(?(1)b|c)", "r2.txt"]),
    dict(name="r2_cond_name", inputs=R2, args=["-O", "-l", r"(?<n>x)?(?(<n>)a|c)b?", "r2.txt"]),
    dict(name="r2_cond_lookahead", inputs=R2, args=["-O", "-l", r"(?(?=a)ab|cb)", "r2.txt"]),
    dict(name="r2_cond_neg_capture", inputs=R2, args=["-O", "-l", r"a(?(?!(b)))", "r2.txt"]),
    dict(name="r2_define_call", inputs=R2, args=["-O", "-l", r"(?(DEFINE)(?<d>[0-9]+))x(?&d)", "r2.txt"]),
    dict(name="r2_recursion_parens", inputs=R2, args=["-O", "-l", r"\((?:[^()]++|(?R))*\)", "r2.txt"]),
    dict(name="r2_recursion_parens_lines", inputs=R2, args=[r"\((?:[^()]|(?R))*\)", "r2.txt"]),
    dict(name="r2_recursion_angle", inputs=R2, args=["-O", "-l", r"<(?:[^<>]+|(?R))*>", "r2.txt"]),
    dict(name="r2_palindrome", inputs=R2, args=["-O", "-l", r"\b(?:(\w)(?:(?R)|\w?)\1)\b", "r2.txt"]),
    dict(name="r2_call_forward", inputs=R2, args=["-O", "-l", r"(?+1)(ab)", "r2.txt"]),
    dict(name="r2_call_in_define_free", inputs=R2, args=["-O", "-l", r"x(?1)?(a|c)", "r2.txt"]),
    dict(name="r2_branch_reset", inputs=R2, args=["-O", "-l", r"(?|(a)b|(c)b)\1?", "r2.txt"]),
    dict(name="r2_branch_reset_nocap", inputs=R2, args=["-O", "-l", r"(?|a|c)b", "r2.txt"]),
    dict(name="r2_anynl", inputs=R2, args=["-O", "-l", r"a\Rb", "r2.txt"]),
    dict(name="r2_extuni", inputs=R2, args=["-O", "-l", r"a\Xb", "r2.txt"]),
    dict(name="r2_prop_letters", inputs=R2, args=["-O", "-l", r"\p{L}{3,}", "r2.txt"]),
    dict(name="r2_prop_upper_lower", inputs=R2, args=["-O", "-l", r"\p{Lu}\p{Ll}|\P{L}\p{Nd}", "r2.txt"]),
    dict(name="r2_fail_verb", inputs=R2, args=["-O", "-l", r"a(*FAIL)|cb", "r2.txt"]),
    dict(name="r2_accept_verb", inputs=R2, args=["-O", "-l", r"a(*ACCEPT)b", "r2.txt"], jit_only=True),  # (8.45 finds a minimum length for it and prints; 8.39, the timing build, does not: Q2)
    dict(name="r2_callout", inputs=R2, args=["-O", "-l", r"ab(?C)c", "r2.txt"]),
    dict(name="r2_ungreedy", inputs=R2, args=["-O", "-l", r"(?U)a+b", "r2.txt"]),
    dict(name="r2_octal_quoted_class", inputs=R2, args=["-O", "-l", r"[\Qa-\E]\o{142}", "r2.txt"]),
    # the JIT build loses the match the interpreter finds ((\d+)-(?1)x on "1-23-4x": DESIGN.md section 2); the product refuses the pattern
    dict(name="r2_jit_lost_match", inputs=R2, args=["-O", "-l", r"(\d+)-(?1)x", "r2.txt"], jit_only=True),
]


def run(binary, args, cwd):
    r = subprocess.run([binary] + args, cwd=cwd, capture_output=True)
    return r.returncode, r.stdout, r.stderr


def main():
    if not os.path.exists(REF_JIT):
        sys.exit("oracle/_ref/grab_jit missing: run `make -C oracle ref` (needs /root/reference)")
    out = {"_about": "outputs of the reference binary (oracle/_ref/grab_jit, PCRE 8.39 JIT) -- generated by tests/golden/make_golden.py; do not edit",
           "cases": []}
    cache = {}
    for case in CASES:
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            for rel, recipe in case["inputs"].items():
                materialize(recipe, os.path.join(td, rel), cache)
            rc, so, se = run(REF_JIT, case["args"], td)
            rc2, so2, se2 = run(REF_NOJIT, case["args"], td)
            if case.get("sort"):
                so = b"".join(sorted(so.splitlines(True)))
                so2 = b"".join(sorted(so2.splitlines(True)))
            if not case.get("jit_only"):
                assert (rc, so) == (rc2, so2), "JIT and non-JIT reference builds disagree on " + case["name"]
        rec = {"name": case["name"], "inputs": case["inputs"], "args": case["args"], "rc": rc,
               "sorted": bool(case.get("sort")), "big": bool(case.get("big")),
               "stderr": se.decode("latin-1"),
               "stdout_md5": hashlib.md5(so).hexdigest(), "stdout_len": len(so), "stdout_lines": so.count(b"\n")}
        if len(so) <= 16384:
            rec["stdout_b64"] = base64.b64encode(so).decode()
        else:
            rec["stdout_head_b64"] = base64.b64encode(so[:512]).decode()
        out["cases"].append(rec)
        print("%-28s rc=%d lines=%d md5=%s" % (case["name"], rc, rec["stdout_lines"], rec["stdout_md5"]))
    with open(os.path.join(ROOT, "tests", "golden", "golden.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
