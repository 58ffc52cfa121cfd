#!/usr/bin/env python3
"""A zoo of everyday grep patterns through the product's compiler and its host-side pipeline, on CPU: is the pattern accepted, is its
minlen PCRE's, and is the output of the walk over what the kernels are specified to list (tests/inputs.py: the resolved list for a
database with info.resolve, the VM-filtered list, or the group starts) byte-identical to libpcre's under the reference's loop -- in
`-O -l`, `-O` and line mode -- over a text that HAS such things in it (log lines, C, JSON, mail headers, dates, addresses) and
over the SURVEY.md 8d corpus.  One JSON line per pattern + a summary.  `oracle/` is the checker."""
import ctypes as C
import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("GSCAN_MATCH_LIMIT", "5000000")
from grab_amd import engine, filegrep, synth  # noqa: E402
import inputs  # noqa: E402

ZOO = [
    # networks, identifiers, secrets
    r"\b(?:\d{1,3}\.){3}\d{1,3}\b", r"\b(?:25[0-5]|2[0-4]\d|1?\d?\d)(?:\.(?:25[0-5]|2[0-4]\d|1?\d?\d)){3}\b", r"(?:[0-9a-fA-F]{2}:){5}[0-9a-fA-F]{2}",
    r"[A-Za-z0-9._%+-]+@[A-Za-z0-9.-]+\.[A-Za-z]{2,}", r"https?://[^\s\"'<>]+", r"(?i)\bhttps?://(?:www\.)?[a-z0-9-]+(?:\.[a-z0-9-]+)+(?:/\S*)?", r"\bwww\.[a-z0-9-]+\.[a-z]{2,}\b",
    r"[0-9a-fA-F]{8}-[0-9a-fA-F]{4}-[0-9a-fA-F]{4}-[0-9a-fA-F]{4}-[0-9a-fA-F]{12}", r"\b[0-9a-f]{40}\b", r"\b[0-9a-f]{32}\b", r"0x[0-9a-fA-F]+", r"\b0[xX][0-9a-fA-F]{1,8}\b",
    r"AKIA[0-9A-Z]{16}", r"(?i)(?:api|secret)[_-]?key\s*[=:]\s*\S+", r"(?i)password\s*[=:]\s*\S+", r"-----BEGIN [A-Z ]+-----", r"(?i)bearer\s+[a-z0-9._-]+", r"eyJ[A-Za-z0-9_-]+\.[A-Za-z0-9_-]+",
    r"\b[A-Z]{2}\d{2}[A-Z0-9]{10,30}\b", r"\b\d{3}-\d{2}-\d{4}\b", r"\b\d{4}[ -]?\d{4}[ -]?\d{4}[ -]?\d{4}\b", r"\+?\d{1,3}[ -]?\(?\d{2,4}\)?[ -]?\d{3}[ -]?\d{4}", r"\(\d{3}\) \d{3}-\d{4}", r"555-\d{4}",
    # dates, times, numbers
    r"\b\d{4}-\d{2}-\d{2}\b", r"\b\d{2}:\d{2}:\d{2}\b", r"\d{4}-\d{2}-\d{2}[T ]\d{2}:\d{2}:\d{2}(?:\.\d+)?(?:Z|[+-]\d{2}:?\d{2})?", r"\b\d{1,2}/\d{1,2}/\d{2,4}\b",
    r"(?i)\b(?:jan|feb|mar|apr|may|jun|jul|aug|sep|oct|nov|dec)[a-z]* \d{1,2},? \d{4}\b", r"(?i)\b(?:mon|tue|wed|thu|fri|sat|sun)[a-z]*\b", r"[-+]?\d+\.\d+(?:[eE][-+]?\d+)?", r"\b\d+(?:,\d{3})+\b", r"\b\d+%", r"\$\d+(?:\.\d{2})?",
    r"\b\d+(?:\.\d+)? ?(?:ms|us|ns|s|KiB|MiB|GiB|kB|MB|GB)\b", r"\b[1-9]\d{3,}\b", r"\b0\d+\b", r"^\d+$", r"(?m)^\d+\s", r"\d+(?=%)", r"(?<=\$)\d+", r"(?<![\d.])\d+(?![\d.])",
    # logs
    r"error|warning|fatal|critical", r"(?i)\b(?:error|warn(?:ing)?|fatal|crit(?:ical)?)\b", r"(?i)\bexception\b.*", r"(?m)^\[?(?:ERROR|WARN|INFO|DEBUG)\]?", r"\[(?:ERROR|WARN)\]\s+\S+", r"(?i)failed to \w+", r"(?i)\btimed? ?out\b",
    r"(?i)connection (?:refused|reset|closed)", r"\bpid[=: ]\d+", r"\buser=\w+", r"\bstatus=(?:4|5)\d\d\b", r"\" (?:4|5)\d\d \d+", r"(?m)^\S+ \S+ \S+ \[[^\]]+\] \"[A-Z]+ [^\"]+\" \d{3}", r"\bGET\b|\bPOST\b|\bPUT\b|\bDELETE\b", r"(?i)\bsegfault\b|\bsegmentation fault\b",
    r"Traceback \(most recent call last\)", r"(?m)^\s+at [\w.$]+\(", r"(?m)^\s*File \"[^\"]+\", line \d+", r"\b[A-Z][a-zA-Z]+(?:Exception|Error)\b", r"(?i)out of memory", r"oom-?kill", r"\bkernel: \[\s*\d+\.\d+\]",
    # code
    r"\bif\s*\(", r"\bfor\s*\([^;]*;[^;]*;[^)]*\)", r"\bwhile\s*\(", r"#include\s*<[^>]+>", r"#include\s*\"[^\"]+\"", r"(?m)^#\s*define\s+\w+", r"(?m)^#\s*if(?:n?def)?\b", r"\breturn\b.*;", r"\w+\s*=\s*\w+\s*\(", r"\w+(?=\()",
    r"\b[A-Za-z_]\w*\s*\(", r"\b(?:int|long|char|float|double|void|unsigned)\b\s+\**\w+", r"\bstruct\s+\w+\s*\{", r"\btypedef\b.*;", r"->\w+", r"\w+::\w+", r"\bstd::\w+", r"\bnew\s+\w+", r"\bdelete\s*(?:\[\])?\s*\w+", r"\bTODO\b|\bFIXME\b|\bXXX\b",
    r"(?i)\b(?:todo|fixme|xxx)\b", r"//.*", r"/\*.*?\*/", r"\"[^\"\n]*\"", r"'[^'\n]*'", r"\"(?:[^\"\\\n]|\\.)*\"", r"\b[a-z]+(?:[A-Z][a-z]+)+\b", r"\b(?:[a-z]+_)+[a-z]+\b", r"\b[A-Z][A-Z0-9_]{2,}\b", r"\b_[A-Za-z]\w*", r"\b\w{12,}\b",
    r"[A-Za-z_][A-Za-z0-9_]{15,}", r"(?<![A-Za-z0-9_])[A-Z]{2,}(?![A-Za-z0-9_])", r"\bdef\s+\w+\s*\(", r"\bclass\s+\w+", r"\bimport\s+\w+", r"\bfrom\s+[\w.]+\s+import\b", r"\blambda\b[^:]*:", r"\bself\.\w+", r"(?m)^\s*@\w+", r"\bfunction\s+\w*\s*\(",
    r"=>", r"\bconst\s+\w+\s*=", r"\bvar\s+\w+", r"\bconsole\.log\(", r"\$\{?\w+\}?", r"\$\(\w+\)", r"(?m)^\s*[a-z_]+\s*=", r"(?m)^[A-Z_]+=", r"[;,]\s*$", r"(?m)[ \t]+$", r"(?m)^\s*$", r"\t", r"\s{2,}\S", r"\s\w{8,}\s", r"[{][^{}]*[}]", r"\([^()]*\)", r"\[[^\]\n]*\]",
    r"(?<=\()[^()\n]+(?=\))", r"\b(\w+)\s+\1\b", r"\b(\w)\w*\1\b", r"(a|b)+c", r"(?:ab)+c", r"\b[a-z]{3,}\b", r"\b[A-Z][a-z]+\b", r"\b[A-Z][a-z]+ [A-Z][a-z]+\b", r"\bthe\b.{0,20}\bthe\b", r"(?i)\bselect\b.+\bfrom\b", r"(?i)\binsert\s+into\s+\w+", r"(?i)\bwhere\s+\w+\s*=",
    # JSON / config / markup
    r"\"\w+\"\s*:\s*\"[^\"]*\"", r"\"\w+\"\s*:\s*\d+", r"\"(?:id|name|type)\"\s*:", r"<[a-z][a-z0-9]*[^>]*>", r"</[a-z]+>", r"<!--.*?-->", r"&[a-z]+;", r"\bhref=\"[^\"]*\"", r"(?m)^\[[^\]]+\]$", r"(?m)^\w+:\s+\S", r"(?m)^---$", r"(?m)^- \w+", r"(?m)^#{1,6} .+",
    r"\*\*[^*]+\*\*", r"`[^`\n]+`", r"\bv?\d+\.\d+\.\d+(?:-[\w.]+)?\b", r"\b[\w.-]+\.(?:c|h|cc|py|js|json|ya?ml|txt|log)\b", r"(?:/[\w.-]+){2,}", r"[A-Za-z]:\\(?:[\w.-]+\\)*[\w.-]+", r"~/[\w./-]+", r"\.\./", r"\b\w+\.so(?:\.\d+)*\b",
    # PCRE extras
    r"\p{Lu}\p{Ll}+", r"[[:upper:]][[:lower:]]+", r"[[:digit:]]+[[:space:]][[:alpha:]]+", r"\d++\.", r"(?>\w+)\(", r"a*+b", r"(?i:foo)bar", r"foo(?!bar)", r"(?<!foo)bar", r"\Afoo", r"bar\z", r"\bfoo\b|\bbar\b", r"\Bfoo", r"(?s)a.b", r"(?x) f o o  # comment",
    r"(?U)a+", r"a+?b", r"\x41\x42", r"\101", r"[\x00-\x1f]", r"[^\x20-\x7e\n]", r"\R", r"\h+", r"\N+", r"(?|(a)|(b))c", r"(?P<w>\w+) (?P=w)", r"(?:(?:a|b)(?:c|d)){2,}", r"(foo|bar|baz)+qux", r"\Qa.b\E", r"(*UTF8)a", r"(*ANYCRLF)a$", r"a(*SKIP)b", r"(?R)?x", r"\Kfoo", r"(\d+)(?(1)a|b)",
    # feature variety: empty alternatives, counted / lazy / possessive repeats, anchors and (?m), case folding, high bytes, look-around,
    # atomic groups, nested repeats, back references by number / name / relative, conditions, class syntax corners, escapes, settings
    r"a|", r"|a", r"(?:a|)b", r"(?:|a)+b", r"a{0}b", r"a{0,}b", r"a{,3}", r"a{2}{3}", r"x{1,2}?y", r"(?:ab){2,3}?c", r"[a-c]{2,4}+d", r"\d{3,}?\.", r"\w*?\(", r".*?;", r".+?=", r"[^;]*+;", r"(?s).{3}\n", r"(?s)\{.*?\}", r"(?s)/\*.*?\*/",
    r"(?m)^$", r"(?m)^.{0,3}$", r"(?m)^.*error.*$", r"(?m)^(?!#).+", r"(?m)^(?=.*foo)(?=.*bar).*$", r"(?m)^\s*#.*$", r"(?m)^[^#\n]*=", r"(?m)\S$", r"(?m)^\t+", r"(?m)^ {4}\S", r"(?m)^.{80,}$", r"(?m)^[A-Z].*\.$", r"\A\d+", r"\d+\Z", r"\n\n", r"\r?\n", r"[\r\n]+",
    r"(?i)ERROR", r"(?i)[a-f]{4}", r"(?i)\bcon(?:nection|fig)\b", r"(?i:a)B", r"(?i)a(?-i)B", r"((?i)a)B", r"(?i)[^a-z]{2}[a-z]", r"(?i)\x41b", r"(?i)straße|strasse", r"(?i)\bé\w+", r"[à-ÿ]+", r"[\x80-\xff]{2,}", r"[^\x00-\x7f]",
    r"\bfoo\B", r"\B\w\B", r"\b\d+\b(?!\.)", r"(?<=\s)\w+(?=\s)", r"(?<=^|,)\w+", r"(?<=ab|cd)e", r"(?<=a{2})b", r"(?<!\\)\"", r"(?<![a-z])[a-z]{2}(?![a-z])", r"(?<=\d)(?=(?:\d{3})+\b)", r"(?=\d{4})\d{2}", r"(?!0)\d+", r"\w+(?<!ing)\b", r"\b\w+(?<=ed)\b",
    r"(?>a+)b", r"(?>\d+)\.(?>\d+)", r"(?>[a-z]+|[0-9]+)x", r"a++b", r"[a-z]*+\d", r"(?:a+)+b", r"(?:a*)*b", r"(?:a|aa)+b", r"(a+)+$", r"(\w+\s?)+$", r"(?:\w+\s)*\w+\.", r"(?:[a-z]+,)*[a-z]+;",
    r"(\w+)=\1", r"(['\"]).*?\1", r"(a)(b)?\2", r"(?:(a)|b)\1", r"(\d)\d\1", r"\b(\w)(\w)\2\1\b", r"(?<n>\d+)-\k<n>", r"(?'q'['\"])\w+\k'q'", r"\g{1}(x)", r"(a)\g{-1}", r"(?i)(foo)\s\1",
    r"(?(?=\d)\d{2}|[a-z]{2})", r"(a)?(?(1)b|c)", r"(?<q>\")?\w+(?(q)\")", r"(?(?<=a)b|c)",
    r"[]]", r"[^]]", r"[]a]+", r"[a\]]+", r"[\[\]]", r"[a-]", r"[-a]", r"[a\-z]", r"[\w-]+", r"[\d.]+", r"[\s\S]", r"[^\W\d]+", r"[[:alnum:]_]+", r"[^[:space:]]+", r"[[:^digit:]]{3}", r"[[:punct:]]{2,}", r"[[:xdigit:]]{8}", r"[a-z&&[^aeiou]]",
    r"\.", r"\\", r"\/", r"\-", r"a\ b", r"\e", r"\a", r"\f", r"\cA", r"\x{41}", r"\o{101}", r"\N{U+41}", r"\p{L}+", r"\P{L}+", r"\pL\pN", r"\p{Lu}", r"\X", r"\C", r"\v+", r"\H+", r"\V+", r"\D{3}", r"\S+@\S+", r"\W{2,}",
    r"foo.*bar", r"foo.+bar", r"foo.{1,10}bar", r"foo[^\n]*bar", r"foo(?:.|\n)*?bar", r"^foo", r"foo$", r"^foo$", r"(?m)^foo$", r"foo\n", r"foo(?=\n)", r"\bfoo\b.*\bbar\b", r".*foo", r".*", r".+", r".", r"\w", r"a", r"ab", r"abc", r"abcd", r"abcde",
    r"(?#comment)foo", r"(?x)\d+ \s* # digits\n [a-z]+", r"(?xx)[a b]c", r"(?J)(?<n>a)|(?<n>b)", r"(?-m)^a", r"(?s-i:a.)b", r"(?:(?i)a)b", r"(?^)a",
    # syntax corners: what pcre_compile rejects the product's compiler has to reject too (and the other way round)
    r"\cA", r"\c", r"\c[", r"\x", r"\xg", r"\x4", r"\x{", r"\x{41", r"\x{100}", r"\x{}", r"\0", r"\00", r"\012", r"\0123", r"\o{}", r"\o{400}",
    r"[a-\d]", r"[\d-z]", r"[z-a]", r"[a-a]", r"[\x41-\x43]", r"[[:foo:]]", r"[[.a.]]", r"[[=a=]]", r"[:alpha:]", r"[[:alpha:][:digit:]]", r"[[:alpha:]-z]", r"[a-[:digit:]]", r"[", r"[]", r"[^]", r"[a", r"[\]", r"a]", r"[\b]", r"[\B]", r"[\R]", r"[\X]", r"[\N]", r"[\Qa-c\E]", r"[\E]", r"[a\Q]\E]",
    r"(?<=a+)b", r"(?<=a|bc)d", r"(?<=(?:a|bc))d", r"(?<=a*)b", r"(?<=a{2,3})b", r"(?<=\Ka)b", r"(?<=\b)a", r"(?<!^)a", r"(?<=a(?=b))b", r"(?<=(a))b", r"(?<=\1)(a)", r"(?<=a\C)b", r"(?<=\R)a", r"(?<=\X)a",
    r"a**", r"a+*", r"a?*", r"a*?+", r"a{2}*", r"a{2,1}", r"a{65536}", r"a{65535}", r"x{1001}y", r"*a", r"+a", r"?a", r"{1}a", r"a{1", r"a{1,", r"a{,}", r"a|*", r"(*)", r"(+)", r"()", r"(?:)", r"(|)", r"()*", r"(a)*?", r"^*", r"$+", r"\b+", r"(?=a)*", r"(?=a)+b", r"(?!a){2}b", r"\A*a",
    r"(?P=n)", r"(?P<n>a)(?P=n)", r"(?P<n>a)(?P>n)", r"(?<n>a)\k{n}", r"(?<n>a)\g{n}", r"\g", r"\g1", r"\g{", r"\g{0}", r"\g{-1}", r"\g-1(a)", r"(a)\g+1(b)", r"\k", r"\k<n>", r"(?<1a>x)", r"(?<n>a)(?<n>b)", r"(?<>a)", r"(?P<n", r"(?&n)(?<n>a)", r"(?1)(a)", r"(a)(?1)", r"(?R)", r"(?0)", r"(?+1)(a)", r"(?-1)",
    r"(?", r"(?<", r"(?a)", r"(?i", r"(?i-", r"(?-)", r"(?i-i)a", r"(?im-sx)a", r"(?i:)", r"(?i:a", r"(?#", r"(?#)a", r"(a", r"a)", r"(?:a", r"(?>a", r"(?|a", r"(?=", r"(?!)", r"(?!)a", r"(*", r"(*FOO)", r"(*ACCEPT)", r"(*FAIL)", r"(*F)a", r"a(*COMMIT)b", r"(*PRUNE)a", r"(*THEN)a", r"(*MARK:x)a", r"(*:x)a", r"(*LF)a", r"(*CR)a", r"(*CRLF)a", r"(*ANY)a", r"(*BSR_ANYCRLF)\R", r"(*BSR_UNICODE)\R", r"(*NO_START_OPT)a", r"(*NO_AUTO_POSSESS)a+b", r"(*UCP)\w", r"(*UTF)a", r"(*LIMIT_MATCH=10)a", r"(*LIMIT_RECURSION=10)a",
    r"\Qabc", r"\Qa\Eb\Qc", r"\E", r"a\E+", r"\Q\E", r"\Q\Ea", r"a\Q\E*", r"\Q*\E+", r"\Qa|b\E",
    r"\p{Foo}", r"\p", r"\pZ", r"\p{Z}", r"\p{^Lu}", r"\P{^Lu}", r"\p{L&}", r"\p{Any}", r"\p{Xan}", r"\p{Xps}", r"\p{Xsp}", r"\p{Xwd}", r"\p{Latin}", r"\p{Greek}", r"\p{Nd}+", r"[\p{Lu}\d]+", r"[^\p{L}]",
    r"(?C)a", r"(?C12)a", r"a(?C1)b", r"(?C256)a", r"\Ga", r"a\G", r"\G", r"(?:\Ga|b)c", r"\L", r"\l", r"\U", r"\u", r"\ua", r"\i", r"\j", r"\y", r"\_", r"\ ", r"\~", r"\%", r"(?X)\j", r"(?X)a",
    r"(?x)a b", r"(?x)a\ b", r"(?x)[a b]", r"(?x)a#b\nc", r"(?x)a#b", r"(?x) ", r"(?x)(?# c) a", r"(?x)a {2}", r"(?x)a{ 2}", r"(?x)\Q a \E",
    r"a\z", r"a\Z", r"\za", r"\Za", r"$a", r"a^", r"a^b", r"(?m)a$b", r"(?m)a$\nb", r"a$\n", r"(?m)$\n^", r"^^a", r"a$$", r"(?m)^$^$",
    r"", r"(?i)", r"(?:)", r"a?", r"a*", r"\b", r"^", r"$", r"(?=a)", r"\K", r"a\Kb", r"(?<=\K)a",
]


def sample_text(n, seed):
    rng = random.Random(seed)
    words = "the quick brown fox jumps over lazy dog error warning value index count buffer size user name path file line foo bar baz helloWorld snake_case_name MAX_SIZE main init parse_config".split()
    parts = []
    def w(): return rng.choice(words)
    gens = [
        lambda: "2026-%02d-%02d %02d:%02d:%02d [%s] %s: %s %s pid=%d user=%s status=%d took %d ms" % (rng.randint(1, 12), rng.randint(1, 28), rng.randint(0, 23), rng.randint(0, 59), rng.randint(0, 59), rng.choice(["ERROR", "WARN", "INFO", "DEBUG"]), w(), w(), w(), rng.randint(1, 99999), w(), rng.choice([200, 404, 500, 301]), rng.randint(0, 5000)),
        lambda: "%d.%d.%d.%d - - [10/Oct/2026:13:55:36 +0000] \"%s /%s/%s.html HTTP/1.1\" %d %d" % (rng.randint(1, 255), rng.randint(0, 255), rng.randint(0, 255), rng.randint(0, 300), rng.choice(["GET", "POST", "PUT"]), w(), w(), rng.choice([200, 404, 503]), rng.randint(0, 99999)),
        lambda: "    if (%s == %d) { return %s(%s, %s); } // TODO fix %s" % (w(), rng.randint(0, 99), w(), w(), w(), w()),
        lambda: "for (int i = 0; i < %s; i++) %s[i] = %s->%s + 0x%x;" % (w(), w(), w(), w(), rng.randint(0, 1 << 30)),
        lambda: "#include <%s.h>\n#define %s %d\nstatic int %s(struct %s *p, const char *%s) {" % (w(), w().upper(), rng.randint(0, 999), w(), w(), w()),
        lambda: "def %s(self, %s):\n    import %s\n    self.%s = %s.%s(\"%s\")  # FIXME" % (w(), w(), w(), w(), w(), w(), w()),
        lambda: "{\"id\": %d, \"name\": \"%s\", \"email\": \"%s.%s@%s.com\", \"url\": \"https://www.%s.org/%s?q=%s\", \"price\": %d.%02d}" % (rng.randint(1, 99999), w(), w(), w(), w(), w(), w(), w(), rng.randint(0, 999), rng.randint(0, 99)),
        lambda: "From: %s <%s@%s.net>\nDate: Mon, %d Jan 2026 10:%02d:00 +0100\nSubject: %s %s the the %s" % (w(), w(), w(), rng.randint(1, 28), rng.randint(0, 59), w(), w(), w()),
        lambda: "uuid=%08x-%04x-%04x-%04x-%012x sha=%040x key=AKIA%s api_key = sk_%s $%d.%02d %d%% v%d.%d.%d /usr/lib/%s/lib%s.so.%d" % (rng.getrandbits(32), rng.getrandbits(16), rng.getrandbits(16), rng.getrandbits(16), rng.getrandbits(48), rng.getrandbits(160), "".join(rng.choice("ABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789") for _ in range(16)), w(), rng.randint(0, 999), rng.randint(0, 99), rng.randint(0, 100), rng.randint(0, 9), rng.randint(0, 20), rng.randint(0, 99), w(), w(), rng.randint(1, 9)),
        lambda: "<div class=\"%s\"><a href=\"/%s/%s\">%s &amp; %s</a></div> <!-- %s -->" % (w(), w(), w(), w(), w(), w()),
        lambda: "%s   %s\t%s  \n\n# %s\n- %s: %s\n---" % (w(), w(), w(), w(), w(), w()),
        lambda: "Traceback (most recent call last):\n  File \"/home/%s/%s.py\", line %d, in %s\n    at com.%s.%s(Main.java:%d)\nValueError: %s timed out, connection refused" % (w(), w(), rng.randint(1, 999), w(), w(), w(), rng.randint(1, 99), w()),
        lambda: "Hello World Foo Bar  (x) (y(z)) a@b.co 555-1234 (212) 555-0199 192.168.1.%d 2001:db8::1 00:1A:2b:3C:4d:5E C:\\Users\\%s\\file.txt ~/src/%s ../%s 1,234,567 12:34:56" % (rng.randint(0, 255), w(), w(), w()),
    ]
    size = 0
    while size < n:
        s = rng.choice(gens)() + "\n"
        parts.append(s)
        size += len(s)
    return np.frombuffer("".join(parts).encode("latin-1")[:n], np.uint8).copy()


def main():
    L = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    L.oracle_minlen.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
    L.oracle_scan_chunk.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_longlong, C.c_uint, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.oracle_free.argtypes = [C.c_void_p]
    L.oracle_resource_errors.restype = C.c_long
    texts = [("everyday text", sample_text(int(sys.argv[1]) if len(sys.argv) > 1 else 400_000, 7)), ("8d corpus", synth.text(300_000, 3))]
    tally = {"patterns": 0, "pcre_rejects": 0, "refused": 0, "accepted": 0, "minlen_differs": 0, "compared": 0, "different": 0, "skipped_resource": 0}
    for pat in ZOO:
        tally["patterns"] += 1
        rec = {"pattern": pat}
        pb = pat.encode("latin-1")
        ml = C.c_int(-9)
        if L.oracle_minlen(pb, C.byref(ml)) != 0:
            rec["pcre"] = "rejects the pattern"
            tally["pcre_rejects"] += 1
            try:
                engine.Database(pat)
                rec["product"] = "ACCEPTS what pcre rejects"
                tally["different"] += 1
            except ValueError:
                pass
            print(json.dumps(rec), flush=True)
            continue
        try:
            db = engine.Database(pat)
        except ValueError as ex:
            rec["refused"] = str(ex)[:160]
            tally["refused"] += 1
            print(json.dumps(rec), flush=True)
            continue
        tally["accepted"] += 1
        info = db.info
        rec.update({"tier": info.tier, "resolve": info.resolve, "exact": info.exact, "vm": info.vm, "minlen": db.minlen, "pcre_minlen": ml.value})
        if db.minlen != ml.value:
            tally["minlen_differs"] += 1
        bad = []
        for tname, data in texts:
            if db.minlen < 0:
                continue
            if info.resolve:
                starts, ends = inputs.resolved_list(db, data)
            else:
                starts, ends = inputs.engine_list(db, data), None
            for flags in (3, 1, 0):
                e0, g0 = L.oracle_resource_errors(), engine.resource_errors()
                out, n = C.c_void_p(), C.c_size_t()
                want = b""
                if ml.value <= data.size:
                    assert L.oracle_scan_chunk(pb, b"", data.ctypes.data, data.size, 0, flags, C.byref(out), C.byref(n)) == 0
                    want = C.string_at(out, n.value)
                    L.oracle_free(out)
                got = filegrep.report_chunk(db, flags, b"", data, 0, starts, ends=ends) if db.minlen <= data.size else b""
                if L.oracle_resource_errors() != e0 or engine.resource_errors() != g0:
                    tally["skipped_resource"] += 1
                    continue
                tally["compared"] += 1
                if got != want:
                    bad.append({"text": tname, "flags": flags, "got_lines": got.count(b"\n"), "want_lines": want.count(b"\n")})
            rec.setdefault("lines", {})[tname] = int(len(starts))
        if bad:
            rec["DIFFERENT"] = bad
            tally["different"] += 1
        print(json.dumps(rec), flush=True)
    print(json.dumps({"summary": tally}))


if __name__ == "__main__":
    main()
