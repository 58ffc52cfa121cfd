"""scan_oracle.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement (numpy + Python's `re`) of stealth/grab's per-file match loop, independent
of both libpcre and of the product's own pattern compiler.  Only tests/, the smoke test and
bench.py's checker may import it; nothing under grab_amd/ does.

Restated (citations relative to /root/reference):
  chunk geometry ............ src/grab.cc:151-159, src/grab.h:48, src/main.cc:114,131-136,172-173
  inner loop / advance ...... src/grab.cc:171-213
  offset + line extents ..... src/grab.cc:182-207
  per-chunk flush, -s ....... src/grab.cc:217-234
  minlen skip ............... src/grab.cc:133-135

The regex arithmetic is libpcre's (third-party, not in the reference tree; 8.39/8.45 in this
image).  For the pattern subset the engine accepts -- single-byte atoms with fixed counts
plus one greedy tail -- Python's `re` on bytes has the same semantics (leftmost, greedy,
ASCII \\d\\w\\s, '.' excludes LF), so `re` stands in for pcre_exec here.  This module is
pinned against the reference binary's outputs frozen in tests/golden/ (test_oracle.py).
"""
import re

import numpy as np

OVERLAP = 0x1000          # grab.cc:151
DEFAULT_CHUNK = 1 << 30   # grab.h:48
CONTEXT = 511             # grab.cc:173: char before[512], after[512]

F_OFFSETS, F_NOLINE, F_SINGLE, F_PREFIX, F_COLOR = 1, 2, 4, 8, 16


def chunk_size(n_L=0, cores=0):
    """main.cc:114,131-136,172-173: 1 GiB, halved per -L down to 32 MiB, quartered when -n > 1."""
    c = 1 << 30
    for _ in range(n_L):
        c = max(c >> 1, 1 << 25)
    if cores > 1:
        c >>= 2
    return c


def chunks(size, chunk=DEFAULT_CHUNK):
    """grab.cc:154-159: (offset, length) of every window."""
    out = []
    off = 0
    while off < size:
        out.append((off, min(chunk, size - off)))
        off += chunk - OVERLAP
    return out


def compile_bytes(pattern):
    if isinstance(pattern, str):
        pattern = pattern.encode("latin-1")
    return re.compile(pattern)


def py_minlen(pattern):
    """Minimum match length via sre's own analysis (== PCRE_INFO_MINLENGTH on the subset); -1 if 0."""
    try:
        import re._parser as sp  # py3.11+
    except ImportError:  # pragma: no cover
        import sre_parse as sp
    if isinstance(pattern, str):
        pattern = pattern.encode("latin-1")
    lo, _ = sp.parse(pattern).getwidth()
    return lo if lo > 0 else -1


def all_starts(pattern, data):
    """Every p such that the pattern matches AT p (subject taken to start at p, quirk Q4)."""
    rx = re.compile(b"(?=(?:" + (pattern if isinstance(pattern, bytes) else pattern.encode("latin-1")) + b"))")
    return np.fromiter((m.start() for m in rx.finditer(bytes(data))), dtype=np.int64)


def window_starts(data, tables):
    """Vectorised candidate set for a fixed window: tables[i][b] says byte b may sit at position i."""
    buf = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else data
    m = len(tables)
    n = buf.size - m + 1
    if n <= 0:
        return np.zeros(0, np.int64)
    ok = np.ones(n, bool)
    for i, t in enumerate(tables):
        ok &= np.asarray(t, bool)[buf[i:i + n]]
    return np.flatnonzero(ok).astype(np.int64)


def group_starts(cands):
    """Candidates whose predecessor offset is not a candidate (what the engine must at least report)."""
    c = np.asarray(cands, np.int64)
    if c.size == 0:
        return c
    keep = np.ones(c.size, bool)
    keep[1:] = c[1:] != c[:-1] + 1
    return c[keep]


def check_reported(got, cands):
    """The engine's contract: every group start is reported, and nothing but candidates."""
    got = np.asarray(got, np.int64)
    cands = np.asarray(cands, np.int64)
    if got.size and (np.any(np.diff(got) <= 0) or not np.all(np.isin(got, cands))):
        return False
    return bool(np.all(np.isin(group_starts(cands), got)))


def scan_chunk(rx, minlen, flags, path, content, off):
    """grab.cc:171-213 for one chunk; returns the bytes the reference appends to its ostringstream.
    Every search runs on content[s:] -- the subject STARTS at the restart position (grab.cc:178 passes subject =
    start, startoffset 0), which is what ^ \\b \\B and look-behind see (quirk Q4); a zero-copy memoryview slice."""
    out = []
    clen = len(content)
    view = memoryview(content)
    s = 0
    while s + minlen < clen:  # strict (Q3)
        m = rx.search(view[s:])
        if m is None or m.lastindex:  # a set capturing group: pcre_exec returns 0 with ovector[3] (grab.cc:171,179, quirk Q5)
            break
        b, e = s + m.start(), s + m.end()
        if flags & F_PREFIX:
            out.append(path + b":")
        if flags & F_OFFSETS:
            out.append(b"Match at offset %d\n" % (off + b))
        a = 0
        if not flags & F_NOLINE:
            lo = b
            while lo - 1 >= s and content[lo - 1] != 0x0A and b - lo < CONTEXT:
                lo -= 1
            while e + a < clen and content[e + a] != 0x0A and a < CONTEXT:
                a += 1
            out.append(content[lo:b])
            if flags & F_COLOR:
                out.append(b"\x1b[7m")
            out.append(content[b:e])
            if flags & F_COLOR:
                out.append(b"\x1b[27m")
            out.append(content[e:e + a] + b"\n")
        elif not flags & F_OFFSETS:
            out.append(b"matches\n")
            break
        s = e + a
        if flags & F_SINGLE:
            break
    return b"".join(out)


def grab_file(pattern, data, flags=0, chunk=DEFAULT_CHUNK, path=b"", minlen=None):
    """grab.cc:131-239 over an in-memory file: everything the reference writes to stdout for it."""
    rx = compile_bytes(pattern)
    if minlen is None:
        minlen = py_minlen(pattern)
    size = len(data)
    if minlen < 0 or minlen > size:  # (size_t)-1 > size: every file skipped (Q2)
        return b""
    data = bytes(data)
    res = []
    for off, clen in chunks(size, chunk):
        text = scan_chunk(rx, minlen, flags, path, data[off:off + clen], off)
        if text:
            res.append(text)
            if flags & F_SINGLE:
                break
    return b"".join(res)


def offsets_nl(pattern, data, chunk=DEFAULT_CHUNK):
    """The `-O -l` offsets (the configuration BASELINE's parity cases use), as an int64 array."""
    rx = compile_bytes(pattern)
    minlen = py_minlen(pattern)
    size = len(data)
    if minlen < 0 or minlen > size:
        return np.zeros(0, np.int64)
    data = bytes(data)
    out = []
    for off, clen in chunks(size, chunk):
        s = 0
        view = memoryview(data)[off:off + clen]
        while s + minlen < clen:
            m = rx.search(view[s:])  # the subject starts at the restart position (Q4)
            if m is None or m.lastindex:
                break
            out.append(off + s + m.start())
            s += m.end()
    return np.asarray(out, np.int64)
