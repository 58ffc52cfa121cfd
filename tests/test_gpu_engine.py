"""GPU parity, engine level: candidate starts produced by the HIP kernels (through the C ABI, include/gscan.h) on seeded
inputs -- bit-exact, every kernel variant, ragged sizes, tile/wave boundaries, dense outputs, multi-segment arenas.
Two checkers: libpcre itself (liboracle.oracle_all_starts: test_kernels_against_libpcre, one case per kernel form) and, for
the bulk of the size / variant matrix, the candidate set evaluated from the database's class tables (table_candidates)."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT
from grab_amd import engine, synth
from inputs import resolved_list, db_candidates, engine_list

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import scan_oracle as so  # noqa: E402

pytestmark = pytest.mark.gpu

VARIANTS = [0, 1, 2, 4, 5, 6, 13, 14, 38]  # KiB per wave {16, 8, 12} x nontemporal loads {off, on}; 13: 768-thread table kernels; 14: table kernels prefetch the next tile; 38 (the default): 6 + K2's lane-table form
DEFAULT_VARIANT = 38
PATTERNS = ["foobardoesnotexist", "foo", "e", "xy", "[A-Za-z_][A-Za-z0-9_]{15,}", "[a-z]{2,5}", "abc[0-9]*", r"\d{3}-\d{4}",
            "[Ll]inus", "a.c", "[^x]{5,}", "[0-9a-f]{32}", "[0-9A-F]{6}[a-z]", "e+", r"\w\s\w\s\w", "[a-z][0-9][A-Z][.,][;:]q",
            "[ab][cd][ef][gh]{20}", "[0-9]{17}", "[0-9]{18}", "[a-z_]{49}",
            # K3 (bucket filter): alternations, optional / bounded repeats in the middle, (?i), > 4 classes
            "foo|bar", "foobardoesnotexist|Linus|555-1234", "colou?r|axc", "(?i)linus", "[ab]{1,3}c", "(?:foo|bar)baz?",
            "a|ab", "(?:a|b|c|d|e|f|x|y|0|1)x", "[a-z][0-9][A-Z][.,][;:]", "(?i)foobar|k7Q,;q|[0-9]{12}x?", "e|ee|eee",
            "(?:ab|cd|ef|gh|ij|kl|mn|op){2}", "a 1 b|ABCDEF012x|acegg+", "[0-9a-f]{30}(?:ab|cd)",
            # context positions (\b ^ $ ...): device windows carry the byte before / after the match
            r"\bfoo\b", r"\Bfoo", "(?m)^[a-z]{3}", "(?m)[a-z]{3}$", r"\b\w+\b", r"\b[a-z.]o|Linus$|^abc", r"\bfoobardoesnotexist\b", r"e\B",
            # gapped alternatives: the device window is one repeat byte + the rest
            "e+f", r"[0-9]+\.[0-9]+", "foo.*bar", r"[a-z]+\b", r"\w+@\w", "a 1.*c",
            # inexact: the alternatives are what a match must begin with (the host matcher confirms); the kernels' job is the same
            r"\w+@\w+\.com", "(?:foo|bar)+baz", "a.*b.*c", "(?:ab|cd)+?e", "e++f", "[0-9]{1,40}x"]


@pytest.fixture(scope="module")
def ctx(built):
    c = engine.Context(0, 1 << 30)
    yield c
    c.close()


def table_candidates(db, data):
    """Every candidate offset according to the DATABASE'S OWN class tables (numpy evaluation of gscan_db_dev_window): what the
    kernels are asked to find.  This checks the kernels against the compiler's output -- fast enough for every variant and
    size -- and is blind to a compiler bug by construction; test_kernels_against_libpcre below and the CLI-level tests
    (reference-generated goldens, liboracle side by side) are what pin the compiler."""
    return db_candidates(db, data)


def pcre_starts(liboracle, pattern, data):
    """Every offset p at which libpcre matches with the subject starting at p (oracle_all_starts: pcre_exec ANCHORED,
    /root/reference/src/grab.cc:178 semantics) -- the candidate set by the REFERENCE's definition, no product code involved."""
    import ctypes as C

    buf = np.ascontiguousarray(data)
    cap = buf.size + 1
    out = np.zeros(cap, np.uint32)
    n = liboracle.oracle_all_starts(pattern.encode("latin-1"), buf.ctypes.data, buf.size, out.ctypes.data, None, cap)
    assert 0 <= n <= cap
    return out[:n].astype(np.int64)


def same(got, want):
    """The engine's contract (include/gscan.h, gscan_wait): ascending, candidates only, and every start
    of a group of consecutive candidates present."""
    return so.check_reported(got, want)


def as_specified(db, got, data):
    """What ctx.scan() has to return for `data`: the contract above against the candidate set -- or, for a database whose
    candidates the device confirms itself (info.vm), EXACTLY the device hits its VM filter keeps, no more, no fewer
    (tests/inputs.py engine_list runs the kernel's cold path, same source, on the host)."""
    if db.info.resolve:  # the device settles the matches: EXACTLY the offsets at which the pattern's VM program finds one (or gives up), run on the host here
        return np.array_equal(np.asarray(got, np.uint32), resolved_list(db, np.ascontiguousarray(data))[0])
    if db.info.vm:
        return np.array_equal(np.asarray(got, np.uint32), engine_list(db, np.ascontiguousarray(data)))
    return same(got, table_candidates(db, data))


def sample(n, seed):
    """Text with every test pattern planted a few times, also across 1 KiB / 16 KiB / 64 KiB boundaries."""
    rng = np.random.default_rng(seed)
    buf = synth.text(n, seed % 1000)
    plants = [b"foobardoesnotexist", b"foo", b"Linus", b"linus", b"555-1234", b"abc0123456789", b"a\nc", b"axc",
              b"0123456789abcdef0123456789abcdef", b"ABCDEF012x", b"a 1 b 2 c", b"k7Q,;q", b"acegggggggggggggggggggggggg",
              b"12345678901234567890", b"abcdefghijklmnopqrstuvwxyz_abcdefghijklmnopqrstuvwxyz", b"eeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeeee",
              b"0123456789abcdef0123456789abcdabcd", b"a7Q.;x5R,:", b"abefghghghghghghghghghghghghgh", b"ijopabklmn"]
    if n > 400:
        spots = list(rng.integers(0, n - 64, 40)) + [b - d for b in (1024, 4096, 16384, 32768, 65536, 131072) for d in (1, 3, 9, 17, 30) if b < n - 64]
        for i, at in enumerate(spots):
            p = plants[i % len(plants)]
            at = int(max(0, min(at, n - len(p))))
            buf[at:at + len(p)] = np.frombuffer(p, np.uint8)
    return buf


# one pattern per kernel form (exact patterns without context or capturing groups: for those the candidate set IS "libpcre
# matches at p"), checked against libpcre directly
KERNEL_FORMS = [
    ("K1 literal", "foobardoesnotexist", engine.TIER_LITERAL),
    ("K1 class sequence with an anchor", "[Ll]inus", engine.TIER_LITERAL),
    ("K2 lane form, two runs", "[A-Za-z_][A-Za-z0-9_]{15,}", engine.TIER_CLASSRUN),
    ("K2 lane form, one run", "[0-9]{17}", engine.TIER_CLASSRUN),
    ("K2 lane form, three runs", "[a-z][0-9][a-z]{9}", engine.TIER_CLASSRUN),
    ("K2 lane form, 3 classes", "[a-z][0-9][A-Z]{3}", engine.TIER_CLASSRUN),
    ("K2 lane form, 4 classes", "[a-z][0-9][A-Z]{2}[.,;]", engine.TIER_CLASSRUN),
    ("K2 pair form, wide", "[0-9a-f]{32}", engine.TIER_CLASSRUN),
    ("K2 general form, wide", "[ab][cd][ef][gh]{20}", engine.TIER_CLASSRUN),
    ("K3, tables exact", "foobardoesnotexist|Linus|555-1234", engine.TIER_BUCKET),
    ("K3, three filter positions", "foo|bar", engine.TIER_BUCKET),
    ("K3 + settle: shared buckets", "(?:ab|cd|ef|gh|ij|kl|mn|op){2}", engine.TIER_BUCKET),
    ("K3 + settle: windows beyond the confirm tables", "[0-9a-f]{30}(?:ab|cd)", engine.TIER_BUCKET),
    ("K3, > 4 classes", "[a-z][0-9][A-Z][.,][;:]", engine.TIER_BUCKET),
]


@pytest.mark.parametrize("name,pattern,tier", KERNEL_FORMS, ids=[k[0] for k in KERNEL_FORMS])
def test_kernels_against_libpcre(ctx, liboracle, name, pattern, tier):
    """ctx.scan() -- kernels through the C ABI -- against oracle_all_starts() (libpcre, the reference's engine): no table of
    the product's compiler sits between the two."""
    db = engine.Database(pattern)
    assert db.info.tier == tier and db.info.exact and not db.info.has_context, (name, db.info.tier)
    for seed, n in ((1, 300_007), (7, 65_536 + 17), (9, 1_200_001)):
        data = sample(n, seed)
        want = pcre_starts(liboracle, pattern, data)
        assert want.size > 0, "the sample holds matches of every form"
        got = ctx.scan(db, data)
        assert same(got, want), (name, n, len(got), len(want))
        assert np.array_equal(want, table_candidates(db, data)), "the compiler's tables and libpcre agree as well"


_PCRE_WANT = {}


@pytest.mark.parametrize("variant", VARIANTS)
def test_parity_patterns(ctx, liboracle, variant):
    """Every kernel variant, every pattern: against the candidate set from the database's tables AND, wherever the candidate
    set is by definition "libpcre matches at p" (exact patterns without context), against libpcre itself
    (oracle_all_starts) -- no product table between the kernel and the reference's engine for any variant.  Databases the
    device confirms itself (info.vm): everything libpcre matches is kept, nothing outside the filter's hits is reported.
    (Gapped alternatives -- a+b, foo.*bar -- are listed by where the part behind the repeat begins, not by match starts:
    for them the table check above and the CLI-level differentials are the checks.)"""
    ctx.set_option("variant", variant)
    data = sample(300_007, 1)
    for pattern in PATTERNS:
        db = engine.Database(pattern)
        got = ctx.scan(db, data)
        assert got.dtype == np.uint32
        assert as_specified(db, got, data), (pattern, variant, len(got))
        if not db.info.has_context and not db.info.gapped and (db.info.exact or db.info.vm) and "(" not in pattern.replace("(?:", "").replace("(?i)", ""):
            if pattern not in _PCRE_WANT:
                _PCRE_WANT[pattern] = pcre_starts(liboracle, pattern, data)
            want = _PCRE_WANT[pattern]
            if db.info.exact:
                assert same(got, want), (pattern, variant, len(got), len(want))
            else:
                g = np.asarray(got, np.int64)
                assert np.all(np.isin(want, g)), (pattern, variant, "a match libpcre finds is missing from the device's list")
                assert np.all(np.isin(g, table_candidates(db, data))), (pattern, variant)
    ctx.set_option("variant", DEFAULT_VARIANT)


@pytest.mark.parametrize("depth", [3, 4])
def test_k3_filter_depths(ctx, liboracle, depth):
    """K3 with three or four filter positions whatever the compiler would pick (option "k3_depth"): the filter only decides what
    reaches the confirm tables, so both depths list the same records -- against libpcre for the exact forms, against the
    compiler's tables for all, ragged sizes included.  (Two positions: built, green under this test and slower -- session AA,
    profiles/r06_aa_*.)"""
    ctx.set_option("k3_depth", depth)
    try:
        for name, pattern, tier in KERNEL_FORMS:
            if tier != engine.TIER_BUCKET:
                continue
            db = engine.Database(pattern)
            for seed, n in ((1, 300_007), (7, 65_536 + 17), (3, 12_289), (5, 1)):
                data = sample(n, seed)
                got = ctx.scan(db, data)
                assert same(got, pcre_starts(liboracle, pattern, data)), (name, depth, n, len(got))
        data = sample(300_007, 1)
        for pattern in PATTERNS:
            db = engine.Database(pattern)
            if db.info.tier == engine.TIER_BUCKET:
                assert as_specified(db, ctx.scan(db, data), data), (pattern, depth)
    finally:
        ctx.set_option("k3_depth", 0)
    with pytest.raises(engine.EngineError):
        ctx.set_option("k3_depth", 2)


def runs_text(n, seed):
    """Digits, lower-case letters and a few others in runs of every length from 1 to 40: what the run programs of the
    lane-table kernels (1..5 doubling steps, one or two runs) have to tell apart."""
    rng = np.random.default_rng(seed)
    out = np.empty(n + 64, np.uint8)
    kinds = [np.frombuffer(b"0123456789", np.uint8), np.frombuffer(b"abcxyz", np.uint8), np.frombuffer(b" _.\n", np.uint8)]
    at = 0
    while at < n:
        k = kinds[int(rng.integers(0, 3))]
        ln = int(rng.integers(1, 41))
        out[at:at + ln] = k[rng.integers(0, k.size, ln)]
        at += ln
    return out[:n]


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 7, 8, 9, 12, 15, 16, 17])
def test_lane_run_programs_against_libpcre(ctx, liboracle, n):
    """One instantiation of the lane-table kernel per (number of runs, doubling steps per run): a run of n digits alone
    (n >= 2), in front of a letter, behind one, and between two runs of letters -- every step count 0..5, both orders of
    the two-run programs, the generic three-run form -- against libpcre."""
    data = runs_text(400_003, 100 + n)
    pats = ["[0-9]{%d}[a-z]" % n, "[a-z][0-9]{%d}" % n]
    if n >= 2:
        pats.append("[0-9]{%d}" % n)
    if n <= 13:
        pats += ["[a-z]{2}[0-9]{%d}[a-z]{2}" % n, "[a-z]{%d}[0-9]{%d}" % (min(n, 17 - n), n) if n <= 8 else "[a-z]{3}[0-9]{%d}" % n]
    if n <= 14:
        pats.append("[a-z][0-9]{%d}[_ .]" % n)  # three classes: the four-class table layout
    for pattern in pats:
        db = engine.Database(pattern)
        if db.info.tier != engine.TIER_CLASSRUN:
            continue
        want = pcre_starts(liboracle, pattern, data)
        assert want.size > 0, pattern
        for variant in (DEFAULT_VARIANT, 6):
            ctx.set_option("variant", variant)
            got = ctx.scan(db, data)
            assert same(got, want), (pattern, variant, len(got), len(want))
    ctx.set_option("variant", DEFAULT_VARIANT)


SIZES = [0, 1, 2, 3, 4, 5, 15, 16, 17, 18, 19, 31, 32, 33, 63, 64, 65, 1023, 1024, 1025, 1039, 1040, 4095, 4096, 4097,
         16383, 16384, 16385, 16400, 32767, 32768, 32769, 65535, 65536, 65537, 65553, 131071, 131072, 131073, 200000]


_RAGGED_WANT = {}  # libpcre's answers, shared by the variants


@pytest.mark.parametrize("variant", [0, 1, 2, DEFAULT_VARIANT])
def test_ragged_sizes(ctx, liboracle, variant):
    """Every length around lane / wave-step / sub-tile / tile boundaries; matches planted at the very end (an identifier
    and a digit run too: the lane-table form's tail masking and halo see every ragged end).  Against the database's tables
    and -- all six patterns are exact and context-free -- against libpcre itself."""
    ctx.set_option("variant", variant)
    pats = ["foo", "foobardoesnotexist", "[a-z]{2,5}", "[A-Za-z_][A-Za-z0-9_]{15,}", "[0-9a-f]{32}", "e+", "[0-9]{17}", "[a-z][0-9][A-Z]{3}"]
    dbs = [engine.Database(p) for p in pats]
    base = sample(200_000, 2)
    for n in SIZES:
        data = base[:n].copy()
        if n >= 40:
            data[n - 3:] = np.frombuffer(b"foo", np.uint8)
            data[n - 40:n - 22] = np.frombuffer(b"foobardoesnotexist", np.uint8)
        if n >= 120:
            data[n - 64:n - 47] = np.frombuffer(b"01234567890123456", np.uint8)  # exactly 17 digits ...
            data[n - 47] = ord(" ")
            data[n - 100:n - 95] = np.frombuffer(b"q7XYZ", np.uint8)
        for db, p in zip(dbs, pats):
            # (shorter than the window: the host never submits it; the engine must still answer "nothing")
            got = ctx.scan(db, data)
            want = table_candidates(db, data)
            assert same(got, want), (p, n, variant)
            if n:
                if (p, n, 0) not in _RAGGED_WANT:
                    _RAGGED_WANT[(p, n, 0)] = pcre_starts(liboracle, p, data)
                assert same(got, _RAGGED_WANT[(p, n, 0)]), (p, n, variant, "libpcre")
        if n >= 64:  # the window that ENDS with the chunk: an identifier / a digit run in the last bytes
            tail = data.copy()
            tail[n - 17:] = np.frombuffer(b"01234567890123456", np.uint8)
            tail[n - 18] = ord(" ")
            for db, p in zip(dbs, pats):
                got = ctx.scan(db, tail)
                if (p, n, 1) not in _RAGGED_WANT:
                    _RAGGED_WANT[(p, n, 1)] = pcre_starts(liboracle, p, tail)
                assert same(got, _RAGGED_WANT[(p, n, 1)]), (p, n, variant, "tail")
    ctx.set_option("variant", DEFAULT_VARIANT)


def test_dense_output_and_regrow(ctx):
    """Every position matches: 4 bytes of output per input byte, record buffer overflow -> regrow -> rescan."""
    n = 3_000_001
    data = np.full(n, ord("a"), np.uint8)
    for pattern in ["a", "aa", "[a-z]{3}", "a+"]:
        db = engine.Database(pattern)
        got = ctx.scan(db, data)
        assert got[0] == 0 and same(got, np.arange(n - db.minlen + 1)), pattern
        assert len(got) <= n // 16 + 1, "one group: at most one repeat per lane (K1) / per sub-tile (K2)"
    # dense and NOT compressible: every other position starts a group -> record buffer overflow -> regrow -> rescan
    data = np.frombuffer(b"ab", np.uint8)[np.arange(n) % 2].copy()
    db = engine.Database("a")
    got = ctx.scan(db, data)
    assert np.array_equal(got, np.arange(0, n, 2, dtype=np.uint32))
    data[::7] = ord("\n")
    db = engine.Database("[^\\n]{3}")
    assert same(ctx.scan(db, data), table_candidates(db, data))


def test_adversarial_anchor(ctx):
    """The anchor hits everywhere but the window rarely matches (K1 verify path on every lane)."""
    n = 500_000
    data = np.frombuffer(b"abcd", np.uint8)[np.arange(n) % 4].copy()
    data[123456:123464] = np.frombuffer(b"abcdabcX", np.uint8)
    for pattern in ["abcdabcX", "bcdabcX", "[ab]bcdabcX"]:
        db = engine.Database(pattern)
        assert same(ctx.scan(db, data), table_candidates(db, data)), pattern


def test_binary_data(ctx):
    rng = np.random.default_rng(5)
    data = rng.integers(0, 256, 1_000_003, dtype=np.uint8)
    data[5000:5004] = [0, 255, 0, 255]
    for pattern in [r"\x00\xff\x00\xff", r"[\x80-\xff]{6}", r"\x00[^\x00]{3}\x00", r"[\x00-\x1f][\x7f-\xff]{2,}"]:
        db = engine.Database(pattern)
        assert same(ctx.scan(db, data), table_candidates(db, data)), pattern


def test_pipelined_submits(ctx):
    """GSCAN_SLOTS = 3 chunks in flight come back in submission order with their own results."""
    a, b, c3 = sample(1 << 20, 3), sample((1 << 20) + 77, 4), sample((1 << 19) + 5, 5)
    db = engine.Database("[a-z]{2,5}")
    ctx.submit(db, a, tag=11)
    ctx.submit(db, b, tag=22)
    ctx.submit(db, c3, tag=33)
    with pytest.raises(engine.EngineError):
        ctx.submit(db, a, tag=44)  # GSCAN_EBUSY: every slot in flight
    t1, s1 = ctx.wait()
    t2, s2 = ctx.wait()
    t3, s3 = ctx.wait()
    assert (t1, t2, t3) == (11, 22, 33)
    assert same(s1, table_candidates(db, a))
    assert same(s2, table_candidates(db, b))
    assert same(s3, table_candidates(db, c3))
    with pytest.raises(engine.EngineError):
        ctx.wait()  # GSCAN_EEMPTY


def test_pattern_switch(ctx):
    """Alternating databases on one context re-uploads the program each time."""
    data = sample(100_000, 6)
    dbs = [engine.Database(p) for p in ("foo", "[a-z]{2,5}", "foobardoesnotexist", "e+")]
    for _ in range(3):
        for db in dbs:
            assert same(ctx.scan(db, data), table_candidates(db, data))


def test_device_resident_segments(ctx, liboracle):
    """gscan_scan_device over an arena of ragged, 16-byte aligned segments == per-segment oracle (the path bench.py times):
    every variant incl. the shipped one, against the tables and against libpcre."""
    import torch

    lens = [0, 5, 17, 1000, 65536, 65537, 70001, 300000, 16, 131072, 1]
    offs, pos = [], 0
    for ln in lens:
        offs.append(pos)
        pos += (ln + 15) // 16 * 16 + 16 * (len(offs) % 3)
    host = sample(pos + 64, 9)
    host[offs[5] + 65536 - 2: offs[5] + 65536 + 1] = np.frombuffer(b"foo", np.uint8)
    arena = torch.from_numpy(host).cuda()
    segs = list(zip(offs, lens))
    ctx.set_capacity(1 << 20)
    for pattern in ["foo", "[a-z]{2,5}", "[A-Za-z_][A-Za-z0-9_]{15,}", "[0-9]{17}", "[a-z][0-9][A-Z]{3}", "foobardoesnotexist|Linus|555-1234"]:
        db = engine.Database(pattern)
        wants = [pcre_starts(liboracle, pattern, host[o:o + ln]) if ln else np.zeros(0, np.int64) for o, ln in segs]
        for variant in (0, 1, 2, DEFAULT_VARIANT):
            ctx.set_option("variant", variant)
            res = ctx.scan_device(db, arena.data_ptr(), segs)
            total, overflow = ctx.dev_sync(res)
            assert not overflow
            n = 0
            for i, (o, ln) in enumerate(segs):
                got = ctx.dev_fetch(res, i)
                want = table_candidates(db, host[o:o + ln])
                assert same(got, want), (pattern, variant, i)
                assert same(got, wants[i]), (pattern, variant, i, "libpcre")
                n += len(got)
            assert n == total
    ctx.set_option("variant", DEFAULT_VARIANT)
    ms, launches = ctx.kernel_time()
    assert launches > 0 and ms > 0
    # a deliberately small record buffer reports overflow instead of writing out of bounds
    ctx.set_capacity(64)
    db = engine.Database("[a-z]{2,5}")
    res = ctx.scan_device(db, arena.data_ptr(), segs)
    total, overflow = ctx.dev_sync(res)
    assert overflow and total > 64
    ctx.set_capacity(1 << 20)
    res = ctx.scan_device(db, arena.data_ptr(), segs)
    total2, overflow = ctx.dev_sync(res)
    assert not overflow and total2 == total


def test_grid_shapes(ctx):
    """One workgroup per tile vs. persistent grid-stride give identical results."""
    data = sample(5_000_000, 8)
    db = engine.Database("[a-z]{2,5}")
    want = table_candidates(db, data)
    for bpc in (0, 1, 2, 8, 16):
        ctx.set_option("blocks_per_cu", bpc)
        assert same(ctx.scan(db, data), want), bpc
    ctx.set_option("blocks_per_cu", 0)


def test_submit_fd_ranges(ctx, tmp_path):
    """gscan_submit_fd: file ranges read by the engine's reader threads piece by piece (asynchronously: the last reader
    launches the scan) -- sizes around the piece boundary, ranges that start inside the file, three ranges in flight, and
    a range beyond the end of the file (the error surfaces at gscan_wait and leaves the context usable)."""
    blk = engine.lib().gscan_block_size()
    assert blk == engine.ingest_info()["block_bytes"] and blk % (1 << 20) == 0
    data = sample(2 * blk + 4097 + 77, 21)
    for at in (blk - 9, blk - 2, 2 * blk - 5):  # matches straddling piece boundaries
        data[at:at + 18] = np.frombuffer(b"foobardoesnotexist", np.uint8)
    path = tmp_path / "f.bin"
    data.tofile(str(path))
    fd = os.open(str(path), os.O_RDONLY)
    try:
        for pattern in ["foobardoesnotexist", "[a-z]{2,5}", "foo|bar"]:
            db = engine.Database(pattern)
            for off, ln in [(0, data.size), (0, blk), (0, blk + 1), (0, blk - 1), (4096, blk + 5000), (blk - 4096, blk + 4096 + 77),
                            (data.size - 1, 1), (12288, 0), (0, 1), (0, 100)]:
                ctx.submit_fd(db, fd, off, ln, tag=off)
                tag, per_seg, has_content = ctx.wait_segs()
                assert tag == off and len(per_seg) == 1 and not has_content
                assert same(per_seg[0], table_candidates(db, data[off:off + ln])), (pattern, off, ln)
        db = engine.Database("[a-z]{2,5}")
        ctx.submit_fd(db, fd, 0, blk + 123, tag=1)
        ctx.submit_fd(db, fd, 4096, 3 * 4096, tag=2)
        t1, s1, _ = ctx.wait_segs()
        t2, s2, _ = ctx.wait_segs()
        assert (t1, t2) == (1, 2)
        assert same(s1[0], table_candidates(db, data[:blk + 123])) and same(s2[0], table_candidates(db, data[4096:4 * 4096]))
        # GSCAN_SLOTS = 3 ranges in flight, returned in submission order; a fourth is refused until one has been waited for
        for k in range(3):
            ctx.submit_fd(db, fd, k * 4096, 2 * blk - 5000 + k, tag=10 + k)
        with pytest.raises(engine.EngineError, match="no free slot"):
            ctx.submit_fd(db, fd, 0, 100)
        for k in range(3):
            t, s_, _ = ctx.wait_segs()
            assert t == 10 + k and same(s_[0], table_candidates(db, data[k * 4096:k * 4096 + 2 * blk - 5000 + k]))
        # two databases in flight: the device program is one per context -- a range still being read when another database
        # is submitted must be launched with ITS program (the second submit waits for the first range to arrive and launch)
        dba, dbb, dbc = engine.Database("[a-z]{2,5}"), engine.Database("foobardoesnotexist"), engine.Database("[0-9A-F]{6}[a-z]")
        for rep in range(3):
            ctx.submit_fd(dba, fd, 0, data.size, tag=31)
            ctx.submit_fd(dbb, fd, 0, data.size, tag=32)
            ctx.submit_fd(dbc, fd, 4096, blk + 77, tag=33)
            for tag_want, dbx, lo, ln in ((31, dba, 0, data.size), (32, dbb, 0, data.size), (33, dbc, 4096, blk + 77)):
                t, s_, _ = ctx.wait_segs()
                assert t == tag_want and same(s_[0], table_candidates(dbx, data[lo:lo + ln])), (rep, tag_want)
        ctx.submit_fd(db, fd, data.size - 10, 4096)  # beyond the end of the file: queued ...
        ctx.submit_fd(db, fd, 0, 5000, tag=77)
        with pytest.raises(engine.EngineError, match="shrank"):
            ctx.wait_segs()  # ... reported when the range is waited for; the chunk is dropped, its slot is free again
        t, s_, _ = ctx.wait_segs()
        assert t == 77 and same(s_[0], table_candidates(db, data[:5000]))
        ctx.submit_fd(db, fd, 0, 1000)  # the context is still usable
        assert same(ctx.wait_segs()[1][0], table_candidates(db, data[:1000]))
    finally:
        os.close(fd)


def test_submit_batch_segments(ctx, liboracle):
    """gscan_acquire + gscan_submit_segs: many small inputs in one pinned block, one launch; every segment behaves
    like a chunk of its own (matches never cross a segment boundary, ragged and empty segments)."""
    rng = np.random.default_rng(33)
    base = sample(1_200_000, 22)
    lens = [0, 1, 2, 15, 16, 17, 18, 31, 33, 1000, 4095, 4096, 4097, 70001, 0, 300000, 5, 49151, 49152, 49153, 98304, 3]
    parts, pos = [], 0
    for ln in lens:
        parts.append(base[pos:pos + ln].copy())
        pos += ln
    parts[9][-3:] = np.frombuffer(b"foo", np.uint8)   # ends a segment ...
    parts[10][:3] = np.frombuffer(b"bar", np.uint8)   # ... and the next one starts with another word
    parts[12][-2:] = np.frombuffer(b"fo", np.uint8)   # "fo" + "o...": must NOT match across the boundary
    parts[13][0] = ord("o")
    for pattern in ["foo", "foobardoesnotexist", "[a-z]{2,5}", "[A-Za-z_][A-Za-z0-9_]{15,}", "foo|bar", "e+", "[0-9]{17}", "[a-z][0-9][A-Z]{3}"]:
        db = engine.Database(pattern)
        wants = [pcre_starts(liboracle, pattern, part) if len(part) else np.zeros(0, np.int64) for part in parts]
        for variant in (1, 6, DEFAULT_VARIANT):
            ctx.set_option("variant", variant)
            segs = ctx.submit_batch(db, parts, tag=7)
            assert all(o % 16 == 0 for o, _ in segs)
            tag, per_seg, has_content = ctx.wait_segs()
            assert tag == 7 and has_content and len(per_seg) == len(parts)
            for i, (part, got) in enumerate(zip(parts, per_seg)):
                assert same(got, table_candidates(db, part)), (pattern, variant, i, len(part))
                assert same(got, wants[i]), (pattern, variant, i, len(part), "libpcre")
    ctx.set_option("variant", DEFAULT_VARIANT)
    # thousands of tiny segments (more tiles than len / tile size), and an empty batch
    tiny = [base[i * 37:i * 37 + int(rng.integers(0, 37))].copy() for i in range(3000)]
    db = engine.Database("[a-z]{2,5}")
    ctx.submit_batch(db, tiny)
    _, per_seg, _ = ctx.wait_segs()
    assert len(per_seg) == 3000
    for part, got in zip(tiny, per_seg):
        assert same(got, table_candidates(db, part))
    ctx.submit_batch(db, [])
    _, per_seg, _ = ctx.wait_segs()
    assert len(per_seg) == 1 and per_seg[0].size == 0
    # a batch and a file range in flight together come back in submission order
    ctx.submit_batch(db, parts[:5], tag=1)
    ctx.submit(db, base[:5000], tag=2)
    assert ctx.wait_segs()[0] == 1 and ctx.wait()[0] == 2


def test_submit_files(ctx, liboracle, tmp_path):
    """gscan_submit_files: small files handed over by NAME -- the device's reader threads open, read and close them, runs of
    files that fit one staging block travel in one DMA, the last reader launches ONE scan over the segment table.  Ragged
    and empty files, a batch of several pieces (> one block), a file that does not exist and one that is shorter than
    announced in the MIDDLE of a batch (their own errors; every other segment intact), descriptors instead of names, an
    empty batch, a batch beyond max_chunk, and batches in flight together with other submits."""
    import errno

    blk = engine.lib().gscan_block_size()
    rng = np.random.default_rng(77)
    lens = [0, 1, 2, 15, 16, 17, 31, 33, 1000, 4095, 4096, 4097, 70001, 0, 300000, 5, 49151, 49152, 49153, 98304, 3]
    lens += [int(x) for x in rng.integers(200_000, 900_000, 32)]
    base = sample(sum(lens) + 64, 23)
    assert sum(lens) > blk, "the batch spans more than one staging block: several pieces"
    parts, pos = [], 0
    for ln in lens:
        parts.append(base[pos:pos + ln].copy())
        pos += ln
    parts[9][-3:] = np.frombuffer(b"foo", np.uint8)   # ends a file ...
    parts[10][:3] = np.frombuffer(b"bar", np.uint8)   # ... and the next one starts with another word
    parts[11][-2:] = np.frombuffer(b"fo", np.uint8)   # "fo" + "o...": must NOT match across the boundary
    parts[12][0] = ord("o")
    paths = []
    for i, part in enumerate(parts):
        f = tmp_path / ("f%03d.bin" % i)
        part.tofile(str(f))
        paths.append(str(f))
    files = list(zip(paths, lens))
    for pattern in ["foo", "foobardoesnotexist", "[a-z]{2,5}", "[A-Za-z_][A-Za-z0-9_]{15,}", "foo|bar", "[0-9]{17}"]:
        db = engine.Database(pattern)
        wants = [pcre_starts(liboracle, pattern, part) if 0 < len(part) <= 100_000 else None for part in parts]
        for variant in (1, DEFAULT_VARIANT):
            ctx.set_option("variant", variant)
            ctx.submit_files(db, files, tag=5)
            tag, per_seg, has_content = ctx.wait_segs()
            assert tag == 5 and not has_content and len(per_seg) == len(parts)
            assert ctx.last_file_errors() == [0] * len(parts)
            for i, (part, got) in enumerate(zip(parts, per_seg)):
                assert same(got, table_candidates(db, part)), (pattern, variant, i, len(part))
                if wants[i] is not None:
                    assert same(got, wants[i]), (pattern, variant, i, "libpcre")
    ctx.set_option("variant", DEFAULT_VARIANT)
    db = engine.Database("[a-z]{2,5}")
    # a file that is not there, one that is shorter than the walk said, one that is longer (the first `len` bytes count)
    broken = list(files)
    broken[25] = (str(tmp_path / "gone.bin"), 12345)
    broken[30] = (paths[30], lens[30] + 1000)
    broken[31] = (paths[31], lens[31] - 777)
    ctx.submit_files(db, broken, tag=6)
    tag, per_seg, _ = ctx.wait_segs()
    errs = ctx.last_file_errors()
    assert tag == 6 and errs[25] == errno.ENOENT and errs[30] == -1 and sum(1 for e in errs if e) == 2
    for i, (part, got) in enumerate(zip(parts, per_seg)):
        if i == 31:
            assert same(got, table_candidates(db, part[:lens[31] - 777]))
        elif i not in (25, 30):
            assert same(got, table_candidates(db, part)), i
    # open descriptors instead of names (the caller keeps them open until the batch is back)
    fds = [os.open(p, os.O_RDONLY) for p in paths[:12]]
    try:
        ctx.submit_files(db, list(zip(fds, lens[:12])), tag=7)
        tag, per_seg, _ = ctx.wait_segs()
        assert tag == 7 and ctx.last_file_errors() == [0] * 12
        for part, got in zip(parts[:12], per_seg):
            assert same(got, table_candidates(db, part))
    finally:
        for fd in fds:
            os.close(fd)
    # thousands of tiny files, an empty batch, a batch that cannot fit
    tiny = []
    for i in range(1500):
        ln = int(rng.integers(0, 37))
        f = tmp_path / ("t%04d" % i)
        base[i * 37:i * 37 + ln].tofile(str(f))
        tiny.append((str(f), ln))
    ctx.submit_files(db, tiny)
    _, per_seg, _ = ctx.wait_segs()
    assert len(per_seg) == 1500
    for i, got in enumerate(per_seg):
        assert same(got, table_candidates(db, base[i * 37:i * 37 + tiny[i][1]]))
    ctx.submit_files(db, [])
    _, per_seg, _ = ctx.wait_segs()
    assert len(per_seg) == 1 and per_seg[0].size == 0 and ctx.last_file_errors() is None
    with pytest.raises(engine.EngineError):
        ctx.submit_files(db, [(paths[0], blk + 1)])  # larger than a staging block: gscan_submit_fd's business
    # three in flight, of three kinds, back in submission order; the error array belongs to the batch only
    fd = os.open(paths[-1], os.O_RDONLY)
    try:
        ctx.submit_files(db, files[:8], tag=1)
        ctx.submit_fd(db, fd, 0, lens[-1], tag=2)
        ctx.submit(db, base[:5000], tag=3)
        t1, s1, _ = ctx.wait_segs()
        assert t1 == 1 and len(s1) == 8 and ctx.last_file_errors() == [0] * 8
        t2, s2, _ = ctx.wait_segs()
        assert t2 == 2 and ctx.last_file_errors() is None and same(s2[0], table_candidates(db, parts[-1]))
        assert ctx.wait()[0] == 3
    finally:
        os.close(fd)


def test_shared_buckets_second_pass(ctx):
    """More than 8 alternatives share K3's 8 filter buckets: the tables alone would accept cross-products of a bucket's
    alternatives ("alota" from alpha + iota).  The second pass (k3_settle) strikes them out; also with a record buffer
    that overflows first (regrow -> rescan -> settle again) and through the device-resident API (total excludes them)."""
    import torch

    words = ["alpha", "betaa", "gamma", "delta", "epsil", "zetaa", "etaaa", "theta", "iotaa", "kappa", "lambd", "muuuu"]
    pattern = "|".join(words)
    # (since round 6 such a pattern goes through the resolve pass, which settles every record with the pattern's VM program;
    # k3_settle remains for databases without one -- GSCAN_NO_RESOLVE=1 at compile time is that path's switch)
    assert engine.Database(pattern).info.resolve
    os.environ["GSCAN_NO_RESOLVE"] = "1"
    try:
        db = engine.Database(pattern)
        db2 = engine.Database("|".join("abcdefghijkl") + "|zz")
    finally:
        del os.environ["GSCAN_NO_RESOLVE"]
    assert db.info.tier == engine.TIER_BUCKET and db.info.n_alts == 12 and not db.info.resolve
    data = sample(2_000_003, 31)
    rng = np.random.default_rng(31)
    crosses = [b"alpaa", b"iotha", b"iopha", b"alota", b"betpa", b"kapaa", b"kaaaa", b"betaa", b"iotaa", b"alpha", b"muuuu", b"lambd"]
    for i in range(3000):
        w = crosses[i % len(crosses)]
        at = int(rng.integers(0, data.size - 8))
        data[at:at + len(w)] = np.frombuffer(w, np.uint8)
    want = table_candidates(db, data)
    got = ctx.scan(db, data)
    assert same(got, want) and len(got) > 500
    # dense: every byte is a hit of some single-letter alternative -> overflow -> regrow -> settle on the rescan
    assert db2.info.n_alts == 13 and not db2.info.resolve
    got = ctx.scan(db2, data[:700_001])
    assert same(got, table_candidates(db2, data[:700_001]))
    # device-resident: total counts the survivors only, fetch skips the struck records
    arena = torch.from_numpy(data).cuda()
    ctx.set_capacity(1 << 20)
    res = ctx.scan_device(db, arena.data_ptr(), [(0, data.size)])
    total, overflow = ctx.dev_sync(res)
    fetched = ctx.dev_fetch(res, 0)
    assert not overflow and total == len(fetched) and same(fetched, want)


def _orbit_lines(pattern, data):
    """The reference's loop in line-printing mode (grab.cc:171-213) as (m0, m1, lb, le) tuples: scan_oracle's restatement, unformatted."""
    import re
    rx = re.compile(pattern.encode())
    minlen = so.py_minlen(pattern)
    content = data.tobytes()
    clen = len(content)
    view = memoryview(content)
    out, s = [], 0
    while s + minlen < clen:
        m = rx.search(view[s:])
        if m is None:
            break
        b, e = s + m.start(), s + m.end()
        lo = b
        while lo - 1 >= s and content[lo - 1] != 0x0A and b - lo < so.CONTEXT:
            lo -= 1
        a = 0
        while e + a < clen and content[e + a] != 0x0A and a < so.CONTEXT:
            a += 1
        out.append((b, e, lo, e + a))
        s = e + a
    return out


def test_line_extents_on_device(ctx):
    """k_lines: for patterns whose classes exclude newline the device marks the records the reference's
    line-printing loop prints and gives their line extents; where it answers "ask the host" (lines running past the
    511-byte caps) everything before that point must still be exact."""
    data = sample(900_001, 41)
    data[200_000:203_000] = ord("q")                      # a 3000-byte line: the caps of grab.cc:173 come into play
    data[200_100:200_118] = np.frombuffer(b"foobardoesnotexist", np.uint8)
    data[201_500:201_518] = np.frombuffer(b"foobardoesnotexist", np.uint8)
    data[500_000:500_050] = ord("\n")                     # empty lines
    ctx.set_option("line_extents", 1)
    try:
        for pattern in ["foobardoesnotexist", "foo", "[A-Za-z_][A-Za-z0-9_]{15,}", "[0-9A-F]{6}[a-z]", "e+", "[a-z]{2,5}"]:
            db = engine.Database(pattern)
            assert db.info.lines_ok
            for variant in (1, 6, DEFAULT_VARIANT):
                ctx.set_option("variant", variant)
                starts = ctx.scan(db, data)
                ext = ctx.last_ext(len(starts))
                assert ext is not None and ext.shape == (len(starts), 4)
                text = ctx.last_gather()
                assert text is not None
                want = _orbit_lines(pattern, data)
                got = []
                asked = False
                for p, (m1, lb, le, goff) in zip(starts.tolist(), ext.tolist()):
                    if m1 == 0:
                        continue
                    if lb == 0xFFFFFFFF:
                        asked = True
                        break
                    got.append((p, m1, lb, le))
                    assert goff != 0xFFFFFFFF and np.array_equal(text[goff:goff + le - lb], data[lb:le]), (pattern, variant, p)  # the line's text, gathered by the device
                assert got == want[:len(got)], (pattern, variant)
                if not asked:
                    assert len(got) == len(want), (pattern, variant)
                else:
                    assert len(got) >= sum(1 for w in want if w[0] < 199_000)  # nothing before the long line at 200000 needs the host
        for pattern in ["fo[^x]", "(foo)", r"\bfoo", "foo|bar", "a+b"]:  # newline in a class / capture / context / several alternatives
            assert not engine.Database(pattern).info.lines_ok
    finally:
        ctx.set_option("line_extents", 0)
        ctx.set_option("variant", DEFAULT_VARIANT)


def test_match_ends_on_device(ctx):
    """k_ends (-O -l without the text): with "match_ends" on, every listed start of an ends_ok pattern comes with the end of
    its match -- window + greedy tail, cut at the segment end -- exactly gscan_match_end's (the host rule pinned against
    libpcre in tests/test_pattern.py); a tail that runs on for more than 4 KiB is left to the host (0).  Single chunks and
    batches of segments, the default kernel and an older variant."""
    data = sample(700_001, 43)
    data[300_000:309_000] = ord("q")                      # a 9000-byte identifier: beyond what the device follows
    data[650_000:] = ord("e")                             # a tail that runs into the end of the chunk
    ctx.set_option("match_ends", 1)
    try:
        for pattern in ["[A-Za-z_][A-Za-z0-9_]{15,}", "e+", r"foo\w*", "[0-9A-F]{2}[0-9A-Fa-f]*", r"[a-z][0-9][A-Z]\w*"]:
            db = engine.Database(pattern)
            assert db.info.ends_ok, pattern
            for variant in (DEFAULT_VARIANT, 5):
                ctx.set_option("variant", variant)
                starts = ctx.scan(db, data)
                ends = ctx.last_ends(len(starts))
                assert ends is not None and len(ends) == len(starts) and len(starts) > 0, pattern
                asked = 0
                for p, e in zip(starts.tolist(), ends.tolist()):
                    want = db.match_end(data, p)
                    if e == 0:
                        assert want - p > 4096, (pattern, p, want)
                        asked += 1
                    else:
                        assert e == want, (pattern, variant, p, e, want)
                assert asked < len(starts), (pattern, asked)  # (only starts inside the two long runs; how many of those are listed is the kernel's business)
            # a batch of segments: ends are segment-relative like the starts
            segs = [(0, 100_000), (100_000, 1), (100_016, 250_000), (350_016, 0), (350_016, 349_985)]
            parts = [data[o:o + ln] for o, ln in segs]
            ctx.submit_batch(db, parts)
            _, per, _ = ctx.wait_segs()
            ends = ctx.last_ends(sum(len(x) for x in per))
            assert ends is not None
            at = 0
            for i, part in enumerate(parts):
                for p, e in zip(per[i].tolist(), ends[at:at + len(per[i])].tolist()):
                    want = db.match_end(part, p)
                    assert e == want or (e == 0 and want - p > 4096), (pattern, i, p, e, want)
                at += len(per[i])
        for pattern in ["foo", "[a-f]{3}", "abc[0-9]*", "(f)o+", r"\bfoo\w*", "foo|ba+"]:
            db = engine.Database(pattern)
            assert not db.info.ends_ok
            starts = ctx.scan(db, data)
            if db.info.resolve:  # (the resolve pass's ends come with every chunk, option or no option: tests/test_gpu_resolve.py)
                assert np.array_equal(ctx.last_ends(len(starts)), resolved_list(db, data)[1]), pattern
                continue
            assert ctx.last_ends(len(starts)) is None, pattern
    finally:
        ctx.set_option("match_ends", 0)
        ctx.set_option("variant", DEFAULT_VARIANT)


def test_prefaulted_staging_blocks(built, tmp_path):
    """gscan_prefault: staging memory mapped and touched before the runtime is up, then registered by the reader pool instead of
    allocated -- more blocks wanted than were made ahead (the pool falls back to the runtime's allocator for the rest), file
    ranges of several pieces, a batch of small files; and GSCAN_PREFAULT=0 (a no-op).  In a process of its own: the arena
    is made once per process, before anything else."""
    import subprocess

    driver = (
        "import os, sys\n"
        "import numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))\n"
        "from grab_amd import engine, synth\n"
        "from inputs import db_candidates\n"
        "sys.path.insert(0, os.path.join(%r, 'oracle'))\n"
        "import scan_oracle as so\n"
        "assert engine.lib().gscan_prefault(3) == 0\n"
        "assert engine.lib().gscan_prefault(5) == 0  # once per process: the second call changes nothing\n"
        "blk = engine.lib().gscan_block_size()\n"
        "data = synth.text(5 * blk + 12345, 77)\n"
        "data[blk - 9:blk + 9] = np.frombuffer(b'foobardoesnotexist', np.uint8)\n"
        "d = sys.argv[1]\n"
        "data.tofile(os.path.join(d, 'f.bin'))\n"
        "small = []\n"
        "for i in range(12):\n"
        "    part = data[i * 70001:(i + 1) * 70001]\n"
        "    part.tofile(os.path.join(d, 's%%02d' %% i))\n"
        "    small.append((os.path.join(d, 's%%02d' %% i), part.size))\n"
        "ctx = engine.Context(0, 1 << 30)\n"
        "fd = os.open(os.path.join(d, 'f.bin'), os.O_RDONLY)\n"
        "for pattern in ('foobardoesnotexist', '[A-Za-z_][A-Za-z0-9_]{15,}'):\n"
        "    db = engine.Database(pattern)\n"
        "    for rep in range(3):\n"
        "        ctx.submit_fd(db, fd, 0, data.size, tag=1)\n"
        "        ctx.submit_files(db, small, tag=2)\n"
        "        ctx.submit_fd(db, fd, 4096, 2 * blk + 5, tag=3)\n"
        "        t, s, _ = ctx.wait_segs(); assert t == 1 and so.check_reported(s[0], db_candidates(db, data))\n"
        "        t, s, _ = ctx.wait_segs(); assert t == 2 and all(so.check_reported(s[i], db_candidates(db, data[i * 70001:(i + 1) * 70001])) for i in range(12))\n"
        "        t, s, _ = ctx.wait_segs(); assert t == 3 and so.check_reported(s[0], db_candidates(db, data[4096:4096 + 2 * blk + 5]))\n"
        "ctx.close()\n"
        "print('PREFAULT OK')\n"
    ) % (ROOT, ROOT, ROOT)
    for env in ({}, {"GSCAN_PREFAULT": "0"}, {"GSCAN_READERS": "2"}):
        r = subprocess.run([sys.executable, "-c", driver, str(tmp_path)], capture_output=True, text=True, env=dict(os.environ, **env))
        assert r.returncode == 0 and "PREFAULT OK" in r.stdout, (env, r.stdout[-300:], r.stderr[-800:])
