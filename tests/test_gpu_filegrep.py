"""GPU parity, drop-in level: the `grab` binary and the FileGrep mirror must print exactly what
the reference prints -- checked against the golden outputs of the reference binary
(tests/golden/golden.json) and, on fresh random trees, against the oracle run side by side."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, golden_ids, split_args
from grab_amd import engine, filegrep, synth
from inputs import materialize

sys.path.insert(0, os.path.join(ROOT, "oracle"))

pytestmark = pytest.mark.gpu


def _engine_pattern(case):
    _, pattern, _ = split_args(case["args"])
    try:
        engine.Database(pattern)
        return True
    except ValueError:
        return False


def _run(binary, args, cwd):
    r = subprocess.run([binary] + args, cwd=cwd, capture_output=True, env=dict(os.environ, GRAB_DIAG="1", GSCAN_MATCH_LIMIT="200000000"))
    return r.returncode, r.stdout, r.stderr


@pytest.mark.parametrize("case", GOLDEN, ids=golden_ids(GOLDEN))
def test_cli_matches_reference(case, built, tmp_path):
    cache = {}
    for rel, recipe in case["inputs"].items():
        materialize(recipe, str(tmp_path / rel), cache)
    rc, out, err = _run(built.bin_path(), case["args"], str(tmp_path))
    if case["name"] in ("bad_regex", "missing_file", "n2_without_r", "dir_without_r"):
        assert (rc, out) == (case["rc"], b"")
        assert err.decode("latin-1") == case["stderr"]
        return
    if not _engine_pattern(case):
        # valid PCRE outside the engine's subset: refuse loudly, never scan on the CPU
        assert rc == 255 and out == b"" and b"outside the GPU engine's subset" in err
        return
    if case["sorted"]:
        out = b"".join(sorted(out.splitlines(True)))
    assert rc == case["rc"], err
    assert len(out) == case["stdout_len"]
    assert hashlib.md5(out).hexdigest() == case["stdout_md5"]
    if "stdout" in case:
        assert out == case["stdout"]


def test_filegrep_interface(built, tmp_path):
    """The ctypes mirror of FileGrep: same calls the reference's main() makes (main.cc:231-252)."""
    f = tmp_path / "t1.txt"
    f.write_bytes(b"hello foo world\nno match here\nfoo at start and foo again\ntail foo")
    outp = tmp_path / "out"
    fd = os.open(str(outp), os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
    g = filegrep.FileGrep()
    g.config({"offsets": 1, "noline": 1, "chunk_size": 1 << 30, "out_fd": fd})
    assert g.prepare("foo") == 0, g.why()
    assert g.find(str(f)) == 0
    assert g.find(str(tmp_path / "missing")) == -1 and g.why().startswith("FileGrep::find::stat: ")
    os.close(fd)
    assert outp.read_bytes() == b"Match at offset 6\nMatch at offset 30\nMatch at offset 47\nMatch at offset 62\n"
    g2 = filegrep.FileGrep()
    assert g2.prepare(r"a(*COMMIT)b") == -1 and "outside the GPU engine's subset" in g2.why()
    g3 = filegrep.FileGrep()
    assert g3.prepare("a(") == -1 and g3.why() == "FileGrep::prepare::pcre_compile error"


def _tree(root, rng, nfiles):
    names = []
    for i in range(nfiles):
        d = root / ("d%d" % (i % 5)) / ("s%d" % (i % 3))
        d.mkdir(parents=True, exist_ok=True)
        n = int(rng.choice([0, 1, 7, 100, 5000, 70000, 300000]))
        buf = synth.text(n, 100 + i)
        if n >= 5000:
            synth.plant(buf, b"foobardoesnotexist", 3, i, gap=200)
        p = d / ("f%03d.txt" % i)
        buf.tofile(str(p))
        names.append(p)
    os.symlink(str(names[0]), str(root / "link.txt"))
    return names


@pytest.mark.parametrize("args", [["-r", "-O"], ["-r", "-O", "-l"], ["-r"], ["-r", "-l"], ["-r", "-s"], ["-n", "2", "-r", "-O", "-l"], ["-n", "3", "-r"]])
@pytest.mark.parametrize("pattern", ["foobardoesnotexist", "[A-Za-z_][A-Za-z0-9_]{15,}", "[0-9A-F]{6}[a-z]",
                                     "foobardoes(?:not)?exist|[0-9A-F]{7}[a-z]?|(?i:xyzzy)", r"(?m)^[a-z]{3}\b|\b[0-9A-F]{5}$|^foo", r"[0-9]+\.[0-9]+|foo.*exist|\b[a-z_]+ ?= ?[0-9A-F]{1,4};",
                                     r"(?:foo|bar)+does.*exist|[0-9]+\.[0-9]+\.[0-9]+|\b(?:[a-z]+_)+[a-z]+\b"])
def test_tree_differential(args, pattern, built, oracle_built, tmp_path):
    """Random tree, recursive + threaded modes: sorted output == the oracle's (the reference's own
    criterion for -n, README.md:206-216); the real reference binary is compared too when present."""
    rng = np.random.default_rng(42)
    root = tmp_path / "tree"
    root.mkdir()
    _tree(root, rng, 24)
    argv = args + [pattern, "tree"]
    rc, out, err = _run(built.bin_path(), argv, str(tmp_path))
    orc, oout, _ = _run(os.path.join(oracle_built, "grab_oracle"), argv, str(tmp_path))
    assert rc == orc == 0, err
    assert sorted(out.splitlines(True)) == sorted(oout.splitlines(True))
    if "-n" not in args:
        assert out == oout  # serial walk: same nftw order, byte-identical
    ref = os.path.join(ROOT, "oracle", "_ref", "grab_jit")
    if os.path.exists(ref):
        rrc, rout, _ = _run(ref, argv, str(tmp_path))
        assert rrc == 0 and sorted(out.splitlines(True)) == sorted(rout.splitlines(True))


def test_multichunk_pipeline(built, oracle_built, tmp_path):
    """A 100 MiB file at 32 MiB chunks: 4 chunks through the 2-slot pipeline, dense output, -O with lines."""
    buf = synth.text(100 << 20, 7)
    p = tmp_path / "f"
    buf.tofile(str(p))
    for flags in (["-O", "-l"], ["-O"], []):
        argv = ["-L"] * 5 + flags + ["[A-Za-z_][A-Za-z0-9_]{15,}", "f"]
        rc, out, err = _run(built.bin_path(), argv, str(tmp_path))
        orc, oout, _ = _run(os.path.join(oracle_built, "grab_oracle"), argv, str(tmp_path))
        assert rc == orc == 0, err
        assert hashlib.md5(out).hexdigest() == hashlib.md5(oout).hexdigest() and len(out) == len(oout)


def test_literal_flag_S(built, tmp_path):
    p = tmp_path / "f"
    p.write_bytes(b"a.c abc a.c|x\n(a.c)\n")
    rc, out, _ = _run(built.bin_path(), ["-S", "-O", "-l", "a.c", "f"], str(tmp_path))
    assert rc == 0 and out == b"Match at offset 0\nMatch at offset 8\nMatch at offset 15\n"
    rc, out, _ = _run(built.bin_path(), ["-H", "-2", "-S", "-O", "-l", "(a.c)", "f"], str(tmp_path))
    assert rc == 0 and out == b"Match at offset 14\n"


def test_octal_escapes_through_the_cli(built, oracle_built, tmp_path):
    r"""\NN outside a class that is no back reference (round 6: pcre_compile's rule, tests/test_pattern.py) end to end: K1 / K2 / K3
    databases made from such patterns print what the oracle prints."""
    p = tmp_path / "f"
    p.write_bytes((b"xxAxx A1 \n yy\t81 9 S \xff z aa\t a\n1 AB 8 80 a8 xS1 \x018 ABBA \tx\n" * 400).ljust(40000, b"."))
    for pattern in (r"\101", r"\1011", r"x\1231", r"\81", r"\8|\101B", r"\11x|\12\61", r"[\101-\103]{2,}", r"\b\1011\b"):
        for flags in (["-O", "-l"], []):
            rc, out, err = _run(built.bin_path(), flags + [pattern, "f"], str(tmp_path))
            orc, oout, _ = _run(os.path.join(oracle_built, "grab_oracle"), flags + [pattern, "f"], str(tmp_path))
            assert rc == orc == 0, err
            assert out == oout and len(out) > 0, (pattern, flags)


def test_random_patterns_cli_vs_oracle(built, oracle_built, tmp_path):
    """End to end on the GPU with random patterns of the supported grammar (tests/test_fuzz.py's generator): whatever tier
    the compiler picks, `grab` prints what the oracle (libpcre under the reference's loop) prints for the same file."""
    import random

    from test_fuzz import gen

    rng = random.Random(2024)
    nrng = np.random.default_rng(2024)
    alpha = np.frombuffer(b"abcxA01 .\n\nab  ", np.uint8)
    data = alpha[nrng.integers(0, alpha.size, 300_000)]
    data[1000:1003] = np.frombuffer(b"abc", np.uint8)
    p = tmp_path / "f"
    data.tofile(str(p))
    done = 0
    for _ in range(400):
        pat = gen(rng)
        try:
            db = engine.Database(pat)
        except ValueError:
            continue
        if db.minlen < 0:
            continue
        flags = [["-O", "-l"], ["-O"], []][done % 3]
        orc, oout, oerr = _run(os.path.join(oracle_built, "grab_oracle"), flags + [pat, "f"], str(tmp_path))
        if orc != 0:
            continue  # libpcre rejects the pattern (the product reports the same error; covered elsewhere)
        rc, out, err = _run(built.bin_path(), flags + [pat, "f"], str(tmp_path))
        assert rc == 0, (pat, err)
        if b"gave up" in oerr or b"abandoned" in err:
            continue  # GRAB_DIAG: libpcre hit its match limit / the host matcher its own -- where an engine gives up is its own business
        assert out == oout, (pat, flags, len(out), len(oout))
        done += 1
        if done == 45:
            break
    assert done == 45


def test_calls_and_conditions_cli_vs_oracle(built, oracle_built, tmp_path):
    """The constructs taken in the second half of round 2 -- conditional groups, subroutine calls and recursion, branch reset,
    \\R \\X \\p{..}, (*FAIL), callouts, (?U) -- end to end on the GPU: the hand-made list of tests/test_pattern.py and random
    draws of tests/test_fuzz.py's second grammar, `grab` against the oracle (libpcre under the reference's loop)."""
    import random

    from test_fuzz import gen_calls_and_conditions
    from test_pattern import CALLS_AND_CONDITIONS, NEWLINE_SEQUENCES

    nrng = np.random.default_rng(77)
    alpha = np.frombuffer(b"abcxA01 .\n\nab  ()<>]-", np.uint8)
    data = alpha[nrng.integers(0, alpha.size, 200_000)]
    at = 500
    for _, text in CALLS_AND_CONDITIONS:  # every hand-made text, planted a few times
        if text is None:
            continue
        for rep in range(3):
            data[at:at + len(text)] = np.frombuffer(text, np.uint8)
            at += len(text) + 37
    for w in (b"a\r\nb a\nb a\rb a\x0bb a\x85b a\r\n\nb", b"ab\xe9\xc9 12\xb2 x_\xb5\xaa\xd7"):
        data[at:at + len(w)] = np.frombuffer(w, np.uint8)
        at += len(w) + 11
    (tmp_path / "f").write_bytes(data.tobytes())
    rng = random.Random(99)
    pats = [pt for pt, text in CALLS_AND_CONDITIONS if text is not None] + NEWLINE_SEQUENCES + [r"\p{Lu}\p{Ll}+", r"[\p{Nd}x]{2}\P{L}"]
    pats += [gen_calls_and_conditions(rng) for _ in range(250)]
    done = 0
    for k, pat in enumerate(pats):
        try:
            db = engine.Database(pat)
        except ValueError:
            continue
        if db.minlen < 0:
            continue
        flags = [["-O", "-l"], ["-O"], []][k % 3]
        orc, oout, oerr = _run(os.path.join(oracle_built, "grab_oracle"), flags + [pat, "f"], str(tmp_path))
        if orc != 0:
            continue
        rc, out, err = _run(built.bin_path(), flags + [pat, "f"], str(tmp_path))
        assert rc == 0, (pat, err)
        if b"gave up" in oerr or b"abandoned" in err:
            continue
        assert out == oout, (pat, flags, len(out), len(oout))
        done += 1
        if done == 130:
            break
    assert done >= 100


def test_line_pass_equals_host_walk(built, oracle_built, tmp_path):
    """Line-printing modes with the device's line pass (the default: k_lines picks the printed matches, finds their line
    extents and gathers the lines' text; records it leaves to the host go through the reference's loop, after which the pass
    takes over again) and without it (GRAB_LINE_PASS=0: the host's walk) print the same bytes, and both equal the oracle:
    long lines (511-byte caps), empty lines, dense matches, several chunks, batches."""
    rng = np.random.default_rng(9)
    buf = synth.text(70 << 20, 11)
    buf[1_000_000:1_004_000] = ord("z")                   # one 4000-byte line with two needles in it
    buf[1_000_700:1_000_718] = np.frombuffer(b"foobardoesnotexist", np.uint8)
    buf[1_002_900:1_002_918] = np.frombuffer(b"foobardoesnotexist", np.uint8)
    buf[(32 << 20) - 5000:(32 << 20) + 3000] = ord("k")   # a long line across the first 32 MiB chunk boundary
    buf[(32 << 20) - 10:(32 << 20) + 8] = np.frombuffer(b"foobardoesnotexist", np.uint8)
    synth.plant(buf[2_000_000:], b"foobardoesnotexist", 200, 3, gap=300)
    (tmp_path / "d").mkdir()
    buf.tofile(str(tmp_path / "d" / "big"))
    for i in range(40):                                   # small files: the batched path
        n = int(rng.integers(1, 200_000))
        synth.text(n, 500 + i).tofile(str(tmp_path / "d" / ("s%02d" % i)))
    for pattern in ["foobardoesnotexist", "[A-Za-z_][A-Za-z0-9_]{15,}"]:
        for flags in (["-r", "-O"], ["-r"], ["-L", "-L", "-L", "-L", "-L", "-r", "-O"], ["-r", "-s"]):
            argv = flags + [pattern, "d"]
            rc, out, err = _run(built.bin_path(), argv, str(tmp_path))
            r = subprocess.run([built.bin_path()] + argv, cwd=str(tmp_path), capture_output=True, env=dict(os.environ, GRAB_LINE_PASS="0"))
            orc, oout, _ = _run(os.path.join(oracle_built, "grab_oracle"), argv, str(tmp_path))
            assert rc == r.returncode == orc == 0, err
            assert out == r.stdout, (pattern, flags)
            assert out == oout, (pattern, flags)


def test_line_pass_at_its_look_back_bound(built, oracle_built, tmp_path):
    """k_lines follows a match's tail and looks for a line's start 4096 bytes (kLineBack) at most before it asks the host: tails
    of exactly 4095 / 4096 / 4097 bytes of the tail class, a line that starts exactly 4096 bytes in front of its first record
    (and one byte more, one less), the same again right behind a 32-byte step of the searches -- with the pass and without it the
    output is the oracle's (the boundary had no test)."""
    parts = []
    for tail in (4095, 4096, 4097, 4064, 4128):
        parts.append(b"xx\n" + b"Q" + b"a" * tail + b" rest of the line\nnext Qab\n")
    for back in (4095, 4096, 4097, 4064, 4128, 33):
        parts.append(b"\n" + b"." * back + b"Qabc and more\nQz\n")
    for back in (4096, 4097):  # ... and a second record in such a line: not printed (the first one of the line is)
        parts.append(b"\n" + b"Qfirst" + b"." * back + b"Qsecond\n")
    data = b"".join(parts) * 3
    (tmp_path / "f").write_bytes(data)
    for pattern in ["Q[a-z]*", "Q[a-z]{2,}"]:
        for flags in (["-O"], [], ["-O", "-l"]):
            argv = flags + [pattern, "f"]
            rc, out, err = _run(built.bin_path(), argv, str(tmp_path))
            r = subprocess.run([built.bin_path()] + argv, cwd=str(tmp_path), capture_output=True, env=dict(os.environ, GRAB_LINE_PASS="0", GRAB_NO_ENDS="1"))
            orc, oout, _ = _run(os.path.join(oracle_built, "grab_oracle"), argv, str(tmp_path))
            assert rc == r.returncode == orc == 0, err
            assert out == oout, (pattern, flags)
            assert r.stdout == oout, (pattern, flags)


def test_offsets_without_the_text_equals_host_walk(built, oracle_built, tmp_path):
    """-O -l with the match ends from the device (k_ends, the default: the walk never looks at the window) and with the host
    walk over the mapped text (GRAB_NO_ENDS=1) print the same bytes, and both equal the oracle: identifiers longer than the
    4 KiB the device follows, one across a 32 MiB chunk boundary (printed by both chunks, the second as its suffix), one
    that runs into the end of the file, several chunks, batches of small files, -s."""
    rng = np.random.default_rng(19)
    buf = synth.text(70 << 20, 13)
    buf[3_000_000:3_009_000] = ord("q")                   # a 9000-byte identifier
    buf[(32 << 20) - 5000:(32 << 20) + 3000] = ord("k")   # one across the first 32 MiB chunk boundary (4 KiB overlap)
    buf[-40:] = ord("w")                                  # one that ends with the file
    (tmp_path / "d").mkdir()
    buf.tofile(str(tmp_path / "d" / "big"))
    for i in range(40):                                   # small files: the batched path
        n = int(rng.integers(1, 200_000))
        synth.text(n, 700 + i).tofile(str(tmp_path / "d" / ("s%02d" % i)))
    for pattern in ["[A-Za-z_][A-Za-z0-9_]{15,}", r"foo\w*", "[0-9]{3}[0-9a-f]*"]:
        for flags in (["-r", "-O", "-l"], ["-L", "-L", "-L", "-L", "-L", "-r", "-O", "-l"], ["-r", "-O", "-l", "-s"]):
            argv = flags + [pattern, "d"]
            rc, out, err = _run(built.bin_path(), argv, str(tmp_path))
            r = subprocess.run([built.bin_path()] + argv, cwd=str(tmp_path), capture_output=True, env=dict(os.environ, GRAB_NO_ENDS="1"))
            orc, oout, _ = _run(os.path.join(oracle_built, "grab_oracle"), argv, str(tmp_path))
            assert rc == r.returncode == orc == 0, err
            assert out == r.stdout, (pattern, flags)
            assert out == oout, (pattern, flags)


def test_small_files_through_the_reader_pool(built, oracle_built, tmp_path):
    """Small files are queued by name and read by the device's reader threads (gscan_submit_files; VERDICT r3 task 2): the same
    bytes as round 3's path (GRAB_BATCH_READ=worker: the worker read(2)s them into a pinned block) and as the oracle, in every
    output mode, with batches of 1 MiB (dozens of them, files straddling pieces) and the default 32 MiB, files above the
    batching limit in between, patterns whose report needs the text (mapped on demand), does not (match ends from the device)
    or looks beyond the match (context)."""
    rng = np.random.default_rng(321)
    root = tmp_path / "t"
    for i in range(260):
        d = root / ("d%d" % (i % 7)) / ("s%d" % (i % 3))
        d.mkdir(parents=True, exist_ok=True)
        n = int(rng.choice([0, 1, 17, 18, 400, 5000, 70_000, 300_000, 524_288, 900_000])) if i % 40 else 3_000_000  # (3 MB: above the 2 MiB batching limit)
        buf = synth.text(n, 900 + i)
        if n >= 5000:
            synth.plant(buf, b"foobardoesnotexist", 2, i, gap=100)
        if n >= 400:
            buf[-18:] = np.frombuffer(b"foobardoesnotexist", np.uint8)  # ends with the file
        buf.tofile(str(d / ("f%03d.txt" % i)))
    for pattern in ["foobardoesnotexist", "[A-Za-z_][A-Za-z0-9_]{15,}", r"\bfoobardoesnotexist$|^[a-z]{4}\b"]:
        for flags in (["-r", "-O", "-l"], ["-r", "-O"], ["-r"], ["-r", "-s"], ["-n", "4", "-r", "-O", "-l"], ["-n", "3", "-r"]):
            argv = flags + [pattern, "t"]
            orc, oout, _ = _run(os.path.join(oracle_built, "grab_oracle"), argv, str(tmp_path))
            assert orc == 0
            for env in ({}, {"GRAB_BATCH_MIB": "1"}, {"GRAB_BATCH_READ": "worker"}):
                r = subprocess.run([built.bin_path()] + argv, cwd=str(tmp_path), capture_output=True, env=dict(os.environ, **env))
                assert r.returncode == 0, r.stderr
                if "-n" in flags:
                    if "-l" in flags:
                        assert sorted(r.stdout.splitlines()) == sorted(oout.splitlines()), (pattern, flags, env)
                    else:
                        assert len(r.stdout) == len(oout) and sorted(r.stdout.splitlines()) == sorted(oout.splitlines()), (pattern, flags, env)
                else:
                    assert r.stdout == oout, (pattern, flags, env)


def test_small_file_errors_surface_per_file(built, tmp_path):
    """A file of a batch that has vanished (or shrunk) between the walk's stat and the readers' open: its own error, said the
    way the reference's walk says it (grab.cc:267-268: "path: why" on stderr, the walk goes on), every other file of the
    batch printed as usual; an explicit path returns it from find(); the -n workers stay silent (main.cc:97)."""
    d = tmp_path / "t"
    d.mkdir()
    for i in range(6):
        (d / ("f%d" % i)).write_bytes(b"x" * 100 + b"foo %d\n" % i + b"y" * 50)
    driver = (
        "import os, sys\n"
        "from grab_amd import filegrep\n"
        "mode, d = sys.argv[1], sys.argv[2]\n"
        "g = filegrep.FileGrep()\n"
        "cfg = {'offsets': 1, 'noline': 1, 'chunk_size': 1 << 30}\n"
        "if mode == 'workers': cfg['silent_errors'] = 1\n"
        "g.config(cfg)\n"
        "if mode != 'explicit': g.recurse()\n"
        "assert g.prepare('foo') == 0, g.why()\n"
        "names = sorted(os.listdir(d))\n"
        "stats = [filegrep.c_stat(os.path.join(d, n)) for n in names]\n"
        "os.unlink(os.path.join(d, names[2]))\n"                      # vanished after the walk saw it
        "open(os.path.join(d, names[4]), 'wb').write(b'short')\n"     # shrank after the walk saw it
        "rcs = [g.find3(os.path.join(d, n), st) for n, st in zip(names, stats)]\n"
        "frc = g.flush()\n"
        "sys.stdout.flush()\n"
        "print('RC', rcs, frc, repr(g.why()))\n"
    )
    for mode in ("walk", "workers", "explicit"):
        for i in range(6):
            (d / ("f%d" % i)).write_bytes(b"x" * 100 + b"foo %d\n" % i + b"y" * 50)
        r = subprocess.run([sys.executable, "-c", driver, mode, str(d)], cwd=ROOT, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        lines = r.stdout.splitlines()
        rc_line = [ln for ln in lines if ln.startswith("RC")][0]
        hits = [ln for ln in lines if "Match at offset 100" in ln]
        assert len(hits) == 4, r.stdout  # the four files that were still what the walk saw
        if mode == "walk":
            assert "f2: FileGrep::find::open: No such file or directory" in r.stderr and "f4: FileGrep::find::read: file shrank while reading" in r.stderr, r.stderr
            assert "RC [0, 0, 0, 0, 0, 0] 0" in rc_line
        elif mode == "workers":
            assert "FileGrep::find" not in r.stderr, r.stderr
        else:
            assert "FileGrep::find::read: file shrank" in rc_line or "FileGrep::find::open" in rc_line, rc_line


def test_a_file_error_is_returned_once(built, tmp_path):
    """an error that belongs to one explicit path is returned by the find() during which it surfaces and by no
    later one -- the instance goes on with the next path as if nothing had happened."""
    for n in ("bad", "good1", "good2"):
        (tmp_path / n).write_bytes(b"x" * 100 + b"foo\n" + b"y" * 50)
    driver = (
        "import os, sys\n"
        "from grab_amd import filegrep\n"
        "d = sys.argv[1]\n"
        "g = filegrep.FileGrep()\n"
        "g.config({'offsets': 1, 'noline': 1, 'chunk_size': 1 << 30})\n"
        "assert g.prepare('foo') == 0, g.why()\n"
        "st = filegrep.c_stat(os.path.join(d, 'bad'))\n"
        "open(os.path.join(d, 'bad'), 'wb').write(b'short')\n"   # shrank after it was stat()ed
        "r0 = g.find3(os.path.join(d, 'bad'), st)\n"             # queued in a batch: nothing has looked at it yet
        "r1 = g.find(os.path.join(d, 'good1')); w1 = g.why()\n"   # the batch retires here: bad's error comes out of THIS call
        "r2 = g.find(os.path.join(d, 'good2'))\n"
        "sys.stdout.flush()\n"
        "print('RC', r0, r1, r2, repr(w1))\n"
    )
    r = subprocess.run([sys.executable, "-c", driver, str(tmp_path)], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rc_line = [ln for ln in r.stdout.splitlines() if ln.startswith("RC")][0]
    assert rc_line.startswith("RC 0 -1 0 ") and "file shrank" in rc_line, r.stdout + r.stderr
    assert r.stdout.count("Match at offset 100") == 2, r.stdout  # good1 (same batch as the bad file) and good2


def test_read_ahead_while_the_runtime_starts(built, oracle_built, tmp_path):
    """Explicit files are read into staging memory by helper threads WHILE hipInit runs (gscan_prefault_files) and their
    pieces adopted by the reader pool: same bytes as without (GRAB_NO_READ_AHEAD=1) and as the oracle -- a file of three
    pieces with a ragged last one, two files on one command line (the second one's pieces are beyond the arena), a file
    that fills more than the arena holds, needles across piece boundaries."""
    blk = 8 << 20
    files = []
    for i, n in enumerate((2 * blk + 3, 5 * blk, 45 * blk + 17)):
        buf = synth.text(n, 600 + i)
        synth.plant(buf, synth.NEEDLE, 40, i, gap=100)
        for edge in range(blk, n - 20, blk):  # across and right behind every piece boundary
            buf[edge - 7:edge - 7 + len(synth.NEEDLE)] = np.frombuffer(synth.NEEDLE, np.uint8)
            buf[edge + 100:edge + 100 + len(synth.NEEDLE)] = np.frombuffer(synth.NEEDLE, np.uint8)
        p = tmp_path / ("f%d.bin" % i)
        buf.tofile(p)
        files.append(p.name)
    for pattern, flags in ((synth.NEEDLE.decode(), ["-O", "-l"]), (synth.NEEDLE.decode(), ["-O"]), (synth.IDENT_RE, ["-O", "-l"])):
        for names in (files[:1], files[:2], files[2:], files):
            argv = flags + [pattern] + names
            orc, oout, _ = _run(os.path.join(oracle_built, "grab_oracle"), argv, str(tmp_path))
            assert orc == 0
            for env in ({}, {"GRAB_NO_READ_AHEAD": "1"}, {"GSCAN_BLOCK_MIB": "3"}):
                r = subprocess.run([built.bin_path()] + argv, cwd=str(tmp_path), capture_output=True, env=dict(os.environ, **env))
                assert r.returncode == 0 and r.stdout == oout, (pattern, flags, names, env, r.stderr[-300:])
