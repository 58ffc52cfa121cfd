"""Host-side pieces that need no device: the parallel tree walk of `grab -n` against what nftw(FTW_PHYS) reports,
the NUMA map (device -> local CPUs) the readers and workers are placed by, pattern validation without a device,
and the shape of the ingest configuration."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from grab_amd import engine, filegrep


def _mk_tree(root, rng, depth=3, fan=4, files=6):
    want = {}

    def fill(d, level):
        for i in range(int(rng.integers(0, files + 1))):
            p = os.path.join(d, "f%d_%d.txt" % (level, i))
            n = int(rng.integers(0, 2000))
            with open(p, "wb") as f:
                f.write(b"x" * n)
            want[os.fsencode(p)] = n
        if level < depth:
            for i in range(int(rng.integers(1, fan + 1))):
                sub = os.path.join(d, "d%d" % i)
                os.mkdir(sub)
                fill(sub, level + 1)

    fill(root, 0)
    return want


@pytest.mark.parametrize("threads", [1, 2, 4, 7])
def test_parallel_walk_reports_what_nftw_reports(threads, built, tmp_path):
    """Regular files only, symbolic links neither followed nor reported, empty and unreadable directories harmless,
    every file exactly once whatever the number of walkers (main.cc:74-83,178)."""
    rng = np.random.default_rng(threads)
    root = str(tmp_path / "tree")
    os.mkdir(root)
    want = _mk_tree(root, rng)
    os.mkdir(os.path.join(root, "empty"))
    os.symlink(os.path.join(root, "d0"), os.path.join(root, "link_to_dir"))
    first = sorted(want)[0] if want else None
    if first:
        os.symlink(first, os.path.join(root, "link_to_file"))
    os.mkfifo(os.path.join(root, "fifo"))
    got = filegrep.walk_parallel(root, threads)
    assert len(got) == len(set(p for p, _ in got)), "a file was reported twice"
    assert dict(got) == want
    # trailing slashes of the root are dropped, as nftw does; a symbolic link as the root is not followed; a file as the root is itself
    assert dict(filegrep.walk_parallel(root + "//", threads)) == want
    assert filegrep.walk_parallel(os.path.join(root, "link_to_dir"), threads) == []
    if first:
        assert filegrep.walk_parallel(os.fsdecode(first), threads) == [(first, want[first])]
    assert filegrep.walk_parallel(os.path.join(root, "nope"), threads) == []


def test_parallel_walk_equals_the_reference_walk(built, oracle_built, tmp_path):
    """The path strings are the ones nftw builds: `-l` output of the oracle's serial walk lists them."""
    rng = np.random.default_rng(99)
    root = tmp_path / "t"
    root.mkdir()
    want = _mk_tree(str(root), rng, depth=2)
    for p in want:
        with open(p, "wb") as f:
            f.write(b"needle\n")
    r = subprocess.run([os.path.join(oracle_built, "grab_oracle"), "-r", "-l", "needle", "t/"], cwd=str(tmp_path), capture_output=True)
    assert r.returncode == 0
    ref_paths = sorted(l[:-len(b":matches")] for l in r.stdout.splitlines())
    os.chdir(str(tmp_path))
    try:
        got = sorted(p for p, _ in filegrep.walk_parallel("t/", 3))
    finally:
        os.chdir(ROOT)
    assert got == ref_paths


def test_cpulist_parse(built):
    assert engine.parse_cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11]
    assert engine.parse_cpulist("5") == [5]
    assert engine.parse_cpulist("0-31,64-95\n") == list(range(32)) + list(range(64, 96))
    assert engine.parse_cpulist("") == []
    assert engine.parse_cpulist("garbage") == []


def test_numa_map_from_sysfs(built, tmp_path):
    """Device -> CPUs of its NUMA node, read from <pci root>/<bus id>/local_cpulist; the bus id as the HIP runtime
    prints it (upper-case hex) finds the lower-case sysfs directory.  Two-socket layout of an 8-GPU node."""
    for i in range(8):
        d = tmp_path / ("0000:%02x:00.0" % (0x0c + 0x10 * i))
        d.mkdir()
        (d / "local_cpulist").write_text("0-47,96-143\n" if i < 4 else "48-95,144-191\n")
        (d / "numa_node").write_text("%d\n" % (i // 4))
    assert engine.pci_cpulist(str(tmp_path), "0000:0C:00.0") == "0-47,96-143"
    assert engine.pci_cpulist(str(tmp_path), "0000:4c:00.0") == "48-95,144-191"
    assert engine.pci_cpulist(str(tmp_path), "0000:ff:00.0") is None
    cpus = engine.parse_cpulist(engine.pci_cpulist(str(tmp_path), "0000:7C:00.0"))
    assert cpus[0] == 48 and len(cpus) == 96 and 0 not in cpus


def test_validate_without_device(built):
    assert filegrep.validate("foo") == (0, "")
    assert filegrep.validate("a(") == (-1, "FileGrep::prepare::pcre_compile error")
    rc, why = filegrep.validate(r"a(*COMMIT)b|(x)")  # (valid PCRE, outside the engine: the search-steering verbs)
    assert rc == -2 and "outside the GPU engine's subset" in why
    assert filegrep.validate("a(", literal=True) == (0, "")
    # a text libpcre compiles is never reported as its compile error, whatever the engine's own parser says about it
    # (-n mode answers "pcre_compile error" with a silent exit 0, main.cc:198)
    for pattern in ("(?|(?<n>a)|(?<n>b))", r"\p{Greek}x", "a(*PRUNE)b"):
        rc, why = filegrep.validate(pattern)
        assert rc == -2 and "outside the GPU engine's subset" in why, (pattern, rc, why)


def test_ingest_configuration(built):
    info = engine.ingest_info()
    assert info["block_bytes"] % (1 << 20) == 0 and (1 << 20) <= info["block_bytes"] <= (64 << 20)
    assert 1 <= info["readers"] <= 64 and 1 <= info["copy_streams"] <= 4
    # copy streams: the device's shared ones (GSCAN_SHARED_COPY, the default), or per context (GSCAN_COPY_STREAMS) when sharing is off
    for extra, want in (({"GSCAN_SHARED_COPY": "0", "GSCAN_COPY_STREAMS": "4"}, 4), ({"GSCAN_SHARED_COPY": "2", "GSCAN_COPY_STREAMS": "4"}, 2), ({}, 1)):
        env = dict(os.environ, GSCAN_BLOCK_MIB="32", GSCAN_READERS="3", **extra)
        r = subprocess.run(["python", "-c", "from grab_amd import engine; print(engine.ingest_info())"], cwd=ROOT, env=env, capture_output=True, text=True)
        assert "'block_bytes': 33554432" in r.stdout and "'readers': 3" in r.stdout and "'copy_streams': %d" % want in r.stdout, r.stdout + r.stderr
