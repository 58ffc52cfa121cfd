"""Host-side pieces that need no device: the parallel tree walk of `grab -n` against what nftw(FTW_PHYS) reports,
the NUMA map (device -> local CPUs) the readers and workers are placed by, pattern validation without a device,
and the shape of the ingest configuration."""
import os
import subprocess
import sys

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
    assert 0 <= info["readers"] <= 64 and 1 <= info["copy_streams"] <= 4  # readers 0: auto, per device (gscan_auto_readers)
    # copy streams: per DEVICE, shared by its contexts (GSCAN_COPY_STREAMS: 1, or 2 = a second one once the device has been handed 1 GiB; the scans ride on the first)
    for extra, want in (({"GSCAN_COPY_STREAMS": "2"}, 2), ({}, 2), ({"GSCAN_COPY_STREAMS": "1"}, 1), ({"GSCAN_COPY_STREAMS": "9"}, 2)):
        env = dict(os.environ, GSCAN_BLOCK_MIB="32", GSCAN_READERS="3", **extra)
        r = subprocess.run(["python", "-c", "from grab_amd import engine; print(engine.ingest_info())"], cwd=ROOT, env=env, capture_output=True, text=True)
        assert "'block_bytes': 33554432" in r.stdout and "'readers': 3" in r.stdout and "'copy_streams': %d" % want in r.stdout, r.stdout + r.stderr


def test_eight_gpu_worker_placement(built):
    """`grab -n N -r` on a two-socket node with 8 GPUs (4 per socket), laid out from a faked device -> local-CPU map: worker
    i drives device i mod 8; every device gets the same number of workers (N a multiple of 8) or floor/ceil of N / 8; every
    worker is bound to CPUs of ITS device's socket only, cut to the process's mask; a device whose node is outside the mask
    (or unknown) leaves its workers on the process's own CPUs; GRAB_PIN=cpu is the reference's rule (thread i on CPU i,
    main.cc:200-215); GRAB_PIN=none the process's mask."""
    node0, node1 = "0-63,128-191", "64-127,192-255"
    devs = [node0] * 4 + [node1] * 4
    cpus0 = set(list(range(0, 64)) + list(range(128, 192)))
    cpus1 = set(list(range(64, 128)) + list(range(192, 256)))
    for n in (8, 16, 32, 11, 3):
        pl = filegrep.place_workers(n, devs)
        per_dev = [sum(1 for d, _ in pl if d == k) for k in range(8)]
        assert sum(per_dev) == n and max(per_dev) - min(per_dev) <= 1, per_dev
        for i, (d, cpus) in enumerate(pl):
            assert d == i % 8
            assert set(cpus) == (cpus0 if d < 4 else cpus1), (i, d)
    # the process may only use CPUs 0-31 and 64-95 (a cpuset / taskset): the local lists are cut to that
    pl = filegrep.place_workers(16, devs, allowed="0-31,64-95")
    for d, cpus in pl:
        assert set(cpus) == (set(range(0, 32)) if d < 4 else set(range(64, 96)))
    # ... only socket 0: the devices of socket 1 have no local CPU that is ours -> the process's mask
    pl = filegrep.place_workers(8, devs, allowed="0-15")
    assert all(set(cpus) == set(range(16)) for _, cpus in pl)
    # unknown topology (no sysfs entry): the process's mask
    pl = filegrep.place_workers(8, [None] * 8, allowed="0-7")
    assert all(set(cpus) == set(range(8)) for _, cpus in pl)
    # the reference's rule and no binding at all
    assert [cpus for _, cpus in filegrep.place_workers(6, devs, pin="cpu")] == [[i] for i in range(6)]
    assert all(set(cpus) == set(range(24)) for _, cpus in filegrep.place_workers(6, devs, allowed="0-23", pin="none"))
    # a one-GPU box: every worker on device 0 and its node
    assert all(d == 0 and set(cpus) == cpus0 for d, cpus in filegrep.place_workers(8, [node0]))


def test_reader_threads_scale_with_the_node(built):
    """GSCAN_READERS unset: 8 reader threads per device where the device's share of its NUMA node has CPUs to spare (the
    measured optimum on the one-GPU boxes), fewer on a node where 8 devices x 8 readers would outnumber the CPUs."""
    L = engine.lib()
    assert L.gscan_auto_readers(128, 1, 1) == 8   # the gpurun boxes: device 0 -> CPUs 0-63,128-191
    assert L.gscan_auto_readers(128, 2, 4) == 6   # four devices: 24 readers in all is what the page cache feeds (profiles/r05_a_n8_*)
    assert L.gscan_auto_readers(128, 4, 8) == 3   # 8-GPU node, 4 devices per socket
    assert L.gscan_auto_readers(32, 4, 4) == 4    # a smaller host: 8 CPUs per device, half for readers
    assert L.gscan_auto_readers(8, 4, 4) == 2     # never below 2
    assert L.gscan_auto_readers(0, 1, 1) == 8     # nothing known
    for cpus in (16, 64, 128, 256):
        for sharing, total in ((1, 1), (1, 2), (2, 2), (2, 4), (4, 8), (8, 8)):
            r = L.gscan_auto_readers(cpus, sharing, total)
            assert 2 <= r <= 8 and (r == 2 or (r * sharing * 2 <= cpus and r * total <= 24))  # never more than half the node's CPUs, nor than the host can feed


def test_detached_parent_forwards_signals(built, tmp_path):
    """GRAB_DETACH=1 (opt-in): the scan runs in a child and the parent only waits for its status.  A signal that reaches the
    parent alone (subprocess.terminate(), a supervisor) must end BOTH: the parent passes it on and takes it itself, and the
    child has asked for SIGKILL should the parent vanish.  Runs without a device too: there the child fails fast
    (no HIP device: exit 255), so what is checked is the status hand-over with and without the child, and that no process
    that names the test's directory is left a moment after the parent was signalled."""
    import signal
    import time

    d = tmp_path / "t"
    d.mkdir()
    (d / "f").write_bytes(b"foo\n" * 100)
    for detach in ("0", "1"):
        r = subprocess.run([built.bin_path(), "-r", "foo", str(d)], capture_output=True, env=dict(os.environ, GRAB_DETACH=detach))
        assert r.returncode in (0, 255)  # 255 on a box without a device (prepare fails loudly), 0 with one
    # a parent that is signalled while its child is still busy: use a directory tree large enough to take a moment
    big = tmp_path / "big"
    big.mkdir()
    for i in range(2000):
        (big / ("f%04d" % i)).write_bytes(b"x" * 10)
    p = subprocess.Popen([built.bin_path(), "-n", "2", "-r", "foo", str(big)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, GRAB_DETACH="1"))
    time.sleep(0.05)
    p.send_signal(signal.SIGTERM)
    rc = p.wait(timeout=30)
    assert rc in (-signal.SIGTERM, 0, 255)  # killed by the signal -- or already done
    time.sleep(0.3)
    out = subprocess.run(["pgrep", "-f", str(big)], capture_output=True, text=True).stdout.split()
    assert out == [], "a scanning child outlived its signalled parent: %s" % out


def test_visible_devices_are_narrowed_before_the_runtime_starts(built, tmp_path):
    """hipInit brings up every device it can see; what the input can use is known before the first HIP call (VERDICT r3 task
    3c): explicit files of one window need one device, `-n N` the first N -- taken from the caller's own HIP_VISIBLE_DEVICES
    list when there is one; GRAB_DEVICE / GRAB_DEVICES / GRAB_ALL_DEVICES leave the list alone.  (Runs without a device: the
    mark is printed before the runtime is touched.)"""
    f = tmp_path / "f.txt"
    f.write_bytes(b"foo\n" * 10)
    big = tmp_path / "big.bin"
    with open(big, "wb") as fh:
        fh.truncate((64 << 20) + 5)  # sparse: three 32 MiB windows under -L x 5

    def narrowed(argv, **env):
        e = {k: v for k, v in os.environ.items() if not k.startswith(("HIP_VISIBLE", "GRAB_", "GSCAN_"))}
        e.update(env, GRAB_TIMING="1")
        r = subprocess.run([built.bin_path()] + argv, capture_output=True, text=True, env=e, cwd=str(tmp_path))
        got = [ln.split("narrowed to ")[1] for ln in r.stderr.splitlines() if "visible devices narrowed to" in ln]
        return got[0] if got else None

    assert narrowed(["foo", "f.txt"]) == "0"
    assert narrowed(["foo", "f.txt"], HIP_VISIBLE_DEVICES="3,5,7") == "3"
    assert narrowed(["-L", "-L", "-L", "-L", "-L", "foo", "big.bin"]) == "0,1,2"
    assert narrowed(["-L", "-L", "-L", "-L", "-L", "foo", "big.bin"], HIP_VISIBLE_DEVICES="4,5") == "4,5"
    assert narrowed(["-n", "2", "-r", "foo", "."], HIP_VISIBLE_DEVICES="3,5,7") == "3,5"
    assert narrowed(["-n", "2", "-r", "foo", "."]) == "0,1"
    assert narrowed(["-r", "foo", "."]) is None  # a tree: how many windows it holds is not known up front
    assert narrowed(["foo", "f.txt"], GRAB_DEVICES="2") is None
    assert narrowed(["foo", "f.txt"], GRAB_DEVICE="1") is None
    assert narrowed(["foo", "f.txt"], GRAB_ALL_DEVICES="1") is None


def test_prefault_files_without_a_device(built, tmp_path):
    """gscan_prefault_files makes no HIP call: it can be called first thing in a process, with files that exist, files that do
    not, directories, no files at all, more blocks than the arena may hold -- and a second call is a no-op (once per process)."""
    f = tmp_path / "a.bin"
    f.write_bytes(b"x" * (3 << 20))
    code = (
        "import ctypes as C, sys, time\n"
        "from grab_amd import engine\n"
        "L = engine.lib()\n"
        "paths = [sys.argv[1].encode(), b'/nonexistent/file', sys.argv[2].encode()]\n"
        "arr = (C.c_char_p * len(paths))(*paths)\n"
        "assert L.gscan_prefault_files(1000, arr, len(paths)) == 0\n"
        "assert L.gscan_prefault_files(4, arr, len(paths)) == 0\n"   # once per process
        "assert L.gscan_prefault(18) == 0\n"
        "time.sleep(0.3)\n"                                           # (the helpers read and touch in the background)
        "print('ok')\n"
    )
    for env in ({}, {"GSCAN_BLOCK_MIB": "1"}, {"GSCAN_PREFAULT": "0"}):
        r = subprocess.run([sys.executable, "-c", code, str(f), str(tmp_path)], cwd=ROOT, capture_output=True, text=True, env=dict(os.environ, **env))
        assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-500:]
    r = subprocess.run([sys.executable, "-c", "from grab_amd import engine; L = engine.lib(); assert L.gscan_prefault_files(0, None, 0) == 0; print('ok')"], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-500:]


def test_n8_model_forecast_arithmetic():
    """scripts/n8_model.py: T(N) = F(N) + bytes / min(N x P, D) from measured terms (DESIGN.md 6's table, run L's terms)."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import n8_model

    f = n8_model.forecast(50.98, 150.0, 0.1833, 0.3038, 64 << 30)
    assert f["bound_by"].startswith("D") and abs(f["T1_s"] - 1.531) < 0.002 and abs(f["T8_s"] - 0.945) < 0.002
    assert abs(f["strong_scaling_efficiency"] - 0.202) < 0.002 and abs(f["efficiency_if_F8_were_F1"] - 0.298) < 0.002
    g = n8_model.forecast(50.0, 1000.0, 0.2, 0.0, 64 << 30)   # a host that can feed eight links: the links bound it
    assert g["bound_by"] == "the links" and abs(g["T8_s"] - (0.2 + 68.719476736 / 400.0)) < 0.002
    tree = n8_model.fake_pci_tree(os.path.join("/tmp", "grab_fake_pci_%d" % os.getpid()))
    assert len([d for d in os.listdir(tree) if d.startswith("0000:")]) == 8
    import shutil
    shutil.rmtree(tree)
