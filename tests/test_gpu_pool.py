"""The staging-block pool's slow paths, forced on every run (VERDICT r4 task 3): a pool capped far below the number of
reader threads, so that readers sleep on DMA events and on each other all the time; an allocation that fails while the pool
grows; the non-temporal reader copy.  Eight contexts on four device indices push a mix of file windows and small-file batches
through it, and every chunk's list is compared with libpcre's candidate set (tests/pool_driver.py).  The driver runs as a
subprocess because the ingest configuration is read once per process."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _drive(tmp_path, env, mib, per_device=2):
    # (GSCAN_SECOND_STREAM_MIB: every index's second copy stream is made after 16 MiB -- while pieces are in flight on the first)
    e = dict(os.environ, GSCAN_VIRTUAL_DEVICES="4", GSCAN_BLOCK_MIB="1", GSCAN_READERS="16", GSCAN_SECOND_STREAM_MIB="16", **env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "pool_driver.py"), "--dir", str(tmp_path), "--mib", str(mib), "--per-device", str(per_device)],
                       capture_output=True, text=True, env=e, timeout=900)
    assert r.stdout.strip(), r.stderr[-2000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert r.returncode == 0 and rec["mismatches"] == 0, (rec, r.stderr[-2000:])
    assert rec["devices"] == 4 and rec["contexts"] == 4 * per_device and rec["bytes"] >= 4 * per_device * (mib << 20)
    return rec


def test_pool_at_its_cap_with_every_block_in_flight(built, oracle_built, tmp_path):
    """Four blocks for sixteen readers per device, 2 GiB in all: a reader finds the free list empty almost every time -- it
    reaps, sleeps on the oldest DMA's event (the block taken out of its lane first) or, when the other readers hold every
    block, on the condition variable."""
    rec = _drive(tmp_path, {"GSCAN_POOL_CAP": "4"}, 256)
    for d, st in rec["pool"].items():
        assert st["cap"] == 4 and st["allocated"] <= 4, (d, st)
        assert st["event_waits"] > 50 and st["reader_waits"] > 50, (d, st)  # the slow paths ran, hundreds of times


def test_pool_stops_growing_when_the_runtime_refuses_a_block(built, oracle_built, tmp_path):
    """The third staging block of every device cannot be had: the pool stays at two blocks (no retry per piece) and every
    result is still right."""
    rec = _drive(tmp_path, {"GSCAN_POOL_CAP": "8", "GSCAN_FAIL_ALLOC_AFTER": "2"}, 96)
    for d, st in rec["pool"].items():
        assert st["allocated"] == 2 and st["cap"] == 2, (d, st)
        assert st["event_waits"] + st["reader_waits"] > 50, (d, st)


def test_no_staging_block_at_all_is_an_error_not_a_hang(built, oracle_built, tmp_path):
    """Not one staging block: every chunk fails with a device error (GSCAN_EHIP through gscan_wait), nothing hangs."""
    e = dict(os.environ, GSCAN_BLOCK_MIB="1", GSCAN_READERS="4", GSCAN_FAIL_ALLOC_AFTER="0", GSCAN_PREFAULT="0")
    code = ("import os, sys; sys.path.insert(0, %r); from grab_amd import engine\n"
            "open(%r, 'wb').write(b'x' * (3 << 20))\n"
            "c = engine.Context(0, 64 << 20); db = engine.Database('foobar'); fd = os.open(%r, os.O_RDONLY)\n"
            "c.submit_fd(db, fd, 0, 3 << 20)\n"
            "try:\n    c.wait_segs(); print('no error')\n"
            "except engine.EngineError as e:\n    print('error:', e)\n"
            "c.submit_fd(db, fd, 0, 100)\n"
            "try:\n    c.wait_segs(); print('no error')\n"
            "except engine.EngineError as e:\n    print('error:', e)\n") % (ROOT, str(tmp_path / "f.bin"), str(tmp_path / "f.bin"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, timeout=300)
    assert r.returncode == 0 and r.stdout.count("error:") == 2 and "staging block" in r.stdout, r.stdout + r.stderr[-2000:]


def test_non_temporal_reader_copy(built, oracle_built, tmp_path):
    """GSCAN_NT_COPY=1 (off by default: measured slower than pread at every reader count): pread into a bounce buffer, non-temporal
    copy into the block -- same results, through the same starved pool."""
    rec = _drive(tmp_path, {"GSCAN_POOL_CAP": "6", "GSCAN_NT_COPY": "1"}, 96)
    assert all(st["cap"] == 6 for st in rec["pool"].values())
