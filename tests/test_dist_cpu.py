"""Multi-process plumbing of bench.py on CPU (gloo, world_size 2): rank setup from the env,
barrier, MAX-over-ranks timing, SUM of match counts, and file sharding.  No scanning happens
here -- the scan itself needs a HIP device and is covered by the `-m gpu` tests."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch

    import bench

    r, w, local = bench.dist_setup(world)
    dev = torch.device("cpu")
    bench.barrier(w, dev)
    slowest = bench.reduce_max(1.0 + r, w, dev)       # rank 1 is "slower"
    matches = bench.reduce_sum(100.0 * (r + 1), w, dev)
    mine = bench.shard(10, r, w)
    bench.barrier(w, dev)
    out.put((r, w, slowest, matches, mine))
    import torch.distributed as dist

    dist.destroy_process_group()


def test_two_rank_plumbing():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [g[0] for g in got] == [0, 1] and all(g[1] == 2 for g in got)
    assert all(g[2] == 2.0 for g in got), "MAX over ranks"
    assert all(g[3] == 300.0 for g in got), "SUM over ranks"
    assert got[0][4] == [0, 2, 4, 6, 8] and got[1][4] == [1, 3, 5, 7, 9]
    assert sorted(got[0][4] + got[1][4]) == list(range(10)), "every file scanned exactly once"


def test_gpus_must_match_world(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench

    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(SystemExit):
        bench.dist_setup(2)


def test_bench_checker_and_worker_rule():
    """bench.py's own checker (check_span: the timed launch's records against the oracle's candidate set) says "ok" for what the
    contract allows -- group starts, with or without further members of the groups -- and names what is wrong otherwise; the
    `-n` rule is four workers per device, at least eight."""
    import numpy as np

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import bench
    import scan_oracle as so

    text = b"xx abcdefghijklmnopqrstu  yy_0123456789abcdefgh zz short_id x" + b"q" * 40 + b" end"
    pat = "[A-Za-z_][A-Za-z0-9_]{15,}"
    cands = so.all_starts(pat.encode(), text)
    heads = so.group_starts(cands)
    assert len(heads) == 3 and len(cands) > len(heads)
    n = len(text)
    assert bench.check_span(heads, text, 0, 0, n, pat, so) is None
    assert bench.check_span(cands, text, 0, 0, n, pat, so) is None              # more of a group is allowed
    assert "missing" in bench.check_span(heads[1:], text, 0, 0, n, pat, so)     # a group start left out
    assert "not a candidate" in bench.check_span(np.sort(np.append(heads, 1)), text, 0, 0, n, pat, so)
    # a window cut out of a longer segment: text begins at a = 1000, the span checked is [1010, 1000 + n)
    assert bench.check_span(heads + 1000, text, 1000, 1010, 1000 + n, pat, so) in (None,) or heads[0] + 1000 < 1010
    assert [bench.pick_workers(k) for k in (1, 2, 4, 8)] == [8, 8, 16, 32]
