"""Input recipes shared by tests/golden/make_golden.py and the tests: small JSON-able
descriptions that are turned into byte strings / files deterministically."""
import base64
import os

import numpy as np

from grab_amd import synth


def big_txt():
    """SURVEY Appendix A.5 `big.txt`: chunk-boundary fixture for 32 MiB chunks."""
    C = 1 << 25
    stride = C - 4096
    size = 2 * stride + 5_000_000
    buf = np.full(size, ord("."), np.uint8)
    buf[79::80] = 10
    nd = np.frombuffer(b"NEEDLE", np.uint8)
    buf[C:C + 3] = np.frombuffer(b"DLE", np.uint8)
    for at in (100, stride + 1000, C - 6, stride, 2 * stride + 2000, size - 6):
        buf[at:at + 6] = nd
    return buf


def big2_txt():
    """SURVEY Appendix A.5 `big2.txt`: a literal straddling a chunk end, a run straddling a chunk start."""
    C = 1 << 25
    stride = C - 4096
    size = stride + 1_000_000
    buf = np.full(size, ord("."), np.uint8)
    buf[79::80] = 10
    buf[C - 3:C + 3] = np.frombuffer(b"NEEDLE", np.uint8)
    buf[stride - 10:stride + 30] = ord("a")
    return buf


def build(recipe, cache=None):
    """recipe -> uint8 numpy array."""
    key = repr(sorted(recipe.items()))
    if cache is not None and key in cache:
        return cache[key]
    kind = recipe["kind"]
    if kind == "bytes":
        buf = np.frombuffer(base64.b64decode(recipe["b64"]), np.uint8).copy()
    elif kind == "synth":
        buf = synth.text(recipe["nbytes"], recipe.get("k", 0))
        if "plant" in recipe:
            needle, count = recipe["plant"]
            synth.plant(buf, needle.encode(), count, recipe.get("k", 0))
    elif kind == "big":
        buf = big_txt()
    elif kind == "big2":
        buf = big2_txt()
    else:
        raise ValueError(kind)
    if cache is not None:
        cache.clear()  # keep at most one big buffer alive
        cache[key] = buf
    return buf


def materialize(recipe, path, cache=None):
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    build(recipe, cache).tofile(path)


def db_candidates(db, data):
    """Every offset the engine may report for `data` (the candidate set), from the database's own class tables: the
    union over its alternatives of the offsets where that alternative's DEVICE window -- the window plus, for patterns
    with \\b ^ $ ..., one context byte before / after it -- fits and matches, shifted to the match start.  For patterns
    without context that is every offset at which the pattern matches.  (The tables themselves are pinned against libpcre
    and Python's re in tests/test_pattern.py; the host's part of context patterns against the reference's outputs.)"""
    import os
    import sys

    import numpy as np

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import scan_oracle as so

    out = [np.zeros(0, np.int64)]
    for a in range(db.info.n_windows):
        tables, shift = db.dev_window(a)
        out.append(so.window_starts(data, tables) + shift)
    return np.unique(np.concatenate(out))


def engine_list(db, data):
    """The list the engine hands to gscan_next_match for `data`, computed on the host from the candidate set: the start of every
    group of consecutive candidates -- or, for a database whose candidates the device confirms itself (info.vm), EVERY device
    hit its VM filter keeps (gscan_vm_filter is the kernel's cold path, same source)."""
    import numpy as np
    import scan_oracle as so

    cands = db_candidates(db, data)
    if db.info.resolve:  # start windows, every offset where one fits (no group-start compression): what k_resolve is handed
        return cands.astype(np.uint32)
    if db.info.vm:
        return db.vm_filter(np.ascontiguousarray(data), cands.astype(np.uint32)).astype(np.uint32)
    return so.group_starts(cands).astype(np.uint32)


END_ASK, END_CAPTURES, END_LOOK = 0, 0xfffffffe, 0x80000000  # GSCAN_END_* (include/gscan.h)


def resolved_list(db, data):
    """What the device's resolve pass (k_resolve) makes of engine_list() for a database with info.resolve: the offsets at which the
    pattern's VM program -- run on the host here, same source -- finds a match with the chunk's real bytes in front of them, and
    per offset the match's end, END_CAPTURES (the match sets a capturing group) or END_ASK (the VM gave up: the host decides)."""
    import numpy as np

    return db.vm_resolve(np.ascontiguousarray(data), engine_list(db, data))
