"""ctypes binding of include/gscan.h -- the C ABI of the gfx950 scan engine.

Mirrors the header one to one; see the header for what each entry point replaces in
the reference (/root/reference/src/grab.cc:106-178).  No scanning happens in Python and
there is no fallback: if libgscan.so is missing or no HIP device opens, this raises.
"""
import ctypes as C
import os

import numpy as np

from .build import lib_path

OK, UNSUPPORTED = 0, 1
EINVAL, ENOMEM, EHIP, EBUSY, EEMPTY, ETOOBIG, EIO = -1, -2, -3, -4, -5, -6, -7
LITERAL = 1
PCRE_CHECKED = 2  # GSCAN_PCRE_CHECKED: pcre_compile has accepted the text (what FileGrep::prepare passes)
TIER_NULL, TIER_LITERAL, TIER_CLASSRUN, TIER_BUCKET, TIER_ANCHORED = 0, 1, 2, 3, 4
SLOTS = 3  # GSCAN_SLOTS (include/gscan.h): chunks one context keeps in flight

# every symbol include/gscan.h declares (what a binding of the engine needs) ...
SYMBOLS = [
    "gscan_compile", "gscan_free", "gscan_db_info", "gscan_match_info", "gscan_next_match", "gscan_next_listed", "gscan_next_resolved", "gscan_db_first", "gscan_resource_errors",
    "gscan_open", "gscan_close", "gscan_strerror", "gscan_device_count", "gscan_device_cpulist", "gscan_parse_cpulist",
    "gscan_acquire", "gscan_block_size", "gscan_prefault", "gscan_prefault_files", "gscan_ingest_info", "gscan_submit", "gscan_submit_segs", "gscan_submit_fd", "gscan_submit_files",
    "gscan_last_file_errors", "gscan_wait", "gscan_wait_segs", "gscan_last_ext", "gscan_last_gather", "gscan_last_ends",
    "gscan_scan_device", "gscan_dev_sync", "gscan_dev_fetch", "gscan_set_capacity", "gscan_set_option", "gscan_kernel_time",
]
# ... and include/gscan_test.h (test and diagnostic hooks: the compiler's tables, the matcher's and the VM's verdict at one offset, the pool's counters)
TEST_SYMBOLS = [
    "gscan_db_class", "gscan_db_alt_class", "gscan_match_at", "gscan_match_end", "gscan_tail_positions", "gscan_db_dev_window",
    "gscan_vm_verdict", "gscan_vm_match", "gscan_vm_resolve", "gscan_vm_filter", "gscan_vm_pair", "gscan_prefix_viable",
    "gscan_pool_stats", "gscan_auto_readers", "gscan_pci_cpulist",
]


class Info(C.Structure):
    _fields_ = [("tier", C.c_int), ("minlen", C.c_int), ("n_classes", C.c_int), ("has_tail", C.c_int),
                ("tail_extra", C.c_uint32), ("anchor_off", C.c_int), ("anchor_len", C.c_int),
                ("is_literal", C.c_int), ("n_alts", C.c_int), ("has_context", C.c_int), ("lines_ok", C.c_int), ("exact", C.c_int), ("vm", C.c_int), ("gapped", C.c_int), ("textfree", C.c_int), ("ends_ok", C.c_int), ("resolve", C.c_int), ("reach", C.c_int), ("n_windows", C.c_int)]


class Cursor(C.Structure):
    """gscan_cursor (include/gscan.h): per-chunk state of gscan_next_match; `ready = 0` before the first call."""
    _fields_ = [("li", C.c_size_t), ("ntails", C.c_uint32), ("ready", C.c_uint32), ("tails", C.c_uint32 * 132),
                ("next_at", C.c_uint32 * 65), ("next_known", C.c_uint8 * 65)]


class Seg(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("len", C.c_uint32), ("_pad", C.c_uint32)]


class File(C.Structure):
    """gscan_file (include/gscan.h): one small file of a gscan_submit_files batch."""
    _fields_ = [("path", C.c_char_p), ("fd", C.c_int), ("oflags", C.c_int), ("len", C.c_uint32)]


class DevResult(C.Structure):
    _fields_ = [("recs", C.c_void_p), ("desc", C.c_void_p),
                ("n_tiles", C.c_uint64), ("tile_bytes", C.c_uint32), ("total", C.c_uint64),
                ("overflow", C.c_int), ("ends", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = lib_path("libgscan.so")
        try:
            # torch bundles a HIP runtime with the same SONAME (libamdhip64.so.7); loading torch first
            # makes libgscan.so bind to that one copy, so device pointers and streams are shared.
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(path):
            raise RuntimeError("libgscan.so is not built (%s): run __graft_entry__.build()" % path)
        L = C.CDLL(path)
        L.gscan_compile.argtypes = [C.c_char_p, C.c_size_t, C.c_uint, C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                                    C.c_char_p, C.c_size_t]
        L.gscan_compile.restype = C.c_int
        L.gscan_free.argtypes = [C.c_void_p]
        L.gscan_free.restype = None
        L.gscan_db_info.argtypes = [C.c_void_p, C.POINTER(Info)]
        L.gscan_db_class.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.gscan_db_alt_class.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
        L.gscan_match_at.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32]
        L.gscan_match_at.restype = C.c_int
        L.gscan_match_end.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32]
        L.gscan_match_end.restype = C.c_uint32
        L.gscan_match_info.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        L.gscan_tail_positions.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.gscan_next_match.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32,
                                       C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.gscan_tail_positions.restype = C.c_size_t
        L.gscan_db_dev_window.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.gscan_open.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
        L.gscan_close.argtypes = [C.c_void_p]
        L.gscan_close.restype = None
        L.gscan_strerror.argtypes = [C.c_void_p]
        L.gscan_strerror.restype = C.c_char_p
        L.gscan_device_count.restype = C.c_int
        L.gscan_acquire.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.gscan_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64]
        L.gscan_wait.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.POINTER(C.c_uint32)),
                                 C.POINTER(C.c_size_t), C.POINTER(C.c_void_p)]
        L.gscan_block_size.restype = C.c_size_t
        L.gscan_last_ext.argtypes = [C.c_void_p]
        L.gscan_last_ext.restype = C.POINTER(C.c_uint32)
        L.gscan_last_gather.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        L.gscan_last_gather.restype = C.c_void_p
        L.gscan_last_ends.argtypes = [C.c_void_p]
        L.gscan_last_ends.restype = C.POINTER(C.c_uint32)
        L.gscan_next_listed.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(Cursor), C.c_uint32,
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.gscan_submit_segs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Seg), C.c_size_t, C.c_uint64]
        L.gscan_submit_fd.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_size_t, C.c_uint64]
        L.gscan_prefault.argtypes = [C.c_size_t]
        L.gscan_submit_files.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(File), C.c_size_t, C.c_uint64]
        L.gscan_last_file_errors.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        L.gscan_last_file_errors.restype = C.POINTER(C.c_int)
        L.gscan_wait_segs.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.POINTER(C.c_uint32)),
                                      C.POINTER(C.POINTER(C.c_size_t)), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p)]
        L.gscan_scan_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Seg), C.c_size_t, C.c_void_p,
                                        C.POINTER(DevResult)]
        L.gscan_dev_sync.argtypes = [C.c_void_p, C.POINTER(DevResult)]
        L.gscan_dev_fetch.argtypes = [C.c_void_p, C.POINTER(DevResult), C.c_size_t, C.c_void_p, C.c_size_t]
        L.gscan_dev_fetch.restype = C.c_long
        L.gscan_set_capacity.argtypes = [C.c_void_p, C.c_size_t]
        L.gscan_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_long]
        L.gscan_kernel_time.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]
        L.gscan_resource_errors.argtypes = []
        L.gscan_resource_errors.restype = C.c_uint64
        L.gscan_ingest_info.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.gscan_ingest_info.restype = None
        L.gscan_prefault_files.argtypes = [C.c_size_t, C.POINTER(C.c_char_p), C.c_size_t]
        L.gscan_pool_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.gscan_device_cpulist.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
        L.gscan_pci_cpulist.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
        L.gscan_parse_cpulist.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.c_size_t]
        L.gscan_parse_cpulist.restype = C.c_long
        L.gscan_vm_verdict.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32]
        L.gscan_vm_match.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
        L.gscan_vm_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        L.gscan_vm_resolve.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.gscan_vm_resolve.restype = C.c_long
        L.gscan_vm_filter.restype = C.c_long
        L.gscan_vm_pair.argtypes = [C.c_void_p, C.c_uint, C.c_uint]
        L.gscan_prefix_viable.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        _lib = L
    return _lib


class Unsupported(ValueError):
    """The pattern is valid but outside the GPU engine's subset (GSCAN_UNSUPPORTED)."""


class EngineError(RuntimeError):
    pass


class Database:
    """A compiled pattern (gscan_db). Host only; needs no device."""

    def __init__(self, pattern, literal=False, pcre_checked=False):
        if isinstance(pattern, str):
            pattern = pattern.encode("latin-1")
        self.pattern = pattern
        self._h = C.c_void_p()
        ml = C.c_int(0)
        err = C.create_string_buffer(200)
        rc = lib().gscan_compile(pattern, len(pattern), (LITERAL if literal else 0) | (PCRE_CHECKED if pcre_checked else 0), C.byref(self._h), C.byref(ml), err, 200)
        if rc == UNSUPPORTED:
            raise Unsupported(err.value.decode())
        if rc != OK:
            raise ValueError("malformed pattern: " + err.value.decode())
        self.minlen = ml.value
        info = Info()
        lib().gscan_db_info(self._h, C.byref(info))
        self.info = info

    def class_table(self, pos, alt=0):
        t = np.zeros(256, np.uint8)
        rc = lib().gscan_db_alt_class(self._h, alt, pos, t.ctypes.data, None)
        if rc != OK:
            raise ValueError("no class at position %d of alternative %d" % (pos, alt))
        return t.astype(bool)

    def alt_len(self, alt):
        """Window length of alternative `alt` (alternatives are in PCRE's priority order)."""
        t = np.zeros(256, np.uint8)
        n = C.c_int()
        if lib().gscan_db_alt_class(self._h, alt, 0, t.ctypes.data, C.byref(n)) != OK:
            raise ValueError("no alternative %d" % alt)
        return n.value

    def match_at(self, content, p):
        buf = np.frombuffer(content, np.uint8)
        return bool(lib().gscan_match_at(self._h, buf.ctypes.data, buf.size, p))

    def match_end(self, content, start):
        buf = np.frombuffer(content, np.uint8)
        return int(lib().gscan_match_end(self._h, buf.ctypes.data, buf.size, start))

    def match_info(self, content, p, subject_start=None):
        """(kind, end) of a match attempt AT p with the subject starting at subject_start (default: at p itself):
        kind 0 no match, 1 match, 2 match through a capturing group (the reference ends the chunk there)."""
        buf = np.frombuffer(content, np.uint8)
        e = C.c_uint32()
        s0 = p if subject_start is None else subject_start
        return int(lib().gscan_match_info(self._h, buf.ctypes.data, buf.size, s0, p, C.byref(e))), e.value

    def next_match(self, content, starts, cursor, s):
        """gscan_next_match: one pcre_exec call of the reference's loop -- (rc, m0, m1), rc 0 none / 1 match / 2 match that set a
        capturing group.  `starts`: the uint32 list the engine returned for the chunk; `cursor`: a Cursor() kept across the
        calls for one chunk (s may only grow)."""
        buf = np.frombuffer(content, np.uint8)
        st = np.ascontiguousarray(starts, np.uint32)
        m0, m1 = C.c_uint32(), C.c_uint32()
        rc = lib().gscan_next_match(self._h, buf.ctypes.data, buf.size, st.ctypes.data, st.size, C.byref(cursor), s, C.byref(m0), C.byref(m1))
        return int(rc), m0.value, m1.value

    def vm_verdict(self, content, p, subject_start=0):
        """The device VM's answer AT p, run on the host: 0 no match starts at p, 1 one does, 2 gave up, -1 no VM program."""
        buf = np.frombuffer(content, np.uint8)
        return int(lib().gscan_vm_verdict(self._h, buf.ctypes.data, buf.size, subject_start, p))

    def vm_match(self, content, p, subject_start=0):
        """The VM's full answer AT p: (verdict, end, captures) -- end / captures are meaningful for verdict 1 only."""
        buf = np.frombuffer(content, np.uint8)
        end, cap = C.c_uint32(0), C.c_int(0)
        v = int(lib().gscan_vm_match(self._h, buf.ctypes.data, buf.size, subject_start, p, C.byref(end), C.byref(cap)))
        return v, int(end.value), int(cap.value)

    def vm_resolve(self, content, hits):
        """k_resolve on the host: (starts, ends) of the hits at which the VM finds a match or gives up."""
        buf = np.ascontiguousarray(np.frombuffer(content, np.uint8))
        h = np.ascontiguousarray(np.asarray(hits, np.uint32))
        st, en = np.zeros(h.size, np.uint32), np.zeros(h.size, np.uint32)
        k = lib().gscan_vm_resolve(self._h, buf.ctypes.data if buf.size else None, buf.size, h.ctypes.data if h.size else None, h.size,
                                   st.ctypes.data if h.size else C.c_void_p(8), en.ctypes.data if h.size else C.c_void_p(8))
        if k < 0:
            raise EngineError("gscan_vm_resolve: no VM program")
        return st[:k].copy(), en[:k].copy()

    def vm_pair(self, b0, b1):
        """The device's two-byte table: 1 a match may begin with b0 b1, 0 none can, -1 no table."""
        return int(lib().gscan_vm_pair(self._h, b0, b1))

    def prefix_viable(self, prefix):
        """May a match begin at offset 0 of some subject that starts with `prefix`? (the probe the two-byte table is built from)"""
        buf = np.frombuffer(bytes(prefix), np.uint8)
        return bool(lib().gscan_prefix_viable(self._h, buf.ctypes.data if buf.size else None, buf.size))

    def vm_filter(self, content, hits):
        """What K3 does with its filter hits when info.vm is set: the hits it keeps (uint32 array)."""
        buf = np.frombuffer(content, np.uint8)
        h = np.ascontiguousarray(hits, np.uint32)
        out = np.zeros(h.size, np.uint32)
        n = lib().gscan_vm_filter(self._h, buf.ctypes.data if buf.size else None, buf.size, h.ctypes.data if h.size else None, h.size, out.ctypes.data if h.size else None)
        if n < 0:
            raise ValueError("the pattern's candidates are not confirmed on the device")
        return out[:n]

    def dev_window(self, alt):
        """What the kernels scan for alternative `alt`: ([membership table per device window position], shift)."""
        n, sh = C.c_int(), C.c_int()
        if lib().gscan_db_dev_window(self._h, alt, 0, None, C.byref(n), C.byref(sh)) != OK:
            raise ValueError("no alternative %d" % alt)
        tabs = []
        for pos in range(n.value):
            t = np.zeros(256, np.uint8)
            lib().gscan_db_dev_window(self._h, alt, pos, t.ctypes.data, None, None)
            tabs.append(t.astype(bool))
        return tabs, sh.value

    def tail_positions(self, clen):
        out = np.zeros(130, np.uint32)
        n = lib().gscan_tail_positions(self._h, clen, out.ctypes.data, out.size)
        return out[:n]

    def close(self):
        if self._h:
            lib().gscan_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """A device context (gscan_ctx): one per worker thread."""

    def __init__(self, device=0, max_chunk=1 << 30):
        self._h = C.c_void_p()
        rc = lib().gscan_open(device, max_chunk, C.byref(self._h))
        if rc != OK:
            raise EngineError("gscan_open(device=%d) failed with %d: no usable HIP device" % (device, rc))
        self._keep = None

    def _chk(self, rc, what):
        if rc != OK:
            raise EngineError("%s failed (%d): %s" % (what, rc, lib().gscan_strerror(self._h).decode()))

    def set_option(self, name, value):
        self._chk(lib().gscan_set_option(self._h, name.encode(), value), "gscan_set_option(%s)" % name)

    def set_capacity(self, n):
        self._chk(lib().gscan_set_capacity(self._h, n), "gscan_set_capacity")

    # ---- host-chunk path ----
    def submit(self, db, data, tag=0):
        buf = np.ascontiguousarray(np.frombuffer(data, np.uint8))
        self._keep = buf
        self._chk(lib().gscan_submit(self._h, db._h, buf.ctypes.data if buf.size else None, buf.size, tag), "gscan_submit")

    def wait(self):
        tag = C.c_uint64()
        ptr = C.POINTER(C.c_uint32)()
        n = C.c_size_t()
        content = C.c_void_p()
        self._chk(lib().gscan_wait(self._h, C.byref(tag), C.byref(ptr), C.byref(n), C.byref(content)), "gscan_wait")
        starts = np.ctypeslib.as_array(ptr, shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint32)
        return tag.value, starts

    def scan(self, db, data):
        """Candidate group starts of one chunk (ascending uint32; see gscan_wait in include/gscan.h)."""
        self.submit(db, data)
        return self.wait()[1]

    def last_ext(self, n):
        """The line extents {m1, lb, le, goff} of the n records the last wait() returned (option "line_extents"), or None."""
        p = lib().gscan_last_ext(self._h)
        if not p:
            return None
        return np.ctypeslib.as_array(p, shape=(n * 4,)).copy().reshape(n, 4) if n else np.zeros((0, 4), np.uint32)

    def last_gather(self):
        """The text of the printed lines the device gathered for the last wait()'s chunk (uint8 array), or None."""
        n = C.c_size_t()
        p = lib().gscan_last_gather(self._h, C.byref(n))
        if not p:
            return None
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint8)

    def last_ends(self, n):
        """The match ends of the n records the last wait() returned (option "match_ends"; 0 = left to the host), or None."""
        p = lib().gscan_last_ends(self._h)
        if not p:
            return None
        return np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, np.uint32)

    def submit_fd(self, db, fd, offset, length, tag=0):
        """A range of an open file, read by the engine's reader threads straight into pinned blocks (gscan_submit_fd)."""
        self._chk(lib().gscan_submit_fd(self._h, db._h, fd, offset, length, tag), "gscan_submit_fd")

    def submit_files(self, db, files, tag=0):
        """Small files by name (or open descriptor), read by the engine's reader threads, one launch over all of them
        (gscan_submit_files).  files: (path | fd, length) pairs."""
        arr = (File * max(1, len(files)))()
        for i, (src, ln) in enumerate(files):
            if isinstance(src, int):
                arr[i].path, arr[i].fd = None, src
            else:
                arr[i].path, arr[i].fd = os.fsencode(src), -1
            arr[i].oflags, arr[i].len = os.O_RDONLY, ln
        self._chk(lib().gscan_submit_files(self._h, db._h, arr, len(files), tag), "gscan_submit_files")

    def last_file_errors(self):
        """Per-segment status of the file batch the last wait_segs() returned (None: it was not one)."""
        n = C.c_size_t()
        p = lib().gscan_last_file_errors(self._h, C.byref(n))
        return [p[i] for i in range(n.value)] if p else None

    def submit_batch(self, db, parts, tag=0):
        """Several small inputs (bytes-like) packed into the slot's pinned block at 16-byte aligned offsets and
        scanned in one launch (gscan_acquire + gscan_submit_segs).  Returns the (offset, len) table used."""
        cap = lib().gscan_block_size()
        buf = C.c_void_p()
        self._chk(lib().gscan_acquire(self._h, cap, C.byref(buf)), "gscan_acquire")
        segs, at = [], 0
        for part in parts:
            arr = np.frombuffer(part, np.uint8)
            at = (at + 15) & ~15
            if at + arr.size > cap:
                raise ValueError("batch does not fit one %d-byte block" % cap)
            C.memmove(buf.value + at, arr.ctypes.data, arr.size)
            segs.append((at, arr.size))
            at += arr.size
        arr = self.make_segs(segs)
        self._chk(lib().gscan_submit_segs(self._h, db._h, buf, arr, len(segs), tag), "gscan_submit_segs")
        return segs

    def wait_segs(self):
        """(tag, [starts of segment 0, starts of segment 1, ...], has_content) of the oldest chunk in flight."""
        tag = C.c_uint64()
        ptr = C.POINTER(C.c_uint32)()
        first = C.POINTER(C.c_size_t)()
        nseg = C.c_size_t()
        content = C.c_void_p()
        self._chk(lib().gscan_wait_segs(self._h, C.byref(tag), C.byref(ptr), C.byref(first), C.byref(nseg), C.byref(content)), "gscan_wait_segs")
        out = []
        for i in range(nseg.value):
            a, b = first[i], first[i + 1]
            out.append(np.ctypeslib.as_array(ptr, shape=(b,))[a:b].copy() if b > a else np.zeros(0, np.uint32))
        return tag.value, out, bool(content.value)

    # ---- device-resident path ----
    @staticmethod
    def make_segs(segs):
        arr = (Seg * len(segs))()
        for i, (off, ln) in enumerate(segs):
            arr[i].offset, arr[i].len = off, ln
        return arr

    def scan_device(self, db, dev_ptr, segs, stream=None):
        """segs: list of (offset, len) or an array from make_segs (reuse it across steps)."""
        arr = segs if isinstance(segs, C.Array) else self.make_segs(segs)
        res = DevResult()
        self._chk(lib().gscan_scan_device(self._h, db._h, dev_ptr, arr, len(segs), stream, C.byref(res)), "gscan_scan_device")
        self._segarr = arr
        return res

    def dev_sync(self, res):
        self._chk(lib().gscan_dev_sync(self._h, C.byref(res)), "gscan_dev_sync")
        return int(res.total), bool(res.overflow)

    def dev_fetch(self, res, seg):
        n = lib().gscan_dev_fetch(self._h, C.byref(res), seg, None, 0)
        if n < 0:
            self._chk(int(n), "gscan_dev_fetch")
        out = np.zeros(max(n, 1), np.uint32)
        n2 = lib().gscan_dev_fetch(self._h, C.byref(res), seg, out.ctypes.data, out.size)
        if n2 < 0:
            self._chk(int(n2), "gscan_dev_fetch")
        return out[:n2]

    def pool_stats(self):
        """{allocated, cap, event_waits, reader_waits} of the device's staging-block pool (gscan_pool_stats)."""
        out = (C.c_uint64 * 4)()
        self._chk(lib().gscan_pool_stats(self._h, out), "gscan_pool_stats")
        return {"allocated": out[0], "cap": out[1], "event_waits": out[2], "reader_waits": out[3]}

    def kernel_time(self, reset=True):
        s = C.c_double()
        n = C.c_uint64()
        self._chk(lib().gscan_kernel_time(self._h, C.byref(s), C.byref(n), 1 if reset else 0), "gscan_kernel_time")
        return s.value, n.value

    def close(self):
        if self._h:
            lib().gscan_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_count():
    return int(lib().gscan_device_count())


def ingest_info():
    """{block_bytes, readers, copy_streams} of the host -> HBM ingest (GSCAN_BLOCK_MIB / GSCAN_READERS / GSCAN_COPY_STREAMS)."""
    b, r, c = C.c_size_t(), C.c_int(), C.c_int()
    lib().gscan_ingest_info(C.byref(b), C.byref(r), C.byref(c))
    return {"block_bytes": b.value, "readers": r.value, "copy_streams": c.value}


def parse_cpulist(text):
    """sysfs cpulist ("0-3,8,10-11") -> [cpu, ...]."""
    if isinstance(text, str):
        text = text.encode()
    n = lib().gscan_parse_cpulist(text, None, 0)
    buf = (C.c_int * max(n, 1))()
    lib().gscan_parse_cpulist(text, buf, n)
    return list(buf[:n])


def pci_cpulist(sysfs_root, busid):
    """local_cpulist of PCI device `busid` under `sysfs_root` (None if unknown): where a GPU's host-side threads belong."""
    buf = C.create_string_buffer(1024)
    n = lib().gscan_pci_cpulist(os.fsencode(sysfs_root), busid.encode(), buf, 1024)
    return buf.value.decode() if n >= 0 else None


def device_cpulist(device=0):
    buf = C.create_string_buffer(1024)
    n = lib().gscan_device_cpulist(device, buf, 1024)
    return buf.value.decode() if n >= 0 else None


def resource_errors():
    """Match attempts the host matcher abandoned at its resource limits so far (process-wide)."""
    return int(lib().gscan_resource_errors())
