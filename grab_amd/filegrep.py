"""ctypes mirror of the reference's FileGrep interface (/root/reference/src/grab.h:55-85).

Same method names, argument meaning and error behaviour (0 / -1 + why()); the work is done
by libgrabhost.so -> libgscan.so on a HIP device.  Output goes to `out_fd` (default: the
process's stdout, like the reference).
"""
import ctypes as C
import os

from .build import lib_path

SYMBOLS = [
    "grab_filegrep_new", "grab_filegrep_free", "grab_filegrep_why", "grab_filegrep_recurse",
    "grab_filegrep_show_path", "grab_filegrep_config", "grab_filegrep_prepare", "grab_filegrep_find",
    "grab_filegrep_find_recursive", "grab_filegrep_engine_option", "grab_report_chunk_c", "grab_report_chunk_ends_c", "grab_report_chunk_ext_c", "grab_free", "grab_place_workers_c",
    "grab_filegrep_find3", "grab_filegrep_flush", "grab_walk_parallel", "grab_validate",
]

FTW_F = 0  # <ftw.h>
WALK_FN = C.CFUNCTYPE(None, C.c_char_p, C.c_void_p, C.c_void_p)

OFFSETS, NOLINE, SINGLE, PREFIX, COLOR = 1, 2, 4, 8, 16

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = lib_path("libgrabhost.so")
        if not os.path.exists(path):
            raise RuntimeError("libgrabhost.so is not built (%s): run __graft_entry__.build()" % path)
        C.CDLL(lib_path("libgscan.so"), mode=C.RTLD_GLOBAL)
        L = C.CDLL(path)
        L.grab_filegrep_new.restype = C.c_void_p
        L.grab_filegrep_free.argtypes = [C.c_void_p]
        L.grab_filegrep_free.restype = None
        L.grab_filegrep_why.argtypes = [C.c_void_p]
        L.grab_filegrep_why.restype = C.c_char_p
        L.grab_filegrep_recurse.argtypes = [C.c_void_p]
        L.grab_filegrep_recurse.restype = None
        L.grab_filegrep_show_path.argtypes = [C.c_void_p, C.c_int]
        L.grab_filegrep_show_path.restype = None
        L.grab_filegrep_config.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.grab_filegrep_config.restype = None
        L.grab_filegrep_prepare.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.grab_filegrep_find.argtypes = [C.c_void_p, C.c_char_p]
        L.grab_filegrep_find_recursive.argtypes = [C.c_void_p, C.c_char_p]
        L.grab_filegrep_engine_option.argtypes = [C.c_void_p, C.c_char_p, C.c_long]
        L.grab_report_chunk_c.argtypes = [C.c_void_p, C.c_uint, C.c_char_p, C.c_void_p, C.c_size_t, C.c_longlong,
                                          C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.grab_report_chunk_ends_c.argtypes = [C.c_void_p, C.c_uint, C.c_char_p, C.c_void_p, C.c_size_t, C.c_longlong,
                                               C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.grab_report_chunk_ext_c.argtypes = [C.c_void_p, C.c_uint, C.c_char_p, C.c_void_p, C.c_size_t, C.c_longlong,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.grab_place_workers_c.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_char_p), C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.c_void_p, C.c_size_t]
        L.grab_free.argtypes = [C.c_void_p]
        L.grab_free.restype = None
        L.grab_filegrep_find3.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
        L.grab_filegrep_flush.argtypes = [C.c_void_p]
        L.grab_walk_parallel.argtypes = [C.c_char_p, C.c_int, WALK_FN, C.c_void_p]
        L.grab_walk_parallel.restype = C.c_long
        L.grab_validate.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_char_p, C.c_size_t]
        _lib = L
    return _lib


class FileGrep:
    def __init__(self):
        self._h = C.c_void_p(lib().grab_filegrep_new())

    def why(self):
        return lib().grab_filegrep_why(self._h).decode("latin-1")

    def recurse(self):
        lib().grab_filegrep_recurse(self._h)

    def show_path(self, on):
        lib().grab_filegrep_show_path(self._h, 1 if on else 0)

    def config(self, cfg):
        """cfg: dict like the reference's map<string,size_t> (color noline offsets single low_mem chunk_size ...)."""
        for k, v in cfg.items():
            lib().grab_filegrep_config(self._h, k.encode(), int(v))

    def prepare(self, regex):
        if isinstance(regex, str):
            regex = regex.encode("latin-1")
        return lib().grab_filegrep_prepare(self._h, regex, len(regex))

    def find(self, path):
        return lib().grab_filegrep_find(self._h, os.fsencode(path))

    def find3(self, path, st=None, typeflag=FTW_F):
        """find(path, st, typeflag) -- the per-file entry the nftw callback and the worker threads use (grab.h:82).
        `st` is a `struct stat` buffer (c_stat(path) makes one); work may stay in flight until flush()."""
        if st is None:
            st = c_stat(path)
        return lib().grab_filegrep_find3(self._h, os.fsencode(path), st, typeflag)

    def flush(self):
        return lib().grab_filegrep_flush(self._h)

    def find_recursive(self, path):
        return lib().grab_filegrep_find_recursive(self._h, os.fsencode(path))

    def engine_option(self, name, value):
        return lib().grab_filegrep_engine_option(self._h, name.encode(), value)

    def close(self):
        if self._h:
            lib().grab_filegrep_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def c_stat(path):
    """A C `struct stat` of `path` (stat(2) through libc), as grab_filegrep_find3 wants it."""
    libc = C.CDLL(None, use_errno=True)
    buf = C.create_string_buffer(512)  # sizeof(struct stat) is 144 on x86-64 glibc
    if libc.stat(os.fsencode(path), buf) != 0:
        raise OSError(C.get_errno(), os.strerror(C.get_errno()), path)
    return buf


def walk_parallel(root, threads=4):
    """The `grab -n` tree walk: [(path, size)] of every regular file, in no particular order."""
    import threading

    got, lock = [], threading.Lock()

    def on_file(path, st, _arg):
        size = C.c_longlong.from_address(st + 48).value  # st_size: offset 48 in x86-64 glibc's struct stat
        with lock:
            got.append((path, size))

    fn = WALK_FN(on_file)
    n = lib().grab_walk_parallel(os.fsencode(root), threads, fn, None)
    assert n == len(got), (n, len(got))
    return got


def validate(regex, literal=False):
    """(rc, why): 0 fine, -1 PCRE rejects the pattern (the reference's message), -2 outside the engine's subset."""
    if isinstance(regex, str):
        regex = regex.encode("latin-1")
    why = C.create_string_buffer(512)
    rc = lib().grab_validate(regex, len(regex), 1 if literal else 0, why, 512)
    return rc, why.value.decode("latin-1")


def report_chunk_ext(db, flags, path, content, off, starts, ext, gather=None):
    """grab_report_chunk_ext_c: the chunk's output with the device's line pass (ext: n x 4 uint32 {m1, lb, le, goff}; gather: the
    printed lines' text or None)."""
    import numpy as np

    buf = np.ascontiguousarray(np.frombuffer(content, np.uint8))
    st = np.ascontiguousarray(np.asarray(starts, np.uint32))
    ex = np.ascontiguousarray(np.asarray(ext, np.uint32).reshape(-1))
    assert ex.size == 4 * st.size
    ga = np.ascontiguousarray(np.frombuffer(gather, np.uint8)) if gather is not None else None
    out = C.c_void_p()
    n = C.c_size_t()
    rc = lib().grab_report_chunk_ext_c(db._h, flags, os.fsencode(path) if path is not None else None, buf.ctypes.data if buf.size else None, buf.size, off,
                                       st.ctypes.data if st.size else None, None, ex.ctypes.data if ex.size else None,
                                       ga.ctypes.data if ga is not None and ga.size else (C.c_char_p(b"") if ga is not None else None), st.size, C.byref(out), C.byref(n))
    if rc != 0:
        raise RuntimeError("grab_report_chunk_ext_c failed")
    data = C.string_at(out, n.value)
    lib().grab_free(out)
    return data


def report_chunk(db, flags, path, content, off, starts, ends=None, clen=None):
    """grab_report_chunk_c: what the reference prints for one chunk, from the candidate list. Pure host.
    ends: the match ends the device measured (grab_report_chunk_ends_c); content may then be None (clen = its length)."""
    import numpy as np

    buf = np.ascontiguousarray(np.frombuffer(content, np.uint8)) if content is not None else np.zeros(0, np.uint8)
    st = np.ascontiguousarray(np.asarray(starts, np.uint32))
    out = C.c_void_p()
    n = C.c_size_t()
    if ends is None:
        rc = lib().grab_report_chunk_c(db._h, flags, os.fsencode(path) if path is not None else None,
                                       buf.ctypes.data if buf.size else None, buf.size, off,
                                       st.ctypes.data if st.size else None, st.size, C.byref(out), C.byref(n))
    else:
        en = np.ascontiguousarray(np.asarray(ends, np.uint32))
        assert en.size == st.size
        rc = lib().grab_report_chunk_ends_c(db._h, flags, os.fsencode(path) if path is not None else None,
                                            buf.ctypes.data if buf.size else None, buf.size if content is not None else clen, off,
                                            st.ctypes.data if st.size else None, en.ctypes.data if en.size else None, st.size,
                                            C.byref(out), C.byref(n))
    if rc != 0:
        raise RuntimeError("grab_report_chunk_c failed")
    data = C.string_at(out, n.value)
    lib().grab_free(out)
    return data


def place_workers(workers, dev_cpulists, allowed=None, pin=None, ncpu=512):
    """grab_place_workers_c: [(device, sorted CPU list)] for the workers of `grab -n workers` on a node whose devices have
    these local CPU lists (sysfs cpulist strings)."""
    import numpy as np

    ndev = len(dev_cpulists)
    arr = (C.c_char_p * ndev)(*[(x.encode() if x else None) for x in dev_cpulists])
    devs = (C.c_int * max(workers, 1))()
    nb = (ncpu + 7) // 8
    bits = np.zeros(max(workers, 1) * nb, np.uint8)
    rc = lib().grab_place_workers_c(workers, ndev, arr, allowed.encode() if allowed else None, pin.encode() if pin else None, devs, bits.ctypes.data, nb)
    if rc != 0:
        raise RuntimeError("grab_place_workers_c failed")
    out = []
    for i in range(workers):
        row = np.unpackbits(bits[i * nb:(i + 1) * nb], bitorder="little")
        out.append((devs[i], np.nonzero(row)[0].tolist()))
    return out
