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
    "grab_filegrep_find_recursive", "grab_filegrep_engine_option", "grab_report_chunk_c", "grab_free",
]

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
        L.grab_free.argtypes = [C.c_void_p]
        L.grab_free.restype = None
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


def report_chunk(db, flags, path, content, off, starts):
    """grab_report_chunk_c: what the reference prints for one chunk, from the candidate list. Pure host."""
    import numpy as np

    buf = np.ascontiguousarray(np.frombuffer(content, np.uint8))
    st = np.ascontiguousarray(np.asarray(starts, np.uint32))
    out = C.c_void_p()
    n = C.c_size_t()
    rc = lib().grab_report_chunk_c(db._h, flags, os.fsencode(path) if path is not None else None,
                                   buf.ctypes.data if buf.size else None, buf.size, off,
                                   st.ctypes.data if st.size else None, st.size, C.byref(out), C.byref(n))
    if rc != 0:
        raise RuntimeError("grab_report_chunk_c failed")
    data = C.string_at(out, n.value)
    lib().grab_free(out)
    return data
