import base64
import ctypes as C
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
# the host matcher's step limit per attempt (include/gscan.h: gscan_resource_errors): the differential tests skip inputs
# on which an engine gives up, so a low limit only makes the pathological random patterns cheap
os.environ.setdefault("GSCAN_MATCH_LIMIT", "5000000")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (run on the MI355X box)")


def pytest_collection_modifyitems(config, items):
    """No test may run for ever: a child process that never ends (round 6, session Z: a lost wake-up in the reader pools) took a
    whole GPU session with it.  With pytest-timeout loaded every test without a limit of its own gets 15 minutes (the slowest
    takes about one); a --timeout on the command line stands."""
    if not config.pluginmanager.hasplugin("timeout") or config.getoption("timeout", None):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(900))


@pytest.fixture(scope="session")
def built():
    """Product libraries + CLI (hipcc cross-compiles without a GPU)."""
    import grab_amd

    grab_amd.build()
    return grab_amd


@pytest.fixture(scope="session")
def oracle_built():
    """The C restatement (oracle/grab_oracle + liboracle.so). Test infrastructure only."""
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "all"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return os.path.join(ROOT, "oracle")


@pytest.fixture(scope="session")
def liboracle(oracle_built):
    L = C.CDLL(os.path.join(oracle_built, "liboracle.so"))
    L.oracle_minlen.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
    L.oracle_scan_chunk.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_longlong, C.c_uint,
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.oracle_all_starts.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
    L.oracle_all_starts.restype = C.c_long
    L.oracle_free.argtypes = [C.c_void_p]
    L.oracle_free.restype = None
    L.oracle_resource_errors.restype = C.c_long
    return L


def load_golden():
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases:
        if "stdout_b64" in c:
            c["stdout"] = base64.b64decode(c["stdout_b64"])
    return cases


GOLDEN = load_golden()


def golden_ids(cases):
    return [c["name"] for c in cases]


def split_args(args):
    """reference argv -> (flags list, pattern, paths)."""
    flags, rest = [], []
    i = 0
    while i < len(args):
        a = args[i]
        if a.startswith("-") and not rest:
            flags.append(a)
            if a == "-n":
                i += 1
                flags.append(args[i])
        else:
            rest.append(a)
        i += 1
    return flags, rest[0], rest[1:]


def has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False
