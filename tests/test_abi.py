"""The C-ABI libraries load and export every symbol the headers declare; without a HIP
device the product refuses to run (no CPU scanning path)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT, has_gpu


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(g(?:scan|rab)_[a-z_0-9]+)\s*\(", text)))


def _exported(lib):
    """The library's dynamic symbol table: every defined global function / object a client could bind."""
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    return sorted(ln.split()[2] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] in "TtDdBbRrWwVv")


def test_gscan_exports(built):
    """libgscan.so exports EXACTLY what its two headers declare -- gscan.h, the engine's C ABI, and gscan_test.h, the test and
    diagnostic hooks -- and nothing else: no C++ internals, no unprefixed helpers (built with -fvisibility=hidden, GSCAN_API)."""
    from grab_amd import engine

    names, hooks = _declared("gscan.h"), _declared("gscan_test.h")
    assert sorted(engine.SYMBOLS) == names, "engine.py binds exactly what gscan.h declares"
    assert sorted(engine.TEST_SYMBOLS) == hooks, "... and gscan_test.h"
    assert not set(names) & set(hooks)
    L = C.CDLL(built.lib_path("libgscan.so"))
    for n in names + hooks:
        assert hasattr(L, n), n
    extra = [x for x in _exported(built.lib_path("libgscan.so")) if x not in names + hooks and not x.startswith("__hip_") and x not in ("_init", "_fini")]
    assert extra == [], "libgscan.so exports symbols no header declares: %s" % extra[:10]


def test_grab_host_exports(built):
    from grab_amd import filegrep

    names = _declared("grab_host.h")
    assert sorted(filegrep.SYMBOLS) == names
    C.CDLL(built.lib_path("libgscan.so"), mode=C.RTLD_GLOBAL)
    L = C.CDLL(built.lib_path("libgrabhost.so"))
    for n in names:
        assert hasattr(L, n), n


def test_engine_has_no_pcre_dependency(built):
    out = subprocess.run(["ldd", built.lib_path("libgscan.so")], capture_output=True, text=True).stdout
    assert "pcre" not in out


def test_product_does_not_touch_oracle():
    """Nothing under grab_amd/ may import, link or execute oracle/ (the checker is not the product)."""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "grab_amd")):
        for f in files:
            if f.endswith((".py", ".cc", ".h", ".hip", "Makefile")):
                text = open(os.path.join(dp, f), errors="replace").read()
                if re.search(r"scan_oracle|liboracle|grab_oracle|oracle/", text):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


@pytest.mark.skipif(has_gpu(), reason="only meaningful on a box without a HIP device")
def test_fails_loudly_without_device(built, tmp_path):
    from grab_amd import engine, filegrep

    with pytest.raises(engine.EngineError):
        engine.Context()
    g = filegrep.FileGrep()
    assert g.prepare("foo") == -1
    assert "HIP device" in g.why()
    f = tmp_path / "t.txt"
    f.write_text("foo\n")
    r = subprocess.run([built.bin_path(), "foo", str(f)], capture_output=True, text=True)
    assert r.returncode == 255 and r.stdout == "" and "HIP device" in r.stderr
    # the same through both process models of the command line: the scan in a child that hands its status back (the
    # default, GRAB_DETACH) and in the calling process; usage errors leave before either
    for detach in ("1", "0"):
        env = dict(os.environ, GRAB_DETACH=detach)
        r = subprocess.run([built.bin_path(), "-n", "2", "-r", "foo", str(tmp_path)], capture_output=True, text=True, env=env)
        assert r.returncode == 255 and r.stdout == "" and "HIP device" in r.stderr, (detach, r)
        r = subprocess.run([built.bin_path(), "a(", str(f)], capture_output=True, text=True, env=env)
        assert r.returncode == 255 and "pcre_compile error" in r.stderr, (detach, r)
        r = subprocess.run([built.bin_path(), "-n", "2", "foo", str(f)], capture_output=True, text=True, env=env)
        assert r.returncode == 255 and "Multicore support only for recursive grabs" in r.stderr, (detach, r)
    r = subprocess.run([built.bin_path()], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage" in r.stderr + r.stdout
