"""In-tree build of the HIP engine and host libraries (hipcc cross-compiles gfx950 without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "lib")
BIN = os.path.join(_HERE, "bin")


def lib_path(name):
    return os.path.join(LIB, name)


def bin_path(name="grab"):
    return os.path.join(BIN, name)


def build(verbose=False):
    """Run csrc/Makefile (no-op when up to date). Raises on failure."""
    env = dict(os.environ)
    env.setdefault("HIPCC", "/opt/rocm/bin/hipcc")
    r = subprocess.run(["make", "-C", CSRC, "all"], env=env, capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
        print(r.stderr)
    if r.returncode != 0:
        raise RuntimeError("grab_amd build failed (make -C %s)" % CSRC)
    for f in (lib_path("libgscan.so"), lib_path("libgrabhost.so"), bin_path()):
        if not os.path.exists(f):
            raise RuntimeError("build did not produce %s" % f)
