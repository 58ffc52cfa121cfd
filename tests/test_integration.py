"""INTEGRATION.md recipe A, executed: the REFERENCE's own main.cc (/root/reference/src/main.cc) compiled against the
product's FileGrep (grab_amd/csrc/filegrep.h in place of grab.h) and linked with libgrabhost.so / libgscan.so.

CPU: the recipe compiles and links (needs /root/reference: this container only).
GPU: the resulting binary -- the reference's command line and threading code driving the gfx950 engine through the
FileGrep interface, incl. find(path, st, typeflag) from its pthreads -- prints what the reference prints on goldens."""
import hashlib
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT
from inputs import materialize

BIN = os.path.join(ROOT, "oracle", "_ref", "grab_ref_main_on_product")


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference sources are only in the build container")
def test_recipe_a_compiles_and_links(built):
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref_cli"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert os.path.exists(BIN)
    out = subprocess.run(["ldd", BIN], capture_output=True, text=True).stdout
    assert "libgrabhost.so" in out and "libgscan.so" in out and "not found" not in out
    # usage comes from the reference's main(): nothing of the product's CLI is in this binary
    r = subprocess.run([BIN], capture_output=True, text=True)
    assert r.returncode == 1 and r.stdout.startswith("Usage: ")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(BIN), reason="built from /root/reference in the build container (oracle/Makefile ref_cli)")
@pytest.mark.parametrize("name", ["t1_O", "t1_two_paths", "tree_rO", "tree_n2", "syn8_ident_O", "big_Ol_L5", "big_s_L5", "q5_capture"])
def test_reference_main_over_product_filegrep(name, built, tmp_path):
    case = next(c for c in GOLDEN if c["name"] == name)
    cache = {}
    for rel, recipe in case["inputs"].items():
        materialize(recipe, str(tmp_path / rel), cache)
    r = subprocess.run([BIN] + case["args"], cwd=str(tmp_path), capture_output=True)
    out = r.stdout
    if case["sorted"]:
        out = b"".join(sorted(out.splitlines(True)))
    assert r.returncode == case["rc"], r.stderr
    assert len(out) == case["stdout_len"]
    assert hashlib.md5(out).hexdigest() == case["stdout_md5"]
