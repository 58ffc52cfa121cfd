"""grab_amd -- MI355X (gfx950) scan engine behind stealth/grab's per-file match loop.

Layout
  csrc/     HIP kernels, engine, pattern compiler, FileGrep host + CLI (C++/HIP)
  lib/      built shared objects (git-ignored): libgscan.so (C ABI include/gscan.h),
            libgrabhost.so (C facade include/grab_host.h)
  bin/grab  the drop-in command line
  engine.py   ctypes binding of include/gscan.h
  filegrep.py ctypes mirror of the reference's FileGrep interface
  synth.py    seeded synthetic corpus generator

There is no CPU scanning path in this package: everything that scans goes through
libgscan.so and a HIP device, and fails loudly without them.
"""
from .build import build, lib_path, bin_path  # noqa: F401

__all__ = ["build", "lib_path", "bin_path"]
