#!/usr/bin/env python3
"""Round 6, session AC: `grab -L -L -O NEEDLE big.bin` under GSCAN_VIRTUAL_DEVICES=8 (tests/test_gpu_geometry.py's 2.5 GiB file)
never ended in sessions Z and AB -- every thread asleep on a futex.  Runs it with the engine's trace on; if it is still there
after 40 s, once more as rocgdb's child and interrupted after 25 s: the backtrace of every thread."""
import os
import shutil
import signal
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_geometry as tg  # noqa: E402
from grab_amd import bin_path  # noqa: E402

d = "/dev/shm/grab_ac_%d" % os.getpid()
os.makedirs(d)
try:
    size = 2 * tg.STRIDE + (tg.CHUNK >> 1) + 12345
    tg.boundary_file(size).tofile(os.path.join(d, "big.bin"))
    for argv, extra in ((["-L", "-L", "-O", "NEEDLE", "big.bin"], {}), (["-L", "-L", "-O", "NEEDLE", "big.bin"], {"GSCAN_READERS": "4"}), (["-O", "-l", "NEEDLE", "big.bin"], {})):
        env = dict(os.environ, GSCAN_VIRTUAL_DEVICES="8", GSCAN_TRACE="1", GRAB_TIMING="1", GSCAN_TIMING="1", **extra)
        t0 = time.time()
        with open(os.path.join(d, "trace.txt"), "wb") as tr:
            p = subprocess.Popen([bin_path()] + argv, cwd=d, env=env, stdout=subprocess.DEVNULL, stderr=tr)
            try:
                rc = p.wait(timeout=40)
            except subprocess.TimeoutExpired:
                rc = "HUNG"
                p.kill()
                p.wait()
        print("== %s %s -> rc %s in %.1f s" % (argv, extra, rc, time.time() - t0), flush=True)
        if rc == "HUNG":
            lines = open(os.path.join(d, "trace.txt"), "rb").read().decode("latin-1").splitlines()
            print("trace: %d lines; the last 120:" % len(lines))
            print("\n".join(lines[-120:]), flush=True)
            env.pop("GSCAN_TRACE")
            g = subprocess.Popen(["/opt/rocm/bin/rocgdb", "-batch", "-ex", "set pagination off", "-ex", "run", "-ex", "thread apply all bt 16", "--args", bin_path()] + argv,
                                 cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            time.sleep(25)
            kids = subprocess.run(["pgrep", "-P", str(g.pid)], capture_output=True, text=True).stdout.split()
            for k in kids:
                os.kill(int(k), signal.SIGINT)
            try:
                out = g.communicate(timeout=120)[0].decode("latin-1")
            except subprocess.TimeoutExpired:
                g.kill()
                out = g.communicate()[0].decode("latin-1")
            keep = [ln for ln in out.splitlines() if not ln.startswith("[New Thread") and not ln.startswith("[Thread") and "Match at" not in ln]
            print("\n".join(keep[-400:]), flush=True)
            break
finally:
    shutil.rmtree(d, ignore_errors=True)
