#!/usr/bin/env python3
"""Watchdog for a GPU test session: every few seconds looks for processes of this user that are children (any depth) of the given
root pid, older than --age seconds and not the root itself, and writes what they are doing -- command line, every thread's
wchan / state, and a `rocgdb` backtrace of all threads when ptrace lets it -- to --out, once per pid.  A test that hangs in a
child process (the CLI, a `python -c` driver) then leaves more behind than a timeout.

    scripts/hang_watch.py --root PID --age 90 --out gpurun_out/hang.txt &
"""
import argparse
import os
import subprocess
import time


def children(root):
    kids, ppid = {}, {}
    for p in os.listdir("/proc"):
        if not p.isdigit():
            continue
        try:
            with open("/proc/%s/stat" % p) as f:
                st = f.read()
            rest = st[st.rindex(")") + 2:].split()
            ppid[int(p)] = (int(rest[1]), int(rest[19]))  # ppid, starttime (clock ticks since boot)
        except (OSError, ValueError):
            pass
    out = []
    for p, (pp, start) in ppid.items():
        q, depth = p, 0
        while q in ppid and q != root and depth < 32:
            q = ppid[q][0]
            depth += 1
        if q == root and p != root:
            out.append((p, start))
    return out


def dump(pid, out):
    with open(out, "a") as f:
        f.write("==== pid %d, %s\n" % (pid, time.strftime("%H:%M:%S")))
        try:
            f.write("cmdline: %s\n" % open("/proc/%d/cmdline" % pid, "rb").read().replace(b"\0", b" ").decode("latin-1")[:600])
            for t in sorted(os.listdir("/proc/%d/task" % pid), key=int):
                base = "/proc/%d/task/%s/" % (pid, t)
                comm = open(base + "comm").read().strip()
                wchan = open(base + "wchan").read().strip()
                state = [ln for ln in open(base + "status").read().splitlines() if ln.startswith("State")][0]
                f.write("  tid %s %-16s %-24s wchan %s\n" % (t, comm, state, wchan))
        except OSError as ex:
            f.write("  (gone: %s)\n" % ex)
            return
        f.flush()
        try:
            r = subprocess.run(["/opt/rocm/bin/rocgdb", "-batch", "-ex", "set pagination off", "-ex", "thread apply all bt 14", "-p", str(pid)],
                               capture_output=True, text=True, timeout=120)
            f.write(r.stdout[-24000:])
            f.write("\n[rocgdb stderr] " + r.stderr[-2000:] + "\n")
        except Exception as ex:  # noqa: BLE001 (a diagnostic: whatever goes wrong is written down)
            f.write("rocgdb: %r\n" % (ex,))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", type=int, required=True)
    ap.add_argument("--age", type=float, default=90.0)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    hz = os.sysconf("SC_CLK_TCK")
    seen = set()
    while os.path.exists("/proc/%d" % a.root):
        up = float(open("/proc/uptime").read().split()[0])
        for pid, start in children(a.root):
            if pid in seen or up - start / hz < a.age:
                continue
            seen.add(pid)
            dump(pid, a.out)
        time.sleep(5)


if __name__ == "__main__":
    main()
