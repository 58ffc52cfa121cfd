#!/usr/bin/env python3
"""A/B runs of one `grab` command under several environments: wall clock of the whole process (min and median of --reps,
after one untimed pass), the GRAB_TIMING marks of the best run, the per-worker split of worker 0.

    scripts/ab_run.py --reps 3 --bytes N --env "" --env "GPU_MAX_HW_QUEUES=1" -- grab_amd/bin/grab -n 8 -r PATTERN DIR
"""
import argparse
import json
import os
import re
import statistics
import subprocess
import sys
import time


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--bytes", type=int, default=0)
    ap.add_argument("--env", action="append", default=[])
    ap.add_argument("--sleep", type=float, default=0.0, help="seconds of quiet before every run (the previous process's teardown goes on in the kernel after its parent has seen it exit)")
    ap.add_argument("--interleave", action="store_true", help="round-robin over the environments instead of one after the other")
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    envs = a.env or [""]
    runs = {e: [] for e in envs}
    order = [e for _ in range(a.reps + 1) for e in envs] if a.interleave else [e for e in envs for _ in range(a.reps + 1)]
    seen = set()
    for e in order:
        env = dict(os.environ, GRAB_TIMING="1")
        for kv in e.split():
            k, _, v = kv.partition("=")
            env[k] = v
        if a.sleep:
            time.sleep(a.sleep)
        with open("/dev/null", "wb") as out:
            t0 = time.perf_counter()
            r = subprocess.run(cmd, stdout=out, stderr=subprocess.PIPE, env=env)
            dt = time.perf_counter() - t0
        if e not in seen:  # the untimed pass
            seen.add(e)
            continue
        runs[e].append((dt, r.returncode, r.stderr))
    for e in envs:
        best = min(runs[e], key=lambda x: x[0])
        err = best[2].decode("latin-1")
        marks = dict((m.group(2), float(m.group(1))) for m in re.finditer(r"\[grab timing\] \+([0-9.]+) s ([^\n]+)", err))
        w0 = [ln for ln in err.splitlines() if "files" in ln and "launches" in ln][:1]
        rd = [ln for ln in err.splitlines() if "[gscan timing] device" in ln][:1]
        walls = [x[0] for x in runs[e]]
        rec = {"env": e, "rc": best[1], "wall_min_s": round(min(walls), 4), "wall_median_s": round(statistics.median(walls), 4),
               "GBps_min_wall": a.bytes and round(a.bytes / min(walls) / 1e9, 2), "marks": marks,
               "after_last_mark_s": marks and round(min(walls) - max(marks.values()), 4),
               "worker0": w0[0][14:] if w0 else None, "readers": rd[0][15:] if rd else None}
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
