#!/usr/bin/env python3
"""Where does the host -> HBM link idle?  `rocprofv3 --memory-copy-trace --kernel-trace` over one `grab` run, then the time
line of the H2D copies: how long each took, how much of the span between the first and the last the link was busy (union of
the copies' intervals), how the idle time is distributed (gaps between consecutive copies), and what else ran.

    scripts/copy_trace.py --out gpurun_out/copytrace -- grab_amd/bin/grab -n 8 -r PATTERN DIR
"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    os.makedirs(a.out, exist_ok=True)
    env = dict(os.environ, GRAB_NORMAL_EXIT="1", TMPDIR="/tmp")  # (the profiler writes its results from exit handlers)
    r = subprocess.run(["rocprofv3", "--memory-copy-trace", "--kernel-trace", "--output-format", "csv", "-d", a.out, "--"] + cmd,
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
    files = glob.glob(os.path.join(a.out, "**", "*memory_copy_trace.csv"), recursive=True)
    if r.returncode != 0 or not files:
        print(json.dumps({"error": "rocprofv3 failed", "rc": r.returncode, "stderr": r.stderr[-400:].decode("latin-1")}))
        return 1
    rows = list(csv.DictReader(open(files[0])))
    cols = list(rows[0].keys()) if rows else []
    h2d = []
    other = {}
    for row in rows:
        kind = row.get("Direction") or row.get("Kind") or "?"
        s, e = int(row["Start_Timestamp"]), int(row["End_Timestamp"])
        nbytes = int(row.get("Bytes") or row.get("Size") or 0)
        if "HOST_TO_DEVICE" in kind.upper() or kind.upper().endswith("H2D"):
            h2d.append((s, e, nbytes))
        else:
            k = other.setdefault(kind, [0, 0])
            k[0] += 1
            k[1] += e - s
    h2d.sort()
    big = [x for x in h2d if (x[2] >= (1 << 20) if x[2] else (x[1] - x[0]) > 50_000)]
    out = {"columns": cols, "h2d_copies": len(h2d), "h2d_big": len(big), "other": {k: {"n": v[0], "busy_ms": round(v[1] / 1e6, 3)} for k, v in other.items()}}
    if big:
        span = big[-1][1] - big[0][0]
        # union of the intervals
        busy, cur_s, cur_e = 0, big[0][0], big[0][1]
        gaps = []
        overl = 0
        for s, e, _ in big[1:]:
            if s > cur_e:
                busy += cur_e - cur_s
                gaps.append(s - cur_e)
                cur_s, cur_e = s, e
            else:
                overl += min(e, cur_e) - s
                cur_e = max(cur_e, e)
        busy += cur_e - cur_s
        durs = sorted(e - s for s, e, _ in big)
        nb = sum(b for _, _, b in big)
        gaps.sort()
        out.update({
            "span_ms": round(span / 1e6, 3), "link_busy_ms": round(busy / 1e6, 3), "busy_frac": round(busy / span, 4),
            "bytes": nb, "GBps_over_span": nb and round(nb / span, 2), "GBps_while_busy": nb and round(nb / busy, 2),
            "copy_us": {"min": durs[0] / 1e3, "p10": durs[len(durs) // 10] / 1e3, "median": durs[len(durs) // 2] / 1e3, "p90": durs[len(durs) * 9 // 10] / 1e3, "max": durs[-1] / 1e3},
            "sum_of_copy_ms": round(sum(durs) / 1e6, 3), "overlapped_ms": round(overl / 1e6, 3),
            "gaps": {"n": len(gaps), "total_ms": round(sum(gaps) / 1e6, 3), "median_us": gaps and gaps[len(gaps) // 2] / 1e3, "p90_us": gaps and gaps[len(gaps) * 9 // 10] / 1e3,
                     "max_us": gaps and gaps[-1] / 1e3, "over_100us": sum(1 for g in gaps if g > 100_000), "over_100us_total_ms": round(sum(g for g in gaps if g > 100_000) / 1e6, 3)},
        })
        # the first 40 copies as a time line (relative to the first): start, duration, gap to the previous end
        t0 = big[0][0]
        out["first_copies_us"] = [[round((s - t0) / 1e3, 1), round((e - s) / 1e3, 1)] for s, e, _ in big[:40]]
        mid = len(big) // 2
        out["mid_copies_us"] = [[round((s - t0) / 1e3, 1), round((e - s) / 1e3, 1)] for s, e, _ in big[mid:mid + 40]]
    kfiles = glob.glob(os.path.join(a.out, "**", "*kernel_trace.csv"), recursive=True)
    if kfiles:
        per = {}
        for row in csv.DictReader(open(kfiles[0])):
            name = row["Kernel_Name"].split("(")[0][-60:]
            k = per.setdefault(name, [0, 0])
            k[0] += 1
            k[1] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        out["kernels"] = {k: {"n": v[0], "total_ms": round(v[1] / 1e6, 3), "avg_us": round(v[1] / v[0] / 1e3, 2)} for k, v in per.items()}
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
