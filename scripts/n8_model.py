#!/usr/bin/env python3
"""The terms of DESIGN.md 6's N-GPU model that ONE GPU can measure (VERDICT r4 task 1): T(N) = F(N) + bytes / min(N x P, D).

  F -- the fixed cost with eight device indices' worth of reader pools, streams, slots and pinned blocks going through ONE
       HIP runtime: `grab -n 32 -r` under GSCAN_VIRTUAL_DEVICES=8 (+ a faked two-socket sysfs tree for the placement) against
       `grab -n 8 -r` on the one index, same corpus: F(8 indices) - F(1) = how much later the LAST index queues its first
       DMA + how much longer the exit takes (the wall clocks also differ by what eight pools lose on ONE shared link, which
       eight GPUs would not: reported beside it, not counted).
  D -- the host's page cache -> pinned ceiling with the DMA and the scan stubbed out (GSCAN_DIAG=1: the readers fill their
       blocks and hand them straight back), eight pools on both sockets, 1 / 2 / 4 / 8 readers per pool = 8 ... 64 readers,
       plain pread against pread + non-temporal copy (GSCAN_NT_COPY).

    scripts/n8_model.py --dir /dev/shm/corpus [--bytes N]      prints one JSON object (bench.py embeds it as "n8_model")
"""
import argparse
import json
import os
import re
import shutil
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fake_pci_tree(root):
    """sysfs as an 8-GPU, two-socket node shows it, for the bus ids GSCAN_VIRTUAL_DEVICES gives its indices: devices 0-3 on
    the first NUMA node's CPUs, 4-7 on the second's (one node: halves of the CPU list)."""
    lists = []
    try:
        for n in sorted(os.listdir("/sys/devices/system/node")):
            if re.fullmatch(r"node\d+", n):
                lists.append(open("/sys/devices/system/node/%s/cpulist" % n).read().strip())
    except OSError:
        pass
    if len(lists) < 2:
        cpus = sorted(os.sched_getaffinity(0))
        half = max(1, len(cpus) // 2)
        lists = [",".join(map(str, cpus[:half])), ",".join(map(str, cpus[half:] or cpus[:half]))]
    for v in range(8):
        d = os.path.join(root, "0000:%02x:00.0" % (0x0c + 0x10 * v))
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "local_cpulist"), "w") as f:
            f.write(lists[(v // 4) % len(lists)] + "\n")
    return root


def run(argv, env, reps, pause=0.5):
    """min-wall run of `reps` (after one untimed pass): (wall, marks, first DMA per device index, stderr)."""
    best = None
    for it in range(reps + 1):
        time.sleep(pause)
        t0 = time.perf_counter()
        r = subprocess.run(argv, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
        dt = time.perf_counter() - t0
        if r.returncode != 0:
            return None
        if it and (best is None or dt < best[0]):
            best = (dt, r.stderr)
    dt, err = best
    marks = dict((m.group(2).decode(), float(m.group(1))) for m in re.finditer(rb"\[grab timing\] \+([0-9.]+) s ([^\n]+)", err))
    first = dict((int(m.group(1)), float(m.group(2))) for m in re.finditer(rb"device (\d+): first piece queued for DMA at \+([0-9.]+) s", err))
    return dt, marks, first, err


def summary(got, nbytes):
    dt, marks, first, _ = got
    t_up, t_done = marks.get("runtime up"), marks.get("workers joined")
    out = {"wall_s": round(dt, 4), "startup_s": t_up, "worker0_context_open_s": marks.get("worker 0: context open"),
           "workers_joined_s": t_done, "exit_s": t_done is not None and round(dt - t_done, 4)}
    if first:
        v = sorted(first.values())
        out["first_dma_s"] = {"first_index": v[0], "median_index": round(statistics.median(v), 4), "last_index": v[-1], "indices": len(v)}
    if t_up is not None and t_done is not None and t_done > t_up:
        out["scan_phase_GBps"] = round(nbytes / (t_done - t_up) / 1e9, 2)
    return out


def forecast(P, D, F1, dF, nbytes, n=8):
    """T(N) = F(N) + bytes / min(N x P, D) with the measured terms: P = the one index's scan phase (GB/s), D = the
    best host copy rate (an UPPER bound of what the host can feed: with the DMA reading the same memory a reader moves half of
    what it moves alone), F(N) = F(1) + dF (the measured difference between eight indices and one)."""
    gb = nbytes / 1e9
    F8 = F1 + dF
    T1, T8 = F1 + gb / P, F8 + gb / min(n * P, D)
    return {"P_GBps": P, "D_GBps": D, "F1_s": round(F1, 3), "F8_s": round(F8, 3), "T1_s": round(T1, 3), "T8_s": round(T8, 3),
            "GBps": round(gb / T8, 1), "strong_scaling_efficiency": round(T1 / (n * T8), 3),
            "bound_by": "D (the host's page cache -> pinned copy)" if D < n * P else "the links",
            "efficiency_if_F8_were_F1": round(T1 / (n * (F1 + gb / min(n * P, D))), 3)}


def measure(grab, d, nbytes, pattern="foobardoesnotexist", reps=2, host_copy=True):
    tmp = tempfile.mkdtemp(prefix="grab_n8_", dir="/tmp")
    try:
        cpus = len(os.sched_getaffinity(0))
        env1 = dict(os.environ, GRAB_TIMING="1", GSCAN_TIMING="1")
        env8 = dict(env1, GSCAN_VIRTUAL_DEVICES="8", GSCAN_SYSFS_PCI=fake_pci_tree(os.path.join(tmp, "pci")))
        w1, w8 = str(min(8, cpus)), str(min(32, cpus))
        one = run([grab, "-n", w1, "-r", pattern, d], env1, reps)
        eight = run([grab, "-n", w8, "-r", pattern, d], env8, reps)
        if not one or not eight:
            return {"error": "grab failed"}
        o1, o8 = summary(one, nbytes), summary(eight, nbytes)
        out = {"bytes": nbytes,
               "one_index": dict(o1, command="grab -n %s -r" % w1),
               "eight_indices": dict(o8, command="GSCAN_VIRTUAL_DEVICES=8 grab -n %s -r (eight reader pools, their streams and 96 slots through one runtime -- and, here, one link)" % w8)}
        # F(8 indices) - F(1), from the marks: how much later the LAST index's first DMA is queued than the one index's, plus
        # how much longer the exit takes.  (The wall clocks' difference also holds what eight pools lose by sharing ONE link
        # and ONE GPU's queues on this box -- scan_phase_GBps of the two runs says how much -- which eight GPUs would not.)
        try:
            ramp = o8["first_dma_s"]["last_index"] - o1["first_dma_s"]["last_index"]
            out["F8_minus_F1_measured_s"] = round(ramp + (o8["exit_s"] - o1["exit_s"]), 4)
            out["F8_minus_F1_parts_s"] = {"last_index_first_dma_later_by": round(ramp, 4), "exit_longer_by": round(o8["exit_s"] - o1["exit_s"], 4)}
        except (KeyError, TypeError):
            out["F8_minus_F1_measured_s"] = None
        out["wall_delta_s"] = round(eight[0] - one[0], 4)
        out["F1_measured_s"] = round(one[0] - nbytes / (o1["scan_phase_GBps"] * 1e9), 4) if o1.get("scan_phase_GBps") else None
        if host_copy:
            table = {}
            best = None
            for nt in (0, 1):
                for per in (1, 2, 4, 8):
                    e = dict(env8, GSCAN_DIAG="1", GSCAN_TEST_HOOKS="1", GSCAN_READERS=str(per), GSCAN_NT_COPY=str(nt))
                    got = run([grab, "-n", w8, "-r", pattern, d], e, 1, pause=0.2)
                    if not got:
                        continue
                    marks = got[1]
                    t_up, t_done = marks.get("runtime up"), marks.get("workers joined")
                    if t_up is None or t_done is None or t_done <= t_up:
                        continue
                    rate = round(nbytes / (t_done - t_up) / 1e9, 1)
                    table["%d readers, %s" % (8 * per, "non-temporal copy" if nt else "pread")] = rate
                    if best is None or rate > best[0]:
                        best = (rate, nt, 8 * per)
            out["host_copy_GBps_by_readers"] = table
            if best:
                out["host_copy_best"] = {"GBps": best[0], "nt_copy": best[1], "readers": best[2]}
                out["read_mode_best"] = "non-temporal copy" if best[1] else "pread"
            out["host_copy_what"] = "GSCAN_DIAG=1: page cache -> pinned blocks only (no DMA, no scan), eight reader pools bound to the two sockets' CPUs by a faked sysfs tree, corpus pages interleaved"
        try:
            out["forecast_N8"] = forecast(o1["scan_phase_GBps"], out["host_copy_best"]["GBps"], out["F1_measured_s"], out["F8_minus_F1_measured_s"], nbytes)
        except (KeyError, TypeError, ZeroDivisionError):
            pass
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", required=True)
    ap.add_argument("--bytes", type=int, default=0)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--no-host-copy", action="store_true")
    a = ap.parse_args()
    sys.path.insert(0, ROOT)
    from grab_amd import bin_path

    nbytes = a.bytes
    if not nbytes:
        for dp, _, files in os.walk(a.dir):
            nbytes += sum(os.path.getsize(os.path.join(dp, f)) for f in files)
    print(json.dumps(measure(bin_path(), a.dir, nbytes, reps=a.reps, host_copy=not a.no_host_copy)))


if __name__ == "__main__":
    main()
