#!/usr/bin/env python3
"""Reduce a rocprofv3 kernel trace (CSV with Start_Timestamp / End_Timestamp / Queue_Id / Kernel_Name) of tools/train_bench.py
to the timeline of its LAST training step: every launch with its start offset, duration, queue, the idle gap in front of it on
its queue, and per-queue busy time -- enough to see which stream the step's wall time sits on."""
import csv
import sys


def short(name):
    name = name.replace("dispu::", "").replace("void ", "")
    return name[:78]


def main(path):
    rows = list(csv.DictReader(open(path)))
    ks = []
    for r in rows:
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
    ks.sort()
    # steps are delimited by the Adam launch
    adam = [i for i, k in enumerate(ks) if "adam_kernel" in k[3]]
    if len(adam) < 2:
        print("fewer than two adam launches in the trace")
        return
    lo, hi = adam[-2] + 1, adam[-1] + 1
    step = ks[lo:hi]
    t0 = step[0][0]
    print("last step: %d launches, wall %.1f us, kernel time %.1f us" % (len(step), (step[-1][1] - t0) / 1e3, sum(e - s for s, e, _, _ in step) / 1e3))
    qs = sorted({k[2] for k in step})
    for q in qs:
        mine = [k for k in step if k[2] == q]
        print("queue %s: %d launches, busy %.1f us, first %.1f last-end %.1f" % (q, len(mine), sum(e - s for s, e, _, _ in mine) / 1e3, (mine[0][0] - t0) / 1e3,
                                                                                   (mine[-1][1] - t0) / 1e3))
    last_end = {}
    print("%9s %8s %7s %3s  %s" % ("start_us", "dur_us", "gap_us", "q", "kernel"))
    for s, e, q, n in step:
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        print("%9.1f %8.1f %7.1f %3s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, qs.index(q), short(n)))


if __name__ == "__main__":
    main(sys.argv[1])
