#!/usr/bin/env python3
"""One step of a rocprofv3 kernel trace, queue by queue: every launch with its start offset, duration and the gap to the previous launch
on the same queue - what sits between the persistent launches of a chain.
    python tools/chain_trace.py <kernel_trace.csv> [--step 3] [--min-us 0]
(step boundaries: the began_step_kernel launch, one per AAS step)"""
import argparse
import collections
import csv
import re


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*", "", n)
    return n[:58]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--step", type=int, default=3)
    ap.add_argument("--min-us", type=float, default=0.0)
    a = ap.parse_args()
    rows = []
    for r in csv.DictReader(open(a.csv)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(r["Kernel_Name"]),
                     "%sx%s" % (r["Grid_Size_X"], r["Workgroup_Size_X"])))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if r[3].startswith("began_step_kernel")]
    lo = marks[a.step - 1] + 1
    hi = marks[a.step] + 1
    t0 = rows[lo][0]
    print("step %d: %d launches, %.3f ms" % (a.step, hi - lo, (rows[hi - 1][1] - t0) / 1e6))
    byq = collections.defaultdict(list)
    for r in rows[lo:hi]:
        byq[r[2]].append(r)
    for q, rs in sorted(byq.items()):
        busy = sum(r[1] - r[0] for r in rs) / 1e6
        print("\n== queue %d: %d launches, busy %.3f ms" % (q, len(rs), busy))
        prev = None
        for r in rs:
            gap = (r[0] - prev) / 1e3 if prev is not None else 0.0
            d = (r[1] - r[0]) / 1e3
            if d >= a.min_us:
                print("  %9.3f ms  %8.1f us  gap %7.1f us  %-58s %s" % ((r[0] - t0) / 1e6, d, gap, r[3], r[4]))
            prev = r[1]


if __name__ == "__main__":
    main()
