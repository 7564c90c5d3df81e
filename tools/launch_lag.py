#!/usr/bin/env python3
"""Host-side launch time against device-side start time of every kernel of one steady-state step.
    rocprofv3 --kernel-trace --hip-trace --output-format csv -d DIR -o run -- python3 bench.py ...
    python tools/launch_lag.py DIR/.../run_kernel_trace.csv DIR/.../run_hip_api_trace.csv [out.csv]
lag = kernel start - return of the launching HIP call: small (tens of us) where the device waits for the host,
large (ms) where the host runs ahead and the device works through its queues."""
import csv
import re
import sys


def main():
    kpath, apath = sys.argv[1], sys.argv[2]
    api = {}
    with open(apath) as f:
        for r in csv.DictReader(f):
            fn = r.get("Function", "")
            if "Launch" in fn or "Memset" in fn or "Memcpy" in fn:
                api[r["Correlation_Id"]] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]), fn)
    rows = []
    with open(kpath) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r["Correlation_Id"]))
    rows.sort()
    short = lambda n: re.sub(r"\(.*", "", re.sub(r"void |\(anonymous namespace\)::", "", n))[:44]
    # one step: between two G-optimiser ticks late in the run (adam_tick_kernel launches come in pairs/triples per step)
    ticks = [r[0] for r in rows if "began_step" in r[2]]
    if len(ticks) < 6:
        print("too few steps"); return
    lo, hi = ticks[-4], ticks[-3]
    one = [r for r in rows if lo <= r[0] < hi]
    out = open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout
    out.write("start_ms,dur_us,queue,lag_us,host_call_ms,kernel\n")
    for s, e, n, q, cid in one:
        a = api.get(cid)
        lag = (s - a[1]) / 1e3 if a else float("nan")
        hc = (a[0] - lo) / 1e6 if a else float("nan")
        out.write("%.4f,%.1f,%s,%.1f,%.4f,%s\n" % ((s - lo) / 1e6, (e - s) / 1e3, q, lag, hc, short(n)))
    print("step %.3f ms, %d launches" % ((hi - lo) / 1e6, len(one)), file=sys.stderr)


if __name__ == "__main__":
    main()
