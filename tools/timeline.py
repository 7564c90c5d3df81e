#!/usr/bin/env python3
"""Timeline analysis of a rocprofv3 --kernel-trace CSV of bench.py: per-kernel totals plus, for the steady-state steps,
how much wall time has 0 / 1 / 2+ persistent recurrent launches in flight, GPU-idle gaps, and the overlap of GEMM time with
recurrent launches.  Usage: python tools/timeline.py <kernel_trace.csv> [--skip-ms 3000]"""
import csv
import re
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))))
    rows.sort()
    t_begin, t_end = rows[0][0], max(r[1] for r in rows)
    print("kernels: %d, span %.1f ms" % (len(rows), (t_end - t_begin) / 1e6))
    adam = [r for r in rows if "began_step" in r[2]] or [r for r in rows if "adam" in r[2]]   # one per step when present
    # steady-state window: between the Adam launches of two steps in the middle of the run
    ad = sorted(set(r[0] for r in adam))
    steps = []
    last = None
    for t in ad:
        if last is None or t - last > 5e6:
            steps.append(t)
        last = t
    if len(steps) < 8:
        print("too few steps found (%d)" % len(steps)); return
    nsteps = min(8, len(steps) - 5)
    first = len(steps) - 1 - nsteps        # the last `nsteps` complete steps of the run (steady state)
    lo, hi = steps[first], steps[first + nsteps]
    win = [r for r in rows if r[0] >= lo and r[1] <= hi]
    span = (hi - lo) / 1e6
    print("window: %d steps, %.2f ms / step, %d launches / step" % (nsteps, span / nsteps, len(win) / nsteps))
    short = lambda n: re.sub(r"\(.*", "", re.sub(r"void |\(anonymous namespace\)::", "", n))[:60]
    agg = defaultdict(lambda: [0, 0.0])
    for s, e, n, q in win:
        a = agg[short(n)]
        a[0] += 1; a[1] += (e - s) / 1e6
    print("%-62s %8s %10s %9s" % ("kernel", "calls/st", "ms/step", "avg us"))
    for n, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print("%-62s %8.1f %10.3f %9.1f" % (n, c / nsteps, ms / nsteps, 1e3 * ms / c))
    is_rnn = lambda n: "rnn_" in n
    is_gemm = lambda n: "gemm" in n
    ev = []
    for s, e, n, q in win:
        k = "rnn" if is_rnn(n) else ("gemm" if is_gemm(n) else "other")
        ev.append((s, 1, k)); ev.append((e, -1, k))
    ev.sort()
    cnt = defaultdict(int)
    hist = defaultdict(float)
    prev = lo
    for t, d, k in ev:
        dt = (t - prev) / 1e6
        if dt > 0:
            key = ("rnn%d" % min(cnt["rnn"], 3), "gemm" if cnt["gemm"] > 0 else "-", "other" if cnt["other"] > 0 else "-")
            hist[key] += dt
        cnt[k] += d
        prev = t
    print("\nwall time per step by what is in flight (persistent recurrent launches, any GEMM, any other kernel):")
    for key, ms in sorted(hist.items(), key=lambda kv: -kv[1]):
        print("  %-8s %-5s %-6s %8.3f ms" % (key[0], key[1], key[2], ms / nsteps))
    # Gantt of the persistent launches of one step (offsets from the step's first kernel)
    one = [r for r in rows if r[0] >= steps[first + 1] and r[0] < steps[first + 2]]
    if one:
        t0 = one[0][0]
        print("\nrecurrent launches of one step (start ms, duration ms, queue, kernel):")
        for s_, e_, n_, q_ in one:
            if is_rnn(n_) or "ctc" in n_ or "adam" in n_:
                print("  %7.3f  %6.3f  q%-3s %s" % ((s_ - t0) / 1e6, (e_ - s_) / 1e6, q_, short(n_)))
    if "--dump-step" in sys.argv and one:
        outp = sys.argv[sys.argv.index("--dump-step") + 1]
        with open(outp, "w") as f:
            f.write("start_ms,dur_us,queue,kernel\n")
            for s_, e_, n_, q_ in one:
                f.write("%.4f,%.1f,%s,%s\n" % ((s_ - t0) / 1e6, (e_ - s_) / 1e3, q_, short(n_)))
    idle = sum(ms for k, ms in hist.items() if k == ("rnn0", "-", "-")) / nsteps
    no_rnn = sum(ms for k, ms in hist.items() if k[0] == "rnn0") / nsteps
    print("GPU idle %.3f ms / step; no recurrent launch in flight %.3f ms / step" % (idle, no_rnn))


if __name__ == "__main__":
    main()
