"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-kernel HBM bytes per launch.

    python tools/pmc_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> "<command that was profiled>" [steps in the run]

Units (MI355X_MICROARCH.md, HBM / rocprofv3 section): both counters are in KiB per dispatch; on gfx950 FETCH_SIZE reports
half of the bytes of wide (16 B/lane) coalesced reads, so fetched bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact."""
import csv
import glob
import json
import re
import sys


def collect(d, counter):
    acc = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            name = re.sub(r"^void ", "", name).split("(")[0]
            if "rnn_" in name:   # the same persistent kernel runs at several batch sizes: tell them apart by their grid
                name += " grid=%s" % r.get("Grid_Size", "?")
            key = (name, r.get("Dispatch_Id"))
            acc[key] = acc.get(key, 0.0) + float(r["Counter_Value"])   # one row per XCD / dimension instance: sum them
    per = {}
    for (name, _), v in acc.items():
        per.setdefault(name, []).append(v)
    return per


# launch classes of bench.py's roofline object (ops.Profiler names) -> kernel-name prefixes of either arithmetic mode
# (persistent grids: P x Q x 2 x threads; the same kernel at two batch sizes has the same grid here, so one entry serves both)
CLASS_PATTERNS = {
    "lstm_bwd[N=60,H=500]": "rnn_bwd_rs_kernel<1, ", "lstm_bwd[N=30,H=500]": ("rnn_bwd_rs_kernel<1, 32, 4, -1, true, true>", "rnn_bwd_rs_kernel<1, "), "gru_bwd[N=30,H=1000]": "rnn_bwd_rs_kernel<3, ",
    "lstm_fwd[N=60,H=500]": "rnn_fwd32_kernel<0, ", "lstm_fwd[N=30,H=500]": "rnn_split_kernel<0, ", "gru_fwd[N=30,H=1000]": "rnn_fwd32_kernel<2, ",
    "gemm_planes_wgrad": "gemm_planes_tn_kernel<", "gemm_planes": "gemm_planes_kernel<256, 256",
    "gemm_tn": "gemm_f32_kernel<false, false", "gemm_nn": "gemm_f32_kernel<true, false", "gemm_nt": "gemm_f32_kernel<true, true"}


def classes_of(kernels):
    by_class = {}
    for cname, pats in CLASS_PATTERNS.items():
        hit = []
        for pat in ((pats,) if isinstance(pats, str) else pats):      # (a tuple: the first pattern that matches anything wins)
            hit = [(k, v) for k, v in kernels.items() if k.startswith(pat)]
            if hit:
                break
        if hit:
            k, v = max(hit, key=lambda kv: kv[1]["dispatches"])
            by_class[cname] = dict(v, kernel=k)
    return by_class


def main():
    if sys.argv[1] == "--rebuild":      # recompute by_class of an existing summary (kernel names changed, same raw numbers)
        j = json.load(open(sys.argv[2]))
        j["by_class"] = classes_of(j["kernels"])
        json.dump(j, open(sys.argv[2], "w"), indent=1)
        return
    fdir, wdir, out, cmd = sys.argv[1:5]
    fe, wr = collect(fdir, "FETCH_SIZE"), collect(wdir, "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(fe) | set(wr)):
        f = fe.get(name, [])
        w = wr.get(name, [])
        fa = sum(f) / len(f) if f else 0.0
        wa = sum(w) / len(w) if w else 0.0
        kernels[name] = {"FETCH_SIZE_KiB_avg": fa, "WRITE_SIZE_KiB_avg": wa, "dispatches": max(len(f), len(w)),
                         "hbm_bytes_per_launch": (2.0 * fa + wa) * 1024.0}
    by_class = classes_of(kernels)
    total = sum(v["hbm_bytes_per_launch"] * v["dispatches"] for v in kernels.values())
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    json.dump({"total_hbm_bytes": total, "steps_in_run": steps, "hbm_bytes_per_step": (total / steps) if steps else None, "by_class": by_class, "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) of `%s`" % cmd,
               "units": "KiB per dispatch; fetched bytes = 2 x FETCH_SIZE x 1024 (gfx950 wide-read correction), written bytes = WRITE_SIZE x 1024",
               "kernels": kernels}, open(out, "w"), indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
        print("%-60s x%-4d %8.1f MB / launch" % (k[:60], v["dispatches"], v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
