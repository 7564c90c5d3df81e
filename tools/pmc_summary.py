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


# launch classes of bench.py's roofline object (ops.Profiler names) -> kernel-name prefixes, per arithmetic mode and kernel
# generation.  A class takes EVERY kernel one of its prefixes matches (template variants of one kernel: tile heights, wave
# counts); the first prefix group that matches anything wins (newer kernels first).  bytes per launch = dispatch-weighted mean.
CLASS_PATTERNS = {
    "lstm_bwd[N=60,H=500]": (("rnn_bwd_rs_kernel<1, 32, 4, -1, true, false,", "rnn_bwd_rs_kernel<1, 32, 4, -1, false, false, true"), ("rnn_bwd_rs_kernel<1, ",)),
    "lstm_bwd[N=30,H=500]": (("rnn_bwd_rs_kernel<1, 32, 4, -1, true, true,",), ("rnn_bwd_rs_kernel<1, ",)),
    "gru_bwd[N=30,H=1000]": (("rnn_bwd_rs_kernel<3, ",),),
    "lstm_fwd[N=60,H=500]": (("rnn_fwd32_kernel<0, ",),), "lstm_fwd[N=30,H=500]": (("rnn_split_kernel<0, ",),), "gru_fwd[N=30,H=1000]": (("rnn_fwd32_kernel<2, ",),),
    "gemm_planes_wgrad": (("gemm_planes_tn_kernel<",),), "gemm_planes": (("gemm_planes_kernel<256, 256",),),
    "gemm_tn": (("gemm32_kernel<false, false",), ("gemm_f32_kernel<false, false",)),
    "gemm_nn": (("gemm32_kernel<true, false",), ("gemm_f32_kernel<true, false",)),
    "gemm_nt": (("gemm32_kernel<true, true",), ("gemm_f32_kernel<true, true",)),
    "gemm_splitk_reduce": (("gemm32_reduce_kernel",),)}


def classes_of(kernels):
    by_class = {}
    for cname, groups in CLASS_PATTERNS.items():
        for pats in groups:
            hit = [(k, v) for k, v in kernels.items() if any(k.startswith(p_) for p_ in pats)]
            if hit:
                n = sum(v["dispatches"] for _, v in hit)
                mean = lambda key: sum(v[key] * v["dispatches"] for _, v in hit) / max(n, 1)
                by_class[cname] = {"FETCH_SIZE_KiB_avg": mean("FETCH_SIZE_KiB_avg"), "WRITE_SIZE_KiB_avg": mean("WRITE_SIZE_KiB_avg"), "dispatches": n,
                                   "hbm_bytes_per_launch": mean("hbm_bytes_per_launch"), "kernels": sorted(k for k, _ in hit)}
                break
    return by_class


def summarise(fdir, wdir, cmd, steps=0):
    """-> the summary dict of one FETCH_SIZE and one WRITE_SIZE pass (directories of rocprofv3 csv output)."""
    fe, wr = collect(fdir, "FETCH_SIZE"), collect(wdir, "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(fe) | set(wr)):
        f = fe.get(name, [])
        w = wr.get(name, [])
        fa = sum(f) / len(f) if f else 0.0
        wa = sum(w) / len(w) if w else 0.0
        kernels[name] = {"FETCH_SIZE_KiB_avg": fa, "WRITE_SIZE_KiB_avg": wa, "dispatches": max(len(f), len(w)),
                         "hbm_bytes_per_launch": (2.0 * fa + wa) * 1024.0}
    total = sum(v["hbm_bytes_per_launch"] * v["dispatches"] for v in kernels.values())
    return {"total_hbm_bytes": total, "steps_in_run": steps, "hbm_bytes_per_step": (total / steps) if steps else None, "by_class": classes_of(kernels),
            "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) of `%s`" % cmd,
            "units": "KiB per dispatch; fetched bytes = 2 x FETCH_SIZE x 1024 (gfx950 wide-read correction), written bytes = WRITE_SIZE x 1024",
            "kernels": kernels}


def main():
    if sys.argv[1] == "--rebuild":      # recompute by_class of an existing summary (kernel names changed, same raw numbers)
        j = json.load(open(sys.argv[2]))
        j["by_class"] = classes_of(j["kernels"])
        json.dump(j, open(sys.argv[2], "w"), indent=1)
        return
    fdir, wdir, out, cmd = sys.argv[1:5]
    j = summarise(fdir, wdir, cmd, int(sys.argv[5]) if len(sys.argv) > 5 else 0)
    json.dump(j, open(out, "w"), indent=1)
    for k, v in sorted(j["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
        print("%-60s x%-4d %8.1f MB / launch" % (k[:60], v["dispatches"], v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
