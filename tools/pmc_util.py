#!/usr/bin/env python3
"""Per-kernel MFMA utilisation and LDS bank-conflict rate from one rocprofv3 pass
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d DIR -o run --output-format csv -- python3 bench.py ...
    python tools/pmc_util.py DIR out.md
MFMA utilisation = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) * 1024 SIMDs): chip-wide, i.e. a launch that occupies half
of the CUs can reach 50 % at most; LDS conflict rate = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (cycles)."""
import csv
import glob
import re
import sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    acc = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            name = re.sub(r"^void ", "", name).split("(")[0]
            key = (name, r["Dispatch_Id"])
            c = acc.setdefault(key, {})
            v = float(r["Counter_Value"])
            cn = r["Counter_Name"]
            if cn == "GRBM_GUI_ACTIVE":
                c[cn] = max(c.get(cn, 0.0), v)
            else:
                c[cn] = c.get(cn, 0.0) + v
    per = {}
    for (name, _), c in acc.items():
        p = per.setdefault(name, dict(n=0, mfma=0.0, gui=0.0, conf=0.0, idx=0.0))
        p["n"] += 1
        p["mfma"] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        p["gui"] += c.get("GRBM_GUI_ACTIVE", 0.0)
        p["conf"] += c.get("SQ_LDS_BANK_CONFLICT", 0.0)
        p["idx"] += c.get("SQ_LDS_IDX_ACTIVE", 0.0)
    rows = sorted(per.items(), key=lambda kv: -kv[1]["gui"])
    lines = ["| kernel | launches | GPU cycles / launch | MFMA utilisation (chip-wide) | LDS bank-conflict cycles / LDS active cycles |", "|---|---|---|---|---|"]
    for name, p in rows[:20]:
        if p["gui"] <= 0:
            continue
        lines.append("| `%s` | %d | %.0f | %.1f %% | %s |" % (name[:70], p["n"], p["gui"] / p["n"], 100.0 * p["mfma"] / (p["gui"] * 1024.0),
                                                             ("%.3f" % (p["conf"] / p["idx"])) if p["idx"] > 0 else "-"))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
