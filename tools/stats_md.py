#!/usr/bin/env python3
"""rocprofv3 `*_kernel_stats.csv` -> the markdown table kept beside it under profiles/.
  python tools/stats_md.py profiles/r03_f32_kernel_stats.csv "fp32 mode (bench.py --precision 0 ...)" """
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"\(.*$", "", name)[:110]


def main(path, title):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    out = ["# rocprofv3 --kernel-trace --stats: %s" % title, "", "GPU kernel time in the run: %.1f ms" % (tot / 1e6), "",
           "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
        out.append("| `%s` | %s | %.2f | %.1f | %.1f |" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                          float(r["AverageNs"]) / 1e3, 100.0 * float(r["TotalDurationNs"]) / tot))
    open(re.sub(r"\.csv$", ".md", path), "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else sys.argv[1])
