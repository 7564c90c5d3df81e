#!/bin/bash
# per-kernel medians of tools/glue_bench.py under rocprofv3 (run on the GPU box):  bash tools/glue_prof.sh
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/glue_prof
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/glue_prof -o glue -- python3 $R/tools/glue_bench.py > /dev/null 2>&1
python3 - $(find /tmp/glue_prof -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if "bn_" in n:
        key = (n.split("(")[0], r["Grid_Size_X"], r["Grid_Size_Y"])
        d[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items()):
    v.sort()
    print("%-28s grid %6s x %-6s calls %4d  median %.1f us" % (k[0], k[1], k[2], len(v), v[len(v) // 2] / 1e3))
PY
