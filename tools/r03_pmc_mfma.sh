#!/bin/bash
# MFMA utilisation (derived metric MfmaUtil) + LDS bank conflicts per kernel inside the fp32 step -> gpurun_out/<tag>/
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03h}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="bench.py --precision 0 --steps 8 --warmup 3 --no-cpu-baseline --no-extras --profile-steps 0 --no-traffic"
rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d /tmp/p_m -o run -- python3 $R/$CMD > /dev/null 2>&1
python3 - <<PY > $O/f32_pmc_mfmautil.md
import csv, glob, re
acc = {}
for f in glob.glob("/tmp/p_m/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        if r["Counter_Name"] != "MfmaUtil":
            continue
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); name = re.sub(r"^void ", "", name).split("(")[0]
        acc.setdefault(name, []).append(float(r["Counter_Value"]))
print("| kernel | dispatches | MfmaUtil % (chip-wide, mean over dispatches) |\n|---|---|---|")
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) > 0:
        print("| \`%s\` | %d | %.1f |" % (k[:80], len(v), sum(v) / len(v)))
PY
cat $O/f32_pmc_mfmautil.md
head -3 $(find /tmp/p_m -name "*counter_collection.csv" | head -1)
