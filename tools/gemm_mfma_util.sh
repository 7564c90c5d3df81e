#!/bin/bash
# MfmaUtil of the fp32 GEMM alone on the chip (is the matrix pipe already saturated at 117 TFLOP/s?)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d /tmp/p_g -o run -- python3 $R/tools/rnn_bench.py --precision 0 --skip-rnn --gemm > /tmp/g.txt 2>&1
grep gemm /tmp/g.txt
python3 - <<PY
import csv, glob, re
acc = {}
for f in glob.glob("/tmp/p_g/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "MfmaUtil": continue
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
        key = (name, r["Grid_Size"])
        acc.setdefault(key, []).append(float(r["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) > 0: print("%-70s grid=%-9s n=%3d MfmaUtil %.1f" % (k[0][:70], k[1], len(v), sum(v)/len(v)))
PY
