#!/bin/bash
# kernel statistics of the one-rank data-parallel step (AAS_DP_FORCE=1): which launches does the DP path add to the single-process step?
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05/dp_stats; mkdir -p $O; rm -rf /tmp/p_dp
export AAS_DP_FORCE=1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_dp -o run -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --profile-steps 0 --no-traffic > $O/bench.json 2> $O/bench.err
cp $(find /tmp/p_dp -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python3 - <<PY
import csv, json
rows = list(csv.DictReader(open("$O/kernel_stats.csv")))
print(open("$O/bench.json").read().strip().splitlines()[-1][:300])
for r in rows:
    n = r["Name"]
    if not any(k in n for k in ("gemm32", "rnn_", "bn_")):
        print("%6s calls %8.1f us avg  %s" % (r["Calls"], float(r["AverageNs"]) / 1e3, n[:110]))
PY
