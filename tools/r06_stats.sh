#!/bin/bash
# Round 6: rocprofv3 kernel statistics of every timed path OTHER than the headline - config 4 (FSEGAN), config 5 (AM), config 1 (DCE),
# trainable A, the one-rank RCCL data-parallel step - and, for each, the launches that are not library kernels (at::native, fills).
#   tools/r06_stats.sh [tag]      -> gpurun_out/<tag>/{config4,config5,config1,trainableA,dp}_kernel_stats.csv + non_library_launches.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}
O=$R/gpurun_out/$TAG; mkdir -p $O
COMMON="--steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 --no-traffic"
run() {   # name, extra env (or "-"), bench args
  local name=$1; shift; local envs=$1; shift
  rm -rf /tmp/p_$name
  if [ "$envs" != "-" ]; then export $envs; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -o run -- python3 $R/bench.py $COMMON "$@" > $O/${name}_under_rocprof.json 2> $O/${name}.err
  if [ "$envs" != "-" ]; then unset ${envs%%=*}; fi
  cp $(find /tmp/p_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
  cp $(find /tmp/p_$name -name "*kernel_trace.csv" | head -1) /tmp/${name}_trace.csv
}
run config4 - --config 4
run config5 - --config 5
run config1 - --config 1
run trainableA - --trainable-asr --no-extras
run dp AAS_DP_FORCE=1 --no-extras
python3 - <<PY > $O/non_library_launches.txt
import csv, json
for name in ("config4", "config5", "config1", "trainableA", "dp"):
    rows = list(csv.DictReader(open("$O/%s_kernel_stats.csv" % name)))
    try:
        line = open("$O/%s_under_rocprof.json" % name).read().strip().splitlines()[-1]
        d = json.loads(line)
        print("== %s: %.3f ms / step under rocprof (%s)" % (name, d.get("ms_per_step", float("nan")), d.get("config", {}).get("workload", "")[:90]))
    except Exception as e:
        print("== %s: no bench line (%r)" % (name, e))
    tot = 0
    for r in rows:
        n = r["Name"]
        if n.startswith("void at::") or "at::native" in n or "fillBuffer" in n or "rocclr" in n or "Cijk" in n or "elementwise_kernel" in n:
            tot += int(r["Calls"])
            print("%7s calls %8.1f us avg  %s" % (r["Calls"], float(r["AverageNs"]) / 1e3, n[:120]))
    print("   -> %d non-library launches in the whole run (25 steps + construction + gate)" % tot)
PY
python3 - <<PY > $O/non_library_context.txt
# where in a steady-state step the non-library launches sit: the kernels right before / after each one on its queue
import csv, re
def short(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*$", "", n)[:70]
for name in ("config1", "config4", "config5", "trainableA", "dp"):
    try:
        rows = list(csv.DictReader(open("/tmp/%s_trace.csv" % name)))
    except Exception as e:
        print("==", name, "no trace", e); continue
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    byq = {}
    for r in rows:
        byq.setdefault(r.get("Queue_Id", "0"), []).append(r)
    print("==", name, len(rows), "launches")
    seen = {}
    for q, rs in byq.items():
        n = len(rs)
        for i, r in enumerate(rs):
            k = r["Kernel_Name"]
            if not (k.startswith("void at::") or "at::native" in k or "rocclr" in k):
                continue
            if i < 0.6 * n:        # steady state only (the tail of the run)
                continue
            key = (short(k), short(rs[i - 1]["Kernel_Name"]) if i else "-", short(rs[i + 1]["Kernel_Name"]) if i + 1 < n else "-")
            seen[key] = seen.get(key, 0) + 1
    for (k, a, b), c in sorted(seen.items(), key=lambda kv: -kv[1])[:40]:
        print("%4d x  %-60s  after [%s]  before [%s]" % (c, k, a, b))
PY
cat $O/non_library_launches.txt
