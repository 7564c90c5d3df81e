#!/bin/bash
# One-rank RCCL timeline of the data-parallel step (AAS_DP_FORCE=1: world 1 over RCCL - the data-parallel code path, bucket hooks and
# collective calls included, on one GPU): when does each gradient bucket's all-reduce become eligible relative to the weight-gradient
# products, E's backward and the end of the step?  (With one rank RCCL moves no data: the durations are the model's, below.)
#   tools/r05_dp_timeline.sh <tag>   -> gpurun_out/<tag>_dp_onerank_timeline.txt
set -u
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT
cd $R
AAS_DP_FORCE=1 python3 tools/event_timeline.py --steps 6 --classes rnn,coll > /tmp/dp_ev.txt 2> /tmp/dp_ev.err
python3 - <<'PY' > gpurun_out/${TAG}_dp_onerank_timeline.txt
import re
L = open("/tmp/dp_ev.txt").read().splitlines()
print("one-rank RCCL (AAS_DP_FORCE=1), config 2 fp32, Trainer.train_step_async: HIP events on the issuing streams (tools/event_timeline.py)")
print(L[0])
rows = []
for ln in L:
    m = re.match(r"\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(\S.*)", ln)
    if m:
        rows.append((float(m.group(1)), float(m.group(2)), m.group(4)))
step_end = max(r[1] for r in rows)
print("%9s %9s  %s" % ("start_ms", "end_ms", "launch"))
for s, e, n in rows:
    if n.startswith("allreduce") or "_bwd" in n:
        print("%9.3f %9.3f  %s" % (s, e, n))
print()
colls = [(s, n) for s, e, n in rows if n.startswith("allreduce")]
last_bptt = max(e for s, e, n in rows if "_bwd" in n)
print("last BPTT launch ends at %.3f ms; buckets become eligible at:" % last_bptt)
tot = 0.0
for s, n in colls:
    mb = float(re.search(r"([\d.]+) MB", n).group(1))
    tot += mb
    ring = 2 * (7 / 8) * mb * 2 ** 20 / 153e9 * 1e3      # 8-rank ring, one xGMI link direction
    direct = 2 * (7 / 8) * mb * 2 ** 20 / (7 * 153e9) * 1e3   # reduce-scatter + all-gather over all 7 links
    print("  %8.3f ms  %6.1f MB   modelled 8-GPU duration: ring %.3f ms, all-links %.3f ms   -> done by %.3f / %.3f ms" % (s, mb, ring, direct, s + ring, s + direct))
print("total %.1f MB per step" % tot)
PY
cat gpurun_out/${TAG}_dp_onerank_timeline.txt
