#!/bin/bash
# Round 6: the weight-gradient stream confined to one CU half (knobs.WGRAD_LANE: a hipExtStreamCreateWithCUMask stream, 16 CUs on each
# XCD) - so that a persistent recurrent launch capped at the other half always finds whole CUs.  Same box, one process per setting.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() {
  local tag="$1"; shift
  local args="$1"; shift
  local out
  out=$(env AAS_ABLATION=1 "$@" python3 bench.py $args --no-cpu-baseline --no-traffic --profile-steps 0 --steps 30 --warmup 12 --allow-ablation 2>/dev/null | tail -1)
  python3 - "$tag" "$out" <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[2]); print("%-52s %.3f ms / step  gate %s" % (sys.argv[1], d["ms_per_step"], d["parity_gate"]["status"]))
except Exception as e:
    print("%-52s FAILED %r" % (sys.argv[1], e))
PY
}
for rep in 1 2; do
run "config 5 (AM step): shipped" "--config 5" AAS_NOTHING=1
run "config 5: WGRAD_LANE=1" "--config 5" AAS_WGRAD_LANE=1
run "config 5: WGRAD_LANE=0" "--config 5" AAS_WGRAD_LANE=0
done
run "config 2 headline: shipped" "--no-extras" AAS_NOTHING=1
run "config 2 headline: WGRAD_LANE=1" "--no-extras" AAS_WGRAD_LANE=1
run "config 2 headline: WGRAD_LANE=0" "--no-extras" AAS_WGRAD_LANE=0
run "config 2 trainable A: shipped" "--no-extras --trainable-asr" AAS_NOTHING=1
run "config 2 trainable A: WGRAD_LANE=1" "--no-extras --trainable-asr" AAS_WGRAD_LANE=1
run "config 4 (FSEGAN): shipped" "--config 4" AAS_NOTHING=1
run "config 4: WGRAD_LANE=1 FSEGAN_BWD_CUS=128" "--config 4" AAS_WGRAD_LANE=1 AAS_FSEGAN_BWD_CUS=128
