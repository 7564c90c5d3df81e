#!/bin/bash
# Round 6: config 4 (FSEGAN step) - CU budget of the BPTT launches and held-back discriminator layers.  Same box, one process each.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() {
  local tag="$1"; shift
  local out
  out=$(env AAS_ABLATION=1 "$@" python3 bench.py --config 4 --no-cpu-baseline --no-traffic --profile-steps 0 --steps 30 --warmup 12 --allow-ablation 2>/dev/null | tail -1)
  python3 - "$tag" "$out" <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[2]); print("%-44s %.3f ms / step  gate %s" % (sys.argv[1], d["ms_per_step"], d["parity_gate"]["status"]))
except Exception as e:
    print("%-44s FAILED %r" % (sys.argv[1], e))
PY
}
run "shipped defaults" AAS_NOTHING=1
for c in 128 160 192; do run "FSEGAN_BWD_CUS=$c" AAS_FSEGAN_BWD_CUS=$c; done
for d in 1 2 4; do run "FSEGAN_DEFER_D=$d" AAS_FSEGAN_DEFER_D=$d; done
run "FSEGAN_BWD_CUS=128 FSEGAN_DEFER_D=2" AAS_FSEGAN_BWD_CUS=128 AAS_FSEGAN_DEFER_D=2
run "FSEGAN_BWD_CUS=160 FSEGAN_DEFER_D=2" AAS_FSEGAN_BWD_CUS=160 AAS_FSEGAN_DEFER_D=2
run "FSEGAN_BWD_CUS=192 FSEGAN_DEFER_D=4" AAS_FSEGAN_BWD_CUS=192 AAS_FSEGAN_DEFER_D=4
run "shipped defaults (again)" AAS_NOTHING=1
