#!/bin/bash
# Round 6: held-back weight-gradient layers with a TRAINABLE acoustic model (the reference's default, allow_ASR_update_iter = 0): the
# shipped DEFER_D_LAYERS = 2 / DEFER_A_LAYERS = 0 were tuned for the frozen-A headline.  Same box, one process per setting.
#   tools/r06_trainableA_sweep.sh [1|2] > gpurun_out/r06_trainableA_sweep.txt      (2: the second pass around the first's winner)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() {
  local tag="$1"; shift
  local out
  out=$(env AAS_ABLATION=1 "$@" python3 bench.py --trainable-asr --no-extras --no-cpu-baseline --no-traffic --profile-steps 0 --steps 30 --warmup 12 --allow-ablation 2>/dev/null | tail -1)
  python3 - "$tag" "$out" <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[2]); print("%-40s %.3f ms / step  gate %s" % (sys.argv[1], d["ms_per_step"], d["parity_gate"]["status"]))
except Exception as e:
    print("%-40s FAILED %r" % (sys.argv[1], e))
PY
}
python3 bench.py --trainable-asr --no-extras --no-cpu-baseline --no-traffic --profile-steps 0 --steps 30 --warmup 12 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-40s %.3f ms / step  gate %s' % ('shipped defaults', d['ms_per_step'], d['parity_gate']['status']))"
if [ "${1:-1}" = "1" ]; then
for a in 1 2 3 5; do run "DEFER_A_LAYERS=$a" AAS_DEFER_A_LAYERS=$a; done
for d in 0 1 3 4; do run "DEFER_D_LAYERS=$d" AAS_DEFER_D_LAYERS=$d; done
run "DEFER_A_LAYERS=2 DEFER_D_LAYERS=1" AAS_DEFER_A_LAYERS=2 AAS_DEFER_D_LAYERS=1
run "DEFER_A_LAYERS=1 DEFER_D_LAYERS=3" AAS_DEFER_A_LAYERS=1 AAS_DEFER_D_LAYERS=3
run "EARLY_ADAM=0" AAS_EARLY_ADAM=0
run "EBWD_CUS=160" AAS_EBWD_CUS=160
run "EBWD_CUS=96" AAS_EBWD_CUS=96
else    # second pass: around the winner of the first (all of A's layers held back)
run "DEFER_A_LAYERS=5" AAS_DEFER_A_LAYERS=5
run "DEFER_A_LAYERS=4" AAS_DEFER_A_LAYERS=4
run "DEFER_A_LAYERS=5 DEFER_D_LAYERS=0" AAS_DEFER_A_LAYERS=5 AAS_DEFER_D_LAYERS=0
run "DEFER_A_LAYERS=5 DEFER_D_LAYERS=1" AAS_DEFER_A_LAYERS=5 AAS_DEFER_D_LAYERS=1
run "DEFER_A_LAYERS=5 DEFER_D_LAYERS=3" AAS_DEFER_A_LAYERS=5 AAS_DEFER_D_LAYERS=3
run "DEFER_A_LAYERS=5 DEFER_D_LAYERS=4" AAS_DEFER_A_LAYERS=5 AAS_DEFER_D_LAYERS=4
run "DEFER_A_LAYERS=5 EBWD_CUS=112" AAS_DEFER_A_LAYERS=5 AAS_EBWD_CUS=112
run "DEFER_A_LAYERS=5 EBWD_CUS=144" AAS_DEFER_A_LAYERS=5 AAS_EBWD_CUS=144
run "DEFER_A_LAYERS=5 EARLY_ADAM=0" AAS_DEFER_A_LAYERS=5 AAS_EARLY_ADAM=0
run "DEFER_A_LAYERS=5 (again)" AAS_DEFER_A_LAYERS=5
fi
python3 bench.py --trainable-asr --no-extras --no-cpu-baseline --no-traffic --profile-steps 0 --steps 30 --warmup 12 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-40s %.3f ms / step  gate %s' % ('shipped defaults (again)', d['ms_per_step'], d['parity_gate']['status']))"
