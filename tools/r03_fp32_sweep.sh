#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03e}; mkdir -p $O
cd $R
run() { echo "== $*" >> $O/sweep.txt; env AAS_ABLATION=1 "$@" timeout 300 python bench.py --allow-ablation --precision 0 --no-extras --no-cpu-baseline --steps 10 --profile-steps 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['ms_per_step'],2), j['parity_gate']['status'])" >> $O/sweep.txt 2>&1; }
run AAS_X=0
run AAS_DEFER_D_LAYERS=0
run AAS_DEFER_D_LAYERS=1
run AAS_DEFER_D_LAYERS=0 AAS_EBWD_CUS=0
run AAS_DEFER_D_LAYERS=0 AAS_EBWD_CUS=0 AAS_DEBUG_FLAGS=512
run AAS_DEFER_D_LAYERS=0 AAS_DEBUG_FLAGS=512
run AAS_DEFER_D_LAYERS=0 AAS_EBWD_CUS=192
run AAS_DEFER_D_LAYERS=0 AAS_TWO_LANES=1
run AAS_DEFER_D_LAYERS=0 AAS_WGRAD_PRIO=0
run AAS_DEFER_D_LAYERS=0 AAS_OVERLAP_ASR=0
cat $O/sweep.txt
