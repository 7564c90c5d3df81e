#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; cd $R
python tools/rnn_bench.py --precision 0 --skip-rnn --gemm --gflags 0,268435456 2>&1 | grep gemm
for f in 0 268435456; do echo "== AAS_DEBUG_FLAGS=$f"; AAS_ABLATION=1 AAS_DEBUG_FLAGS=$f python bench.py --allow-ablation --precision 0 --no-extras --no-cpu-baseline --steps 10 --profile-steps 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['ms_per_step'],2), j['parity_gate']['status'])"; done
