#!/bin/bash
# same-box A/B of bench.py under environment settings: tools/ab.sh "ENV1=a ENV2=b" "ENV1=c" ...   (each run twice, interleaved)
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do for e in "$@"; do
  printf "%-60s " "$e"; env AAS_ABLATION=1 $e python bench.py --allow-ablation ${AB_ARGS:---no-extras --no-cpu-baseline --steps 10 --profile-steps 0} 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['ms_per_step'],2), j['dtype'], j['parity_gate']['status'])"
done; done
