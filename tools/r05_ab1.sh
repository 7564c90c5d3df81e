cd $GRAFT_REPO_ROOT
export AB_ARGS="--no-extras --no-cpu-baseline --steps 12 --profile-steps 0 --no-traffic"
bash tools/ab.sh "AAS_X=0" "AAS_CHAIN_PRIO=1" "AAS_DEFER_D_LAYERS=3" "AAS_DEFER_D_LAYERS=1" "AAS_DEFER_D_LAYERS=4" "AAS_EBWD_CUS=160" "AAS_EBWD_CUS=96" > gpurun_out/r05_ab1.txt 2>&1
cat gpurun_out/r05_ab1.txt
