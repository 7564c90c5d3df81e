cd $GRAFT_REPO_ROOT
export AB_ARGS="--no-extras --no-cpu-baseline --steps 12 --profile-steps 0 --no-traffic"
bash tools/ab.sh "AAS_X=0" "AAS_D_WHOLE_CU=fwd" "AAS_D_WHOLE_CU=bwd" "AAS_D_WHOLE_CU=fwd2" "AAS_D_WHOLE_CU=fwd3" "AAS_D_WHOLE_CU=bwd2" "AAS_D_WHOLE_CU=bwd4" > gpurun_out/r05_ab3.txt 2>&1
cat gpurun_out/r05_ab3.txt
