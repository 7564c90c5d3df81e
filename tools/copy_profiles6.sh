#!/bin/bash
# copy one tools/profile_round6.sh result set (gpurun_out/<tag>/) into the tracked profiles/r06_* files and rebuild the derived tables
#   tools/copy_profiles6.sh r06
S=gpurun_out/${1:?tag}
A="--steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 --no-traffic"
cp $S/bench_n1.json profiles/r06_bench_n1.json
cp $S/bench_dp_onerank.json profiles/r06_bench_dp_onerank.json
for f in bench_under_rocprof.json kernel_stats.csv step_launches.csv kernel_timeline.txt pmc_traffic.json event_timeline.txt rnn_bench.txt soak.txt pmc_mfmautil.md; do
  [ -f $S/f32_$f ] && cp $S/f32_$f profiles/r06_f32_$f
done
python tools/pmc_summary.py --rebuild profiles/r06_f32_pmc_traffic.json > /dev/null
cp $S/xcd_stats.txt profiles/r06_xcd_placement_stats.txt
[ -f $S/f32_soak_6000.txt ] && cp $S/f32_soak_6000.txt profiles/r06_f32_soak_6000.txt
[ -f $S/lmfb_ablation.txt ] && grep -v amdgpu.ids $S/lmfb_ablation.txt > profiles/r06_lmfb_ablation.txt
[ -f gpurun_out/${1}_dp_onerank_timeline.txt ] && cp gpurun_out/${1}_dp_onerank_timeline.txt profiles/r06_dp_onerank_timeline.txt
for c in config1 config4 config5; do cp $S/${c}_kernel_stats.csv profiles/r06_${c}_kernel_stats.csv; done
cp $S/trainableA_kernel_stats.csv profiles/r06_f32_trainableA_kernel_stats.csv
cp $S/dp_kernel_stats.csv profiles/r06_dp_kernel_stats.csv
cp $S/non_library_launches.txt profiles/r06_non_library_launches.txt
cp $S/non_library_context.txt profiles/r06_non_library_context.txt
cp $S/HEAD.txt profiles/r06_profiled_commit.txt 2>/dev/null
python tools/stats_md.py profiles/r06_f32_kernel_stats.csv "fp32 headline (bench.py --precision 0 --no-extras $A)"
python tools/stats_md.py profiles/r06_config1_kernel_stats.csv "config 1 minimize_DCE, fp32 (bench.py --config 1 $A)"
python tools/stats_md.py profiles/r06_config4_kernel_stats.csv "config 4 FSEGAN, fp32 (bench.py --config 4 $A)"
python tools/stats_md.py profiles/r06_config5_kernel_stats.csv "config 5 acoustic-model training, fp32 (bench.py --config 5 $A)"
python tools/stats_md.py profiles/r06_f32_trainableA_kernel_stats.csv "config 2 with a trainable A, fp32 (bench.py --trainable-asr --no-extras $A)"
python tools/stats_md.py profiles/r06_dp_kernel_stats.csv "config 2 data parallel on a one-rank RCCL group, fp32 (AAS_DP_FORCE=1 bench.py --no-extras $A)"
