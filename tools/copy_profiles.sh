#!/bin/bash
# copy one tools/profile_round.sh result set (gpurun_out/<tag>/) into the tracked profiles/r03_* files and rebuild the derived tables
#   tools/copy_profiles.sh r03o
S=gpurun_out/${1:?tag}
A="--steps 20 --warmup 5 --no-cpu-baseline --no-extras --profile-steps 0"
cp $S/bench_n1.json profiles/r03_bench_n1.json
for m in f32 bf16x3 f32eq; do
  for f in bench_under_rocprof.json event_timeline.txt kernel_stats.csv pmc_traffic.json rnn_bench.txt; do cp $S/${m}_$f profiles/r03_${m}_$f; done
  python tools/pmc_summary.py --rebuild profiles/r03_${m}_pmc_traffic.json > /dev/null
done
for c in 4 5; do cp $S/config${c}_kernel_stats.csv profiles/r03_config${c}_kernel_stats.csv; done
cp $S/x6_error.txt profiles/r03_x6_error.txt; cp $S/lmfb_ablation.txt profiles/r03_lmfb_ablation.txt
python tools/stats_md.py profiles/r03_f32_kernel_stats.csv "fp32 headline (bench.py --precision 0 $A)"
python tools/stats_md.py profiles/r03_bf16x3_kernel_stats.csv "split-bf16 fast mode (bench.py --precision 1 $A)"
python tools/stats_md.py profiles/r03_f32eq_kernel_stats.csv "fp32-equivalent mode (bench.py --precision 2 $A)"
python tools/stats_md.py profiles/r03_config4_kernel_stats.csv "config 4 FSEGAN, fp32 (bench.py --config 4 --steps 20)"
python tools/stats_md.py profiles/r03_config5_kernel_stats.csv "config 5 acoustic-model training, fp32 (bench.py --config 5 --steps 20)"
