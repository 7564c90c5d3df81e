#!/bin/bash
# One GPU-box call that regenerates the round's measurement artefacts under gpurun_out/<tag>/ (copy what is judged into profiles/).
#   tools/profile_round.sh r03       headline = fp32 (library default); the split-bf16 fast mode is profiled beside it
set -u
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err
for P in 0 1 2; do
  N=$([ $P = 0 ] && echo f32 || ([ $P = 1 ] && echo bf16x3 || echo f32eq))
  CMD="bench.py --precision $P --steps 20 --warmup 5 --no-cpu-baseline --no-extras --profile-steps 0"
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats_$N -o run -- python3 $R/$CMD > $O/${N}_bench_under_rocprof.json 2> /dev/null
  cp $(find /tmp/p_stats_$N -name "*kernel_stats.csv" | head -1) $O/${N}_kernel_stats.csv
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_f_$N -o run -- python3 $R/$CMD > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_w_$N -o run -- python3 $R/$CMD > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py /tmp/p_f_$N /tmp/p_w_$N $O/${N}_pmc_traffic.json "python $CMD" 26 > $O/${N}_pmc_top.txt 2>&1
  AAS_PRECISION=$P python3 $R/tools/event_timeline.py > $O/${N}_event_timeline.txt 2>&1
  python3 $R/tools/rnn_bench.py --precision $P --flags 0,64 --cus 128 > $O/${N}_rnn_bench.txt 2>&1
done
for c in 4 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c$c -o run -- python3 $R/bench.py --config $c --steps 20 --no-cpu-baseline --profile-steps 0 > $O/config${c}_under_rocprof.json 2> /dev/null
  cp $(find /tmp/p_c$c -name "*kernel_stats.csv" | head -1) $O/config${c}_kernel_stats.csv
done
python3 $R/tools/x6_error.py > $O/x6_error.txt 2>&1
python3 $R/tools/lmfb_bench.py 2048 > $O/lmfb_ablation.txt 2>&1
bash $R/tools/r03_pmc_mfma.sh $TAG > /dev/null 2>&1
ls -la $O
