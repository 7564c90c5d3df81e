#!/bin/bash
# Round 5: one GPU-box call that regenerates the fp32 headline's measurement artefacts under gpurun_out/<tag>/.
#   tools/profile_round5.sh r05
set -u
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err
CMD="bench.py --precision 0 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --profile-steps 0 --no-traffic"
rm -rf /tmp/p_stats /tmp/p_f /tmp/p_w /tmp/p_m /tmp/p_c4 /tmp/p_c5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o run -- python3 $R/$CMD > $O/f32_bench_under_rocprof.json 2> /dev/null
cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) $O/f32_kernel_stats.csv
python3 $R/tools/timeline.py $(find /tmp/p_stats -name "*kernel_trace.csv" | head -1) --dump-step $O/f32_step_launches.csv > $O/f32_kernel_timeline.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_f -o run -- python3 $R/$CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_w -o run -- python3 $R/$CMD > /dev/null 2>&1
python3 $R/tools/pmc_summary.py /tmp/p_f /tmp/p_w $O/f32_pmc_traffic.json "python $CMD" 26 > $O/f32_pmc_top.txt 2>&1
python3 $R/tools/event_timeline.py > $O/f32_event_timeline.txt 2>&1
python3 $R/tools/rnn_bench.py --precision 0 --flags 0,64 --cus 128 > $O/f32_rnn_bench.txt 2>&1
LD_LIBRARY_PATH=$R/aas_enhancement_amd/lib $R/tools/bin/gemm32_bench all > $O/gemm32_bench.txt 2>&1
python3 $R/tools/xcd_stats.py > $O/xcd_stats.txt 2>&1
for c in 4 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c$c -o run -- python3 $R/bench.py --config $c --steps 20 --no-cpu-baseline --profile-steps 0 > $O/config${c}_under_rocprof.json 2> /dev/null
  cp $(find /tmp/p_c$c -name "*kernel_stats.csv" | head -1) $O/config${c}_kernel_stats.csv
done
bash $R/tools/r03_pmc_mfma.sh $TAG > /dev/null 2>&1
bash $R/tools/r05_dp_timeline.sh $TAG > /dev/null 2>&1
python3 $R/tools/soak.py 1500 > $O/f32_soak.txt 2>&1
ls -la $O
