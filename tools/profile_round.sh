#!/bin/bash
# One GPU-box call that regenerates the round's measurement artefacts under gpurun_out/<tag>/ (copy what is judged into profiles/).
#   tools/profile_round.sh r02d
set -u
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err
for c in 1 4 5; do python3 $R/bench.py --config $c > $O/bench_config$c.json 2> $O/bench_config$c.err; done
CMD="bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --profile-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o run -- python3 $R/$CMD > $O/bench_under_rocprof.json 2> /dev/null
cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_f -o run -- python3 $R/$CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_w -o run -- python3 $R/$CMD > /dev/null 2>&1
python3 $R/tools/pmc_summary.py /tmp/p_f /tmp/p_w $O/pmc_traffic.json "python $CMD" 26 > $O/pmc_top.txt 2>&1
python3 $R/tools/event_timeline.py > $O/event_timeline.txt 2>&1
(cd $R/tools && python3 xcd_check.py) > $O/xcd_ab.txt 2>&1
python3 $R/tools/wgrad_bench.py > $O/wgrad_bench.txt 2>&1
python3 $R/tools/rnn_bench.py --flags 0,64 --cus 128 > $O/rnn_bench.txt 2>&1
ls -la $O
