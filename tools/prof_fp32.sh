#!/bin/bash
# GPU-box call: where does the exact-fp32 step (aas_set_precision(0)) spend its time?  -> gpurun_out/<tag>/
set -u
TAG=${1:-r03a}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --precision 0 --no-extras --no-cpu-baseline --steps 10 > $O/bench_fp32.json 2> $O/bench_fp32.err
CMD="bench.py --precision 0 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --profile-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o run -- python3 $R/$CMD > $O/bench_fp32_under_rocprof.json 2> /dev/null
cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) $O/fp32_kernel_stats.csv
ls -la $O
