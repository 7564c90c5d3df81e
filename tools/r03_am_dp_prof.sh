#!/bin/bash
# kernel stats of the AM step (config 5) with and without the data-parallel code path on a one-rank RCCL group -> gpurun_out/am_dp/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/am_dp; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --profile-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/amdp0 -o run -- python3 $R/$CMD > $O/nodp.json 2>/dev/null
cp $(find /tmp/amdp0 -name "*kernel_stats.csv" | head -1) $O/nodp_kernel_stats.csv
export AAS_DP_FORCE=1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/amdp1 -o run -- python3 $R/$CMD > $O/dp.json 2>/dev/null
cp $(find /tmp/amdp1 -name "*kernel_stats.csv" | head -1) $O/dp_kernel_stats.csv
