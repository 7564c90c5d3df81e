#!/bin/bash
# Round 6: one GPU-box call that regenerates EVERY measurement artefact kept under profiles/r06_* from the tree as it is
# (run at the final commit; tools/copy_profiles6.sh copies the result set into profiles/).
#   tools/profile_round6.sh r06
set -u
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
git -C $R rev-parse HEAD > $O/HEAD.txt 2>/dev/null || true
# 1. the default bench line (what the driver runs)
python3 $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err
# 2. the one-rank RCCL data-parallel step, timed (no profiler)
AAS_DP_FORCE=1 python3 $R/bench.py --no-extras --no-cpu-baseline --no-traffic > $O/bench_dp_onerank.json 2> $O/bench_dp_onerank.err
# 3. kernel statistics + one step launch by launch: the headline
CMD="bench.py --precision 0 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --profile-steps 0 --no-traffic"
rm -rf /tmp/p_stats /tmp/p_f /tmp/p_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o run -- python3 $R/$CMD > $O/f32_bench_under_rocprof.json 2> /dev/null
cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) $O/f32_kernel_stats.csv
python3 $R/tools/timeline.py $(find /tmp/p_stats -name "*kernel_trace.csv" | head -1) --dump-step $O/f32_step_launches.csv > $O/f32_kernel_timeline.txt 2>&1
# 4. HBM traffic of the same command (two PMC passes) and MFMA utilisation
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_f -o run -- python3 $R/$CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_w -o run -- python3 $R/$CMD > /dev/null 2>&1
python3 $R/tools/pmc_summary.py /tmp/p_f /tmp/p_w $O/f32_pmc_traffic.json "python $CMD" 26 > $O/f32_pmc_top.txt 2>&1
bash $R/tools/r03_pmc_mfma.sh $TAG > /dev/null 2>&1
# 5. every other timed path: config 4 / 5 / 1, trainable A, one-rank RCCL (kernel statistics + the launches that are not library kernels)
bash $R/tools/r06_stats.sh $TAG > /dev/null 2>&1
# 6. event timelines (no tracer), isolated recurrent launches, XCD placement, data-parallel bucket timing
python3 $R/tools/event_timeline.py > $O/f32_event_timeline.txt 2>&1
python3 $R/tools/rnn_bench.py --precision 0 --flags 0,64 --cus 128 > $O/f32_rnn_bench.txt 2>&1
# (E's forward over the whole chip: the launch with its input projection inside beside GEMM + launch; duration against T)
python3 $R/tools/rnn_bench.py --precision 0 --flags 0,64 --only 0 >> $O/f32_rnn_bench.txt 2>&1
python3 $R/tools/rnn_bench.py --precision 0 --tsweep >> $O/f32_rnn_bench.txt 2>&1
python3 $R/tools/xcd_stats.py > $O/xcd_stats.txt 2>&1
bash $R/tools/r05_dp_timeline.sh $TAG > /dev/null 2>&1
# 7. soak: 1500 steps, schedules alternating
python3 $R/tools/soak.py 1500 > $O/f32_soak.txt 2>&1
python3 $R/tools/soak.py 6000 > $O/f32_soak_6000.txt 2>&1
python3 $R/tools/lmfb_bench.py 2048 > $O/lmfb_ablation.txt 2>&1
ls -la $O
