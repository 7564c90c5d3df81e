#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03d}; mkdir -p $O
cd $R
AAS_PRECISION=0 timeout 600 python tools/event_timeline.py > $O/fp32_event_timeline.txt 2>&1
echo "== U=16 BPTT (flag 512), precision 0, cus 128 / 0" > $O/rnn_bench_fp32_u16.txt
timeout 600 python tools/rnn_bench.py --precision 0 --cus 128 --flags 0,512 --only 0,2 >> $O/rnn_bench_fp32_u16.txt 2>&1
timeout 600 python tools/rnn_bench.py --precision 0 --cus 0 --flags 0,512 --only 0,2 >> $O/rnn_bench_fp32_u16.txt 2>&1
cat $O/rnn_bench_fp32_u16.txt
timeout 600 python tools/rnn_bench.py --precision 0 --skip-rnn --gemm > $O/gemm_fp32.txt 2>&1
cat $O/gemm_fp32.txt
