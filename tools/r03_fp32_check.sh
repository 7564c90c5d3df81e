#!/bin/bash
# exact-fp32 data-is-the-flag kernels: parity tests in fp32 mode, then old (debug bit 134217728) vs new timings, isolated
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03c}; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "fp32 or bptt or xcd" 2>&1 | tail -15 > $O/pytest_fp32.txt
cat $O/pytest_fp32.txt
for cus in 0 128; do
  echo "== precision 0, cus=$cus, flags 0 = new exact kernels, 134217728 = counter-based kernels" >> $O/rnn_bench_fp32.txt
  timeout 600 python tools/rnn_bench.py --precision 0 --cus $cus --flags 0,134217728,64 >> $O/rnn_bench_fp32.txt 2>&1
done
cat $O/rnn_bench_fp32.txt
timeout 600 python bench.py --precision 0 --no-extras --no-cpu-baseline --steps 10 > $O/bench_fp32.json 2> $O/bench_fp32.err; tail -2 $O/bench_fp32.err
python - <<PY
import json
j=json.loads(open("$O/bench_fp32.json").read().strip().splitlines()[-1])
print(j["ms_per_step"], j["parity_gate"], j["roofline"]["critical_path_ms"])
for k,v in j["roofline"]["kernels"].items(): print(k, {a:round(b,3) for a,b in v.items()})
PY
