#!/bin/bash
# tile height x split-K sweep of the LDS-DMA fp32 GEMM on the step's shapes (tools/gemm32_bench.cpp)
B=tools/bin/gemm32_bench
echo "== check"; $B check | grep -c " ok"; $B check | grep -v " ok"
echo "== default (chooser)"; $B time
export GEMM32_SKIP_LEGACY=1
for cfg in ${SWEEP:-"128 1" "128 2" "128 4" "128 8" "64 1" "64 2" "64 3" "64 6"}; do set -- $cfg
  echo "== BM=$1 SK=$2"; AAS_ABLATION=1 AAS_GEMM32_BM=$1 AAS_GEMM32_SK=$2 $B time | grep "^time" | sed 's/legacy.*|//'
done
