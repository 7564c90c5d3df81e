#!/bin/bash
B=tools/bin/gemm32_bench
export GEMM32_SKIP_LEGACY=1
for cfg in "128 1" "128 4" "128 8"; do set -- $cfg
for fl in 0 64 16 80; do
  echo "== BM=$1 SK=$2 flags=$fl (64: no loads after the first, 16: no MFMA / LDS reads)"; GEMM32_FLAGS=$fl AAS_ABLATION=1 AAS_GEMM32_BM=$1 AAS_GEMM32_SK=$2 $B time | grep "^time" | grep -E "tn 2000x500x6000|nt 6000x2000x500 \(|nt 8192|nn 6000x500x4000" | sed 's/legacy.*|//'
done; done
