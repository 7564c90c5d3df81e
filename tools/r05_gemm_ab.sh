#!/bin/bash
# fp32 GEMM alone: where do the 10-12 % that the operand loads cost go?  flags 1 = every k-step refetches the tile's first one
# (same instruction stream, same LDS-DMA writes, sources L2-resident), 64 = no loads after the first stage.
B=tools/bin/gemm32_bench
export GEMM32_SKIP_LEGACY=1 LD_LIBRARY_PATH=aas_enhancement_amd/lib
for fl in ${FLS:-0 1 64 0 1 64}; do
  echo "== flags=$fl"; GEMM32_FLAGS=$fl AAS_ABLATION=1 $B time | grep "^time" | grep -E "tn 2000x500x6000|nt 6000x2000x500 \(|nt 12000x2000x500|nt 8192|nn 6000x500x4000" | sed 's/legacy.*|//'
done
