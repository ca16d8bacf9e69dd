#!/bin/bash
# Runs the reduced LDS-DMA write-after-read reproducer (tools/lds_dma_war.hip) over the protocol x occupancy x injected-skew matrix on the GPU
# box and writes one JSON line per configuration to gpurun_out/lds_dma_war.jsonl (copied to profiles/r03_lds_dma_war.jsonl).
set -u
mkdir -p gpurun_out
out=gpurun_out/lds_dma_war.jsonl
: > $out
T=tools/lds_dma_war
for per_cu in 1 2; do
  for proto in 0 1 2; do
    timeout 120 $T $proto $per_cu -1 0 60 >> $out          # natural skew only
  done
done
# injected skew: one trailing wave (5) held back before its fragment reads; a leading wave (1) as the control
for per_cu in 1 2; do
  for proto in 0 1 2; do
    for d in 8 32 128; do
      timeout 120 $T $proto $per_cu 5 $d 10 >> $out
    done
    timeout 120 $T $proto $per_cu 1 32 10 >> $out
  done
done
cat $out
