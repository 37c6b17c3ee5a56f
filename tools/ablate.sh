#!/bin/bash
# GEMM experiments: variants x ablations on the K=1024 and K=4096 MLP shapes
mkdir -p gpurun_out; : > gpurun_out/ablate.jsonl
for shape in "65536 4096 1024" "65536 1024 4096" "65536 1024 1024"; do
  for nt in 2 3; do
    for abl in 0 1 2 4; do
      CLIPA_GEMM_NT=$nt CLIPA_GEMM_ABL=$abl timeout 120 python tools/gemm_time.py $shape bias >> gpurun_out/ablate.jsonl 2>/dev/null
    done
  done
  CLIPA_GEMM_NT=1 timeout 120 python tools/gemm_time.py $shape bias >> gpurun_out/ablate.jsonl 2>/dev/null
done
cat gpurun_out/ablate.jsonl
