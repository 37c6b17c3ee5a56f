#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/ab.jsonl
for shape in "65536 4096 1024 bias" "65536 3072 1024 bias" "65536 1024 4096 bias" "65536 1024 1024 bias"; do
  timeout 200 python tools/gemm_ab.py $shape 5:0 5:16 3:0 >> gpurun_out/ab.jsonl 2>/dev/null
done
cat gpurun_out/ab.jsonl
