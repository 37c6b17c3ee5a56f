#!/bin/bash
# One batched GPU-box session: parity tests, smoke, micro-benchmarks, a short bench.  Everything lands in
# gpurun_out/ (merged back by gpurun).  Usage: gpurun --timeout 1800 -- 'bash tools/gpu_round.sh [stage...]'
set -u
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
STAGES="${*:-tests smoke micro bench_small}"
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
nproc >> gpurun_out/gpu.txt
for s in $STAGES; do
  case $s in
    tests)
      timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
      echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log ;;
    smoke)
      timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "rc=$?" >> gpurun_out/smoke.log ;;
    micro)
      timeout 900 python tools/microbench.py --quick > gpurun_out/microbench.jsonl 2> gpurun_out/microbench.err ;;
    microfull)
      timeout 1200 python tools/microbench.py > gpurun_out/microbench_full.jsonl 2> gpurun_out/microbench_full.err ;;
    bench_small)
      timeout 600 python bench.py --batch 512 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_small.log 2>&1
      echo "rc=$?" >> gpurun_out/bench_small.log ;;
    bench)
      timeout 1500 python bench.py > gpurun_out/bench.log 2>&1; echo "rc=$?" >> gpurun_out/bench.log ;;
    pmc)
      # hardware counters for the dominant GEMM shape (own passes, no trace domains mixed in)
      mkdir -p gpurun_out/pmc
      R="$PWD"
      for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
                 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE" \
                 "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
        tag=$(echo $set | cut -d' ' -f1)
        (cd /tmp && timeout 300 rocprofv3 --pmc $set -d "$R/gpurun_out/pmc/$tag" -o pmc -- \
           python "$R/tools/gemm_probe.py" nt 65536 4096 1024 bias 3 > "$R/gpurun_out/pmc/$tag.log" 2>&1)
      done
      python tools/pmc_summary.py gpurun_out/pmc > gpurun_out/pmc_summary.txt 2>&1 ;;
    pmcbench)
      # HBM-side traffic counters over the real bench workload (separate passes, counters only)
      mkdir -p gpurun_out/pmcbench
      R="$PWD"
      for set in "FETCH_SIZE" "WRITE_SIZE"; do
        (cd /tmp && timeout 600 rocprofv3 --pmc $set -d "$R/gpurun_out/pmcbench/$set" -o pmc -- \
           python "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --keep-blocks 0,0,23,2 > "$R/gpurun_out/pmcbench/$set.log" 2>&1)
      done
      python tools/pmc_summary.py gpurun_out/pmcbench gemm attn > gpurun_out/pmcbench_summary.txt 2>&1 ;;
    prof)
      mkdir -p gpurun_out/prof
      (cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o r01 -- \
         python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OLDPWD/gpurun_out/prof_bench.log" 2>&1)
      echo "rc=$?" >> gpurun_out/prof_bench.log ;;
  esac
done
echo "=== pytest tail"; tail -n 40 gpurun_out/pytest_gpu.log 2>/dev/null
echo "=== smoke"; tail -n 5 gpurun_out/smoke.log 2>/dev/null
echo "=== micro"; tail -n 60 gpurun_out/microbench.jsonl 2>/dev/null; tail -n 5 gpurun_out/microbench.err 2>/dev/null
echo "=== bench"; tail -n 6 gpurun_out/bench_small.log 2>/dev/null; tail -n 6 gpurun_out/bench.log 2>/dev/null
