#!/bin/bash
# Round-2 GPU session (fp8 path): parity tests, fp8-vs-bf16 GEMM A/B on the production shapes, ViT-H/14 bench in both
# precisions, hardware counters of the two GEMM kernels.  Usage: gpurun --timeout 1500 -- 'bash tools/gpu_round2.sh [stage...]'
set -u
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R="$PWD"
STAGES="${*:-fp8tests f8bench h14 tests pmc}"
for s in $STAGES; do
  case $s in
    fp8tests)
      timeout 600 python -m pytest tests/test_fp8_gpu.py -q --tb=short -s -p no:cacheprovider > gpurun_out/pytest_fp8.log 2>&1
      echo "pytest rc=$?" >> gpurun_out/pytest_fp8.log ;;
    tests)
      timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --deselect tests/test_fp8_gpu.py > gpurun_out/pytest_gpu.log 2>&1
      echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log ;;
    f8bench)
      timeout 400 python tools/f8_bench.py > gpurun_out/f8_bench.jsonl 2> gpurun_out/f8_bench.err ;;
    h14)
      for prec in fp8 bf16; do
        timeout 400 python bench.py --model ViT-H-14 --batch 2048 --precision $prec --steps 2 --warmup 1 --no-cpu-baseline --shapes \
          > gpurun_out/bench_h14_$prec.log 2>&1
        echo "rc=$?" >> gpurun_out/bench_h14_$prec.log
      done ;;
    l16fp8)
      timeout 400 python bench.py --precision fp8 --steps 2 --warmup 1 --no-cpu-baseline --shapes > gpurun_out/bench_l16_fp8.log 2>&1
      echo "rc=$?" >> gpurun_out/bench_l16_fp8.log ;;
    pmc)
      mkdir -p gpurun_out/pmc2
      for kind in nt f8; do
        (cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
           -d "$R/gpurun_out/pmc2/$kind" -o pmc -- python "$R/tools/gemm_probe.py" $kind 806912 4096 1024 gelu 4 > "$R/gpurun_out/pmc2/$kind.log" 2>&1)
      done
      python tools/pmc_summary.py gpurun_out/pmc2 gemm quantize > gpurun_out/pmc2_summary.txt 2>&1
      for kind in nt f8; do
        db=$(find gpurun_out/pmc2/$kind -name '*.db' | head -1)
        [ -n "$db" ] && python tools/rocpd_stats.py "$db" >> gpurun_out/pmc2_durations.txt 2>&1
      done ;;
    bench)
      timeout 900 python bench.py --shapes > gpurun_out/bench.log 2>&1; echo "rc=$?" >> gpurun_out/bench.log ;;
    sharded)
      # pre-flight of the sharded optimizer's RCCL call sequence inside the real step (1-rank group, forced collectives)
      CLIPA_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
        --master-port 29611 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --h2d-steps 0 --optimizer sharded \
        > gpurun_out/bench_sharded.log 2>&1
      echo "rc=$?" >> gpurun_out/bench_sharded.log ;;
    alltests)
      timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
      echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log ;;
    smoke)
      timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "rc=$?" >> gpurun_out/smoke.log ;;
    pmcbench)
      # fabric-side traffic counters over the real bench step (separate passes, counters only)
      mkdir -p gpurun_out/pmcbench
      for set in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && timeout 400 rocprofv3 --pmc $set -d "$R/gpurun_out/pmcbench/$set" -o pmc -- \
           python "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --h2d-steps 0 --keep-blocks 0,0,24,4 > "$R/gpurun_out/pmcbench/$set.log" 2>&1)
      done
      python tools/pmc_summary.py gpurun_out/pmcbench gemm attn > gpurun_out/pmcbench_summary.txt 2>&1 ;;
    attnab)
      echo "tools/attn_ab.py was removed with the experiment (profiles/r02_attention_bwd_pipelining_ab.jsonl)" ;;
    profh14)
      mkdir -p gpurun_out/prof_h14
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_h14" -o r02h14 -- \
         python "$R/bench.py" --model ViT-H-14 --batch 2048 --precision fp8 --steps 2 --warmup 1 --no-cpu-baseline --h2d-steps 0 \
         > "$R/gpurun_out/prof_h14_bench.log" 2>&1)
      echo "rc=$?" >> gpurun_out/prof_h14_bench.log
      db=$(find gpurun_out/prof_h14 -name '*.db' | head -1)
      [ -n "$db" ] && python tools/rocpd_stats.py "$db" gpurun_out/prof_h14_kernel_stats.csv > gpurun_out/prof_h14_kernel_stats.txt 2>&1 ;;
    prof)
      mkdir -p gpurun_out/prof
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof" -o r02 -- \
         python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/prof_bench.log" 2>&1)
      echo "rc=$?" >> gpurun_out/prof_bench.log
      db=$(find gpurun_out/prof -name '*.db' | head -1)
      [ -n "$db" ] && python tools/rocpd_stats.py "$db" gpurun_out/prof_kernel_stats.csv > gpurun_out/prof_kernel_stats.txt 2>&1 ;;
  esac
done
echo "=== fp8 tests"; tail -n 30 gpurun_out/pytest_fp8.log 2>/dev/null
echo "=== all gpu tests"; tail -n 8 gpurun_out/pytest_gpu.log 2>/dev/null
echo "=== f8 bench"; cat gpurun_out/f8_bench.jsonl 2>/dev/null; tail -n 3 gpurun_out/f8_bench.err 2>/dev/null
echo "=== h14"; tail -n 2 gpurun_out/bench_h14_fp8.log 2>/dev/null | cut -c1-1500; tail -n 2 gpurun_out/bench_h14_bf16.log 2>/dev/null | cut -c1-1500
echo "=== bench"; tail -n 3 gpurun_out/bench.log 2>/dev/null | cut -c1-3000
echo "=== l16 fp8"; tail -n 2 gpurun_out/bench_l16_fp8.log 2>/dev/null | cut -c1-1500
echo "=== sharded"; tail -n 3 gpurun_out/bench_sharded.log 2>/dev/null | cut -c1-1500
echo "=== smoke"; tail -n 3 gpurun_out/smoke.log 2>/dev/null
echo "=== prof"; head -n 16 gpurun_out/prof_kernel_stats.txt 2>/dev/null
echo "=== pmcbench"; cat gpurun_out/pmcbench_summary.txt 2>/dev/null | head -30
echo "=== attn ab"; cat gpurun_out/attn_ab.jsonl 2>/dev/null; tail -n 3 gpurun_out/attn_ab.err 2>/dev/null
echo "=== prof h14"; head -n 14 gpurun_out/prof_h14_kernel_stats.txt 2>/dev/null
