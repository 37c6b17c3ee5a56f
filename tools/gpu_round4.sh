#!/bin/bash
# Round-4 GPU sessions.  Usage: gpurun --timeout N -- 'bash tools/gpu_round4.sh [stage...]'
set -u
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R="$PWD"
STAGES="${*:-newtests benchq}"
for s in $STAGES; do
  case $s in
    alltests)
      timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s --durations=15 > gpurun_out/pytest_gpu.log 2>&1
      echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log ;;
    newtests)
      timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_fp8_gpu.py -m gpu -q --tb=short -p no:cacheprovider -s \
        -k "${NEWTESTS_K:-gemm or bf16_mode or handoff or f8a or recompute}" > gpurun_out/pytest_new.log 2>&1
      echo "pytest rc=$?" >> gpurun_out/pytest_new.log ;;
    bench)
      timeout 900 python bench.py --shapes > gpurun_out/bench.log 2>&1; echo "rc=$?" >> gpurun_out/bench.log ;;
    benchq)
      timeout 600 python bench.py --shapes --steps 3 --warmup 1 --no-cpu-baseline --h2d-steps 0 --plain-steps 0 ${BENCHQ_ARGS:-} > gpurun_out/benchq.log 2>&1; echo "rc=$?" >> gpurun_out/benchq.log ;;
    benchq2)
      timeout 600 python bench.py --shapes --steps 3 --warmup 1 --no-cpu-baseline --h2d-steps 0 --plain-steps 0 ${BENCHQ2_ARGS:-} > gpurun_out/benchq2.log 2>&1; echo "rc=$?" >> gpurun_out/benchq2.log ;;
    benchdist)
      # the multi-rank code path (RCCL group, DDP, vote-based keep plan, per-rank record) on one GPU
      CLIPA_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 \
        bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --h2d-steps 0 --plain-steps 0 ${BENCHDIST_ARGS:-} > gpurun_out/bench_dist1.log 2>&1; echo "rc=$?" >> gpurun_out/bench_dist1.log ;;
    gemmab)
      timeout 600 ./tools/probes/gemm_nta_ab ${GEMMAB_ARGS:-} > gpurun_out/gemm_nt_asm_ab.log 2>&1; echo "rc=$?" >> gpurun_out/gemm_nt_asm_ab.log ;;
    flagsweep)
      timeout 600 ./tools/probes/gemm_flag_sweep ${FLAGSWEEP_ARGS:-} > gpurun_out/gemm_flag_sweep.jsonl 2>&1; echo "rc=$?" >> gpurun_out/gemm_flag_sweep.jsonl ;;
    storepol)
      timeout 600 python tools/gemm_lib_ab.py clipa_amd/lib/libclipa_hip.so $(ls clipa_amd/lib/libclipa_var_st*.so) > gpurun_out/store_policy_ab.jsonl 2>&1; echo "rc=$?" >> gpurun_out/store_policy_ab.jsonl ;;
    libab)
      timeout 600 python tools/gemm_lib_ab.py clipa_amd/lib/libclipa_hip.so ${LIBAB_LIBS} > gpurun_out/gemm_lib_ab.jsonl 2>&1; echo "rc=$?" >> gpurun_out/gemm_lib_ab.jsonl ;;
    streamab)
      # grid shape / cache policy sweep of the HBM-streaming kernels, then old-vs-new library (build both with the recipe in tools/stream_lib_ab.py)
      timeout 200 ./tools/probes/stream_ab 806912 ${STREAMAB_PHASE:-2} > gpurun_out/stream_ab.jsonl 2>&1; echo "rc=$?" >> gpurun_out/stream_ab.jsonl
      timeout 360 python tools/stream_lib_ab.py tools/probes/lnvar/libstream_old.so tools/probes/lnvar/libstream_new.so > gpurun_out/stream_lib_ab.jsonl 2>&1; echo "rc=$?" >> gpurun_out/stream_lib_ab.jsonl ;;
    tnab)
      timeout 600 ./tools/probes/gemm_tna_ab ${GEMMAB_ARGS:-} > gpurun_out/gemm_tna_ab.log 2>&1; echo "rc=$?" >> gpurun_out/gemm_tna_ab.log ;;
    fp8conv)
      for seed in ${FP8CONV_SEEDS:-1 2}; do
        timeout 600 python tools/fp8_convergence.py --steps 200 --batch 256 --lr ${FP8CONV_LR:-3e-4} --seed $seed --oracle-steps 0 > gpurun_out/fp8_convergence_seed$seed.jsonl 2> gpurun_out/fp8_convergence_seed$seed.err; echo "rc=$?" >> gpurun_out/fp8_convergence_seed$seed.err
      done ;;
    others)
      timeout 400 python bench.py --model ViT-B-16 --steps 3 --warmup 1 --no-cpu-baseline --h2d-steps 0 --plain-steps 0 > gpurun_out/bench_b16.log 2>&1
      for prec in bf16 fp8; do
        timeout 500 python bench.py --model ViT-H-14 --batch 2048 --precision $prec --steps 2 --warmup 1 --no-cpu-baseline --h2d-steps 0 --plain-steps 0 --shapes > gpurun_out/bench_h14_$prec.log 2>&1
      done
      timeout 400 python bench.py --model ViT-L-16 --image-size 84 --steps 3 --warmup 1 --no-cpu-baseline --h2d-steps 0 --plain-steps 0 > gpurun_out/bench_l16_84.log 2>&1 ;;
    cfg4)
      # BASELINE configs[3] at its own per-GPU batch: ViT-H/14 @ 224, fp8, 8192 pairs per GPU as 4 micro-batches (accum_freq feature cache)
      timeout 700 python bench.py --model ViT-H-14 --batch 8192 --accum-freq 4 --precision fp8 --steps 2 --warmup 1 --no-cpu-baseline --h2d-steps 0 --plain-steps 0 > gpurun_out/bench_h14_B8192_accum4_fp8.log 2>&1; echo "rc=$?" >> gpurun_out/bench_h14_B8192_accum4_fp8.log ;;
    driverbench)
      timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_form.log 2>&1; echo "rc=$?" >> gpurun_out/bench_driver_form.log ;;
    smoke)
      timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "rc=$?" >> gpurun_out/smoke.log ;;
    stats)
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof" -o r04 -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --h2d-steps 0 --plain-steps 0 > "$R/gpurun_out/bench_prof.log" 2>&1)
      tail -3 gpurun_out/bench_prof.log | cut -c1-400
      db=$(find gpurun_out/prof -name '*.db' | head -1)
      [ -n "$db" ] && python tools/rocpd_stats.py "$db" > gpurun_out/kernel_stats.csv 2>&1
      rm -rf gpurun_out/prof ;;      # (only <= 64 MiB of gpurun_out/ travel back: the summaries, not the databases)
    pmcbench)
      # one bench step per counter set, each in its own rocprofv3 pass (MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE never share a
      # pass; no tracing domains next to --pmc): traffic, MFMA busy + wait states + effective clock, LDS conflicts, L2 hit rate
      mkdir -p gpurun_out/pmcbench
      declare -A SETS=( [FETCH_SIZE]="FETCH_SIZE" [WRITE_SIZE]="WRITE_SIZE"
                        [MFMA]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
                        [LDS]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INST_CYCLES_VMEM"
                        [L2]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_STALL_sum" )
      for set in ${PMC_SETS:-FETCH_SIZE WRITE_SIZE MFMA LDS L2}; do
        (cd /tmp && timeout 400 rocprofv3 --pmc ${SETS[$set]} -d "$R/gpurun_out/pmcbench/$set" -o pmc -- \
           python "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --h2d-steps 0 --plain-steps 0 --keep-blocks ${PMC_KEEP:-0,0,24,4} > "$R/gpurun_out/pmcbench/$set.log" 2>&1)
        tail -2 "$R/gpurun_out/pmcbench/$set.log" | cut -c1-300
      done
      python tools/pmc_summary.py gpurun_out/pmcbench gemm attn ln_ > gpurun_out/pmcbench_summary.txt 2>&1
      find gpurun_out/pmcbench -name '*.db' -delete ;;
    *) echo "unknown stage $s" ;;
  esac
done
ls -la gpurun_out | tail -30
