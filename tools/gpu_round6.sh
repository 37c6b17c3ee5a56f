#!/bin/bash
# Round-6 GPU sessions.  Usage: gpurun --timeout N -- 'bash tools/gpu_round6.sh [stage...]'
# Every stage writes under gpurun_out/r06/ (merged back by gpurun); summaries that are kept get copied into profiles/ by hand.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
R="$PWD"
O="$R/gpurun_out/r06"
STAGES="${*:-h14}"
H14="--model ViT-H-14 --batch 2048 --no-cpu-baseline --h2d-steps 0 --plain-steps 0 --exact-steps 0 --unpad-steps 0 --fp8-line-steps 0"
for s in $STAGES; do
  case $s in
    h14)
      # BASELINE configs[3] dimensions at the local batch that fits one GPU: bf16 and fp8, same box, per-shape lines
      for prec in bf16 fp8; do
        timeout 600 python bench.py $H14 --precision $prec --steps ${H14_STEPS:-6} --warmup 2 --shapes > $O/bench_h14_$prec.json 2> $O/bench_h14_$prec.err
        cut -c1-330 $O/bench_h14_$prec.json
      done ;;
    cfg4)
      # BASELINE configs[3] at its own per-GPU batch (65536 / 8 = 8192) as four micro-batches with the reference's feature cache
      timeout 900 python bench.py --model ViT-H-14 --batch 8192 --accum-freq 4 --precision fp8 --steps 2 --warmup 1 --no-cpu-baseline --h2d-steps 0 --plain-steps 0 --exact-steps 0 --unpad-steps 0 --fp8-line-steps 0 > $O/bench_h14_B8192_accum4_fp8.json 2> $O/bench_h14_B8192_accum4_fp8.err
      cut -c1-330 $O/bench_h14_B8192_accum4_fp8.json ;;
    stats8)
      # rocprofv3 kernel trace of the fp8 step (summary only travels back)
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$O/prof8" -o r06f8 -- python "$R/bench.py" $H14 --precision fp8 --steps 3 --warmup 1 > "$O/bench_h14_fp8_under_rocprof.json" 2> "$O/bench_h14_fp8_under_rocprof.err")
      cut -c1-300 $O/bench_h14_fp8_under_rocprof.json
      db=$(find $O/prof8 -name '*.db' | head -1)
      [ -n "$db" ] && python tools/rocpd_stats.py "$db" $O/kernel_stats_h14_fp8.csv | head -16
      rm -rf $O/prof8 ;;
    stats)
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$O/prof" -o r06 -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --h2d-steps 0 --plain-steps 0 --exact-steps 0 --unpad-steps 0 --fp8-line-steps 0 > "$O/bench_B4096_under_rocprof.json" 2> "$O/bench_B4096_under_rocprof.err")
      cut -c1-300 $O/bench_B4096_under_rocprof.json
      db=$(find $O/prof -name '*.db' | head -1)
      [ -n "$db" ] && python tools/rocpd_stats.py "$db" $O/kernel_stats_B4096.csv | head -16
      rm -rf $O/prof ;;
    pmc8|pmc)
      # one bench step per counter set, each in its own rocprofv3 pass (MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE never share a
      # pass; no tracing domains next to --pmc)
      if [ $s = pmc8 ]; then ARGS="$H14 --precision fp8"; D=$O/pmc8; else ARGS="--no-cpu-baseline --h2d-steps 0 --plain-steps 0 --exact-steps 0 --unpad-steps 0 --fp8-line-steps 0"; D=$O/pmc; fi
      mkdir -p $D
      declare -A SETS=( [FETCH_SIZE]="FETCH_SIZE" [WRITE_SIZE]="WRITE_SIZE"
                        [MFMA]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
                        [LDS]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INST_CYCLES_VMEM" )
      for set in ${PMC_SETS:-FETCH_SIZE WRITE_SIZE MFMA LDS}; do
        (cd /tmp && timeout ${PMC_TIMEOUT:-900} rocprofv3 --pmc ${SETS[$set]} -d "$D/$set" -o pmc -- python "$R/bench.py" $ARGS --steps 1 --warmup 0 > "$D/$set.log" 2>&1)
        tail -1 "$D/$set.log" | cut -c1-200
      done
      python tools/pmc_summary.py $D gemm attn ln_ quant > $D.summary.txt 2>&1
      find $D -name '*.db' -delete
      head -40 $D.summary.txt ;;
    driverbench)
      timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_B4096_driver_form.json 2> $O/bench_B4096_driver_form.err
      cut -c1-330 $O/bench_B4096_driver_form.json ;;
    alltests)
      timeout 2000 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s --durations=15 > $O/pytest_gpu.log 2>&1
      echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log ;;
    smoke)
      timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log ;;
    *) echo "unknown stage $s" ;;
  esac
done
