mkdir -p gpurun_out/r06
python -m pytest tests/test_fp8_gpu.py -q -s 2>&1 | grep -E "^\.?\[fp8|passed|failed|Error" > gpurun_out/r06/pytest_fp8_parity.log; tail -14 gpurun_out/r06/pytest_fp8_parity.log | cut -c1-250
H14="--model ViT-H-14 --batch 2048 --no-cpu-baseline --h2d-steps 0 --plain-steps 0 --exact-steps 0 --unpad-steps 0 --precision fp8 --steps 5 --warmup 2 --shapes"
python bench.py $H14 > gpurun_out/r06/bench_h14_fp8_final_default.json 2> gpurun_out/r06/bench_h14_fp8_final_default.err; cut -c1-200 gpurun_out/r06/bench_h14_fp8_final_default.json
python bench.py $H14 --fp8-predicted-scales > gpurun_out/r06/bench_h14_fp8_predicted_scales.json 2> gpurun_out/r06/bench_h14_fp8_predicted_scales.err; cut -c1-200 gpurun_out/r06/bench_h14_fp8_predicted_scales.json
