mkdir -p gpurun_out/r06
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -x -k "gemm or headline or production or light8 or per_tensor" 2>&1 | tail -3
python bench.py --gpus 1 --steps 20 --warmup 5 --shapes > gpurun_out/r06/driver_form_b.json 2> gpurun_out/r06/driver_form_b.err; cut -c1-330 gpurun_out/r06/driver_form_b.json; grep "SHAPE gemm_nt" gpurun_out/r06/driver_form_b.err | head -12
