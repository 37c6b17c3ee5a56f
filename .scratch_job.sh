mkdir -p gpurun_out/r06
(python tools/stream8_bench.py --only ln_bwd; python tools/stream8_bench.py --only ln_bwd --lib tools/probes/var/lnbnopf.so; python tools/stream8_bench.py --rows 806912 --D 1024 --only ln_; python tools/stream8_bench.py --rows 806912 --D 1024 --only ln_bwd --lib tools/probes/var/lnbnopf.so; python tools/stream8_bench.py --rows 315392 --D 768 --only ln_; python tools/stream8_bench.py --rows 315392 --D 768 --only ln_bwd --lib tools/probes/var/lnbnopf.so) > gpurun_out/r06/stream8_e.jsonl 2>/dev/null
cat gpurun_out/r06/stream8_e.jsonl
python -m pytest tests/test_fp8_gpu.py tests/test_kernels_gpu.py -q -s -k "full_dims or layernorm" 2>&1 | grep -E "^\[fp8|passed|failed" | tail -12
FLAGSWEEP_FROM=8 ./tools/probes/gemm_flag_sweep 0 $((4<<20)) $((8<<20)) $((12<<20)) $((16<<20)) $((24<<20)) $((32<<20)) > gpurun_out/r06/vitb_group_sweep.jsonl 2>&1; cat gpurun_out/r06/vitb_group_sweep.jsonl
