mkdir -p gpurun_out/r06
(python tools/stream8_bench.py --only ln_bwd; for n in 1536 2048 3072; do python tools/stream8_bench.py --only ln_bwd --lib tools/probes/var/lnb$n.so; done) > gpurun_out/r06/stream8_f.jsonl 2>/dev/null; cat gpurun_out/r06/stream8_f.jsonl
python tools/tn8_bench.py --shapes all --schedules 1,2 --orders 4096,8192 > gpurun_out/r06/tn8_bench_orders.jsonl 2>/dev/null; cat gpurun_out/r06/tn8_bench_orders.jsonl | cut -c1-400
