"""In-process interleaved A/B of the gemm_tn kernels / work orders (clipa_internal_debug_set flags: 1024 force the 16x16x32
kernel, 2048 force the ping-pong kernel, 4096 slice-per-XCD work order).   python tools/tn_ab.py [M]"""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clipa_amd import ops, lib
h = lib.load()
bf16 = torch.bfloat16
ABLS = (0, 4096, 1024, 1024 | 4096, 2048, 2048 | 4096)
SHAPES = [(806912, 4096, 1024), (806912, 1024, 4096), (806912, 3072, 1024), (806912, 1024, 1024),
          (315392, 3072, 768), (315392, 768, 3072), (315392, 2304, 768), (315392, 768, 768)]
for (M, R, C) in SHAPES:
    torch.manual_seed(0)
    p = torch.randn(M, R, device="cuda").to(bf16)
    q = (torch.randn(M, C, device="cuda") * 0.1).to(bf16)
    outs = {}
    times = {a: [] for a in ABLS}
    for rnd in range(5):
        for abl in ABLS:
            lib.debug_set(0, abl)
            o = ops.gemm_tn(p, q, bf16, want_colsum=True); torch.cuda.synchronize()
            outs[abl] = o
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): ops.gemm_tn(p, q, bf16, want_colsum=True)
            e1.record(); torch.cuda.synchronize()
            times[abl].append(e0.elapsed_time(e1) / 3)
    lib.debug_set(0, 0)
    same = all(torch.allclose(outs[0][0].float(), outs[a][0].float(), rtol=2e-2, atol=2e-2) and torch.allclose(outs[0][1], outs[a][1], rtol=1e-4, atol=1e-2) for a in ABLS)
    for abl in ABLS:
        t = sorted(times[abl]); med = t[len(t) // 2]
        print(json.dumps({"M": M, "R": R, "C": C, "abl": abl, "ms_med": round(med, 4),
                          "tflops_med": round(2 * M * R * C / med / 1e9, 1), "outputs_equal": bool(same)}), flush=True)
    del p, q, outs
