"""In-process interleaved A/B of the two gemm_tn kernels (ablation bit 512 = first generation).
   python tools/tn_ab.py M R C"""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clipa_amd import ops, lib
M, R, C = (int(x) for x in sys.argv[1:4])
h = lib.load()
h.clipa_debug_set.argtypes = [ctypes.c_int, ctypes.c_int]
bf16 = torch.bfloat16
torch.manual_seed(0)
p = torch.randn(M, R, device="cuda").to(bf16)
q = (torch.randn(M, C, device="cuda") * 0.1).to(bf16)
outs = {}
ABLS = (2048, 512, 1024)
times = {a: [] for a in ABLS}
for rnd in range(7):
    for abl in ABLS:
        h.clipa_debug_set(11, abl)
        o = ops.gemm_tn(p, q, bf16, want_colsum=True); torch.cuda.synchronize()
        outs[abl] = o
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): ops.gemm_tn(p, q, bf16, want_colsum=True)
        e1.record(); torch.cuda.synchronize()
        times[abl].append(e0.elapsed_time(e1) / 4)
h.clipa_debug_set(11, 0)
same = all(torch.allclose(outs[2048][0].float(), outs[a][0].float(), rtol=2e-2, atol=2e-2) and torch.allclose(outs[2048][1], outs[a][1], rtol=1e-5, atol=1e-3) for a in ABLS)
for abl in ABLS:
    t = sorted(times[abl]); med = t[len(t) // 2]
    print(json.dumps({"M": M, "R": R, "C": C, "kernel": {2048: "tn2", 512: "tn1", 1024: "tn3"}[abl], "ms_med": round(med, 4),
                      "tflops_med": round(2 * M * R * C / med / 1e9, 1), "outputs_equal": bool(same)}))
