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
times = {0: [], 512: []}
for rnd in range(7):
    for abl in (0, 512):
        h.clipa_debug_set(11, abl)
        o = ops.gemm_tn(p, q, bf16, want_colsum=True); torch.cuda.synchronize()
        outs[abl] = o
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): ops.gemm_tn(p, q, bf16, want_colsum=True)
        e1.record(); torch.cuda.synchronize()
        times[abl].append(e0.elapsed_time(e1) / 4)
h.clipa_debug_set(11, 0)
same = torch.allclose(outs[0][0].float(), outs[512][0].float(), rtol=2e-2, atol=2e-2) and torch.allclose(outs[0][1], outs[512][1], rtol=1e-5, atol=1e-3)
for abl in (0, 512):
    t = sorted(times[abl]); med = t[len(t) // 2]
    print(json.dumps({"M": M, "R": R, "C": C, "kernel": "tn2" if abl == 0 else "tn1", "ms_med": round(med, 4),
                      "tflops_med": round(2 * M * R * C / med / 1e9, 1), "outputs_equal": bool(same)}))
