"""Time one gemm_nt shape (HIP events). Env CLIPA_GEMM_NT / CLIPA_GEMM_ABL select kernel variants.
   python tools/gemm_time.py M N K [epi]"""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clipa_amd import ops
M, N, K = (int(x) for x in sys.argv[1:4])
epi = sys.argv[4] if len(sys.argv) > 4 else "bias"
bf16 = torch.bfloat16
torch.manual_seed(0)
a = torch.randn(M, K, device="cuda").to(bf16)
w = (torch.randn(N, K, device="cuda") * 0.05).to(bf16)
bias = torch.randn(N, device="cuda")
res = torch.randn(M, N, device="cuda").to(bf16)
def run():
    if epi == "gelu": ops.gemm_nt(a, w, bias, epi=ops.EPI_ACT, want_pre=True)
    elif epi == "res": ops.gemm_nt(a, w, bias, epi=ops.EPI_ADD, aux=res)
    elif epi == "none": ops.gemm_nt(a, w)
    else: ops.gemm_nt(a, w, bias)
for _ in range(3): run()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 5)
ms = sorted(ts)[len(ts) // 2]
print(json.dumps({"M": M, "N": N, "K": K, "epi": epi, "nt": os.environ.get("CLIPA_GEMM_NT", ""), "abl": os.environ.get("CLIPA_GEMM_ABL", "0"),
                  "ms": round(ms, 4), "tflops": round(2 * M * N * K / ms / 1e9, 1)}))
