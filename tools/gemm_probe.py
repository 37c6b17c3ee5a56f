"""Run ONE kernel shape repeatedly (for rocprofv3 --pmc / --kernel-trace passes).
    python tools/gemm_probe.py nt M N K [epi] [iters]     epi in {none,bias,gelu,res}
    python tools/gemm_probe.py f8 M N K [epi] [iters]     the fp8 GEMM on pre-quantised operands, same epilogues
    python tools/gemm_probe.py tn M R C
    python tools/gemm_probe.py attn_fwd|attn_bwd B H L causal
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clipa_amd import ops  # noqa: E402

bf16 = torch.bfloat16
kind = sys.argv[1]
a1, a2, a3 = (int(x) for x in sys.argv[2:5])
extra = sys.argv[5] if len(sys.argv) > 5 else "bias"
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
torch.manual_seed(0)
dev = "cuda"
if kind == "nt":
    a = torch.randn(a1, a3, device=dev).to(bf16)
    w = (torch.randn(a2, a3, device=dev) * 0.05).to(bf16)
    bias = torch.randn(a2, device=dev)
    res = torch.randn(a1, a2, device=dev).to(bf16) if extra == "res" else None
    for _ in range(iters):
        if extra == "gelu":
            ops.gemm_nt(a, w, bias, epi=ops.EPI_ACT, want_pre=True)
        elif extra == "res":
            ops.gemm_nt(a, w, bias, epi=ops.EPI_ADD, aux=res)
        else:
            ops.gemm_nt(a, w, bias if extra == "bias" else None)
elif kind == "f8":
    a = torch.randn(a1, a3, device=dev).to(bf16)
    w = (torch.randn(a2, a3, device=dev) * 0.05).to(bf16)
    bias = torch.randn(a2, device=dev)
    res = torch.randn(a1, a2, device=dev).to(bf16) if extra == "res" else None
    qa, sa = ops.quantize_rows(a)
    qw, sw = ops.quantize_rows(w)
    for _ in range(iters):
        if extra == "gelu":
            ops.gemm_nt_f8(qa, sa, qw, sw, bias, epi=ops.EPI_ACT, want_pre=True)
        elif extra == "res":
            ops.gemm_nt_f8(qa, sa, qw, sw, bias, epi=ops.EPI_ADD, aux=res)
        else:
            ops.gemm_nt_f8(qa, sa, qw, sw, bias if extra == "bias" else None)
elif kind == "tn":
    p = torch.randn(a1, a2, device=dev).to(bf16)
    q = torch.randn(a1, a3, device=dev).to(bf16)
    for _ in range(iters):
        ops.gemm_tn(p, q)
else:
    B, H, L = a1, a2, a3
    causal = extra == "1"
    qkv = torch.randn(B * L, 3 * 64 * H, device=dev).to(bf16)
    o, st = ops.attention_fwd(qkv, B, L, H, causal, want_stats=True)
    do = torch.randn_like(o)
    for _ in range(iters):
        if kind == "attn_fwd":
            ops.attention_fwd(qkv, B, L, H, causal)
        else:
            ops.attention_bwd(qkv, o, do, st, B, L, H, causal)
torch.cuda.synchronize()
print("done")
