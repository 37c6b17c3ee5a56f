"""In-process interleaved A/B of gemm_nt variants (kernel experiments).  Variants are (CLIPA_GEMM_NT, ABL) pairs
set through clipa_amd.lib's debug hook.   python tools/gemm_ab.py M N K epi nt:abl nt:abl ..."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clipa_amd import ops, lib
M, N, K = (int(x) for x in sys.argv[1:4])
epi = sys.argv[4]
variants = [tuple(int(v) for v in a.split(":")) for a in sys.argv[5:]]
h = lib.load()
bf16 = torch.bfloat16
torch.manual_seed(0)
PAD = int(os.environ.get("PAD", "0"))      # extra elements in the leading dimension of a and w (channel-camping probe)
a = torch.randn(M, K + PAD, device="cuda").to(bf16)[:, :K]
w = (torch.randn(N, K + PAD, device="cuda") * 0.05).to(bf16)[:, :K]
bias = torch.randn(N, device="cuda")
res = torch.randn(M, N, device="cuda").to(bf16)
def run():
    if epi == "gelu": ops.gemm_nt(a, w, bias, epi=ops.EPI_ACT, want_pre=True)
    elif epi == "res": ops.gemm_nt(a, w, bias, epi=ops.EPI_ADD, aux=res)
    else: ops.gemm_nt(a, w, bias)
times = {v: [] for v in variants}
for rnd in range(7):
    for v in variants:
        lib.debug_set(v[0], v[1])
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): run()
        e1.record(); torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1) / 4)
for v in variants:
    t = sorted(times[v]); med = t[len(t) // 2]
    print(json.dumps({"M": M, "N": N, "K": K, "epi": epi, "nt": v[0], "abl": v[1], "ms_med": round(med, 4), "ms_min": round(t[0], 4),
                      "tflops_med": round(2 * M * N * K / med / 1e9, 1), "pad": PAD}))
