"""Cross-check one gemm_nt variant against another (and an fp64 product) on ragged shapes and every epilogue.
   python tools/gemm_check.py <variant> [<reference variant>]      (exit code 1 on mismatch)"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clipa_amd import ops, lib

var = int(sys.argv[1]); ref = int(sys.argv[2]) if len(sys.argv) > 2 else 2
h = lib.load()
bf16 = torch.bfloat16
shapes = [(256, 256, 96), (512, 768, 128), (300, 264, 104), (1000, 520, 776), (4096, 1024, 1024), (777, 3072, 768),
          (2048, 256, 4096), (197 * 64, 4096, 1024), (65536, 1024, 96), (131, 8, 512)]
bad = 0
for (M, N, K) in shapes:
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda").to(bf16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(bf16)
    bias = torch.randn(N, device="cuda")
    aux = torch.randn(M, N, device="cuda").to(bf16)
    for name, kw in (("none", dict()), ("bias", dict(bias=bias)), ("gelu", dict(bias=bias, epi=ops.EPI_ACT, want_pre=True)),
                     ("res", dict(bias=bias, epi=ops.EPI_ADD, aux=aux)), ("dact", dict(epi=ops.EPI_DACT, aux=aux))):
        outs = []
        for v in (ref, var, var):          # twice: catches ring state carried between launches
            lib.debug_set(v, 0)
            kw2 = dict(kw); b = kw2.pop("bias", None)
            o = ops.gemm_nt(a, w, b, **kw2)
            torch.cuda.synchronize()
            outs.append([t.float() for t in (o if isinstance(o, tuple) else (o,))])
        for which in (1, 2):
            for i, (x, y) in enumerate(zip(outs[0], outs[which])):
                err = (x - y).abs().max().item()
                scale = x.abs().max().item() + 1e-6
                ok = err <= 2e-2 * scale and torch.isfinite(y).all().item()
                if not ok:
                    bad += 1
                    print(f"MISMATCH M={M} N={N} K={K} epi={name} out{i} run{which} err={err:.4g} scale={scale:.4g}")
    if M * N <= 4096 * 1024:
        lib.debug_set(var, 0)
        o = ops.gemm_nt(a, w, bias).double()
        want = a.double() @ w.double().t() + bias.double()
        rel = ((o - want).abs().max() / want.abs().max()).item()
        if rel > 1e-2:
            bad += 1
            print(f"FP64 MISMATCH M={M} N={N} K={K} rel={rel:.4g}")
print("gemm_check variant", var, "vs", ref, ":", "OK" if bad == 0 else f"{bad} mismatches")
sys.exit(1 if bad else 0)
