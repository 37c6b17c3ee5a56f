"""Geometry of gemm_nt mismatches between variants (debug aid):  python tools/gemm_diag.py"""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clipa_amd import ops, lib
import tools.gemm_round2 as R

h = lib.load()
bf16 = torch.bfloat16
for (M, N, K) in [(65536, 4096, 1024), (10317, 1024, 1024), (300, 264, 136)]:
    torch.manual_seed(M + N)
    a = torch.randn(M, K, device="cuda").to(bf16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(bf16)
    bias = torch.randn(N, device="cuda")
    aux = torch.randn(M, N, device="cuda").to(bf16)
    for epi in ("bias", "dact", "res", "gelu+pre"):
        h.clipa_debug_set(11, 0)
        ref = R.call(epi, a, w, bias, aux)
        ref = [t.clone() for t in (ref if isinstance(ref, tuple) else (ref,))]
        for v in (12, 13):
            for rep in range(2):
                h.clipa_debug_set(v, 0)
                got = R.call(epi, a, w, bias, aux)
                got = got if isinstance(got, tuple) else (got,)
                for oi, (x, y) in enumerate(zip(ref, got)):
                    bad = (x.view(torch.int16) != y.view(torch.int16))
                    nb = int(bad.sum())
                    if nb == 0:
                        print(json.dumps(dict(M=M, N=N, K=K, epi=epi, v=v, rep=rep, out=oi, bad=0)))
                        continue
                    TN = 256 if v == 12 else 128
                    r, c = bad.nonzero(as_tuple=True)
                    tiles = torch.unique((r // 256) * ((N + TN - 1) // TN) + c // TN)
                    rows_in = torch.bincount((r % 256), minlength=256)
                    cols_in = torch.bincount((c % TN), minlength=TN)
                    yb = y[bad].float()
                    xb = x[bad].float()
                    print(json.dumps(dict(M=M, N=N, K=K, epi=epi, v=v, rep=rep, out=oi, bad=nb, ntiles=int(tiles.numel()),
                                          tiles=tiles[:40].tolist(), tile_rows=sorted(set((tiles // ((N + TN - 1) // TN)).tolist()))[:20],
                                          rows_nonzero=int((rows_in > 0).sum()), cols_nonzero=int((cols_in > 0).sum()),
                                          rows_hist=rows_in.tolist()[:64:4], cols_hist=cols_in.tolist()[:64:4],
                                          nan=int(torch.isnan(yb).sum()), zero=int((yb == 0).sum()),
                                          sample_got=yb[:6].tolist(), sample_ref=xb[:6].tolist(),
                                          first=[int(r[0]), int(c[0])], last=[int(r[-1]), int(c[-1])])))
h.clipa_debug_set(0, 0)
