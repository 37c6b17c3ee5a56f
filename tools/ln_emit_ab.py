"""LayerNorm backward with and without the emitted forward output (clipa_layernorm_bwd_y) between builds of the library, one
process, interleaved rounds, medians: what does the fusion buy against ln_bwd + a separate ln_fwd?
    python tools/ln_emit_ab.py libA.so [libB.so ...]"""
import ctypes
import json
import os
import statistics
import sys

import torch

libs = [ctypes.CDLL(os.path.abspath(p)) for p in sys.argv[1:]]
P, I64, F, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_int
for L in libs:
    L.clipa_layernorm_fwd.argtypes = [P, P, P, P, I64, I64, F, I, I, P]
    L.clipa_layernorm_bwd_workspace.argtypes = [I64, I64]
    L.clipa_layernorm_bwd_workspace.restype = I64
    L.clipa_layernorm_bwd.argtypes = [P, P, P, P, P, P, P, I64, I64, F, I, I, P, I64, P]
    L.clipa_layernorm_bwd_y.argtypes = [P, P, P, P, P, P, P, P, P, I64, I64, F, I, I, P, I64, P]
dev, bf16 = "cuda", torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
for rows, D in ((806912, 1024), (315392, 768), (806912, 768), (526336, 1280), (630784, 1024), (802816, 384)):
    x, dy, dres = (torch.randn(rows, D, device=dev).to(bf16) for _ in range(3))
    y, dx = torch.empty_like(x), torch.empty_like(x)
    g, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    dg, db = torch.empty(D, device=dev), torch.empty(D, device=dev)
    wsb = max(int(L.clipa_layernorm_bwd_workspace(rows, D)) for L in libs)
    ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    p = lambda t: ctypes.c_void_p(t.data_ptr())

    def fns(L):
        return {"ln_fwd": lambda: L.clipa_layernorm_fwd(p(x), p(g), p(b), p(y), rows, D, 1e-5, 0, 0, st),
                "ln_bwd": lambda: L.clipa_layernorm_bwd(p(x), p(g), p(dy), p(dres), p(dx), p(dg), p(db), rows, D, 1e-5, 0, 0, p(ws), wsb, st),
                "ln_bwd_y": lambda: L.clipa_layernorm_bwd_y(p(x), p(g), p(b), p(dy), p(dres), p(dx), p(y), p(dg), p(db), rows, D, 1e-5, 0, 0, p(ws), wsb, st)}

    allf = [fns(L) for L in libs]
    ts = [{k: [] for k in f} for f in allf]
    for f in allf:
        for fn in f.values():
            assert fn() == 0
    torch.cuda.synchronize()
    for _ in range(7):
        for i, f in enumerate(allf):
            for k, fn in f.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ts[i][k].append(e0.elapsed_time(e1) / 4)
    row = {"rows": rows, "D": D}
    for i, t in enumerate(ts):
        m = {k: round(statistics.median(v), 4) for k, v in t.items()}
        m["fused_saves_ms"] = round(m["ln_bwd"] + m["ln_fwd"] - m["ln_bwd_y"], 4)
        row[os.path.basename(sys.argv[1 + i])] = m
    print(json.dumps(row), flush=True)
