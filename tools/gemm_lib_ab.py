"""A/B of several builds of libclipa_hip.so in ONE process (interleaved rounds, median): clipa_gemm_nt / clipa_gemm_nt_f8 per
epilogue; the first library is the baseline.
    python tools/gemm_lib_ab.py [f8] clipa_amd/lib/libclipa_hip.so clipa_amd/lib/libclipa_var_X.so ..."""
import ctypes
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clipa_amd import ops  # noqa: E402  (quantize_rows for the fp8 operands)

argv = sys.argv[1:]
f8 = bool(argv) and argv[0] == "f8"
paths = argv[1:] if f8 else argv
libs = [ctypes.CDLL(os.path.abspath(p)) for p in paths]
NL = len(libs)
P, I64, F, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_int
for L in libs:
    L.clipa_gemm_nt.argtypes = [P] * 6 + [I64] * 7 + [F, I, I, I, P]
    L.clipa_gemm_nt_f8.argtypes = [P] * 8 + [I64] * 7 + [F, I, I, I, I, P]
dev = "cuda"
M = 200704
st = torch.cuda.current_stream().cuda_stream
for N, K in ((4096, 1024), (1024, 4096), (1024, 1024), (3072, 1024)):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    aux = torch.randn(M, N, device=dev).to(torch.bfloat16)
    out = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in libs]
    pre = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in libs]
    if f8:
        qa, sa = ops.quantize_rows(a)
        qb, sb = ops.quantize_rows(b)
    for name, epi, use_bias, use_aux, use_pre in (("none", 0, False, False, False), ("bias", 0, True, False, False), ("gelu", 1, True, False, False),
                                                  ("gelu+pre", 1, True, False, True), ("add", 2, True, True, False), ("dact", 3, False, True, False)):
        def once(i):
            L = libs[i]
            bp = bias.data_ptr() if use_bias else None
            ap = aux.data_ptr() if use_aux else None
            pp = pre[i].data_ptr() if use_pre else None
            if f8:
                rc = L.clipa_gemm_nt_f8(qa.data_ptr(), qb.data_ptr(), sa.data_ptr(), sb.data_ptr(), out[i].data_ptr(), pp, bp, ap, M, N, K, K, K, N,
                                        N if use_aux else 0, 1.0, epi, 0, 0, 0, st)
            else:
                rc = L.clipa_gemm_nt(a.data_ptr(), b.data_ptr(), out[i].data_ptr(), pp, bp, ap, M, N, K, K, K, N, N if use_aux else 0, 1.0, epi, 0, 0, st)
            assert rc == 0
        for i in range(NL):
            once(i)
        torch.cuda.synchronize()
        same = all(bool(torch.equal(out[0], out[i])) and (not use_pre or bool(torch.equal(pre[0], pre[i]))) for i in range(1, NL))
        ts = [[] for _ in range(NL)]
        for _ in range(5):
            for i in range(NL):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    once(i)
                e1.record()
                torch.cuda.synchronize()
                ts[i].append(e0.elapsed_time(e1) / 3)
        tf = [2.0 * M * N * K / statistics.median(t) / 1e9 for t in ts]
        print(json.dumps({"kernel": "gemm_nt_f8" if f8 else "gemm_nt", "N": N, "K": K, "epi": name, "TF": [round(t, 1) for t in tf],
                          "vs_first": [round(t / tf[0], 3) for t in tf], "bit_identical": same}), flush=True)
