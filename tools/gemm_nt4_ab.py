"""Experiment harness: production clipa_gemm_nt vs an experimental kernel of tools/experiments/ that exports clipa_gemm_<name>
with the same signature (nt4: four waves, gemm_nt4_four_waves.hip; nt6: A3/B2 operand ring, gemm_nt6_a3b2_ring.hip), linked into a
VARIANT library by `tools/build_variant.sh <name> tools/experiments/<file>.hip` and never into libclipa_hip.so: outputs compared bit for
bit, interleaved timing.   python tools/gemm_nt4_ab.py [name=nt4] [M]"""
import ctypes
import json
import os
import statistics
import sys

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prod = ctypes.CDLL(os.path.join(root, "clipa_amd", "lib", "libclipa_hip.so"))
name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].isdigit() else "nt4"
var = ctypes.CDLL(os.path.join(root, "clipa_amd", "lib", f"libclipa_var_{name}.so"))
var_fn = getattr(var, "clipa_gemm_" + name)
P, I64, F, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_int
sig = [P] * 6 + [I64] * 7 + [F, I, I, I, P]
prod.clipa_gemm_nt.argtypes = sig
var_fn.argtypes = sig
var.clipa_last_error.restype = ctypes.c_char_p
fns = [prod.clipa_gemm_nt, var_fn]
dev = "cuda"
M = int(sys.argv[-1]) if len(sys.argv) > 1 and sys.argv[-1].isdigit() else 200704
st = torch.cuda.current_stream().cuda_stream
shapes = [(M, 4096, 1024), (M, 1024, 4096), (M, 1024, 1024), (M, 3072, 1024), (1000, 520, 72), (300, 264, 1032)]
for Mi, N, K in shapes:
    torch.manual_seed(1)
    a = torch.randn(Mi, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    aux = torch.randn(Mi, N, device=dev).to(torch.bfloat16)
    out = [torch.zeros(Mi, N, device=dev, dtype=torch.bfloat16) for _ in fns]
    pre = [torch.zeros(Mi, N, device=dev, dtype=torch.bfloat16) for _ in fns]
    for ename, epi, use_bias, use_aux, use_pre in (("none", 0, False, False, False), ("bias", 0, True, False, False), ("gelu", 1, True, False, False),
                                                  ("gelu+pre", 1, True, False, True), ("add", 2, True, True, False), ("dact", 3, False, True, False)):
        def once(i):
            rc = fns[i](a.data_ptr(), b.data_ptr(), out[i].data_ptr(), pre[i].data_ptr() if use_pre else None,
                        bias.data_ptr() if use_bias else None, aux.data_ptr() if use_aux else None, Mi, N, K, K, K, N, N if use_aux else 0,
                        1.0, epi, 0, 0, st)
            assert rc == 0, var.clipa_last_error()
        for i in range(2):
            out[i].zero_(); pre[i].zero_()
            once(i)
        torch.cuda.synchronize()
        same = bool(torch.equal(out[0], out[1])) and (not use_pre or bool(torch.equal(pre[0], pre[1])))
        row = {"kernel": name, "M": Mi, "N": N, "K": K, "epi": ename, "bit_identical": same}
        if Mi >= 100000:
            ts = [[], []]
            for _ in range(5):
                for i in range(2):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(3):
                        once(i)
                    e1.record()
                    torch.cuda.synchronize()
                    ts[i].append(e0.elapsed_time(e1) / 3)
            tf = [2.0 * Mi * N * K / statistics.median(t) / 1e9 for t in ts]
            row.update({"prod_TF": round(tf[0], 1), "var_TF": round(tf[1], 1), "var_over_prod": round(tf[1] / tf[0], 3)})
        print(json.dumps(row), flush=True)
