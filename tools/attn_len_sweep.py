"""clipa_attention_fwd / _bwd against the sequence length around a tile boundary (what does the 257th token of ViT-L/14 / ViT-H/14
cost?): one JSON line per (B, H, L, dh), ms per launch and ms per 1000 tokens.
    python tools/attn_len_sweep.py [--lib path.so]"""
import argparse
import ctypes
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "clipa_amd", "lib", "libclipa_hip.so"))
ap.add_argument("--cases", default="2048,16,80:224,256,257,288;4096,16,64:192,197,224,256,257")
args = ap.parse_args()
lib = ctypes.CDLL(os.path.abspath(args.lib))
P, I64, F, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_int
lib.clipa_attention_fwd.argtypes = [P] * 5 + [I64] * 6 + [F, I, P]
lib.clipa_attention_bwd.argtypes = [P] * 9 + [I64] * 7 + [F, I, P]
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
for case in args.cases.split(";"):
    head, lens = case.split(":")
    B, H, dh = (int(x) for x in head.split(","))
    for L in (int(x) for x in lens.split(",")):
        D = H * dh
        torch.manual_seed(0)
        qkv = torch.randn(B * L, 3 * D, device=dev).to(torch.bfloat16)
        do = torch.randn(B * L, D, device=dev).to(torch.bfloat16)
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        out = torch.empty(B * L, D, device=dev, dtype=torch.bfloat16)
        stats = torch.empty(B * H * L, 2, device=dev)
        g = torch.empty(B * L, 3 * D, device=dev, dtype=torch.bfloat16)
        scale = dh ** -0.5

        def fwd():
            assert lib.clipa_attention_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), stats.data_ptr(), B, H, L, dh, 3 * D, D, scale, 0, st) == 0

        def bwd():
            assert lib.clipa_attention_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), do.data_ptr(), stats.data_ptr(), g[:, :D].data_ptr(),
                                           g[:, D:2 * D].data_ptr(), g[:, 2 * D:].data_ptr(), B, H, L, dh, 3 * D, D, 3 * D, scale, 0, st) == 0
        fwd(); bwd()
        torch.cuda.synchronize()
        row = {"B": B, "H": H, "L": L, "dh": dh, "lib": os.path.basename(args.lib)}
        for name, fn in (("fwd", fwd), ("bwd", bwd)):
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 3)
            ms = statistics.median(ts)
            row[name + "_ms"] = round(ms, 4)
            row[name + "_us_per_seq"] = round(ms * 1e3 / B, 3)
        print(json.dumps(row), flush=True)
        del qkv, do, out, stats, g
