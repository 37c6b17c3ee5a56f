"""A/B of clipa_attention_fwd / clipa_attention_bwd between builds of libclipa_hip.so loaded side by side (first = baseline):
interleaved rounds, median, outputs compared bit for bit.
    python tools/attn_lib_ab.py clipa_amd/lib/libclipa_hip_head.so clipa_amd/lib/libclipa_hip.so"""
import ctypes
import json
import os
import statistics
import sys

import torch

libs = [ctypes.CDLL(os.path.abspath(p)) for p in sys.argv[1:]]
P, I64, F, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_int
for L_ in libs:
    L_.clipa_attention_fwd.argtypes = [P] * 5 + [I64] * 6 + [F, I, P]
    L_.clipa_attention_bwd.argtypes = [P] * 9 + [I64] * 7 + [F, I, P]
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
NL = len(libs)
for B, H, L, dh, causal in ((4096, 16, 197, 64, 0), (4096, 12, 77, 64, 1), (2048, 16, 257, 80, 0), (4096, 16, 50, 64, 0), (4096, 12, 32, 64, 1)):
    D = H * dh
    torch.manual_seed(0)
    qkv = torch.randn(B * L, 3 * D, device=dev).to(torch.bfloat16)
    do = torch.randn(B * L, D, device=dev).to(torch.bfloat16)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    outs = [torch.empty(B * L, D, device=dev, dtype=torch.bfloat16) for _ in libs]
    stats = [torch.empty(B * H * L, 2, device=dev) for _ in libs]
    dqkv = [torch.empty(B * L, 3 * D, device=dev, dtype=torch.bfloat16) for _ in libs]
    scale = dh ** -0.5

    def fwd(i):
        assert libs[i].clipa_attention_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), outs[i].data_ptr(), stats[i].data_ptr(), B, H, L, dh, 3 * D, D,
                                           scale, causal, st) == 0

    def bwd(i):
        g = dqkv[i]
        assert libs[i].clipa_attention_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), outs[i].data_ptr(), do.data_ptr(), stats[i].data_ptr(),
                                           g[:, :D].data_ptr(), g[:, D:2 * D].data_ptr(), g[:, 2 * D:].data_ptr(), B, H, L, dh, 3 * D, D, 3 * D,
                                           scale, causal, st) == 0
    for i in range(NL):
        dqkv[i].zero_(); outs[i].zero_(); stats[i].zero_()
        fwd(i); bwd(i)
    torch.cuda.synchronize()
    same = all(torch.equal(outs[0], outs[i]) and torch.equal(stats[0], stats[i]) and torch.equal(dqkv[0], dqkv[i]) for i in range(1, NL))
    row = {"B": B, "H": H, "L": L, "dh": dh, "causal": causal, "bit_identical": bool(same)}
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        ts = [[] for _ in libs]
        for _ in range(5):
            for i in range(NL):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    fn(i)
                e1.record()
                torch.cuda.synchronize()
                ts[i].append(e0.elapsed_time(e1) / 3)
        ms = [statistics.median(t) for t in ts]
        row[name + "_ms"] = [round(m, 4) for m in ms]
        row[name + "_speedup"] = [round(ms[0] / m, 3) for m in ms]
    print(json.dumps(row), flush=True)
