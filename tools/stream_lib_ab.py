"""A/B of the HBM-streaming kernels (LayerNorm forward / backward, MLP re-materialisation from bf16 and from e4m3, row
quantisation, LayerNorm + quantise) between builds of the library, in ONE process on torch-allocated buffers at the bench's
sizes, interleaved rounds, median.  Two settings per kernel: "hot" = the same launch back to back; "after_writer" = each launch
preceded by a torch kernel that rewrites the input (what a training step looks like: the input was just produced, its tail is
dirty in L2 / MALL) - there the time is writer + kernel for both libraries, so only the DIFFERENCE is meaningful.
    python tools/stream_lib_ab.py old.so new.so [rows=806912] [D=1024]
The two libraries hold only the streaming kernels (seconds to build):
    git archive <old-commit> clipa_amd/csrc include | tar -x -C /tmp/old
    (cd /tmp/old && hipcc -O3 --offload-arch=gfx950 -shared -fPIC -I clipa_amd/csrc -I include clipa_amd/csrc/{layernorm,misc,quant,runtime}.hip \
         -o $REPO/tools/probes/lnvar/libstream_old.so)
    hipcc ... (same, from the working tree) -o tools/probes/lnvar/libstream_new.so"""
import ctypes
import json
import os
import statistics
import sys

import torch

paths = [p for p in sys.argv[1:] if p.endswith(".so")]
nums = [int(a) for a in sys.argv[1:] if not a.endswith(".so")]
rows = nums[0] if nums else 806912
D = nums[1] if len(nums) > 1 else 1024
libs = [ctypes.CDLL(os.path.abspath(p)) for p in paths]
P, I64, F, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_int
for L in libs:
    L.clipa_layernorm_fwd.argtypes = [P, P, P, P, I64, I64, F, I, I, P]
    L.clipa_layernorm_bwd_workspace.argtypes = [I64, I64]
    L.clipa_layernorm_bwd_workspace.restype = I64
    L.clipa_layernorm_bwd.argtypes = [P, P, P, P, P, P, P, I64, I64, F, I, I, P, I64, P]
    L.clipa_activation_fwd.argtypes = [P, P, I64, I, P]
    L.clipa_activation_fwd_e4m3.argtypes = [P, P, I64, I, P]
    L.clipa_quantize_rows.argtypes = [P, P, P, I64, I64, I64, I64, I, P]
    L.clipa_layernorm_fwd_q8.argtypes = [P, P, P, P, P, P, I64, I64, F, P]
dev = "cuda"
bf16 = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
x = torch.randn(rows, D, device=dev).to(bf16)
src = torch.randn(rows, D, device=dev).to(bf16)
dy = torch.randn(rows, D, device=dev).to(bf16)
dres = torch.randn(rows, D, device=dev).to(bf16)
y = torch.empty_like(x)
g = torch.ones(D, device=dev)
b = torch.zeros(D, device=dev)
dg, db = torch.empty(D, device=dev), torch.empty(D, device=dev)
wsb = max(int(L.clipa_layernorm_bwd_workspace(rows, D)) for L in libs)
ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
H = 4 * D
h = torch.randn(rows, H, device=dev).to(bf16)
hsrc = torch.randn(rows, H, device=dev).to(bf16)
hout = torch.empty_like(h)
h8 = torch.randint(0, 120, (rows, H), device=dev, dtype=torch.uint8)
h8src = h8.clone()
q = torch.empty(rows, H, device=dev, dtype=torch.uint8)
dq = torch.empty(rows, device=dev)
GB = 1e-9


def cases(L):
    return {
        "ln_fwd": (lambda: L.clipa_layernorm_fwd(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), rows, D, 1e-5, 0, 0, st),
                   lambda: x.copy_(src), 4 * rows * D),
        "ln_bwd": (lambda: L.clipa_layernorm_bwd(x.data_ptr(), g.data_ptr(), dy.data_ptr(), dres.data_ptr(), y.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                                 rows, D, 1e-5, 0, 0, ws.data_ptr(), wsb, st),
                   lambda: dy.copy_(src), 8 * rows * D),
        "activation_fwd": (lambda: L.clipa_activation_fwd(h.data_ptr(), hout.data_ptr(), rows * H, 0, st), lambda: h.copy_(hsrc), 4 * rows * H),
        "activation_fwd_e4m3": (lambda: L.clipa_activation_fwd_e4m3(h8.data_ptr(), hout.data_ptr(), rows * H, 0, st), lambda: h8.copy_(h8src), 3 * rows * H),
        "quantize_rows": (lambda: L.clipa_quantize_rows(h.data_ptr(), q.data_ptr(), dq.data_ptr(), rows, H, H, H, 0, st), lambda: h.copy_(hsrc), 3 * rows * H),
        "ln_fwd_q8": (lambda: L.clipa_layernorm_fwd_q8(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), q.data_ptr(), dq.data_ptr(), rows, D, 1e-5, st),
                      lambda: x.copy_(src), 5 * rows * D),
    }


def timed(fn, pre, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        if pre is not None:
            pre()
        rc = fn()
        assert rc == 0, rc
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


table = [cases(L) for L in libs]
for name in table[0]:
    for setting in ("hot", "after_writer"):
        ts = [[] for _ in libs]
        for rnd in range(7):
            for i in range(len(libs)):
                fn, pre, nbytes = table[i][name]
                t = timed(fn, pre if setting == "after_writer" else None, 4)
                if rnd:
                    ts[i].append(t)
        med = [statistics.median(t) for t in ts]
        rec = {"kernel": name, "rows": rows, "D": D, "setting": setting, "ms": [round(m, 4) for m in med], "libs": [os.path.basename(p) for p in paths]}
        if setting == "hot":
            rec["gbps"] = [round(table[0][name][2] / m * 1e-6) for m in med]
        else:
            rec["new_minus_old_ms"] = round(med[-1] - med[0], 4)
        print(json.dumps(rec), flush=True)
