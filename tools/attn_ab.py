"""Attention backward: software-pipelined vs plain schedule (clipa_debug_set flag 16384 = plain), interleaved in one process
on the production shapes; also checks the two give bit-identical gradients.
    python tools/attn_ab.py > gpurun_out/attn_ab.jsonl"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clipa_amd import ops, lib

h = lib.load()
bf16 = torch.bfloat16


def timed(fn, iters=5):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


for (B, H, L, dh, causal) in [(4096, 16, 197, 64, False), (4096, 12, 77, 64, True), (2048, 16, 257, 80, False), (4096, 16, 26, 64, False),
                              (2048, 16, 77, 64, True)]:
    torch.manual_seed(0)
    D = H * dh
    qkv = torch.randn(B * L, 3 * D, device="cuda").to(bf16)
    o, st = ops.attention_fwd(qkv, B, L, H, causal, want_stats=True)
    do = torch.randn_like(o)
    res, t = {}, {}
    for rnd in range(3):
        for name, flag in (("plain", 16384), ("pipe", 0)):
            h.clipa_debug_set(0, flag)
            f = lambda: ops.attention_bwd(qkv, o, do, st, B, L, H, causal)
            res[name] = f()
            t.setdefault(name, []).append(timed(f))
    h.clipa_debug_set(0, 0)
    fl = 10.0 * B * H * L * L * dh * (0.5 if causal else 1.0)
    tp, tq = sorted(t["plain"])[1], sorted(t["pipe"])[1]
    print(json.dumps({"B": B, "H": H, "L": L, "dh": dh, "causal": causal, "plain_ms": round(tp, 3), "pipe_ms": round(tq, 3),
                      "plain_tf": round(fl / tp / 1e9, 1), "pipe_tf": round(fl / tq / 1e9, 1), "speedup": round(tp / tq, 3),
                      "bit_identical": bool(torch.equal(res["plain"], res["pipe"]))}), flush=True)
