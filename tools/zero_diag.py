"""Diagnostic: plain fused AdamW vs ShardedAdamW (flat buffers, optional forced RCCL collectives on a 1-rank group) on the
HIP engine - where do the two first differ?  python tools/zero_diag.py [rccl]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clipa_amd
from clipa_amd.optim import AdamW
from clipa_amd.zero import ShardedAdamW
from oracle import clip_oracle as O

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
use_rccl = len(sys.argv) > 1 and sys.argv[1] == "rccl"
if use_rccl:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29799")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)


def build():
    torch.manual_seed(0)
    m = clipa_amd.create_model("ViT-S-16", precision="bf16", device=dev, force_image_size=112, output_dict=True)
    m.positional_embedding = torch.nn.Parameter(m.positional_embedding[:32].clone())
    m.set_grad_checkpointing(True)
    m.transformer.keep_blocks = 2
    return m


img, txt = O.synthetic_batch(8, 112, 32, 49408, seed=7)
img, txt = img.to(dev), txt.to(dev)


def run(mode, steps=2):
    m = build()
    params = [p for p in m.parameters() if p.requires_grad]
    kw = dict(lr=1e-3, betas=(0.9, 0.95), eps=1e-6, weight_decay=0.1)
    if mode == "plain":
        opt = AdamW(params, **kw)
    else:
        opt = ShardedAdamW(params, bucket_bytes=8 << 20, exchange=mode if mode != "flat" else "reduce_scatter",
                           force_collectives=use_rccl and mode != "flat", **kw)
    trace = []
    for s in range(steps):
        opt.zero_grad()
        out = m(img, txt)
        loss = clipa_amd.ClipLoss(local_loss=True, gather_with_grad=True, rank=0, world_size=1)(**out, output_dict=True)["contrastive_loss"]
        loss.backward()
        torch.cuda.synchronize()
        grads = {n: p.grad.detach().float().clone() for n, p in m.named_parameters() if p.grad is not None}
        opt.step()
        torch.cuda.synchronize()
        trace.append((float(loss), grads, {n: p.detach().float().clone() for n, p in m.named_parameters()}))
    return trace


def cmp(a, b, what):
    for s, ((la, ga, pa), (lb, gb, pb)) in enumerate(zip(a, b)):
        ng = [(n, float((ga[n] - gb[n]).abs().max())) for n in ga if not torch.equal(ga[n], gb[n])]
        npar = [(n, float((pa[n] - pb[n]).abs().max())) for n in pa if not torch.equal(pa[n], pb[n])]
        print(f"{what} step {s}: loss {la:.7f} vs {lb:.7f}; grads differing {len(ng)}/{len(ga)} {ng[:3]}; params differing {len(npar)}/{len(pa)} {npar[:3]}", flush=True)


p1, p2 = run("plain"), run("plain")
cmp(p1, p2, "plain vs plain")
f = run("flat")
cmp(p1, f, "plain vs sharded(flat, no collectives)")
if use_rccl:
    cmp(p1, run("reduce_scatter"), "plain vs sharded(reduce_scatter)")
    cmp(p1, run("all_to_all"), "plain vs sharded(all_to_all)")
    dist.destroy_process_group()
