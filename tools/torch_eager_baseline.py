"""What the reference's own execution model does on this GPU: the same CLIP (ViT-L/16 @ 224 + 12-layer text tower, InfoNCE, AdamW,
per-block gradient checkpointing) written with stock PyTorch-ROCm modules - nn.Linear / nn.LayerNorm / nn.GELU /
F.scaled_dot_product_attention / torch.utils.checkpoint / torch.optim.AdamW(fused) - pure bf16 like `--precision bf16`.
Same synthetic batch shape and the same timing protocol as bench.py; prints one JSON line.  Not part of the product or the tests.
    python tools/torch_eager_baseline.py [--batch 4096] [--model L16|H14] [--steps 2] [--no-checkpoint]"""
import argparse
import json
import math
import time

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint


class Block(nn.Module):
    def __init__(self, d, heads, causal):
        super().__init__()
        self.ln_1, self.ln_2 = nn.LayerNorm(d), nn.LayerNorm(d)
        self.in_proj, self.out_proj = nn.Linear(d, 3 * d), nn.Linear(d, d)
        self.c_fc, self.c_proj = nn.Linear(d, 4 * d), nn.Linear(4 * d, d)
        self.heads, self.causal = heads, causal

    def forward(self, x):                      # [B, L, D]
        B, L, D = x.shape
        q, k, v = self.in_proj(self.ln_1(x)).view(B, L, 3, self.heads, D // self.heads).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(q, k, v, is_causal=self.causal).transpose(1, 2).reshape(B, L, D)
        x = x + self.out_proj(a)
        return x + self.c_proj(F.gelu(self.c_fc(self.ln_2(x))))


class Tower(nn.Module):
    def __init__(self, d, layers, heads, causal):
        super().__init__()
        self.blocks = nn.ModuleList([Block(d, heads, causal) for _ in range(layers)])

    ckpt = True

    def forward(self, x):
        for b in self.blocks:
            x = checkpoint(b, x, use_reentrant=False) if Tower.ckpt else b(x)
        return x


class CLIP(nn.Module):
    def __init__(self, vw, vl, vh, patch, size, tw, tl, th, embed, ctx=77, vocab=49408):
        super().__init__()
        g = size // patch
        self.conv1 = nn.Conv2d(3, vw, patch, patch, bias=False)
        self.cls = nn.Parameter(torch.randn(vw) * vw ** -0.5)
        self.vpos = nn.Parameter(torch.randn(g * g + 1, vw) * vw ** -0.5)
        self.ln_pre, self.ln_post = nn.LayerNorm(vw), nn.LayerNorm(vw)
        self.visual = Tower(vw, vl, vh, False)
        self.vproj = nn.Parameter(torch.randn(vw, embed) * vw ** -0.5)
        self.tok = nn.Embedding(vocab, tw)
        self.tpos = nn.Parameter(torch.randn(ctx, tw) * 0.01)
        self.text = Tower(tw, tl, th, True)
        self.ln_final = nn.LayerNorm(tw)
        self.tproj = nn.Parameter(torch.randn(tw, embed) * tw ** -0.5)
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))

    def forward(self, image, text):
        x = self.conv1(image).flatten(2).transpose(1, 2)
        x = torch.cat([self.cls.expand(x.shape[0], 1, -1), x], 1) + self.vpos
        x = self.ln_post(self.visual(self.ln_pre(x))[:, 0]) @ self.vproj
        t = self.text(self.tok(text) + self.tpos)
        t = self.ln_final(t)[torch.arange(t.shape[0]), text.argmax(-1)] @ self.tproj
        return F.normalize(x, dim=-1), F.normalize(t, dim=-1), self.logit_scale.exp()


ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--model", default="L16")
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--no-checkpoint", action="store_true", help="keep every activation (needs a batch small enough to fit)")
args = ap.parse_args()
Tower.ckpt = not args.no_checkpoint
cfg = {"L16": dict(vw=1024, vl=24, vh=16, patch=16, size=224, tw=768, tl=12, th=12, embed=768),
       "H14": dict(vw=1280, vl=32, vh=16, patch=14, size=224, tw=1024, tl=24, th=16, embed=1024)}[args.model]
dev = "cuda"
torch.manual_seed(0)
model = CLIP(**cfg).to(dev).to(torch.bfloat16)
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.95), eps=1e-6, weight_decay=0.2, fused=True)
B = args.batch
image = torch.randn(B, 3, cfg["size"], cfg["size"], device=dev, dtype=torch.bfloat16)
text = torch.randint(1, 49407, (B, 77), device=dev)
labels = torch.arange(B, device=dev)


def step():
    opt.zero_grad(set_to_none=True)
    i, t, s = model(image, text)
    logits = s * i @ t.T
    loss = (F.cross_entropy(logits.float(), labels) + F.cross_entropy(logits.T.float(), labels)) / 2
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step()
    with torch.no_grad():
        model.logit_scale.clamp_(0, math.log(100))
    return loss


for _ in range(args.warmup):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    loss = step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / args.steps * 1e3
print(json.dumps({"baseline": "stock PyTorch-ROCm modules (hipBLASLt GEMMs, SDPA, torch.utils.checkpoint per block, fused AdamW), pure bf16",
                  "model": args.model, "batch": B, "grad_checkpointing": Tower.ckpt, "ms_per_step": round(ms, 1), "pairs_per_s": round(B / ms * 1e3, 1), "loss": round(float(loss), 4),
                  "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 2**30, 1), "torch": torch.__version__}), flush=True)
