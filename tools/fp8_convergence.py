#!/usr/bin/env python
"""Convergence A/B of the reduced-precision paths: the SAME model, data, seed and optimizer trained for N steps with
precision bf16 (all-recompute = bit-exact gradients), bf16 with the e4m3 pre-activation kept in every block (`bf16_h8`: the
activation plan of bench.py's `value`; VERDICT r4 next #1b) and fp8 (e4m3 operands, e4m3 or e5m2 gradient operands; VERDICT r2
item 6a; since round 6 forward, input-gradient AND weight-gradient products, `fp8_h8` additionally with bench.py's keep plan) on the HIP engine, and - for the first steps - the fp32 CPU oracle (oracle/clip_oracle.py), on LEARNABLE synthetic pairs.

    python tools/fp8_convergence.py --steps 200 --batch 256 > profiles/r03_fp8_convergence_S16_112.jsonl
    python tools/fp8_convergence.py --arms bf16,bf16_h8 --lr 3e-4 --seed 1 > profiles/r05_h8_convergence_S16_112_lr3e-4_seed1.jsonl

Data: C "concepts"; a pair of concept c is (image = a fixed random low-frequency pattern of c + per-sample noise, caption = a
fixed token sequence of c with a few random filler tokens).  With C >= batch the contrastive task is learnable to a loss well
below ln(batch), so a precision problem shows up as a curve that separates from the bf16 one.  Model: ViT-S/16 @ 112 + text
ctx 32 (BASELINE configs[0] dimensions).  Prints one JSON line per run and a summary.
"""
import argparse
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_data(C, S, ctx, vocab, seed):
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(C, 3, S // 8, S // 8, generator=g)
    base = torch.nn.functional.interpolate(base, size=(S, S), mode="bilinear", align_corners=False)      # low-frequency patterns
    toks = torch.randint(1, vocab - 2, (C, ctx), generator=g)
    length = torch.randint(6, ctx - 1, (C,), generator=g)
    for c in range(C):
        toks[c, length[c]] = vocab - 1          # EOT = largest id (model.py:254 argmax)
        toks[c, length[c] + 1:] = 0
    return base, toks, length


def batch_of(base, toks, length, idx, noise, seed, vocab):
    g = torch.Generator().manual_seed(seed)
    img = (base[idx] + noise * torch.randn(base[idx].shape, generator=g)).clamp(0, 1)
    img_u8 = (img * 255).round().to(torch.uint8)
    t = toks[idx].clone()
    for j, c in enumerate(idx.tolist()):        # two random filler tokens inside the caption
        pos = torch.randint(1, int(length[c]), (2,), generator=g)
        t[j, pos] = torch.randint(1, vocab - 2, (2,), generator=g)
    return img_u8, t


def run_engine(precision, grad_fmt, args, base, toks, length, cfg):
    import clipa_amd
    from clipa_amd.optim import AdamW
    dev = torch.device("cuda", 0)
    torch.manual_seed(args.seed)
    m = clipa_amd.CLIP(**cfg, output_dict=True).to(dev)
    pred = precision == "fp8_h8_pred"
    if pred:
        precision = "fp8_h8"
    h8 = precision in ("bf16_h8", "fp8_h8")
    if h8:
        precision = precision[:-3]
    if precision in ("bf16", "fp8"):
        clipa_amd.convert_weights_to_lp(m, torch.bfloat16)
    if h8:    # bench.py's plan at the headline shape: every block keeps the e4m3 pre-activation, x1 and the attention output
        for t in (m.visual.transformer, m.transformer):
            t.keep_counts = dict(t.keep_counts, h8=t.layers, a=t.layers, x1=t.layers)
    if precision == "fp8":
        for t in (m.visual.transformer, m.transformer):
            t.fp8, t.fp8_grad_format, t.fp8_predicted_scales = True, grad_fmt, pred
    m.set_grad_checkpointing(True)
    named = list(m.named_parameters())
    exclude = lambda n, p: p.ndim < 2 or "bn" in n or "ln" in n or "bias" in n or "logit_scale" in n
    opt = AdamW([{"params": [p for n, p in named if exclude(n, p)], "weight_decay": 0.},
                 {"params": [p for n, p in named if not exclude(n, p)], "weight_decay": 0.2}],
                lr=args.lr, betas=(0.9, 0.95), eps=1e-6, grad_clip_norm=1.0, clamp=(m.logit_scale, 0.0, math.log(100)))
    loss_fn = clipa_amd.ClipLoss()
    losses = []
    C = base.shape[0]
    for step in range(args.steps):
        for gp in opt.param_groups:
            gp["lr"] = args.lr * min(1.0, (step + 1) / 20)
        idx = torch.randperm(C, generator=torch.Generator().manual_seed(1000 + step))[:args.batch]
        img, txt = batch_of(base, toks, length, idx, args.noise, 5000 + step, cfg["text_cfg"]["vocab_size"])
        opt.zero_grad(set_to_none=True)
        out = m(img.to(dev), txt.to(dev))
        loss = loss_fn(**out, output_dict=True)["contrastive_loss"]
        loss.backward()
        opt.step()
        losses.append(float(loss))
    return losses


def run_oracle(args, base, toks, length, cfg, steps):
    from oracle import clip_oracle as O
    import clipa_amd
    torch.manual_seed(args.seed)
    m = clipa_amd.CLIP(**cfg)                     # parameter container only: same init as the engine runs
    sd = {k: v.detach().clone().float().requires_grad_(v.requires_grad) for k, v in m.named_parameters()}
    ocfg = O.oracle_cfg(cfg)
    exclude = lambda n, p: p.ndim < 2 or "bn" in n or "ln" in n or "bias" in n or "logit_scale" in n
    opt = torch.optim.AdamW([{"params": [p for n, p in sd.items() if exclude(n, p)], "weight_decay": 0.},
                             {"params": [p for n, p in sd.items() if not exclude(n, p)], "weight_decay": 0.2}],
                            lr=args.lr, betas=(0.9, 0.95), eps=1e-6)
    losses = []
    C = base.shape[0]
    for step in range(steps):
        for gp in opt.param_groups:
            gp["lr"] = args.lr * min(1.0, (step + 1) / 20)
        idx = torch.randperm(C, generator=torch.Generator().manual_seed(1000 + step))[:args.batch]
        img, txt = batch_of(base, toks, length, idx, args.noise, 5000 + step, cfg["text_cfg"]["vocab_size"])
        opt.zero_grad()
        i, t, s = O.clip_forward(sd, ocfg, O.normalize_images(img), txt)
        loss, _ = O.clip_loss(i, t, s)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(sd.values()), 1.0)
        opt.step()
        with torch.no_grad():
            sd["logit_scale"].clamp_(0, math.log(100))
        losses.append(float(loss))
    return losses


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--concepts", type=int, default=1024)
    ap.add_argument("--noise", type=float, default=0.15)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--oracle-steps", type=int, default=12)
    ap.add_argument("--arms", default="bf16,fp8_e4m3_grad,fp8_e5m2_grad", help="comma-separated: bf16, bf16_h8, fp8_e4m3_grad, fp8_e5m2_grad, fp8_h8, fp8_h8_pred (round 6: every fp8 arm with fp8 weight gradients; fp8_h8 = bench.py's fp8 plan, the e4m3 pre-activation kept in every block; _pred = + the predicted-row-scale knob)")
    args = ap.parse_args()
    import clipa_amd
    cfg = clipa_amd.get_model_config("ViT-S-16")
    cfg["vision_cfg"]["image_size"] = 112
    cfg["text_cfg"]["context_length"] = 32
    base, toks, length = make_data(args.concepts, 112, 32, cfg["text_cfg"]["vocab_size"], 42)
    runs = {}
    ARMS = {"bf16": ("bf16", None), "bf16_h8": ("bf16_h8", None), "fp8_e4m3_grad": ("fp8", "e4m3"), "fp8_e5m2_grad": ("fp8", "e5m2"),
            "fp8_h8": ("fp8_h8", "e4m3"), "fp8_h8_pred": ("fp8_h8_pred", "e4m3")}
    for name in args.arms.split(","):
        prec, fmt = ARMS[name]
        runs[name] = run_engine(prec, fmt, args, base, toks, length, cfg)
        print(json.dumps({"run": name, "losses": [round(x, 4) for x in runs[name]]}), flush=True)
    if args.oracle_steps > 0:
        torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
        runs["oracle_fp32_cpu"] = run_oracle(args, base, toks, length, cfg, args.oracle_steps)
        print(json.dumps({"run": "oracle_fp32_cpu", "losses": [round(x, 4) for x in runs["oracle_fp32_cpu"]]}), flush=True)
    tail = lambda xs: float(np.mean(xs[-20:]))
    summ = {"summary": True, "model": "ViT-S-16@112 + text-32", "batch": args.batch, "steps": args.steps, "lr": args.lr, "seed": args.seed, "ln_batch": round(math.log(args.batch), 4),
            "final20_mean": {k: round(tail(v), 4) for k, v in runs.items() if len(v) >= 20},
            "max_abs_gap_vs_bf16": {k: round(max(abs(a - b) for a, b in zip(v, runs["bf16"])), 4) for k, v in runs.items() if k != "bf16"},
            "mean_gap_last50_vs_bf16": {k: round(float(np.mean([a - b for a, b in zip(v[-50:], runs["bf16"][-50:])])), 4)
                                        for k, v in runs.items() if k != "bf16" and len(v) >= 50}}
    print(json.dumps(summ), flush=True)


if __name__ == "__main__":
    main()
