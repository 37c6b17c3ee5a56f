"""AdamW whose update runs in libclipa_hip.so (one fused pass over p / g / m / v, many tensors per launch, instead
of the ~10 elementwise kernels per tensor of the unfused torch optimizer).  Same hyper-parameter semantics as
torch.optim.AdamW as the reference trainer configures it (training/main.py:311-326: two param groups,
weight_decay 0 for ndim<2 / bn / ln / bias / logit_scale; betas, eps from the CLI)."""
import torch

from . import ops


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            # tensors of one group that share dtypes and step count go out in one multi-tensor call
            # (a few launches per group instead of one per tensor: ~1800 -> ~20 per step for ViT-L/16)
            buckets = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros(p.shape, device=p.device, dtype=torch.float32)
                    st["exp_avg_sq"] = torch.zeros(p.shape, device=p.device, dtype=torch.float32)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                key = (p.dtype, g.dtype, st["step"], p.device)
                buckets.setdefault(key, []).append((p, g, st))
            for (_, _, step, _), items in buckets.items():
                ops.adamw_multi_([p.data for p, _, _ in items], [g for _, g, _ in items],
                                 [st["exp_avg"] for _, _, st in items], [st["exp_avg_sq"] for _, _, st in items],
                                 lr=group["lr"], beta1=b1, beta2=b2, eps=group["eps"],
                                 weight_decay=group["weight_decay"], step=step)
                for p, _, _ in items:
                    # the kernel wrote through raw pointers: bump the version counters so autograd's
                    # saved-tensor checks and the engine's WeightCache see the in-place update
                    torch.autograd.graph.increment_version(p)
        return loss
