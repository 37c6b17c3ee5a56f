"""AdamW whose update runs in libclipa_hip.so (one fused pass over p / g / m / v, many tensors per launch, instead
of the ~10 elementwise kernels per tensor of the unfused torch optimizer).  Same hyper-parameter semantics as
torch.optim.AdamW as the reference trainer configures it (training/main.py:311-326: two param groups,
weight_decay 0 for ndim<2 / bn / ln / bias / logit_scale; betas, eps from the CLI).

The optimizer tail of training/train.py:270-286 rides along without host round trips:
  grad_clip_norm=G   torch.nn.utils.clip_grad_norm_(params, G): the total norm and the clip coefficient are computed on
                     the device (ops.grad_clip_coef) and the coefficient multiplies the gradients INSIDE the update
                     (`.grad` itself stays unclipped; `last_grad_norm` holds the norm as a device scalar);
  clamp=(p, lo, hi)  p.clamp_(lo, hi) right after p's update, in the same kernel (logit_scale.clamp_(0, ln 100))."""
import torch

from . import ops


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_clip_norm=None, clamp=None):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_clip_norm = grad_clip_norm
        self.clamp = clamp                    # (parameter, lo, hi) or None
        self.last_grad_norm = None

    def _normalise_state(self):
        """The HIP kernel reads and writes the moments as contiguous `float*` whatever the parameter dtype, and counts
        steps with a Python int.  torch.optim.Optimizer.load_state_dict casts floating-point state to the PARAMETER's
        dtype (bf16 moments for bf16 parameters) and a torch.optim.AdamW checkpoint carries a tensor `step`: bring both
        back to what the kernel expects (training/main.py:338-356 resumes exactly this way)."""
        for p, st in self.state.items():
            if not st:
                continue
            for k in ("exp_avg", "exp_avg_sq"):
                if k in st and (st[k].dtype != torch.float32 or not st[k].is_contiguous() or st[k].device != p.device):
                    st[k] = st[k].to(device=p.device, dtype=torch.float32).contiguous()
            if "step" in st and torch.is_tensor(st["step"]):
                st["step"] = int(st["step"].item())

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._normalise_state()

    def __setstate__(self, state):
        super().__setstate__(state)
        self._normalise_state()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        coef = None
        if self.grad_clip_norm is not None:
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                     for g in self.param_groups for p in g["params"] if p.grad is not None]
            if grads:
                self.last_grad_norm, coef = ops.grad_clip_coef(grads, self.grad_clip_norm)
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
                ci, cl = -1, (0.0, 0.0)
                if self.clamp is not None:
                    for k, (p, _, _) in enumerate(items):
                        if p is self.clamp[0]:
                            ci, cl = k, (self.clamp[1], self.clamp[2])
                ops.adamw_multi_([p.data for p, _, _ in items], [g for _, g, _ in items],
                                 [st["exp_avg"] for _, _, st in items], [st["exp_avg_sq"] for _, _, st in items],
                                 lr=group["lr"], beta1=b1, beta2=b2, eps=group["eps"],
                                 weight_decay=group["weight_decay"], step=step, grad_scale_dev=coef,
                                 clamp_index=ci, clamp=cl)
                for p, _, _ in items:
                    # the kernel wrote through raw pointers: bump the version counters so autograd's
                    # saved-tensor checks and the engine's WeightCache see the in-place update
                    torch.autograd.graph.increment_version(p)
        return loss
