"""TEST INFRASTRUCTURE: torch-CPU stand-ins for clipa_amd.ops with the same signatures and the same bf16
storage points, so the host-side orchestration (autograd glue in clipa_amd.engine / model / loss: shapes,
operand forms, gradient wiring, recompute) can be exercised without a GPU.  Never imported by the product."""
import math

import torch

from oracle import clip_oracle as O

EPI_NONE, EPI_ACT, EPI_ADD, EPI_DACT = 0, 1, 2, 3
ACT_GELU_ERF, ACT_GELU_TANH, ACT_QUICK_GELU = 0, 1, 2
DT_U8, DT_BF16, DT_F32 = 0, 1, 2
POOL_FIRST, POOL_LAST, POOL_INDEX, POOL_MEAN_ALL, POOL_MEAN_PATCH = 0, 1, 2, 3, 4
bf16, f32 = torch.bfloat16, torch.float32
_ACT = {0: "gelu_erf", 1: "gelu_tanh", 2: "quick_gelu"}


def _act(x, a):
    return O.activation(x, _ACT[a])


def _dact(x, a):
    x = x.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        _act(x, a).sum().backward()
    return x.grad


def _e4m3(x):
    return x.float().clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)


def _from_e4m3(x8):
    return x8.view(torch.float8_e4m3fn).float()


def gemm_nt(a, b, bias=None, *, epi=EPI_NONE, act=0, aux=None, alpha=1.0, out_f32=False, want_pre=False, out=None, want_act=False):
    if want_act:
        return gemm_nt(a, b, bias, epi=epi, act=act, aux=aux, alpha=alpha), activation_fwd(aux, act)
    v = (a.float() @ b.float().T) * alpha
    if bias is not None:
        v = v + bias.float()
    if out_f32:
        return v
    v = v.to(bf16)
    pre = _e4m3(v) if want_pre == "e4m3" else v
    if aux is not None and aux.dtype == torch.uint8:
        aux = _from_e4m3(aux)
    if epi == EPI_ACT:
        v = _act(v.float(), act).to(bf16)
    elif epi == EPI_ADD:
        v = (v.float() + aux.float()).to(bf16)
    elif epi == EPI_DACT:
        v = (v.float() * _dact(aux.float(), act)).to(bf16)
    return (v, pre) if want_pre else v


FMT_E4M3, FMT_E5M2 = 0, 1
u8 = torch.uint8
_F8 = {0: (torch.float8_e4m3fn, 448.0), 1: (torch.float8_e5m2, 57344.0)}


def quantize_rows(x, fmt=FMT_E4M3, want_colsum=False, want_rownorm=False):
    dt, fmax = _F8[fmt]
    xf = x.float()
    amax = xf.abs().amax(dim=1)
    s = torch.where(amax > 0, fmax / amax, torch.ones_like(amax))
    dq = torch.where(amax > 0, amax / fmax, torch.zeros_like(amax))
    q = (xf * s[:, None]).to(dt).view(u8)
    if want_colsum and want_rownorm:
        return q, dq, xf.sum(0), xf.norm(dim=1)
    return (q, dq, xf.sum(0)) if want_colsum else (q, dq)


def row_bound(rownorm, wnorm, bmax=None, factor=1.13):
    b = factor * rownorm * wnorm + (bmax if bmax is not None else 0.0)
    return torch.where(b > 0, b / 448.0, torch.zeros_like(b)), torch.where(b > 0, 448.0 / b, torch.zeros_like(b))


def rownorm_max(w):
    return w.float().norm(dim=1).max().reshape(1)


def absmax(v):
    return v.float().abs().max().reshape(1)


def rowscale_max(a, b=None):
    return (a * b if b is not None else a).max().reshape(1).float()


def scale_quantize_rows(x, rowscale, t, act=-1):
    v = _from_e4m3(x) if x.dtype == torch.uint8 else x.float()
    if act >= 0:
        v = _act(v, act).to(bf16).float()
    inv = torch.where(t > 0, 1.0 / t, torch.zeros_like(t))
    return _e4m3(v * (rowscale * inv)[:, None])


def layernorm_fwd_q8s(x, gamma, beta, rowscale, t, eps=1e-5):
    y = layernorm_fwd(x, gamma, beta, eps)
    return scale_quantize_rows(y.reshape(-1, y.shape[-1]), rowscale, t).reshape(y.shape)


def gemm_tn_f8(p8, q8, t=None, alpha=1.0, fmt_p=FMT_E4M3, out_dtype=f32):
    v = p8.view(_F8[fmt_p][0]).float().T @ q8.view(torch.float8_e4m3fn).float()
    return (v * (alpha * (float(t) if t is not None else 1.0))).to(out_dtype)


def layernorm_fwd_q8(x, gamma, beta, eps=1e-5, want_bf16=False, want_rownorm=False):
    y = layernorm_fwd(x, gamma, beta, eps)
    q, dq = quantize_rows(y.reshape(-1, y.shape[-1]))
    r = ((y if want_bf16 else None), q.reshape(y.shape), dq)
    return r + (y.reshape(-1, y.shape[-1]).float().norm(dim=1),) if want_rownorm else r


def gemm_nt_f8_emit(a8, sa, b8, sb, aux8, t, *, act=0, fmt_a=FMT_E4M3):
    return gemm_nt_f8(a8, sa, b8, sb, None, epi=EPI_DACT, act=act, aux=aux8, fmt_a=fmt_a), scale_quantize_rows(aux8, sa, t, act=act)


def gemm_nt_f8(a8, sa, b8, sb, bias=None, *, epi=EPI_NONE, act=0, aux=None, alpha=1.0, want_pre=False, fmt_a=FMT_E4M3,
               fmt_b=FMT_E4M3, out_scale=None, want_colsum=False):
    a = a8.view(_F8[fmt_a][0]).float() * (sa[:, None] if sa is not None else 1.0)
    b = b8.view(_F8[fmt_b][0]).float() * (sb[:, None] if sb is not None else 1.0)
    r = gemm_nt(a, b, bias, epi=epi, act=act, aux=aux, alpha=alpha, want_pre=want_pre)
    if out_scale is None:
        return r
    out, pre = r if want_pre else (r, None)
    res = [_e4m3(out.float() * out_scale[:, None])]
    if want_pre:
        res.append(pre)
    if want_colsum:
        res.append(out.float().sum(0))
    return res[0] if len(res) == 1 else tuple(res)


def gemm_tn(p, q, out_dtype=f32, want_colsum=False):
    out = (p.float().T @ q.float()).to(out_dtype)
    return (out, p.float().sum(0)) if want_colsum else out


def layernorm_fwd(x, gamma, beta, eps=1e-5, out_dtype=None):
    return O.layer_norm(x.float(), gamma, beta, eps).to(out_dtype or x.dtype)


def layernorm_bwd(x, gamma, dy, dres=None, eps=1e-5, beta=None, q8_fmt=None, want_rownorm=False):
    if q8_fmt is not None:
        r = layernorm_bwd(x, gamma, dy, dres, eps, beta)
        return (*r, quantize_rows(r[0].reshape(-1, x.shape[-1]), q8_fmt, want_colsum=True, want_rownorm=want_rownorm))
    xr = x.float().detach().requires_grad_(True)
    g = gamma.detach().clone().requires_grad_(True)
    b = torch.zeros_like(g, requires_grad=True)
    with torch.enable_grad():
        O.layer_norm(xr, g, b, eps).backward(dy.float())
    dx = xr.grad + (dres.float() if dres is not None else 0)
    if beta is not None:      # the emitting form: + LayerNorm(x), exactly what layernorm_fwd returns
        return dx.to(x.dtype), g.grad, b.grad, layernorm_fwd(x, gamma, beta, eps, out_dtype=dy.dtype)
    return dx.to(x.dtype), g.grad, b.grad


def attention_fwd(qkv, B, L, H, causal, want_stats=False):
    D = qkv.shape[1] // 3
    out = O.attention(qkv.float().reshape(B, L, 3 * D), H, causal).reshape(B * L, D).to(bf16)
    return (out, torch.zeros(B * H * L, 2)) if want_stats else out


def attention_bwd(qkv, out, dout, stats, B, L, H, causal):
    D = qkv.shape[1] // 3
    x = qkv.float().reshape(B, L, 3 * D).detach().requires_grad_(True)
    with torch.enable_grad():
        O.attention(x, H, causal).backward(dout.float().reshape(B, L, D))
    return x.grad.reshape(B * L, 3 * D).to(bf16)


class VarLen:
    """Stand-in of clipa_amd.ops.VarLen (same fields, CPU tensors)."""

    def __init__(self, lens, ctx, device):
        lens = lens.to(torch.int64).cpu()
        self.B, self.ctx = int(lens.numel()), int(ctx)
        start = torch.cumsum(lens, 0) - lens
        self.T = int(lens.sum())
        self.rows = max(256, (self.T + 255) // 256 * 256)
        self.lens_host, self.seq_start, self.seq_len = lens, start.to(torch.int32), lens.to(torch.int32)
        src = torch.repeat_interleave(torch.arange(self.B) * self.ctx - start, lens) + torch.arange(self.T)
        self.src_rows = torch.cat([src, torch.full((self.rows - self.T,), -1, dtype=torch.int64)])
        self.last_rows = start + lens - 1
        self.classes = []
        self.sum_len2 = float((lens.double() ** 2).sum())


def attention_fwd_varlen(qkv, vl, H, causal, want_stats=False):
    D = qkv.shape[1] // 3
    out = torch.zeros((vl.rows, D), dtype=bf16)
    for s0, n in zip(vl.seq_start.tolist(), vl.seq_len.tolist()):
        out[s0:s0 + n] = O.attention(qkv[s0:s0 + n].float().reshape(1, n, 3 * D), H, causal).reshape(n, D).to(bf16)
    return (out, torch.zeros(vl.rows * H, 2)) if want_stats else out


def attention_bwd_varlen(qkv, out, dout, stats, vl, H, causal):
    D = qkv.shape[1] // 3
    dq = torch.zeros_like(qkv)
    for s0, n in zip(vl.seq_start.tolist(), vl.seq_len.tolist()):
        x = qkv[s0:s0 + n].float().reshape(1, n, 3 * D).detach().requires_grad_(True)
        with torch.enable_grad():
            O.attention(x, H, causal).backward(dout[s0:s0 + n].float().reshape(1, n, D))
        dq[s0:s0 + n] = x.grad.reshape(n, 3 * D).to(bf16)
    return dq


def patchify(img, P, Kp, mean=None, std=None):
    B, _, S, _ = img.shape
    g = S // P
    x = O.normalize_images(img, mean, std) if mean is not None else img.float()
    K = 3 * P * P
    pt = x[:, :, :g * P, :g * P].reshape(B, 3, g, P, g, P).permute(0, 2, 4, 3, 5, 1).reshape(B * g * g, K)
    return torch.nn.functional.pad(pt, (0, Kp - K)).to(bf16)


def assemble_tokens(patch, cls, pos, B, L):
    D = patch.shape[1]
    x = torch.cat([cls.to(bf16).float().expand(B, 1, D), patch.float().reshape(B, L - 1, D)], 1) + pos.to(bf16).float()
    return x.reshape(B * L, D).to(bf16)


def assemble_tokens_bwd(dtok, B, L, need_pos=True):
    D = dtok.shape[1]
    d = dtok.float().reshape(B, L, D)
    return d[:, 1:].reshape(-1, D).to(bf16), d[:, 0].sum(0), (d.sum(0) if need_pos else None)


def embed_tokens(ids, table, pos):
    B, T = ids.shape
    return (table.to(bf16).float()[ids] + pos.to(bf16).float()).reshape(B * T, -1).to(bf16)


def embed_tokens_bwd(ids, dx, vocab, need_table=True, need_pos=True):
    B, T = ids.shape
    D = dx.shape[1]
    dt = torch.zeros(vocab, D).index_add_(0, ids.reshape(-1), dx.float()) if need_table else None
    dp = dx.float().reshape(B, T, D).sum(0) if need_pos else None
    return dt, dp


def argmax_tokens(ids):
    return ids.argmax(-1).to(torch.int32)


def _pool(x3, mode, idx):
    B = x3.shape[0]
    if mode == POOL_FIRST:
        return x3[:, 0]
    if mode == POOL_LAST:
        return x3[:, -1]
    if mode == POOL_INDEX:
        return x3[torch.arange(B), idx.long()]
    if mode == POOL_MEAN_ALL:
        return x3.mean(1)
    return x3[:, 1:].mean(1)


def pool_fwd(x, B, L, mode, idx=None):
    return _pool(x.float().reshape(B, L, -1), mode, idx)


def pool_bwd(dout, B, L, mode, idx=None):
    x = torch.zeros(B, L, dout.shape[-1], requires_grad=True)
    with torch.enable_grad():
        _pool(x, mode, idx).backward(dout.float())
    return x.grad.reshape(B * L, -1).to(bf16)


def l2norm_fwd(x, eps=1e-12, want_bf16=False):
    n = x.norm(dim=-1).clamp_min(eps)
    y = x / n[:, None]
    return y, (y.to(bf16) if want_bf16 else None), 1.0 / n


def l2norm_bwd(y, inv, dy):
    return inv[:, None] * (dy - y * (y * dy).sum(-1, keepdim=True))


def colsum(dy):
    return dy.float().sum(0)


def to_bf16(t):
    return t.to(bf16)


def to_f32(t):
    return t.float()


def transpose_bf16(t):
    return t.to(bf16).T.contiguous()


def activation_fwd(x, act):
    if x.dtype == torch.uint8:
        x = _from_e4m3(x)
    return _act(x.float(), act).to(bf16)


def simce(rows, cols, n_valid, label0, gscale, scale=None, want_grad=True):
    R = rows.shape[0]
    n8 = (n_valid + 7) // 8 * 8
    s = float(scale.reshape(-1)[0]) if scale is not None else 1.0
    raw = rows.float() @ cols[:n_valid].float().T
    logits = raw * s
    labels = torch.arange(R) + label0
    loss_rows = torch.logsumexp(logits, dim=1) - logits[torch.arange(R), labels]
    if not want_grad:
        return loss_rows, None, None
    p = torch.softmax(logits, dim=1)
    p[torch.arange(R), labels] -= 1.0
    g = p * gscale
    dl = torch.zeros((R, n8), dtype=bf16)
    dl[:, :n_valid] = (g * s).to(bf16)
    return loss_rows, dl, (g * raw).sum(1)


def sum_scale(x, scale, out=None, accumulate=False):
    v = (x.sum() * scale).reshape(())
    if out is None:
        return v
    out.copy_(out + v if accumulate else v)
    return out


def adamw_(param, grad, exp_avg, exp_avg_sq, *, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    g = grad.float() * grad_scale
    p = param.float() * (1 - lr * weight_decay)
    exp_avg.mul_(beta1).add_(g, alpha=1 - beta1)
    exp_avg_sq.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    denom = exp_avg_sq.sqrt() / math.sqrt(1 - beta2 ** step) + eps
    param.copy_((p - lr / (1 - beta1 ** step) * exp_avg / denom).to(param.dtype))


def adamw_multi_(params, grads, exp_avgs, exp_avg_sqs, grad_scale_dev=None, clamp_index=-1, clamp=(0.0, 0.0), **kw):
    if grad_scale_dev is not None:
        kw = dict(kw, grad_scale=kw.get("grad_scale", 1.0) * float(grad_scale_dev))
    for k, (p, g, m, v) in enumerate(zip(params, grads, exp_avgs, exp_avg_sqs)):
        adamw_(p, g, m, v, **kw)
        if k == clamp_index:
            p.clamp_(clamp[0], clamp[1])


def grad_sqnorm(grads, buf=None):
    if buf is None:
        buf = torch.zeros(3)
    buf[0] += sum((g.float() ** 2).sum() for g in grads)
    return buf


def clip_coef(buf, max_norm):
    buf[1] = torch.sqrt(buf[0])
    buf[2] = torch.clamp(max_norm / (buf[1] + 1e-6), max=1.0)
    return buf[1], buf[2]


def grad_clip_coef(grads, max_norm):
    return clip_coef(grad_sqnorm(grads), max_norm)


def reduce_shards(pieces, world, out=None, scale=None, out_dtype=None):
    n = pieces.numel() // world
    r = pieces.float().view(world, n).sum(0) * (1.0 / world if scale is None else scale)
    if out is None:
        return r.to(out_dtype or pieces.dtype)
    out.copy_(r)
    return out


def gather_rows(x, rows):
    out = x[rows.clamp_min(0)].contiguous()
    out[rows < 0] = 0                       # rows outside [0, n) read zeros, as the kernel does
    return out


def scatter_rows(dy, rows, n_dst):
    dx = torch.zeros((n_dst, dy.shape[1]), dtype=dy.dtype)
    ok = rows >= 0
    dx[rows[ok]] = dy[ok]
    return dx
