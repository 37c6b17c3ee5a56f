"""Autograd glue of the MI355X engine: coarse-grained torch.autograd.Functions whose forward and
backward are sequences of HIP launches (clipa_amd.ops).  Mirrors, per function, the reference module
it stands in for (paths relative to /root/reference/clipa_torch):

  ResBlockFn     ResidualAttentionBlock.forward            open_clip/transformer.py:238-250
  LastBlockFn    the same for a tower's last block when the head reads one row per sample (returns those rows only)
  VisionStemFn   conv1 + cls/pos + ln_pre                  open_clip/transformer.py:480-503
  TokenDropFn    PatchDropout (row selection)              open_clip/transformer.py:53-83,501-502
  TextStemFn     token_embedding + positional_embedding    open_clip/model.py:245-247
  HeadFn         pool + ln_post/ln_final + projection      transformer.py:509-529, model.py:251-260
  L2NormFn       F.normalize(dim=-1)                       open_clip/model.py:240,263

Activation policy: a block keeps only its bf16 input and recomputes the rest in backward (what the
reference does with --grad-checkpointing, transformer.py:320-325), so ViT-L/16 at local batch 4096
(806 912 tokens) fits in HBM with whole-batch GEMMs (M = 806 912) instead of micro-batches.
"""
import weakref

import torch

from . import ops

bf16, f32 = torch.bfloat16, torch.float32



class WeightCache:
    """bf16 operand copies of the parameters ([N,K] forward form and [K,N] transposed form for the
    input-gradient GEMMs), refreshed when the parameter's version counter changes (i.e. once per
    optimizer step).  An entry is valid for (this parameter object, its version counter, its storage address): writes
    that go through `p.data` (EMA, manual weight surgery) bump no counter - call `clear()` (CLIP.invalidate_weight_cache)
    after such writes; `load_state_dict` does it by itself."""

    def __init__(self):
        self._c = {}

    def _get(self, p, kind, make):
        key = (id(p), kind)
        ver = p._version
        ent = self._c.get(key)
        if ent is None or ent[0] != ver or ent[1] != p.data_ptr() or ent[3]() is not p:   # id() may be recycled: weakref check
            with torch.no_grad():
                ent = (ver, p.data_ptr(), make(p.detach()), weakref.ref(p))
            self._c[key] = ent
        return ent[2]

    def w(self, p):   # [N,K] bf16
        return self._get(p, "w", lambda t: t if t.dtype == bf16 else ops.to_bf16(t))

    def wt(self, p):  # [K,N] bf16
        return self._get(p, "wt", ops.transpose_bf16)

    def f32(self, p):  # biases / LN affine as f32 vectors
        return self._get(p, "f32", lambda t: t if t.dtype == f32 else ops.to_f32(t))

    def custom(self, p, kind, make):
        return self._get(p, kind, make)

    def clear(self):
        self._c.clear()


def _like_param(g, p):
    return g if g.dtype == p.dtype else g.to(p.dtype)


# ---------------------------------------------------------------------------------------------
def _attn_fwd(qkv, cfg, want_stats):
    """-> (attention output, softmax statistics | None); fixed-length batches or packed variable-length sequences."""
    vl = cfg.get("varlen")
    if vl is not None:
        r = ops.attention_fwd_varlen(qkv, vl, cfg["H"], cfg["causal"], want_stats=want_stats)
    else:
        r = ops.attention_fwd(qkv, cfg["B"], cfg["L"], cfg["H"], cfg["causal"], want_stats=want_stats)
    return r if want_stats else (r, None)


def _attn_bwd(qkv, a, da, stats, cfg):
    vl = cfg.get("varlen")
    if vl is not None:
        return ops.attention_bwd_varlen(qkv, a, da, stats, vl, cfg["H"], cfg["causal"])
    return ops.attention_bwd(qkv, a, da, stats, cfg["B"], cfg["L"], cfg["H"], cfg["causal"])


def _block_forward(x, P, cfg, keep, need_y=True):
    """x [M,D] bf16. P: dict of operand tensors. Returns y and (if keep) the intermediates.
    keep: False (nothing), True / "full" (everything the backward reads), "light" (only the GEMM / attention
    outputs qkv, a, stats, x1, hpre - LayerNorm outputs and the activation are re-materialised in backward),
    "light8" (light with hpre kept as saturating e4m3 bytes written by the c_fc epilogue itself: 14 instead of 18 bytes
    per element of x; the backward takes gelu'(h) and the re-materialised activation from the e4m3 value - ~3 % rms
    rounding of h, the only tier whose gradients are not bit-identical to the recomputed block's) or
    "medium" (light without hpre, 10 bytes per element of x: backward re-runs LN2 + the c_fc GEMM,
    a third of the block's forward FLOPs, and skips the other three GEMMs and attention)."""
    B, L, H, causal, act = cfg["B"], cfg["L"], cfg["H"], cfg["causal"], cfg["act"]
    if cfg.get("fp8"):
        return _block_forward_fp8(x, P, cfg, keep, need_y)[:2]
    # the named tiers are sets of kept tensors; a frozenset names them one by one (round 4: what a byte buys differs per tensor -
    # the e4m3 pre-activation ~1.5 ms per GB, the attention output ~0.94, x1 ~0.91, qkv ~0.79 at ViT-L/16 - so bench.py's planner
    # keeps them independently): "qkv", "a" (with the softmax statistics), "x1", "h" (bf16 pre-activation) or "h8" (e4m3)
    ks = KEEP_SETS.get(keep, keep) if keep and keep is not True and keep != "full" else None
    full = bool(keep) and ks is None
    h1 = ops.layernorm_fwd(x, P["ln1_w"], P["ln1_b"], cfg["eps"])
    qkv = ops.gemm_nt(h1, P["w_in"], P["b_in"])
    a, stats = _attn_fwd(qkv, cfg, full or (ks is not None and "a" in ks))
    x1 = ops.gemm_nt(a, P["w_out"], P["b_out"], epi=ops.EPI_ADD, aux=x)
    h2 = ops.layernorm_fwd(x1, P["ln2_w"], P["ln2_b"], cfg["eps"])
    want_pre = "e4m3" if (ks is not None and "h8" in ks) else (full or (ks is not None and "h" in ks))
    if want_pre:
        g, hpre = ops.gemm_nt(h2, P["w_fc"], P["b_fc"], epi=ops.EPI_ACT, act=act, want_pre=want_pre)
    else:
        g, hpre = ops.gemm_nt(h2, P["w_fc"], P["b_fc"], epi=ops.EPI_ACT, act=act), None
    # (need_y=False: the backward-time recompute of a block wants the intermediates only - its output is the next block's input,
    # which that block kept; the c_proj GEMM would be thrown away)
    y = ops.gemm_nt(g, P["w_proj"], P["b_proj"], epi=ops.EPI_ADD, aux=x1) if need_y else None
    if full:
        return y, (h1, qkv, a, stats, x1, h2, hpre, g)
    if ks:
        return y, (None, qkv if "qkv" in ks else None, a if "a" in ks else None, stats if "a" in ks else None,
                   x1 if "x1" in ks else None, None, hpre, None)
    return y, None


KEEP_SETS = {"light": frozenset(("qkv", "a", "x1", "h")), "light8": frozenset(("qkv", "a", "x1", "h8")),
             "medium": frozenset(("qkv", "a", "x1"))}


def _lin8(xq, xs, P, name, **kw):
    """fp8 linear: row-quantised activation (xq, xs) against the cached row-quantised weight."""
    wq, ws = P["w8_" + name]
    return ops.gemm_nt_f8(xq, xs, wq, ws, P["b_" + name], **kw)


def _gradq8(dy, cfg):
    """The incoming gradient of a linear layer as fp8 operand, quantised ONCE per token for both of its products (input gradient
    and weight gradient), and its column sums = the layer's bias gradient from the same pass: -> (dq, ds, colsum)."""
    return ops.quantize_rows(dy, cfg.get("fp8_grad_fmt", ops.FMT_E4M3), want_colsum=True)


def _dlin8(dq, ds, P, name, cfg, **kw):
    """fp8 input gradient: the per-token quantised gradient against the cached quantised W^T."""
    wq, ws = P["wt8_" + name]
    return ops.gemm_nt_f8(dq, ds, wq, ws, None, fmt_a=cfg.get("fp8_grad_fmt", ops.FMT_E4M3), **kw)


def _wgrad8(dq, ds, sx, emit, out_dtype, cfg):
    """fp8 weight gradient dW = dY^T X (round 6).  The reduction runs over tokens, so the per-token scale ds of the quantised
    gradient cannot leave the product: the activation operand absorbs it before it is quantised, X8[m,:] = e4m3(ds[m] X[m,:] / t),
    with one device scalar t = max_m ds[m] sx[m] (sx = the activation's own row scale of the forward pass: nothing saturates, no
    amax history, recompute reproduces the bytes) and dW = t * dq^T X8.  emit(ds, t) -> X8 (the kernel that has X at hand)."""
    t = ops.rowscale_max(ds, sx)
    return ops.gemm_tn_f8(dq, emit(ds, t), t=t, fmt_p=cfg.get("fp8_grad_fmt", ops.FMT_E4M3), out_dtype=out_dtype)


def _fp8_keep_set(keep):
    """-> (kept tensor names | None, full): the named tiers of the bf16 engine and its per-tensor sets ("qkv", "a" = attention output +
    statistics, "x1", "h" = bf16 pre-activation, "h8" = the same as saturating e4m3 bytes written by the c_fc epilogue) hold in fp8
    mode too (round 6); True / "full" additionally keeps the LayerNorm outputs and the activation in bf16."""
    if not keep:
        return None, False
    if keep is True or keep == "full":
        return KEEP_SETS["light"], True
    return KEEP_SETS.get(keep, keep), False


def _block_forward_fp8(x, P, cfg, keep, need_y=True, scales=None):
    """_block_forward with the four linear layers on the fp8 MFMA path (BASELINE.json configs[3]): the LayerNorms emit the
    e4m3 operand of the GEMM that follows them, the attention output and the MLP activation are quantised per token by
    clipa_quantize_rows; everything between the GEMMs (residual stream, attention, softmax statistics, kept tensors) is
    bf16 exactly as in the bf16 engine.  -> (y, kept tensors | None, scales): scales = the per-token scales (s1, sa, s2, sg) of
    the four GEMM inputs (LN1 output, attention output, LN2 output, activation), a few MB that every block keeps - the fp8
    weight gradients of the backward derive their tensor scale from them (_wgrad8).  need_y=False: no output, only what the
    backward reads (_fp8_fill on nothing)."""
    B, L, H, causal, act = cfg["B"], cfg["L"], cfg["H"], cfg["causal"], cfg["act"]
    if not need_y:
        return None, _fp8_fill(x, (None,) * 8, P, cfg), scales
    ks, full = _fp8_keep_set(keep)
    ks = ks or frozenset()
    h1, q1, s1 = ops.layernorm_fwd_q8(x, P["ln1_w"], P["ln1_b"], cfg["eps"], want_bf16=full)
    qkv = _lin8(q1, s1, P, "in")
    del q1
    a, stats = _attn_fwd(qkv, cfg, "a" in ks)
    qa, sa = ops.quantize_rows(a)
    x1 = _lin8(qa, sa, P, "out", epi=ops.EPI_ADD, aux=x)
    del qa
    h2, q2, s2, rn2 = ops.layernorm_fwd_q8(x1, P["ln2_w"], P["ln2_b"], cfg["eps"], want_bf16=full, want_rownorm=True)      # (the row norms: 4 bytes per row)
    want_pre = "e4m3" if "h8" in ks else ("h" in ks or full)
    # Engine knob `fp8_predicted_scales` (round 6, off by default): the activation leaves the c_fc GEMM as the e4m3 operand of
    # c_proj (no bf16 copy, no row quantiser pass).  A tile of the GEMM cannot know its row's maximum, so the row scale is PREDICTED: |h[m,c]| <= ||LN2(x1)[m,:]|| * max_c ||W_fc[c,:]|| +
    # max |b_fc| (Cauchy-Schwarz; 1.13 covers the e4m3 rounding of both operands) and |gelu(h)| <= |h| - a rigorous bound, a few
    # binades above the row's true maximum, well inside e4m3's range.  Every tier uses the same scales (bit-identical forwards);
    # ragged shapes / the all-stored mode run GEMM + scaled quantiser with the same arithmetic.  Price (tests/test_fp8_gpu.py,
    # DESIGN 4): per-tensor gradient cosines against the fp32 reference 0.001-0.009 lower than with the rows' true maxima.
    sg, so = ops.row_bound(rn2, P["wn_fc"], P["bmax_fc"], 1.13) if cfg.get("fp8_predict") else (None, None)
    if not cfg.get("fp8_predict"):      # the default: bf16 activation + a row quantiser with the row's TRUE maximum
        r = _lin8(q2, s2, P, "fc", epi=ops.EPI_ACT, act=act, want_pre=want_pre)
        g, hpre = r if want_pre else (r, None)
        qg, sg = ops.quantize_rows(g)
        g = g if full else None
    elif full:
        g, hpre = _lin8(q2, s2, P, "fc", epi=ops.EPI_ACT, act=act, want_pre=True)
        qg = ops.scale_quantize_rows(g, so, P["one"])
    else:
        r = _lin8(q2, s2, P, "fc", epi=ops.EPI_ACT, act=act, want_pre=want_pre, out_scale=so)
        (qg, hpre), g = (r if want_pre else (r, None)), None
    del q2
    y = _lin8(qg, sg, P, "proj", epi=ops.EPI_ADD, aux=x1)
    del qg
    sc = (s1, sa, s2, sg)
    if not ks:
        return y, None, sc
    return y, (h1 if full else None, qkv if "qkv" in ks else None, a if "a" in ks else None, stats if "a" in ks else None,
               x1 if "x1" in ks else None, h2 if full else None, hpre, g if full else None), sc


def _fp8_fill(x, kept, P, cfg):
    """Whatever an fp8 block did not keep of (qkv, attention output + statistics, x1, pre-activation), recomputed bit for bit
    as the forward produced it.  The activation itself is never rebuilt in bf16: the weight gradient of c_proj re-materialises
    it as an fp8 operand from the pre-activation (_block_backward_fp8), so a missing pre-activation costs LN2 + the c_fc GEMM
    with a plain epilogue."""
    h1, qkv, a, stats, x1, h2, hpre, g = kept
    if qkv is None:
        _, q1, s1 = ops.layernorm_fwd_q8(x, P["ln1_w"], P["ln1_b"], cfg["eps"], want_bf16=False)
        qkv = _lin8(q1, s1, P, "in")
        del q1
    if a is None:
        a, stats = _attn_fwd(qkv, cfg, True)
    if x1 is None:
        qa, sa = ops.quantize_rows(a)
        x1 = _lin8(qa, sa, P, "out", epi=ops.EPI_ADD, aux=x)
        del qa
    if hpre is None:
        _, q2, s2 = ops.layernorm_fwd_q8(x1, P["ln2_w"], P["ln2_b"], cfg["eps"], want_bf16=False)
        hpre = _lin8(q2, s2, P, "fc")          # bf16(LN2(x1) W^T + b): what the activation epilogue's second output holds
        del q2
    return (h1, qkv, a, stats, x1, h2, hpre, g)


# The LayerNorm-1 backward of block i writes dx = the gradient block i-1 receives - and, in fp8 mode, that gradient's row-quantised
# form with its column sums (ops.layernorm_bwd(q8_fmt=...)): block i-1's backward starts from exactly that (its c_proj products
# and bias gradient).  The autograd edge between two ResBlockFn nodes carries one tensor, so the operand travels beside it: one
# slot, taken (and cleared) by the next fp8 block backward that runs.  The slot holds dx itself, so its memory cannot be
# recycled while the offer stands: a gradient that arrives with dx's address, shape and version IS dx (a summed or copied
# gradient is another allocation and is quantised the ordinary way).
_Q8_HANDOFF = []


def _q8_offer(dx, q8, fmt):
    _Q8_HANDOFF[:] = [(dx, dx._version, fmt, q8)]


def _q8_take(dy, fmt):
    slot = _Q8_HANDOFF[:]
    del _Q8_HANDOFF[:]
    if slot:
        dx, version, f, q8 = slot[0]
        if f == fmt and dy.data_ptr() == dx.data_ptr() and dy.shape == dx.shape and dy.stride() == dx.stride() \
                and dy.dtype == dx.dtype and dy._version == version:      # (the version at the offer: an in-place hook would move it)
            return q8
    return None


def _block_backward_fp8(x, dy, box, P, cfg, scales):
    """Backward of a block in fp8 mode: every matrix product - input gradients AND (round 6) weight gradients - on
    v_mfma_f32_16x16x128_f8f6f4.  Each incoming gradient is quantised once per token (its column sums = the bias gradient ride
    the same pass) and feeds both products of its layer; the activation operand of a weight gradient is emitted as e4m3 by the
    kernel that re-materialises it anyway (LayerNorm, GELU) with the gradient's token scale folded in (_wgrad8)."""
    act, eps = cfg["act"], cfg["eps"]
    h1, qkv, a, stats, x1, h2, hpre, g = box.pop()
    s1, sa, s2, sg = scales
    dy = dy.contiguous()
    fmt = cfg.get("fp8_grad_fmt", ops.FMT_E4M3)
    # y = x1 + c_proj(gelu(hpre))
    dq, ds, d_b_proj, rn = _q8_take(dy, fmt) or ops.quantize_rows(dy, fmt, want_colsum=True, want_rownorm=True)
    # (round 6) the GELU-backward epilogue of the input-gradient GEMM reads the kept e4m3 pre-activation anyway: it also emits the
    # activation operand of this layer's weight gradient (ops.gemm_nt_f8_emit) - the weight gradient then follows the input gradient
    fuse_emit = g is None and hpre is not None and hpre.dtype == torch.uint8 and not (fmt == ops.FMT_E4M3 and cfg.get("fp8_predict"))
    if not fuse_emit:
        emit_g = (lambda r, t: ops.scale_quantize_rows(g, r, t)) if g is not None else (lambda r, t: ops.scale_quantize_rows(hpre, r, t, act=act))
        d_w_proj = _wgrad8(dq, ds, sg, emit_g, P["dt_w_proj"], cfg)
    del g
    if fuse_emit:
        t = ops.rowscale_max(ds, sg)
        wq, ws = P["wt8_proj"]
        dh, x8 = ops.gemm_nt_f8_emit(dq, ds, wq, ws, hpre, t, act=act, fmt_a=fmt)
        d_w_proj = ops.gemm_tn_f8(dq, x8, t=t, fmt_p=fmt, out_dtype=P["dt_w_proj"])
        del x8
        dq, ds, d_b_fc = _gradq8(dh, cfg)
        del dh
    elif fmt == ops.FMT_E4M3 and cfg.get("fp8_predict"):
        # the gradient of the pre-activation leaves the GEMM as the e4m3 operand of the next two products, with its column sums
        # (the bias gradient of c_fc) from the same epilogue; row scale predicted as in the forward: |dh[m,c]| <=
        # ||dy[m,:]|| * max_c ||W_proj[:,c]|| * max gelu' (1.13), times 1.13 for the operands' rounding
        sdh, sodh = ops.row_bound(rn, P["wn_projT"], None, 1.13 * 1.13)
        dq, d_b_fc = _dlin8(dq, ds, P, "proj", cfg, epi=ops.EPI_DACT, act=act, aux=hpre, out_scale=sodh, want_colsum=True)
        ds = sdh
    else:   # e5m2 gradient bytes: bf16 gradient + the row quantiser
        dh = _dlin8(dq, ds, P, "proj", cfg, epi=ops.EPI_DACT, act=act, aux=hpre)     # [M,4D]
        dq, ds, d_b_fc = _gradq8(dh, cfg)
        del dh
    del hpre
    emit_h2 = (lambda r, t: ops.scale_quantize_rows(h2, r, t)) if h2 is not None else \
        (lambda r, t: ops.layernorm_fwd_q8s(x1, P["ln2_w"], P["ln2_b"], r, t, eps))
    d_w_fc = _wgrad8(dq, ds, s2, emit_h2, P["dt_w_fc"], cfg)
    dh2 = _dlin8(dq, ds, P, "fc", cfg)                                            # [M,D]
    del dq, ds, h2
    # (round 6) the two LayerNorm backwards hand the next product its fp8 operand themselves: the rows are in registers
    dx1, d_ln2_w, d_ln2_b, (dq, ds, d_b_out) = ops.layernorm_bwd(x1, P["ln2_w"], dh2, dres=dy, eps=eps, q8_fmt=fmt)
    del dh2, x1
    # x1 = x + out_proj(a)
    d_w_out = _wgrad8(dq, ds, sa, lambda r, t: ops.scale_quantize_rows(a, r, t), P["dt_w_out"], cfg)
    da = _dlin8(dq, ds, P, "out", cfg)
    del dq, ds
    dqkv = _attn_bwd(qkv, a, da, stats, cfg)
    del da, a, qkv, stats
    dq, ds, d_b_in = _gradq8(dqkv, cfg)
    del dqkv
    emit_h1 = (lambda r, t: ops.scale_quantize_rows(h1, r, t)) if h1 is not None else \
        (lambda r, t: ops.layernorm_fwd_q8s(x, P["ln1_w"], P["ln1_b"], r, t, eps))
    d_w_in = _wgrad8(dq, ds, s1, emit_h1, P["dt_w_in"], cfg)
    dh1 = _dlin8(dq, ds, P, "in", cfg)
    del dq, ds, h1
    if cfg.get("q8_handoff"):      # the block in front of this one runs the same engine: its backward starts from (q, dq, colsum, rownorm)
        dx, d_ln1_w, d_ln1_b, q8 = ops.layernorm_bwd(x, P["ln1_w"], dh1, dres=dx1, eps=eps, q8_fmt=fmt, want_rownorm=True)
        _q8_offer(dx, q8, fmt)
    else:
        dx, d_ln1_w, d_ln1_b = ops.layernorm_bwd(x, P["ln1_w"], dh1, dres=dx1, eps=eps)
    return dx, (d_ln1_w, d_ln1_b, d_w_in, d_b_in, d_w_out, d_b_out, d_ln2_w, d_ln2_b, d_w_fc, d_b_fc, d_w_proj,
                d_b_proj)


def _block_backward(x, dy, box, P, cfg):
    """box: one-element list holding the intermediates tuple (popped so they can be freed early).  bf16 engines (the fp8 engine:
    _block_backward_fp8)."""
    B, L, H, causal, act = cfg["B"], cfg["L"], cfg["H"], cfg["causal"], cfg["act"]
    h1, qkv, a, stats, x1, h2, hpre, g = box.pop()
    dy = dy.contiguous()
    # tensors the block did not keep are recomputed here, bit for bit as the forward produced them
    if qkv is None:
        h1 = ops.layernorm_fwd(x, P["ln1_w"], P["ln1_b"], cfg["eps"])
        qkv = ops.gemm_nt(h1, P["w_in"], P["b_in"])
    if a is None:
        a, stats = _attn_fwd(qkv, cfg, True)
    if x1 is None:
        x1 = ops.gemm_nt(a, P["w_out"], P["b_out"], epi=ops.EPI_ADD, aux=x)
    if hpre is None:     # "medium" block: LN2 + c_fc again (one GEMM instead of four + attention)
        h2 = ops.layernorm_fwd(x1, P["ln2_w"], P["ln2_b"], cfg["eps"])
        g, hpre = ops.gemm_nt(h2, P["w_fc"], P["b_fc"], epi=ops.EPI_ACT, act=act, want_pre=True)
    # "light8" block: the GELU-backward epilogue below reads the e4m3 bytes anyway and writes act(hpre) beside its own output
    # (round 6: one more store per chunk instead of an HBM-bound pass over the same bytes; bit for bit what activation_fwd writes)
    fuse_act = g is None and hpre is not None and hpre.dtype == torch.uint8
    if g is None and hpre is not None and not fuse_act:      # "light" block: cheap HBM-bound re-materialisation instead of 4 GEMMs + attention
        g = ops.activation_fwd(hpre, act)
    # LayerNorm outputs that only the weight gradients still need (h2 of a block that kept or re-materialised its MLP
    # intermediates, h1 of a block that kept qkv) are written by the LayerNorm BACKWARD pass, which has every row and its
    # statistics in registers anyway (ops.layernorm_bwd(beta=...): bit for bit ln_fwd's output) - one more store per row instead of
    # a second ln_fwd launch over the same rows; the weight gradient of c_fc / the in-projection then follows its LayerNorm
    # backward instead of preceding it.
    emit2, emit1 = h2 is None, h1 is None
    # y = x1 + c_proj(g).  The weight gradient goes first: g ([M, 4D], the largest transient of the block) is released before
    # the GELU-backward GEMM allocates its output of the same size
    if fuse_act:         # (here g and dh coexist: dy + g + dh = 1.65 GB more than the block's later peak dh + dh2 + dx1 + h2 + dy)
        dh, g = ops.gemm_nt(dy, P["wt_proj"], epi=ops.EPI_DACT, act=act, aux=hpre, want_act=True)
        del hpre
        d_w_proj, d_b_proj = ops.gemm_tn(dy, g, P["dt_w_proj"], want_colsum=True)
        del g
    else:
        d_w_proj, d_b_proj = ops.gemm_tn(dy, g, P["dt_w_proj"], want_colsum=True)
        del g
        dh = ops.gemm_nt(dy, P["wt_proj"], epi=ops.EPI_DACT, act=act, aux=hpre)     # [M,4D]
        del hpre
    dh2 = ops.gemm_nt(dh, P["wt_fc"])       # [M,D]
    if emit2:
        dx1, d_ln2_w, d_ln2_b, h2 = ops.layernorm_bwd(x1, P["ln2_w"], dh2, dres=dy, eps=cfg["eps"], beta=P["ln2_b"])
        del dh2, x1
        d_w_fc, d_b_fc = ops.gemm_tn(dh, h2, P["dt_w_fc"], want_colsum=True)
        del dh, h2
    else:
        d_w_fc, d_b_fc = ops.gemm_tn(dh, h2, P["dt_w_fc"], want_colsum=True)
        del dh, h2
        dx1, d_ln2_w, d_ln2_b = ops.layernorm_bwd(x1, P["ln2_w"], dh2, dres=dy, eps=cfg["eps"])
        del dh2, x1
    # x1 = x + out_proj(a)
    da = ops.gemm_nt(dx1, P["wt_out"])
    d_w_out, d_b_out = ops.gemm_tn(dx1, a, P["dt_w_out"], want_colsum=True)
    dqkv = _attn_bwd(qkv, a, da, stats, cfg)
    del da, a, qkv, stats
    dh1 = ops.gemm_nt(dqkv, P["wt_in"])
    if emit1:
        dx, d_ln1_w, d_ln1_b, h1 = ops.layernorm_bwd(x, P["ln1_w"], dh1, dres=dx1, eps=cfg["eps"], beta=P["ln1_b"])
        d_w_in, d_b_in = ops.gemm_tn(dqkv, h1, P["dt_w_in"], want_colsum=True)
        del dqkv, h1
    else:
        d_w_in, d_b_in = ops.gemm_tn(dqkv, h1, P["dt_w_in"], want_colsum=True)
        del dqkv, h1
        dx, d_ln1_w, d_ln1_b = ops.layernorm_bwd(x, P["ln1_w"], dh1, dres=dx1, eps=cfg["eps"])
    return dx, (d_ln1_w, d_ln1_b, d_w_in, d_b_in, d_w_out, d_b_out, d_ln2_w, d_ln2_b, d_w_fc, d_b_fc, d_w_proj,
                d_b_proj)


BLOCK_PARAM_ORDER = ("ln1_w", "ln1_b", "w_in", "b_in", "w_out", "b_out", "ln2_w", "ln2_b", "w_fc", "b_fc", "w_proj",
                     "b_proj")


def _block_operands(params, cache, fp8=False, fp8_names=("in", "out", "fc", "proj")):
    ln1_w, ln1_b, w_in, b_in, w_out, b_out, ln2_w, ln2_b, w_fc, b_fc, w_proj, b_proj = params
    P = {
        "ln1_w": cache.f32(ln1_w), "ln1_b": cache.f32(ln1_b), "ln2_w": cache.f32(ln2_w), "ln2_b": cache.f32(ln2_b),
        "w_in": cache.w(w_in), "w_out": cache.w(w_out), "w_fc": cache.w(w_fc), "w_proj": cache.w(w_proj),
        "wt_in": cache.wt(w_in), "wt_out": cache.wt(w_out), "wt_fc": cache.wt(w_fc), "wt_proj": cache.wt(w_proj),
        "b_in": cache.f32(b_in), "b_out": cache.f32(b_out), "b_fc": cache.f32(b_fc), "b_proj": cache.f32(b_proj),
        "dt_w_in": w_in.dtype, "dt_w_out": w_out.dtype, "dt_w_fc": w_fc.dtype, "dt_w_proj": w_proj.dtype,
    }
    if fp8:   # e4m3 copies, one scale per output channel of the GEMM they feed (rows of W forward, rows of W^T backward); only
        # of the layers that run on the fp8 path (LastBlockFn: the in-projection alone - its B pooled rows take the bf16 GEMMs)
        for name, w in (("in", w_in), ("out", w_out), ("fc", w_fc), ("proj", w_proj)):
            if name not in fp8_names:
                continue
            P["w8_" + name] = cache.custom(w, "w8", lambda t, w=w: ops.quantize_rows(cache.w(w)))
            P["wt8_" + name] = cache.custom(w, "wt8", lambda t, w=w: ops.quantize_rows(cache.wt(w)))
        P["one"] = cache.custom(w_in, "one", lambda t: torch.ones(1, device=t.device, dtype=f32))
        if "fc" in fp8_names:
            # the two scalars per layer behind the PREDICTED row scales of the MLP's 4 D-wide products (_row_bound): the largest
            # row norm of c_fc.weight and |c_fc.bias| bound the pre-activation, the largest row norm of c_proj.weight^T the
            # gradient that comes back through c_proj; device scalars, refreshed with the operand copies once per optimizer step
            P["wn_fc"] = cache.custom(w_fc, "wn", lambda t: ops.rownorm_max(cache.w(w_fc)))
            P["bmax_fc"] = cache.custom(b_fc, "bmax", lambda t: ops.absmax(cache.f32(b_fc)))
            P["wn_projT"] = cache.custom(w_proj, "wnT", lambda t: ops.rownorm_max(cache.wt(w_proj)))
    return P


class ResBlockFn(torch.autograd.Function):
    """x + MHA(LN1(x)); then + MLP(LN2(.)) on a [M, D] bf16 token matrix (transformer.py:238-250)."""

    @staticmethod
    def forward(ctx, x, cfg, cache, *params):
        P = _block_operands(params, cache, bool(cfg.get("fp8")))
        needs_grad = any(ctx.needs_input_grad)
        keep = False
        if needs_grad:
            keep = cfg.get("keep", "light") if (not cfg["recompute"] or cfg.get("keep_this", False)) else False
        ctx.scales = None
        if cfg.get("fp8"):
            y, inter, scales = _block_forward_fp8(x, P, cfg, keep)
            if needs_grad:
                ctx.scales = scales
        else:
            y, inter = _block_forward(x, P, cfg, keep)
        ctx.cfg, ctx.cache, ctx.params = cfg, cache, params
        if needs_grad:
            ctx.save_for_backward(x)
            ctx.inter = inter
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        cfg, params = ctx.cfg, ctx.params
        P = _block_operands(params, ctx.cache, bool(cfg.get("fp8")))
        box = [ctx.inter]
        ctx.inter = None
        if cfg.get("fp8"):
            scales, ctx.scales = ctx.scales, None
            box[0] = _fp8_fill(x, box[0] if box[0] is not None else (None,) * 8, P, cfg)
            dx, grads = _block_backward_fp8(x, dy, box, P, cfg, scales)
        else:
            if box[0] is None:
                box[0] = _block_forward(x, P, cfg, True, need_y=False)[1]
            dx, grads = _block_backward(x, dy, box, P, cfg)
        grads = tuple(_like_param(g, p) if p.requires_grad else None for g, p in zip(grads, params))
        return (dx, None, None) + grads


# ---------------------------------------------------------------------------------------------
class LastBlockFn(torch.autograd.Function):
    """The LAST residual block of a tower whose head reads ONE row per sample (the class token of a CLS-pooled image tower,
    transformer.py:509-529; the EOT token of the text tower, model.py:251-254): returns that row only, [B, D].

    Every other row of the last block's output is dead - nothing reads it, its gradient is exactly zero - so only the keys and
    values need all tokens: LN1, the in-projection and attention run on every row, the out-projection, LN2 and the MLP on the B
    pooled rows (the reference computes all B * L of them and the pooling discards them; same kind of identity as taking the
    pooled row before ln_post, SURVEY 8a identity 8).  fp8 engines (round 5): the token-level part - LN1, in-projection and its
    input gradient - runs on the fp8 path like every other block; the B pooled rows (0.5 % of a block's rows) run the bf16 GEMMs.  Forward values of the pooled rows and all gradients are those of the full
    block (zero rows add nothing to a weight gradient).  `rows`: int64 device tensor [B], row index of the pooled token of each
    sample in x.  The token-level part (qkv, attention output, softmax statistics) is kept or recomputed like any other block's;
    the B-row part is always kept (a few MB)."""

    @staticmethod
    def forward(ctx, x, rows, cfg, cache, *params):
        P = _block_operands(params, cache, bool(cfg.get("fp8")), fp8_names=("in",))
        needs_grad = any(ctx.needs_input_grad)
        # token-level tensors this block keeps: the block's keep set like any other block's ("qkv", "a" = attention output +
        # softmax statistics; the named tiers all hold both); x1 / the pre-activation exist for the B pooled rows only
        ks = frozenset()
        if needs_grad and (not cfg["recompute"] or cfg.get("keep_this", False)):
            keep = cfg.get("keep", "light")
            ks = KEEP_SETS.get(keep, keep) if isinstance(keep, (str, frozenset)) else KEEP_SETS["light"]
        s1_box = [None]
        qkv, a, stats = LastBlockFn._tokens(x, P, cfg, "a" in ks, box=s1_box)
        ctx.s1 = s1_box[0] if needs_grad else None
        a_c, x_c = ops.gather_rows(a, rows), ops.gather_rows(x, rows)
        x1 = ops.gemm_nt(a_c, P["w_out"], P["b_out"], epi=ops.EPI_ADD, aux=x_c)
        h2 = ops.layernorm_fwd(x1, P["ln2_w"], P["ln2_b"], cfg["eps"])
        g, hpre = ops.gemm_nt(h2, P["w_fc"], P["b_fc"], epi=ops.EPI_ACT, act=cfg["act"], want_pre=True)
        y = ops.gemm_nt(g, P["w_proj"], P["b_proj"], epi=ops.EPI_ADD, aux=x1)
        ctx.cfg, ctx.cache, ctx.params = cfg, cache, params
        if needs_grad:
            ctx.save_for_backward(x, rows)
            ctx.tokens = (qkv if "qkv" in ks else None, a if "a" in ks else None, stats if "a" in ks else None)
            ctx.small = (a_c, x1, h2, hpre, g)
        return y

    @staticmethod
    def _qkv(x, P, cfg, box=None):
        """-> qkv; fp8 engines leave the LayerNorm output's per-token scale in box[0] (the weight gradient's tensor scale, _wgrad8)."""
        if cfg.get("fp8"):
            _, q1, s1 = ops.layernorm_fwd_q8(x, P["ln1_w"], P["ln1_b"], cfg["eps"], want_bf16=False)
            if box is not None:
                box[0] = s1
            return _lin8(q1, s1, P, "in")
        return ops.gemm_nt(ops.layernorm_fwd(x, P["ln1_w"], P["ln1_b"], cfg["eps"]), P["w_in"], P["b_in"])

    @staticmethod
    def _tokens(x, P, cfg, want_stats, qkv=None, box=None):
        if qkv is None:
            qkv = LastBlockFn._qkv(x, P, cfg, box)
        a, stats = _attn_fwd(qkv, cfg, bool(want_stats))
        return qkv, a, stats

    @staticmethod
    def backward(ctx, dy):
        x, rows = ctx.saved_tensors
        cfg, params = ctx.cfg, ctx.params
        P = _block_operands(params, ctx.cache, bool(cfg.get("fp8")), fp8_names=("in",))
        a_c, x1, h2, hpre, g = ctx.small
        ctx.small = None
        dy = dy.contiguous()
        M = x.shape[0]
        dh = ops.gemm_nt(dy, P["wt_proj"], epi=ops.EPI_DACT, act=cfg["act"], aux=hpre)
        d_w_proj, d_b_proj = ops.gemm_tn(dy, g, P["dt_w_proj"], want_colsum=True)
        dh2 = ops.gemm_nt(dh, P["wt_fc"])
        d_w_fc, d_b_fc = ops.gemm_tn(dh, h2, P["dt_w_fc"], want_colsum=True)
        dx1, d_ln2_w, d_ln2_b = ops.layernorm_bwd(x1, P["ln2_w"], dh2, dres=dy, eps=cfg["eps"])
        da_c = ops.gemm_nt(dx1, P["wt_out"])
        d_w_out, d_b_out = ops.gemm_tn(dx1, a_c, P["dt_w_out"], want_colsum=True)
        (qkv, a, stats), ctx.tokens = ctx.tokens, None
        if qkv is None or a is None:      # whatever the block did not keep is recomputed bit for bit
            qkv, a2, stats2 = LastBlockFn._tokens(x, P, cfg, True, qkv=qkv) if a is None else (LastBlockFn._qkv(x, P, cfg), a, stats)
            a, stats = a2, stats2
        dqkv = _attn_bwd(qkv, a, ops.scatter_rows(da_c, rows, M), stats, cfg)
        del qkv, a, stats
        if cfg.get("fp8"):      # token-level products of the block on the fp8 path: input and weight gradient of the in-projection
            s1, ctx.s1 = ctx.s1, None
            dq, ds, d_b_in = _gradq8(dqkv, cfg)
            del dqkv
            d_w_in = _wgrad8(dq, ds, s1, lambda r, t: ops.layernorm_fwd_q8s(x, P["ln1_w"], P["ln1_b"], r, t, cfg["eps"]),
                             P["dt_w_in"], cfg)
            dh1 = _dlin8(dq, ds, P, "in", cfg)
            del dq, ds
        else:
            dh1 = ops.gemm_nt(dqkv, P["wt_in"])
            h1 = ops.layernorm_fwd(x, P["ln1_w"], P["ln1_b"], cfg["eps"])
            d_w_in, d_b_in = ops.gemm_tn(dqkv, h1, P["dt_w_in"], want_colsum=True)
            del dqkv, h1
        # x1 = x[rows] + out_proj(a[rows]): the residual gradient reaches x at the pooled rows only
        dx, d_ln1_w, d_ln1_b = ops.layernorm_bwd(x, P["ln1_w"], dh1, dres=ops.scatter_rows(dx1, rows, M), eps=cfg["eps"])
        grads = (d_ln1_w, d_ln1_b, d_w_in, d_b_in, d_w_out, d_b_out, d_ln2_w, d_ln2_b, d_w_fc, d_b_fc, d_w_proj, d_b_proj)
        grads = tuple(_like_param(gr, p) if p.requires_grad else None for gr, p in zip(grads, params))
        return (dx, None, None, None) + grads


# ---------------------------------------------------------------------------------------------
def _conv_weight_operand(w, Kp):
    """conv1.weight [D,3,P,P] -> bf16 [D,Kp] in (ph,pw,c) element order, zero padded to Kp."""
    D = w.shape[0]
    k = w.shape[1] * w.shape[2] * w.shape[3]
    m = w.permute(0, 2, 3, 1).reshape(D, k)
    if Kp != k:
        m = torch.nn.functional.pad(m, (0, Kp - k))
    return ops.to_bf16(m.contiguous())


def _vision_stem_forward(image, P, cfg):
    patches = ops.patchify(image, cfg["P"], cfg["Kp"], cfg["mean"], cfg["std"])
    pe = ops.gemm_nt(patches, P["w_conv"])
    tok = ops.assemble_tokens(pe, P["cls"], P["pos"], cfg["B"], cfg["L"])
    x0 = ops.layernorm_fwd(tok, P["ln_w"], P["ln_b"], cfg["eps"]) if cfg["ln_pre"] else tok
    return patches, tok, x0


class VisionStemFn(torch.autograd.Function):
    """uint8/float image -> normalise -> patch GEMM -> [cls; patches] + pos -> ln_pre
    (train.py:191-197 + transformer.py:480-503)."""

    @staticmethod
    def forward(ctx, image, cfg, cache, conv_w, cls, pos, ln_w, ln_b):
        P = VisionStemFn._operands(cfg, cache, conv_w, cls, pos, ln_w, ln_b)
        _, _, x0 = _vision_stem_forward(image, P, cfg)
        ctx.cfg, ctx.cache = cfg, cache
        ctx.params = (conv_w, cls, pos, ln_w, ln_b)
        ctx.save_for_backward(image)
        return x0

    @staticmethod
    def _operands(cfg, cache, conv_w, cls, pos, ln_w, ln_b):
        Kp = cfg["Kp"]
        P = {"w_conv": cache.custom(conv_w, "conv%d" % Kp, lambda t: _conv_weight_operand(t, Kp)),
             "cls": cache.f32(cls), "pos": cache.f32(pos)}
        if cfg["ln_pre"]:
            P["ln_w"], P["ln_b"] = cache.f32(ln_w), cache.f32(ln_b)
        return P

    @staticmethod
    def backward(ctx, dx0):
        (image,) = ctx.saved_tensors
        cfg = ctx.cfg
        conv_w, cls, pos, ln_w, ln_b = ctx.params
        P = VisionStemFn._operands(cfg, ctx.cache, conv_w, cls, pos, ln_w, ln_b)
        patches, tok, _ = _vision_stem_forward(image, P, cfg)
        dx0 = dx0.contiguous()
        d_ln_w = d_ln_b = None
        if cfg["ln_pre"]:
            dtok, d_ln_w, d_ln_b = ops.layernorm_bwd(tok, P["ln_w"], dx0, eps=cfg["eps"])
        else:
            dtok = dx0
        dpatch, dcls, dpos = ops.assemble_tokens_bwd(dtok, cfg["B"], cfg["L"], need_pos=pos.requires_grad)
        d_conv = None
        if conv_w.requires_grad:
            D, C, Pp, _ = conv_w.shape
            k = C * Pp * Pp
            dw = ops.gemm_tn(dpatch, patches, f32)[:, :k]                     # [D, (ph,pw,c)]
            d_conv = dw.reshape(D, Pp, Pp, C).permute(0, 3, 1, 2).contiguous()
            d_conv = _like_param(d_conv, conv_w)
        return (None, None, None, d_conv,
                _like_param(dcls, cls) if cls.requires_grad else None,
                _like_param(dpos, pos) if pos.requires_grad else None,
                _like_param(d_ln_w, ln_w) if (cfg["ln_pre"] and ln_w.requires_grad) else None,
                _like_param(d_ln_b, ln_b) if (cfg["ln_pre"] and ln_b.requires_grad) else None)


class TokenDropFn(torch.autograd.Function):
    """PatchDropout (transformer.py:53-83, applied at :501-502): keep the rows `rows` (int64, device) of the [B*L, D] token
    matrix - the class token of every sample and its randomly kept patches.  Forward = row gather, backward = row scatter
    into zeros.  (The reference drops BEFORE ln_pre; LayerNorm is per token, so dropping after it selects the same values.)"""

    @staticmethod
    def forward(ctx, x, rows):
        ctx.save_for_backward(rows)
        ctx.n_src = x.shape[0]
        return ops.gather_rows(x, rows)

    @staticmethod
    def backward(ctx, dy):
        (rows,) = ctx.saved_tensors
        return ops.scatter_rows(dy.contiguous(), rows, ctx.n_src), None


class TextStemFn(torch.autograd.Function):
    """token_embedding(text).to(bf16) + positional_embedding.to(bf16) (model.py:245-247)."""

    @staticmethod
    def forward(ctx, ids, cache, table, pos):
        x0 = ops.embed_tokens(ids, table.detach(), cache.f32(pos))
        ctx.save_for_backward(ids)
        ctx.params = (table, pos)
        return x0

    @staticmethod
    def backward(ctx, dx0):
        (ids,) = ctx.saved_tensors
        table, pos = ctx.params
        dtable, dpos = ops.embed_tokens_bwd(ids, dx0.contiguous(), table.shape[0], need_table=table.requires_grad,
                                            need_pos=pos.requires_grad)
        return (None, None,
                _like_param(dtable, table) if table.requires_grad else None,
                _like_param(dpos, pos) if pos.requires_grad else None)


class HeadFn(torch.autograd.Function):
    """pool -> LayerNorm -> @ proj, returning f32 features [B,E].  Pooling commutes with the per-token
    LayerNorm, so LN runs on B rows instead of B*L (SURVEY 8a identity 8)."""

    @staticmethod
    def forward(ctx, x, idx, cfg, cache, ln_w, ln_b, proj):
        B, L, mode = cfg["B"], cfg["L"], cfg["mode"]
        pooled = ops.pool_fwd(x, B, L, mode, idx)                            # f32 [B,D]
        y = ops.layernorm_fwd(pooled, cache.f32(ln_w), cache.f32(ln_b), cfg["eps"], out_dtype=bf16)
        feat = ops.gemm_nt(y, cache.wt(proj), out_f32=True) if proj is not None else ops.to_f32(y)
        ctx.cfg, ctx.cache, ctx.params = cfg, cache, (ln_w, ln_b, proj)
        ctx.save_for_backward(pooled, y, idx if idx is not None else torch.empty(0, device=x.device))
        ctx.has_idx = idx is not None
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        pooled, y, idx = ctx.saved_tensors
        idx = idx if ctx.has_idx else None
        cfg, cache = ctx.cfg, ctx.cache
        ln_w, ln_b, proj = ctx.params
        dfb = ops.to_bf16(dfeat.contiguous())
        d_proj = None
        if proj is not None:
            dy = ops.gemm_nt(dfb, cache.w(proj))                              # [B,E] @ proj[D,E]^T -> [B,D]
            if proj.requires_grad:
                d_proj = _like_param(ops.gemm_tn(y, dfb, f32), proj)          # [D,E]
        else:
            dy = dfb
        dpooled, d_ln_w, d_ln_b = ops.layernorm_bwd(pooled, cache.f32(ln_w), dy, eps=cfg["eps"])
        dx = ops.pool_bwd(dpooled, cfg["B"], cfg["L"], cfg["mode"], idx)
        return (dx, None, None, None,
                _like_param(d_ln_w, ln_w) if ln_w.requires_grad else None,
                _like_param(d_ln_b, ln_b) if ln_b.requires_grad else None,
                d_proj)


class L2NormFn(torch.autograd.Function):
    """F.normalize(x, dim=-1), eps 1e-12 (model.py:240,263)."""

    @staticmethod
    def forward(ctx, x):
        y, _, inv = ops.l2norm_fwd(x.contiguous())
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        return ops.l2norm_bwd(y, inv, dy.contiguous())
