"""ORACLE - TEST INFRASTRUCTURE ONLY.  CPU restatement of the clipa_torch hot path.

Plain-torch (CPU, fp32/fp64) functional restatement of the reference's CLIP training step, written
from the maths (SURVEY.md 8a identities 1-10), NOT by importing the reference:

  encode_image   open_clip/transformer.py:480-534 (VisionTransformer.forward)
  resblock       open_clip/transformer.py:238-250 + torch nn.MultiheadAttention semantics
  encode_text    open_clip/model.py:242-263
  clip_forward   open_clip/model.py:265-274
  clip_loss      open_clip/loss.py:115-157 incl. the gather variants of loss.py:73-87

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (clipa_amd/) never does.  PINNING: tests/test_oracle_cpu.py checks this restatement against
tests/golden/*.npz, which oracle/make_golden.py produced by running the REAL reference modules
(imported from /root/reference in the build container).  The reference itself ships no tests or golden
vectors, so those generated fixtures are the pin.

`emulate_bf16=True` rounds to bfloat16 at the points where the HIP engine stores bf16 (weights, GEMM
outputs, LN outputs, attention probabilities, residual stream) so kernel-level comparisons can use a
tolerance far below bf16 noise.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)


def _q(t, on):
    """bf16 round-trip when emulating the engine's storage precision."""
    return t.to(torch.bfloat16).to(t.dtype) if on else t


def layer_norm(x, w, b, eps=1e-5):
    """transformer.py:19-34: biased variance, eps inside the sqrt."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def activation(x, kind):
    """nn.GELU(approximate='none'|'tanh') (model.py:128-129) / QuickGELU (transformer.py:37-40)."""
    if kind == "gelu_erf":
        return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))
    if kind == "gelu_tanh":
        return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))
    if kind == "quick_gelu":
        return x * torch.sigmoid(1.702 * x)
    raise ValueError(kind)


def attention(qkv, heads, causal, emulate_bf16=False):
    """qkv [B,L,3D] -> [B,L,D]. q,k,v = consecutive thirds; head h = columns [h*dh,(h+1)*dh);
    softmax(q k^T / sqrt(dh) + triu(1)*-inf) v (transformer.py:223-236, 618-624)."""
    B, L, D3 = qkv.shape
    D = D3 // 3
    dh = D // heads
    q, k, v = qkv.split(D, dim=-1)
    q = q.reshape(B, L, heads, dh).transpose(1, 2)
    k = k.reshape(B, L, heads, dh).transpose(1, 2)
    v = v.reshape(B, L, heads, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    if causal:
        s = s + torch.full((L, L), float("-inf"), dtype=s.dtype).triu(1)
    p = torch.softmax(s, dim=-1)
    if emulate_bf16:   # the engine multiplies un-normalised bf16 exp() by V and divides afterwards
        m = s.max(dim=-1, keepdim=True).values
        e = torch.exp(s - m)
        o = (_q(e, True) @ v) / e.sum(-1, keepdim=True)
    else:
        o = p @ v
    return o.transpose(1, 2).reshape(B, L, D)


def resblock(x, sd, pre, heads, causal, act, emulate_bf16=False):
    """x + out_proj(attn(in_proj(ln_1 x))); then + c_proj(act(c_fc(ln_2 .))) (transformer.py:238-250)."""
    e = emulate_bf16
    W = lambda n: _q(sd[pre + n], e)
    h = _q(layer_norm(x, sd[pre + "ln_1.weight"], sd[pre + "ln_1.bias"]), e)
    qkv = _q(h @ W("attn.in_proj_weight").T + sd[pre + "attn.in_proj_bias"], e)
    a = _q(attention(qkv, heads, causal, e), e)
    x = _q(_q(a @ W("attn.out_proj.weight").T + sd[pre + "attn.out_proj.bias"], e) + x, e)
    h = _q(layer_norm(x, sd[pre + "ln_2.weight"], sd[pre + "ln_2.bias"]), e)
    hp = _q(h @ W("mlp.c_fc.weight").T + sd[pre + "mlp.c_fc.bias"], e)
    g = _q(activation(hp, act), e)
    x = _q(_q(g @ W("mlp.c_proj.weight").T + sd[pre + "mlp.c_proj.bias"], e) + x, e)
    return x


def _n_layers(sd, pre):
    n = 0
    while f"{pre}resblocks.{n}.ln_1.weight" in sd:
        n += 1
    return n


def normalize_images(images_u8, mean=OPENAI_DATASET_MEAN, std=OPENAI_DATASET_STD, dtype=torch.float32):
    """training/train.py:191-197."""
    x = images_u8.to(dtype) / 255.0
    m = torch.tensor(mean, dtype=dtype).view(1, 3, 1, 1)
    s = torch.tensor(std, dtype=dtype).view(1, 3, 1, 1)
    return (x - m) / s


def encode_image(sd, cfg, image, emulate_bf16=False, patch_keep=None):
    """image [B,3,S,S] float (already normalised) -> un-normalised features [B,E].  patch_keep: int64 [B, K] kept patch
    indices of PatchDropout in training mode (transformer.py:53-83, applied at :501-502) or None."""
    e = emulate_bf16
    v = cfg["vision"]
    P, D, H = v["patch_size"], v["width"], v["heads"]
    B, _, S, _ = image.shape
    g = S // P
    # patch embed == GEMM over (c, ph, pw)-flattened VALID patches (identity 1)
    x = image[:, :, :g * P, :g * P].reshape(B, 3, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * P * P)
    x = _q(_q(x, e) @ _q(sd["visual.conv1.weight"], e).reshape(D, -1).T, e)
    cls = _q(sd["visual.class_embedding"], e).expand(B, 1, D)
    x = torch.cat([cls, x], dim=1)
    x = _q(x + _q(sd["visual.positional_embedding"], e), e)
    if patch_keep is not None:       # transformer.py:67-83: the class token is excluded from the draw and put back in front
        cls_tokens, xp = x[:, :1], x[:, 1:]
        xp = xp[torch.arange(B)[..., None], patch_keep]
        x = torch.cat((cls_tokens, xp), dim=1)
    if v.get("ln_pre", True):
        x = _q(layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"]), e)
    for i in range(_n_layers(sd, "visual.transformer.")):
        x = resblock(x, sd, f"visual.transformer.resblocks.{i}.", H, False, v["act"], e)
    style = v.get("pool_style", "open_clip")
    if style == "big_vision_gap":
        pooled = x[:, 1:].mean(1)
    elif style == "big_vision_tok" or not v.get("global_average_pool", False):
        pooled = x[:, 0]
    else:
        pooled = x.mean(1)
    pooled = _q(layer_norm(pooled, sd["visual.ln_post.weight"], sd["visual.ln_post.bias"]), e)
    return pooled @ _q(sd["visual.proj"], e)


def encode_text(sd, cfg, text, emulate_bf16=False):
    """token ids [B,ctx] int64 -> un-normalised features [B,E] (model.py:242-263)."""
    e = emulate_bf16
    t = cfg["text"]
    x = _q(_q(sd["token_embedding.weight"], e)[text] + _q(sd["positional_embedding"], e), e)
    for i in range(_n_layers(sd, "transformer.")):
        x = resblock(x, sd, f"transformer.resblocks.{i}.", t["heads"], t.get("attention_mask", True), t["act"], e)
    style = t.get("pool_style", "open_clip")
    if style == "open_clip":
        pooled = x[torch.arange(x.shape[0]), text.argmax(dim=-1)]   # LN is per-row: pick first (identity 8)
    elif style == "big_vision_tok":
        pooled = x[:, 0]
    else:
        pooled = x[:, -1]
    pooled = _q(layer_norm(pooled, sd["ln_final.weight"], sd["ln_final.bias"]), e)
    return pooled @ _q(sd["text_projection"], e)


def l2_normalize(x, eps=1e-12):
    return x / x.norm(dim=-1, keepdim=True).clamp_min(eps)


def clip_forward(sd, cfg, image, text, emulate_bf16=False, patch_keep=None):
    i = l2_normalize(encode_image(sd, cfg, image, emulate_bf16, patch_keep))
    t = l2_normalize(encode_text(sd, cfg, text, emulate_bf16))
    return i, t, sd["logit_scale"].exp()


def clip_loss(image_features, text_features, logit_scale, emulate_bf16=False):
    """Single-process (world_size 1) InfoNCE: (CE(s I T^T, arange) + CE(s T I^T, arange)) / 2."""
    e = emulate_bf16
    i, t = _q(image_features, e), _q(text_features, e)
    logits_i = logit_scale * i @ t.T
    logits_t = logit_scale * t @ i.T
    labels = torch.arange(i.shape[0])
    return (F.cross_entropy(logits_i, labels) + F.cross_entropy(logits_t, labels)) / 2, logits_i


def clip_loss_rank(img_all, txt_all, logit_scale, rank, world_size, local_loss=True):
    """Loss seen by `rank` when every rank holds B rows of the concatenated [W*B, E] features
    (loss.py:118-120,128-144): local_loss -> local rows x all columns with labels offset by B*rank."""
    B = img_all.shape[0] // world_size
    if local_loss:
        i_loc, t_loc = img_all[rank * B:(rank + 1) * B], txt_all[rank * B:(rank + 1) * B]
        li = logit_scale * i_loc @ txt_all.T
        lt = logit_scale * t_loc @ img_all.T
        labels = torch.arange(B) + B * rank
    else:
        li = logit_scale * img_all @ txt_all.T
        lt = li.T
        labels = torch.arange(img_all.shape[0])
    return (F.cross_entropy(li, labels) + F.cross_entropy(lt, labels)) / 2


# ---- helpers shared by the golden generator and the tests --------------------------------------
def act_name(quick_gelu, approximate):
    return "quick_gelu" if quick_gelu else ("gelu_tanh" if approximate == "tanh" else "gelu_erf")


def oracle_cfg(model_cfg):
    """open_clip-style JSON dict -> the small cfg dict used above."""
    v, t = dict(model_cfg["vision_cfg"]), dict(model_cfg["text_cfg"])
    qg = model_cfg.get("quick_gelu", False)
    return {
        "vision": {"patch_size": v["patch_size"], "width": v["width"], "heads": v["width"] // v.get("head_width", 64),
                   "act": act_name(qg, v.get("gelu_approximate", "none")), "ln_pre": v.get("ln_pre", True),
                   "pool_style": v.get("pool_style", "open_clip"),
                   "global_average_pool": v.get("global_average_pool", False)},
        "text": {"heads": t["heads"], "act": act_name(qg, t.get("gelu_approximate", "none")),
                 "pool_style": t.get("pool_style", "open_clip"), "attention_mask": t.get("attention_mask", True)},
    }


def make_state_dict(shapes, seed, frozen=()):
    """Deterministic, platform-independent weights (numpy MT19937): matrices ~N(0, 0.05^2 .. ), LN weights
    ~1 + 0.1 N, biases / vectors ~0.1 N, logit_scale = ln(1/0.07). `shapes`: ordered {name: shape}."""
    rng = np.random.RandomState(seed)
    sd = {}
    for name, shape in shapes.items():
        shape = tuple(shape)
        if name == "logit_scale":
            arr = np.full(shape, math.log(1 / 0.07))
        elif name in frozen:
            arr = None
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            arr = rng.standard_normal(shape) * min(0.08, 1.5 / math.sqrt(fan_in))
        elif (".ln_" in name or name.startswith("ln_") or "ln_p" in name) and name.endswith("weight"):
            arr = 1.0 + 0.1 * rng.standard_normal(shape)
        else:
            arr = 0.1 * rng.standard_normal(shape)
        if arr is not None:
            sd[name] = torch.from_numpy(np.asarray(arr, dtype=np.float32))
    return sd


def synthetic_batch(B, S, ctx, vocab, seed):
    """LAION-shaped synthetic batch: uint8 images, [SOT, k ids, EOT, 0...] token rows with the
    EOT (= vocab-1) as the row maximum."""
    rng = np.random.RandomState(seed)
    images = torch.from_numpy(rng.randint(0, 256, size=(B, 3, S, S), dtype=np.uint8))
    texts = np.zeros((B, ctx), dtype=np.int64)
    sot, eot = vocab - 2, vocab - 1
    for b in range(B):
        n = int(np.clip(round(rng.normal(min(20, ctx * 0.6), 8)), 3, ctx))
        texts[b, 0] = sot
        texts[b, 1:n - 1] = rng.randint(1, vocab - 2, size=n - 2)
        texts[b, n - 1] = eot
    return images, torch.from_numpy(texts)
