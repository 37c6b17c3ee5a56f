"""CLIP model module with the open_clip surface, computed by the MI355X HIP engine.

Drop-in for `open_clip.model.CLIP` (clipa_torch/open_clip/model.py:200-274): same constructor
arguments, same attributes used by clipa_torch/training (visual.image_size/.image_mean/.image_std,
logit_scale, encode_image/encode_text, set_grad_checkpointing, lock_image_tower, output_dict) and -
because parameters are held by the very same torch.nn containers the reference instantiates
(nn.Conv2d, nn.LayerNorm, nn.MultiheadAttention, nn.Linear, nn.Embedding) - the same state_dict
keys, shapes, dtypes and default initialisation, so checkpoints interchange (main.py:438-455).
Those containers are parameter holders only: their torch forward is never called; every FLOP of
forward and backward runs in libclipa_hip.so via clipa_amd.engine.
"""
import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Optional, Tuple, Union

import os

import numpy as np
import torch
from torch import nn

from . import engine, ops

OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)   # open_clip/constants.py:1
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)   # open_clip/constants.py:2


@dataclass
class CLIPVisionCfg:   # field names / defaults of open_clip/model.py:25-51 (ViT subset)
    layers: Union[Tuple[int, int, int, int], int] = 12
    width: int = 768
    head_width: int = 64
    mlp_ratio: float = 4.0
    patch_size: int = 16
    image_size: Union[Tuple[int, int], int] = 224
    ls_init_value: Optional[float] = None
    patch_dropout: float = 0.
    input_patchnorm: bool = False
    global_average_pool: bool = False
    attentional_pool: bool = False
    n_queries: int = 256
    attn_pooler_heads: int = 8
    timm_model_name: str = None
    timm_model_pretrained: bool = False
    timm_pool: str = 'avg'
    timm_proj: str = 'linear'
    timm_proj_bias: bool = False
    timm_drop: float = 0.
    timm_drop_path: Optional[float] = None
    output_tokens: bool = False
    pos_embed: str = 'learnable'
    gelu_approximate: str = 'none'
    ln_pre: bool = True
    pool_style: str = 'open_clip'


@dataclass
class CLIPTextCfg:     # open_clip/model.py:54-75
    context_length: int = 77
    vocab_size: int = 49408
    width: int = 512
    heads: int = 8
    layers: int = 12
    ls_init_value: Optional[float] = None
    hf_model_name: str = None
    hf_tokenizer_name: str = None
    hf_model_pretrained: bool = True
    proj: str = 'mlp'
    pooler_type: str = 'mean_pooler'
    embed_cls: bool = False
    pad_id: int = 0
    output_tokens: bool = False
    text_mask: str = 'first'
    gelu_approximate: str = 'none'
    pool_style: str = 'open_clip'
    bert_tokenizer: bool = False
    vocab_path: str = None
    attention_mask: bool = True


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False):
    """MAE / MoCo-v3 fixed 2-D sin-cos table (open_clip/pos_embed.py:20-67): first half of the channels
    encodes the w coordinate, second half h; each half is [sin | cos] over 1/10000^(2i/d)."""
    assert embed_dim % 4 == 0
    quarter = embed_dim // 4
    omega = 1.0 / 10000 ** (np.arange(quarter, dtype=np.float64) / quarter)
    ys, xs = np.meshgrid(np.arange(grid_size, dtype=np.float32), np.arange(grid_size, dtype=np.float32), indexing="ij")

    def enc(pos):
        out = pos.reshape(-1).astype(np.float64)[:, None] * omega[None, :]
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)

    emb = np.concatenate([enc(xs), enc(ys)], axis=1)
    if cls_token:
        emb = np.concatenate([np.zeros([1, embed_dim]), emb], axis=0)
    return emb


def _act_code(quick_gelu, approximate):
    if quick_gelu:
        return ops.ACT_QUICK_GELU
    if approximate == 'tanh':
        return ops.ACT_GELU_TANH
    if approximate == 'none':
        return ops.ACT_GELU_ERF
    raise ValueError(f"unsupported gelu_approximate={approximate!r}")


def _unsupported(what):
    raise NotImplementedError(f"clipa_amd: {what} is outside the MI355X hot path (ViT-CLIP training step); "
                              "use the reference implementation for it")


class _ResBlockParams(nn.Module):
    """Parameter holder with the names of ResidualAttentionBlock (transformer.py:195-221)."""

    def __init__(self, d_model, n_head, mlp_ratio=4.0):
        super().__init__()
        self.ln_1 = nn.LayerNorm(d_model)
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ln_2 = nn.LayerNorm(d_model)
        mlp_width = int(d_model * mlp_ratio)
        self.mlp = nn.Sequential(OrderedDict([
            ("c_fc", nn.Linear(d_model, mlp_width)),
            ("c_proj", nn.Linear(mlp_width, d_model)),
        ]))

    def param_tuple(self):
        return (self.ln_1.weight, self.ln_1.bias, self.attn.in_proj_weight, self.attn.in_proj_bias,
                self.attn.out_proj.weight, self.attn.out_proj.bias, self.ln_2.weight, self.ln_2.bias,
                self.mlp.c_fc.weight, self.mlp.c_fc.bias, self.mlp.c_proj.weight, self.mlp.c_proj.bias)

    def forward(self, *a, **k):
        raise RuntimeError("parameter holder: computed by clipa_amd.engine.ResBlockFn")


class Transformer(nn.Module):
    """transformer.py:294-326: a stack of residual attention blocks over a [B*L, D] token matrix."""

    def __init__(self, width, layers, heads, mlp_ratio=4.0, act=ops.ACT_GELU_ERF):
        super().__init__()
        if width % heads != 0 or width // heads not in (64, 80, 88, 104, 112):
            _unsupported(f"head dim {width // heads if heads else '?'} (the fused attention kernels cover 64, 80 and - image "
                         f"towers only - 88 / 104 / 112)")
        self.width, self.layers, self.heads, self.act = width, layers, heads, act
        self.grad_checkpointing = False
        # MI355X engine knob (no reference counterpart): with grad checkpointing on, the first `keep_blocks`
        # blocks still keep their GEMM / attention outputs ("light" keep, ~18*D bytes per token) so their
        # backward skips the recompute; 288 GB of HBM usually has room for several blocks' worth.
        self.keep_blocks = 0
        # ... and the next `medium_blocks` blocks keep the same set minus the MLP pre-activation (~10*D bytes per
        # token) and re-run only LN2 + c_fc in backward.
        self.medium_blocks = 0
        # ... and `light8_blocks` blocks in between keep the light set with the MLP pre-activation as e4m3 bytes (~14*D bytes
        # per token): no GEMM is re-run, gelu'(h) and the re-materialised activation come from the rounded value
        # (engine._block_forward).  Order of the tiers along the depth: light, light8, medium, recompute.
        self.light8_blocks = 0
        # ... or tensor by tensor (what bench.py's planner uses; fp8 engines too since round 6): the LAST keep_counts[t] blocks keep tensor t (their
        # backward runs first, at the memory peak: the blocks that recompute do so after kept tensors have been released)
        # ("h8" e4m3 pre-activation - or "h", the same tensor in bf16: twice the bytes, bit-exact gradients -, "a" attention
        # output + statistics, "x1", "qkv"), on top of the named tiers above
        self.keep_counts = {"h8": 0, "h": 0, "a": 0, "x1": 0, "qkv": 0}
        # fp8 engine mode (create_model(precision="fp8"), BASELINE.json configs[3]): the four linear layers of every block run
        # forward and input-gradient GEMMs on e4m3 operands (engine._block_forward_fp8); "e5m2" switches the gradient operand
        self.fp8 = False
        self.fp8_grad_format = "e4m3"
        # fp8 engine knob (round 6, off): the two 4 D-wide MLP tensors (activation, pre-activation gradient) leave their GEMMs as
        # e4m3 operands with PREDICTED row scales (a Cauchy-Schwarz bound from the producer's row norm and the weight's largest
        # row norm) instead of bf16 + a row quantiser: ~5 % faster, per-tensor gradient cosines 0.001-0.009 lower
        self.fp8_predicted_scales = False
        self.resblocks = nn.ModuleList([_ResBlockParams(width, heads, mlp_ratio) for _ in range(layers)])

    def get_cast_dtype(self):
        return self.resblocks[0].mlp.c_fc.weight.dtype

    def run(self, x, B, L, causal, cache, varlen=None, pooled_rows=None):
        """pooled_rows (int64 device [B]): the head reads only these rows of the tower's output - the last block then computes and
        returns only them ([B, D]; engine.LastBlockFn)."""
        if causal and self.width // self.heads > 80:
            _unsupported(f"causal attention with head dim {self.width // self.heads} (the wide heads are compiled for image towers)")
        base = {"B": B, "L": L, "H": self.heads, "causal": bool(causal), "act": self.act, "eps": 1e-5,
                "recompute": bool(self.grad_checkpointing), "keep": "light", "fp8": bool(self.fp8),
                "fp8_grad_fmt": ops.FMT_E5M2 if self.fp8_grad_format == "e5m2" else ops.FMT_E4M3,
                "fp8_predict": bool(self.fp8 and self.fp8_predicted_scales), "varlen": varlen}
        kept, medium = dict(base, keep_this=True), dict(base, keep_this=True, keep="medium")
        light8 = dict(base, keep_this=True, keep="light8")
        n1, n2 = self.keep_blocks, self.keep_blocks + self.light8_blocks
        last = len(self.resblocks) - 1
        counted = any(self.keep_counts.values())
        for i, blk in enumerate(self.resblocks):
            cfg = kept if i < n1 else (light8 if i < n2 else (medium if i < n2 + self.medium_blocks else base))
            if counted:
                ks = frozenset(t for t, n in self.keep_counts.items() if i >= len(self.resblocks) - n) | engine.KEEP_SETS.get(cfg["keep"] if cfg is not base else None, frozenset())
                if ks:
                    cfg = dict(base, keep_this=True, keep=ks)
            if self.fp8 and i > 0:      # this block's LayerNorm-1 backward hands block i-1 its incoming gradient as fp8 operand
                cfg = dict(cfg, q8_handoff=True)
            if i == last and pooled_rows is not None:
                return engine.LastBlockFn.apply(x, pooled_rows, cfg, cache, *blk.param_tuple())
            x = engine.ResBlockFn.apply(x, cfg, cache, *blk.param_tuple())
        return x if pooled_rows is None else engine.TokenDropFn.apply(x, pooled_rows)

    def light_keep_bytes(self, tokens):
        """HBM bytes one kept block holds between forward and backward for `tokens` rows: qkv, a, x1, hpre (bf16) + softmax
        statistics.  (The MLP width is the c_fc layer's own: 4.36 / 4.92 / 8.57 widths at ViT-g / bigG / e-14, not 4.)"""
        return self.medium_keep_bytes(tokens) + self.tensor_keep_bytes(tokens, "h")

    def light8_keep_bytes(self, tokens):
        return self.medium_keep_bytes(tokens) + self.tensor_keep_bytes(tokens, "h8")     # hpre as e4m3: one byte per element

    def tensor_keep_bytes(self, tokens, name):
        """HBM bytes of one kept tensor of one block (keep_counts)."""
        mlp = self.resblocks[0].mlp.c_fc.weight.shape[0]
        return {"qkv": tokens * 3 * self.width * 2, "a": tokens * self.width * 2 + tokens * self.heads * 8,
                "x1": tokens * self.width * 2, "h8": tokens * mlp, "h": tokens * mlp * 2}[name]

    def medium_keep_bytes(self, tokens):
        return tokens * 5 * self.width * 2 + tokens * self.heads * 8   # qkv, a, x1 (bf16) + softmax stats


def patch_dropout_indices(batch, num_tokens, prob):
    """open_clip/transformer.py:76-81: the kept patch indices of PatchDropout, drawn exactly as the reference draws them
    (`torch.randn(batch, num_tokens)` from the global CPU generator, top-k of the noise) -> int64 [batch, num_keep]."""
    num_keep = max(1, int(num_tokens * (1 - prob)))
    rand = torch.randn(batch, num_tokens)
    return rand.topk(num_keep, dim=-1).indices


class VisionTransformer(nn.Module):
    """transformer.py:329-534 (patch-embed ViT, cls/GAP pooling, ln_post, proj)."""

    def __init__(self, image_size, patch_size, width, layers, heads, mlp_ratio, output_dim, global_average_pool=False,
                 act=ops.ACT_GELU_ERF, pos_embed='learnable', ln_pre=True, pool_style='open_clip', cache=None,
                 patch_dropout=0.):
        super().__init__()
        assert 0 <= patch_dropout < 1.
        self.patch_dropout = float(patch_dropout)      # transformer.py:386-388 (0 = disabled)
        image_size = image_size if isinstance(image_size, (tuple, list)) else (image_size, image_size)
        patch_size = patch_size if isinstance(patch_size, (tuple, list)) else (patch_size, patch_size)
        if image_size[0] != image_size[1] or patch_size[0] != patch_size[1]:
            _unsupported("non-square images / patches")
        self.image_size, self.patch_size = tuple(image_size), tuple(patch_size)
        self.grid_size = (image_size[0] // patch_size[0], image_size[1] // patch_size[1])
        self.output_dim = output_dim
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size[0], stride=patch_size[0], bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        n_tok = self.grid_size[0] * self.grid_size[1] + 1
        if pos_embed == 'learnable':
            self.positional_embedding = nn.Parameter(scale * torch.randn(n_tok, width))
        elif pos_embed == 'sin_cos_2d':
            table = get_2d_sincos_pos_embed(width, self.grid_size[0], cls_token=True)
            self.positional_embedding = nn.Parameter(torch.from_numpy(table).float(), requires_grad=False)
        else:
            raise NotImplementedError(pos_embed)
        self.ln_pre = nn.LayerNorm(width) if ln_pre else nn.Identity()
        self.transformer = Transformer(width, layers, heads, mlp_ratio, act=act)
        self.global_average_pool = global_average_pool
        self.attn_pool = None
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))
        if pool_style not in ('open_clip', 'big_vision_tok', 'big_vision_gap'):
            raise ValueError(pool_style)
        self.pool_style = pool_style
        self.image_mean, self.image_std = OPENAI_DATASET_MEAN, OPENAI_DATASET_STD
        self._cache = cache if cache is not None else engine.WeightCache()

    def lock(self, unlocked_groups=0, freeze_bn_stats=False):
        """transformer.py:415-446 (LiT-style locking)."""
        for p in self.parameters():
            p.requires_grad = False
        if unlocked_groups != 0:
            groups = [[self.conv1, self.class_embedding, self.positional_embedding, self.ln_pre],
                      *self.transformer.resblocks[:-1], [self.transformer.resblocks[-1], self.ln_post], self.proj]

            def _unlock(x):
                if isinstance(x, (list, tuple)):
                    for g in x:
                        _unlock(g)
                elif isinstance(x, torch.nn.Parameter):
                    x.requires_grad = True
                else:
                    for p in x.parameters():
                        p.requires_grad = True
            _unlock(groups[-unlocked_groups:])

    @torch.jit.ignore
    def set_grad_checkpointing(self, enable=True):
        self.transformer.grad_checkpointing = enable

    def _pool_mode(self):
        if self.pool_style == 'big_vision_gap':
            return ops.POOL_MEAN_PATCH
        if self.pool_style == 'big_vision_tok' or not self.global_average_pool:
            return ops.POOL_FIRST
        return ops.POOL_MEAN_ALL

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError(f"expected image batch [B,3,S,S], got {tuple(x.shape)}")
        B, S = x.shape[0], x.shape[2]
        Pp = self.patch_size[0]
        g = S // Pp
        L = g * g + 1
        if L != self.positional_embedding.shape[0]:
            raise RuntimeError(f"image size {S} gives {L} tokens but positional_embedding has "
                               f"{self.positional_embedding.shape[0]} rows")
        k = 3 * Pp * Pp
        has_ln_pre = isinstance(self.ln_pre, nn.LayerNorm)
        cfg = {"B": B, "L": L, "P": Pp, "Kp": (k + 7) // 8 * 8, "eps": 1e-5, "ln_pre": has_ln_pre,
               "mean": self.image_mean if x.dtype == torch.uint8 else None,
               "std": self.image_std if x.dtype == torch.uint8 else None}
        ln_w = self.ln_pre.weight if has_ln_pre else self.class_embedding
        ln_b = self.ln_pre.bias if has_ln_pre else self.class_embedding
        x0 = engine.VisionStemFn.apply(x, cfg, self._cache, self.conv1.weight, self.class_embedding,
                                       self.positional_embedding, ln_w, ln_b)
        if self.training and self.patch_dropout > 0.:
            # transformer.py:501-502: PatchDropout between the positional embedding and ln_pre; LayerNorm is per token, so the
            # row selection is applied to the stem's output.  The rest of the tower runs on 1 + kept tokens per sample.
            keep = patch_dropout_indices(B, L - 1, self.patch_dropout)                    # [B, K] patch indices, CPU
            rows = torch.cat([torch.zeros(B, 1, dtype=torch.int64), keep + 1], dim=1) + torch.arange(B).view(B, 1) * L
            x0 = engine.TokenDropFn.apply(x0, rows.reshape(-1).to(x0.device, non_blocking=True))
            L = 1 + keep.shape[1]
        if self._pool_mode() == ops.POOL_FIRST:
            # the head reads the class token only: the last block computes that row alone (engine.LastBlockFn)
            rows = torch.arange(B, device=x0.device, dtype=torch.int64) * L
            xc = self.transformer.run(x0, B, L, False, self._cache, pooled_rows=rows)
            hcfg = {"B": B, "L": 1, "mode": ops.POOL_FIRST, "eps": 1e-5}
            return engine.HeadFn.apply(xc, None, hcfg, self._cache, self.ln_post.weight, self.ln_post.bias, self.proj)
        xL = self.transformer.run(x0, B, L, False, self._cache)
        hcfg = {"B": B, "L": L, "mode": self._pool_mode(), "eps": 1e-5}
        return engine.HeadFn.apply(xL, None, hcfg, self._cache, self.ln_post.weight, self.ln_post.bias, self.proj)


class CLIP(nn.Module):
    """open_clip/model.py:200-274."""

    def __init__(self, embed_dim, vision_cfg, text_cfg, quick_gelu=False, cast_dtype=None, output_dict=False):
        super().__init__()
        self.output_dict = output_dict
        if isinstance(vision_cfg, dict):
            vision_cfg = CLIPVisionCfg(**vision_cfg)
        if isinstance(text_cfg, dict):
            text_cfg = CLIPTextCfg(**text_cfg)
        if vision_cfg.timm_model_name or isinstance(vision_cfg.layers, (tuple, list)):
            _unsupported("timm / ResNet vision towers")
        if vision_cfg.attentional_pool or vision_cfg.input_patchnorm or \
                vision_cfg.ls_init_value is not None or text_cfg.ls_init_value is not None:
            _unsupported("attentional pool / patchnorm / layer scale")
        if text_cfg.hf_model_name or text_cfg.embed_cls:
            _unsupported("HF text towers / embed_cls")
        self._cache = engine.WeightCache()
        self._gather_partner = None          # weakref to a ClipLoss that opted in with loss.bind(model)
        # engine knob: run the causal text tower on the tokens up to each caption's EOT only.  Set it BEFORE the first iteration
        # under DistributedDataParallel(static_graph=True) (it changes the autograd graph); pass the caption lengths from the
        # loader (`forward(image, text, text_lengths=...)`, a list / numpy array of text.argmax(-1) + 1) to avoid a device-to-host read per step
        self.unpad_text = False
        self.visual = VisionTransformer(
            image_size=vision_cfg.image_size, patch_size=vision_cfg.patch_size, width=vision_cfg.width,
            layers=vision_cfg.layers, heads=vision_cfg.width // vision_cfg.head_width, mlp_ratio=vision_cfg.mlp_ratio,
            output_dim=embed_dim, global_average_pool=vision_cfg.global_average_pool,
            act=_act_code(quick_gelu, vision_cfg.gelu_approximate), pos_embed=vision_cfg.pos_embed,
            ln_pre=vision_cfg.ln_pre, pool_style=vision_cfg.pool_style, cache=self._cache,
            patch_dropout=vision_cfg.patch_dropout)
        # text tower: sub-modules live directly on CLIP (model.py:216-225)
        self.transformer = Transformer(text_cfg.width, text_cfg.layers, text_cfg.heads,
                                       act=_act_code(quick_gelu, text_cfg.gelu_approximate))
        self.context_length = text_cfg.context_length
        self.vocab_size = text_cfg.vocab_size
        self.token_embedding = nn.Embedding(text_cfg.vocab_size, text_cfg.width)
        self.positional_embedding = nn.Parameter(torch.empty(text_cfg.context_length, text_cfg.width))
        self.ln_final = nn.LayerNorm(text_cfg.width)
        self.text_projection = nn.Parameter(torch.empty(text_cfg.width, embed_dim))
        if text_cfg.pool_style not in ('open_clip', 'big_vision_tok', 'big_vision_last'):
            raise ValueError(text_cfg.pool_style)
        self.pool_style = text_cfg.pool_style
        self.causal = bool(text_cfg.attention_mask)
        if text_cfg.attention_mask:
            mask = torch.empty(text_cfg.context_length, text_cfg.context_length).fill_(float("-inf")).triu_(1)
            self.register_buffer('attn_mask', mask, persistent=False)   # kept for state/attr parity; the
        else:                                                           # kernel applies the mask itself
            self.attn_mask = None
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        self._init_text_parameters()
        # operand copies (bf16 / transposed / fp8) are cached per parameter version: drop them whenever weights are loaded
        self.register_load_state_dict_post_hook(lambda module, incompatible_keys: module._cache.clear())

    def invalidate_weight_cache(self):
        """Call after writing weights through `p.data` (EMA, weight surgery): such writes bump no version counter."""
        self._cache.clear()

    def _init_text_parameters(self):
        """TextTransformer.init_parameters (transformer.py:596-612)."""
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        width, layers = self.transformer.width, self.transformer.layers
        proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
        attn_std = width ** -0.5
        fc_std = (2 * width) ** -0.5
        for block in self.transformer.resblocks:
            nn.init.normal_(block.attn.in_proj_weight, std=attn_std)
            nn.init.normal_(block.attn.out_proj.weight, std=proj_std)
            nn.init.normal_(block.mlp.c_fc.weight, std=fc_std)
            nn.init.normal_(block.mlp.c_proj.weight, std=proj_std)
        nn.init.normal_(self.text_projection, std=width ** -0.5)

    def lock_image_tower(self, unlocked_groups=0, freeze_bn_stats=False):
        self.visual.lock(unlocked_groups=unlocked_groups, freeze_bn_stats=freeze_bn_stats)

    @torch.jit.ignore
    def set_grad_checkpointing(self, enable=True):
        self.visual.set_grad_checkpointing(enable)
        self.transformer.grad_checkpointing = enable

    def encode_image(self, image, normalize: bool = False):
        features = self.visual(image)
        return engine.L2NormFn.apply(features) if normalize else features

    def _text_varlen(self, text, text_lengths=None):
        """The index structure of the unpadded text tower (engine knob `unpad_text`), or None.  It needs the caption lengths on
the host (B integers).  `text_lengths` ([B] integers as a list / tuple / numpy array / CPU tensor: `text.argmax(-1) + 1`, the
        position of the EOT token - the LARGEST id, model.py:251-254 - plus one; the loader has the token ids on the host anyway;
        under DistributedDataParallel NOT a tensor, DDP would move it to the device; CLIPA_CHECK_TEXT_LENGTHS=1 verifies the
        contract against the device's argmax) costs nothing; without it the lengths are read back from the device, a blocking copy on the
        compute stream: the host then waits until everything enqueued so far (the previous step's backward and optimizer) has
        drained and loses its launch run-ahead once per step - `forward` at least asks before the image tower is
        enqueued."""
        if not (self.unpad_text and self.causal and self.pool_style == 'open_clip'):
            return None
        if text_lengths is not None:
            if torch.is_tensor(text_lengths) and text_lengths.device.type != "cpu":
                # DistributedDataParallel(device_ids=[...]) moves every TENSOR argument of forward to the device (its _pre_forward
                # runs _to_kwargs), so a CPU lengths tensor arrives here as a device tensor and `.cpu()` below would be exactly the
                # blocking copy the argument exists to avoid: pass a list / tuple / numpy array, which DDP leaves alone
                import warnings
                warnings.warn("clipa_amd.CLIP: text_lengths arrived as a device tensor (under DistributedDataParallel pass a list, "
                              "tuple or numpy array: DDP moves tensor arguments to the GPU) - reading it back blocks the host",
                              RuntimeWarning, stacklevel=3)
            lens = torch.as_tensor(text_lengths).to(torch.int64).cpu()
            if lens.shape != (text.shape[0],) or int(lens.min()) < 1 or int(lens.max()) > text.shape[1]:
                raise RuntimeError(f"text_lengths must be [B] integers in [1, {text.shape[1]}]")
            if os.environ.get("CLIPA_CHECK_TEXT_LENGTHS") == "1":
                # the contract: text_lengths[b] == text[b].argmax() + 1 (the reference pools x[arange, text.argmax(-1)],
                # model.py:251-254); lengths derived otherwise (first EOT, attention mask) move the pooled row silently
                want = ops.argmax_tokens(text.long()).to(torch.int64).cpu() + 1
                if not torch.equal(want, lens):
                    bad = int((want != lens).sum())
                    raise RuntimeError(f"text_lengths disagrees with text.argmax(-1) + 1 on {bad} of {lens.numel()} captions")
            return ops.VarLen(lens, text.shape[1], text.device)
        eot = ops.argmax_tokens(text.long())
        return ops.VarLen(eot.to(torch.int64).cpu() + 1, text.shape[1], text.device)

    def encode_text(self, text, normalize: bool = False, _varlen=None, text_lengths=None):
        if text.dim() != 2 or text.shape[1] != self.positional_embedding.shape[0]:
            raise RuntimeError(f"expected token ids [B,{self.positional_embedding.shape[0]}], got {tuple(text.shape)}")
        text = text.long()
        B, T = text.shape
        x0 = engine.TextStemFn.apply(text, self._cache, self.token_embedding.weight, self.positional_embedding)
        if self.unpad_text and self.causal and self.pool_style == 'open_clip':
            # MI355X engine knob (no reference counterpart, results unchanged): under the causal mask (transformer.py:618-624) no
            # position after a caption's EOT can reach the pooled row x[arange, text.argmax(-1)] (model.py:251-254) or receive
            # gradient, so the tower runs on the tokens up to EOT only, packed back to back (zero-padded to whole GEMM tiles).
            # The lengths come to the host once per forward (B integers) to build the index structure.
            vl = _varlen if _varlen is not None else self._text_varlen(text, text_lengths)
            xp = engine.TokenDropFn.apply(x0, vl.src_rows)
            pooled = self.transformer.run(xp, B, T, True, self._cache, varlen=vl, pooled_rows=vl.last_rows)
            hcfg = {"B": B, "L": 1, "mode": ops.POOL_FIRST, "eps": 1e-5}
            features = engine.HeadFn.apply(pooled, None, hcfg, self._cache, self.ln_final.weight, self.ln_final.bias,
                                           self.text_projection)
            return engine.L2NormFn.apply(features) if normalize else features
        # the head reads one row per caption (EOT / first / last token): the last block computes that row alone (engine.LastBlockFn)
        if self.pool_style == 'open_clip':
            pos = ops.argmax_tokens(text).to(torch.int64)
        elif self.pool_style == 'big_vision_tok':
            pos = torch.zeros(B, device=x0.device, dtype=torch.int64)
        else:
            pos = torch.full((B,), T - 1, device=x0.device, dtype=torch.int64)
        rows = torch.arange(B, device=x0.device, dtype=torch.int64) * T + pos
        pooled = self.transformer.run(x0, B, T, self.causal, self._cache, pooled_rows=rows)
        hcfg = {"B": B, "L": 1, "mode": ops.POOL_FIRST, "eps": 1e-5}
        features = engine.HeadFn.apply(pooled, None, hcfg, self._cache, self.ln_final.weight, self.ln_final.bias,
                                       self.text_projection)
        return engine.L2NormFn.apply(features) if normalize else features

    def forward(self, image, text, text_lengths=None):
        # the reference trainer runs the model under torch.autocast(bfloat16) (train.py:160,203-205; precision.py:6-14):
        # every FLOP here is a HIP kernel with its own fixed operand types, so autocast is switched off for the glue
        with torch.autocast(device_type=image.device.type, enabled=False):
            vl = self._text_varlen(text, text_lengths) if text.dim() == 2 else None
            image_features = self.encode_image(image, normalize=True)
            partner = self._gather_partner() if self._gather_partner is not None else None
            if partner is not None and self.training and torch.is_grad_enabled():
                # MI355X engine: a multi-rank ClipLoss bound to this model (`loss.bind(model)`) starts the RCCL
                # all-gather of the image embeddings now, on a side stream, under the text tower (north_star:
                # "overlapped ... on a side HIP stream").  Single rank / eval / not bound: nothing happens.
                partner.early_gather(image_features)
            text_features = self.encode_text(text, normalize=True, _varlen=vl)
            logit_scale = self.logit_scale.exp()
        if self.output_dict:
            return {"image_features": image_features, "text_features": text_features, "logit_scale": logit_scale}
        return image_features, text_features, logit_scale


def convert_weights_to_lp(model: nn.Module, dtype=torch.bfloat16):
    """open_clip/model.py:329-351: Linear/Conv/MHA weights+biases, proj, text_projection -> low precision;
    LayerNorm affine, embeddings, class/positional embeddings, logit_scale stay fp32."""

    def _convert(l):
        if isinstance(l, (nn.Conv1d, nn.Conv2d, nn.Linear)):
            l.weight.data = l.weight.data.to(dtype)
            if l.bias is not None:
                l.bias.data = l.bias.data.to(dtype)
        if isinstance(l, nn.MultiheadAttention):
            for attr in ("in_proj_weight", "q_proj_weight", "k_proj_weight", "v_proj_weight", "in_proj_bias", "bias_k",
                         "bias_v"):
                t = getattr(l, attr, None)
                if t is not None:
                    t.data = t.data.to(dtype)
        for name in ("text_projection", "proj"):
            attr = getattr(l, name, None)
            if attr is not None and hasattr(attr, "data"):
                attr.data = attr.data.to(dtype)

    model.apply(_convert)


def resize_pos_embed(state_dict, model, interpolation: str = 'bicubic', antialias: bool = True):
    """model.py:452-483: bicubic(antialias) resize of the patch-grid positional table at checkpoint load;
    host-side, one-off (the two-resolution CLIPA schedule)."""
    old = state_dict.get('visual.positional_embedding', None)
    if old is None or not hasattr(model.visual, 'grid_size'):
        return
    gh, gw = model.visual.grid_size
    if gh * gw + 1 == old.shape[0]:
        return
    tok, img = old[:1], old[1:]
    og = int(math.sqrt(len(img)))
    img = img.reshape(1, og, og, -1).permute(0, 3, 1, 2)
    img = torch.nn.functional.interpolate(img.float(), size=(gh, gw), mode=interpolation, antialias=antialias,
                                          align_corners=False)
    img = img.permute(0, 2, 3, 1).reshape(gh * gw, -1).to(old.dtype)
    state_dict['visual.positional_embedding'] = torch.cat([tok, img], dim=0)


def resize_text_pos_embed(state_dict, model, interpolation: str = 'linear', antialias: bool = False):
    """model.py:486-515: linear resize of the text positional table (ctx 8/16 -> 32/77)."""
    old = state_dict.get('positional_embedding', None)
    if old is None:
        return
    num_pos = model.positional_embedding.shape[0]
    if old.shape[1] != model.positional_embedding.shape[1]:
        raise RuntimeError('text pos_embed width changed!')
    if old.shape[0] == num_pos:
        return
    t = old.reshape(1, old.shape[0], old.shape[1]).permute(0, 2, 1).float()
    t = torch.nn.functional.interpolate(t, size=num_pos, mode=interpolation, antialias=antialias, align_corners=False)
    state_dict['positional_embedding'] = t.permute(0, 2, 1)[0].to(old.dtype)
