"""Model-config registry with the open_clip JSON schema (embed_dim / vision_cfg / text_cfg), cf.
clipa_torch/open_clip/factory.py:26-75 and open_clip/model_configs/*.json.  The ViT-CLIP shapes the
hot path covers are generated here; `add_model_config(path)` registers extra JSON files or
directories (e.g. the reference's own model_configs/) exactly like the reference registry does.
"""
import copy
import json
import os
import re

_VIT = {  # name -> (width, layers, head_width)
    "S": (384, 12, 64), "M": (512, 12, 64), "B": (768, 12, 64), "L": (1024, 24, 64), "H": (1280, 32, 80),
}
_TEXT = {  # name -> (embed_dim, text width, heads, layers)
    "S": (384, 384, 6, 12), "M": (512, 512, 8, 12), "B": (512, 512, 8, 12), "L": (768, 768, 12, 12),
    "H": (1024, 1024, 16, 24),
}

_MODEL_CONFIGS = {}


def _vit(size, patch, ctx=77, gap=False, **text_extra):
    w, layers, hw = _VIT[size]
    e, tw, th, tl = _TEXT[size]
    v = {"image_size": 224, "layers": layers, "width": w, "patch_size": patch}
    if hw != 64:
        v["head_width"] = hw
    if gap:
        v["global_average_pool"] = True
    t = {"context_length": ctx, "vocab_size": 49408, "width": tw, "heads": th, "layers": tl}
    t.update(text_extra)
    return {"embed_dim": e, "vision_cfg": v, "text_cfg": t}


def _builtin():
    cfgs = {}
    for size in "SMBLH":
        for patch in (32, 16, 14):
            cfgs[f"ViT-{size}-{patch}"] = _vit(size, patch)
    cfgs["ViT-B-16-CL16"] = _vit("B", 16, ctx=16)
    for ctx in (8, 16):
        cfgs[f"ViT-L-16-CL{ctx}-Syntax-GAP"] = _vit("L", 16, ctx=ctx, gap=True, text_mask="syntax")
    cfgs["ViT-L-16-CL32-GAP"] = _vit("L", 16, ctx=32, gap=True)
    cfgs["ViT-H-14-CL8-SyntaxMask-GAP"] = _vit("H", 14, ctx=8, gap=True, text_mask="syntax")
    cfgs["ViT-H-14-CL32-GAP"] = _vit("H", 14, ctx=32, gap=True)
    # the reference's largest towers (model_configs/ViT-g-14.json, ViT-bigG-14.json - CLIPA-v2's G/14 -, ViT-e-14.json):
    # head widths 88 / 104 / 112, fractional mlp_ratio (MLP widths 6144 / 8192 / 15360)
    for name, (e, w, layers, hw, mlp, tw, th, tl) in {
            "ViT-g-14": (1024, 1408, 40, 88, 4.3637, 1024, 16, 24), "ViT-bigG-14": (1280, 1664, 48, 104, 4.9231, 1280, 20, 32),
            "ViT-e-14": (1280, 1792, 56, 112, 8.5715, 1280, 20, 36)}.items():
        cfgs[name] = {"embed_dim": e,
                      "vision_cfg": {"image_size": 224, "layers": layers, "width": w, "head_width": hw, "mlp_ratio": mlp,
                                     "patch_size": 14},
                      "text_cfg": {"context_length": 77, "vocab_size": 49408, "width": tw, "heads": th, "layers": tl}}
    return cfgs


def _natural_key(s):
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", s.lower())]


def add_model_config(path):
    """Register a JSON file or every *.json in a directory (factory.py:64-69)."""
    files = []
    if os.path.isdir(path):
        files = [os.path.join(path, f) for f in os.listdir(path) if f.endswith(".json")]
    elif os.path.isfile(path):
        files = [path]
    for f in files:
        with open(f) as fh:
            cfg = json.load(fh)
        if all(k in cfg for k in ("embed_dim", "vision_cfg", "text_cfg")):
            _MODEL_CONFIGS[os.path.splitext(os.path.basename(f))[0]] = cfg


def list_models():
    return sorted(_MODEL_CONFIGS.keys(), key=_natural_key)


def get_model_config(name):
    return copy.deepcopy(_MODEL_CONFIGS[name]) if name in _MODEL_CONFIGS else None


_MODEL_CONFIGS.update(_builtin())
