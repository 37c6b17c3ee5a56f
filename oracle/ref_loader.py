"""ORACLE - TEST INFRASTRUCTURE ONLY.  Import the REAL reference modules from /root/reference.

Only usable in the build container (the GPU box has no /root/reference); used by oracle/make_golden.py
to generate tests/golden/*.npz and by the optional CPU tests that compare against the live reference.
The reference's open_clip/__init__.py hard-imports ftfy / tensorflow / torchvision, which are absent,
so the package __init__ is bypassed and torchvision is stubbed; only
open_clip.{transformer,model,loss} are loaded - exactly the files on the hot path.
"""
import importlib
import importlib.machinery
import os
import sys
import types

REF_ROOT = os.environ.get("CLIPA_REFERENCE", "/root/reference/clipa_torch")


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "open_clip"))


def load():
    """Returns (open_clip.model, open_clip.loss, open_clip.transformer) of the reference."""
    if not available():
        raise RuntimeError(f"reference not found at {REF_ROOT}")
    import torch
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    for name in ("torchvision", "torchvision.ops", "torchvision.ops.misc"):
        if name not in sys.modules:
            mod = types.ModuleType(name)
            mod.__spec__ = importlib.machinery.ModuleSpec(name, None)
            sys.modules[name] = mod
    sys.modules["torchvision.ops.misc"].FrozenBatchNorm2d = type("FrozenBatchNorm2d", (torch.nn.Module,), {})
    sys.modules.setdefault("transformers", None)
    if "open_clip" not in sys.modules or not hasattr(sys.modules["open_clip"], "__path__"):
        pkg = types.ModuleType("open_clip")
        pkg.__path__ = [os.path.join(REF_ROOT, "open_clip")]
        sys.modules["open_clip"] = pkg
    model = importlib.import_module("open_clip.model")
    loss = importlib.import_module("open_clip.loss")
    transformer = importlib.import_module("open_clip.transformer")
    return model, loss, transformer
