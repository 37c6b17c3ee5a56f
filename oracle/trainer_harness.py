"""ORACLE - TEST INFRASTRUCTURE ONLY.  Drive the reference's OWN `training.train.train_one_epoch`.

`training/train.py` (clipa_torch/training/train.py:158-314) is importable in the build container once four names it
pulls in at import time are satisfied: `torchvision.transforms.Normalize` (train.py:11,195),
`training.data.DataInfo` (train.py:32), `training.zero_shot.zero_shot_eval` (train.py:30) and the package `open_clip`
(train.py:28: get_cast_dtype, CLIP, CustomTextCLIP).  Two ways to satisfy the last one:

  which="reference"  `open_clip` = the reference's own model / loss modules (oracle.ref_loader)  -> the oracle run
  which="engine"     `open_clip` = clipa_amd (INTEGRATION.md section 1 applied for real)         -> the drop-in run

Everything else (the step loop, autocast context, accum_freq feature cache, clip_grad_norm_, optimizer.step, the logit-scale
clamp, the logging path) is the reference's code, unmodified.  The pieces of `training/main.py` that surround the call
(AdamW parameter groups main.py:311-326, the args namespace) are restated in `make_optimizer` / `make_args`.

Not usable on the GPU box (no /root/reference there): tests/test_trainer_gpu.py runs a restatement of the same loop against
the fixture this harness generated (oracle/make_train_golden.py).
"""
import argparse
import importlib
import importlib.machinery
import math
import os
import sys
import types

import torch

from . import ref_loader


class _Normalize:
    """torchvision.transforms.Normalize on a [B,C,H,W] float tensor: (x - mean[c]) / std[c]  (train.py:195)."""

    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def __call__(self, x):
        mean = torch.as_tensor(self.mean, dtype=x.dtype, device=x.device).view(1, -1, 1, 1)
        std = torch.as_tensor(self.std, dtype=x.dtype, device=x.device).view(1, -1, 1, 1)
        return (x - mean) / std


class DataInfo:
    """Stand-in for training/data.py:66-76 (dataloader + set_epoch)."""

    def __init__(self, dataloader):
        self.dataloader = dataloader

    def set_epoch(self, epoch):
        pass


class ListLoader:
    """An in-memory loader with the two attributes train_one_epoch reads (train.py:172-174)."""

    def __init__(self, batches):
        self.batches = batches
        self.num_batches = len(batches)
        self.num_samples = sum(len(b[0]) for b in batches)

    def __iter__(self):
        return iter(self.batches)


def _purge(prefixes):
    for k in list(sys.modules):
        if any(k == p or k.startswith(p + ".") for p in prefixes):
            del sys.modules[k]


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    mod.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(mod, k, v)
    sys.modules[name] = mod
    return mod


def load_trainer(which):
    """Import the reference's training.train with `open_clip` bound as described above; returns the module."""
    if not ref_loader.available():
        raise RuntimeError("reference not mounted")
    _purge(["training", "open_clip", "torchvision"])
    root = ref_loader.REF_ROOT
    if root not in sys.path:
        sys.path.insert(0, root)
    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms", Normalize=_Normalize)
    tv.ops = _stub("torchvision.ops")
    tv.ops.misc = _stub("torchvision.ops.misc", FrozenBatchNorm2d=type("FrozenBatchNorm2d", (torch.nn.Module,), {}))
    sys.modules.setdefault("transformers", None)
    if which == "reference":
        model, loss, _ = ref_loader.load()
        pkg = sys.modules["open_clip"]
        pkg.get_cast_dtype, pkg.CLIP, pkg.CustomTextCLIP = model.get_cast_dtype, model.CLIP, model.CustomTextCLIP
        pkg.ClipLoss = loss.ClipLoss
    elif which == "engine":
        import clipa_amd
        pkg = _stub("open_clip")
        for k in dir(clipa_amd):
            if not k.startswith("__"):
                setattr(pkg, k, getattr(clipa_amd, k))
        pkg.CustomTextCLIP = type("CustomTextCLIP", (torch.nn.Module,), {})      # out of scope; name only
    else:
        raise ValueError(which)
    # the package `training` without its __init__ side effects; data / zero_shot need webdataset, braceexpand, tensorflow
    tr = types.ModuleType("training")
    tr.__path__ = [os.path.join(root, "training")]
    tr.__spec__ = importlib.machinery.ModuleSpec("training", None, is_package=True)
    sys.modules["training"] = tr
    _stub("training.data", DataInfo=DataInfo)
    _stub("training.zero_shot", zero_shot_eval=lambda *a, **k: {})
    return importlib.import_module("training.train")


def unload():
    _purge(["training", "open_clip", "torchvision"])


def make_args(device, precision="amp_bf16", batch_size=8, accum_freq=1, grad_clip_norm=1.0):
    """The fields of training/params.py that train_one_epoch and after_train_step read."""
    return argparse.Namespace(
        device=str(device), precision=precision, accum_freq=accum_freq, distill=False, skip_scheduler=False,
        to_float_on_device=True, image_mean=None, image_std=None, grad_clip_norm=grad_clip_norm, horovod=False,
        distributed=False, rank=0, local_rank=0, world_size=1, batch_size=batch_size, log_every_n_steps=1, wandb=False,
        val_steps=0, zeroshot_steps=0)


LR, WD, BETAS, EPS = 1e-3, 0.2, (0.9, 0.95), 1e-6
IMAGE_MEAN, IMAGE_STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)   # open_clip/constants.py:1-2


def make_optimizer(model, cls=torch.optim.AdamW):
    """training/main.py:311-326: no weight decay on gains, biases and logit_scale; decided by parameter NAME."""
    exclude = lambda n, p: p.ndim < 2 or "bn" in n or "ln" in n or "bias" in n or "logit_scale" in n
    named = list(model.named_parameters())
    return cls([{"params": [p for n, p in named if exclude(n, p) and p.requires_grad], "weight_decay": 0.},
                {"params": [p for n, p in named if not exclude(n, p) and p.requires_grad], "weight_decay": WD}],
               lr=LR, betas=BETAS, eps=EPS)


def make_scheduler(optimizer, warmup=2):
    """Linear warm-up as training/scheduler.py:4-5,43-53 (`_warmup_lr`, `assign_learning_rate`), constant afterwards."""
    def step_fn(step):
        lr = LR * (step + 1) / warmup if step < warmup else LR
        for g in optimizer.param_groups:
            g["lr"] = lr
        return lr
    return step_fn


def synthetic_batches(n, B, S, ctx, vocab, seed):
    from . import clip_oracle as O
    return [O.synthetic_batch(B, S, ctx, vocab, seed + 17 * i) for i in range(n)]


class LossRecorder:
    """Wraps a loss module; records the scalar of every call (the trainer only logs it)."""

    def __init__(self, loss):
        self.loss, self.values = loss, []

    def __call__(self, *a, **k):
        out = self.loss(*a, **k)
        v = out["contrastive_loss"] if isinstance(out, dict) else out
        self.values.append(float(v.detach().float().cpu()))
        return out


def log_clamp_hi():
    return math.log(100)


def restated_train_one_epoch(model, batches, loss, optimizer, scheduler, args, autocast):
    """The step loop of training/train.py:158-291 restated for hosts without /root/reference (the GPU box): same order of
    operations, same contexts.  tests/test_trainer_cpu.py holds it bit-for-bit to the real train_one_epoch."""
    device = torch.device(args.device)
    model.train()                                                                   # train.py:164
    A = args.accum_freq
    if A > 1:
        accum_images, accum_texts, accum_features = [], [], {}                      # train.py:176-177
    for i, (images, texts) in enumerate(batches):
        step = i // A
        scheduler(step)                                                             # train.py:184-185
        images = images.to(device=device, non_blocking=True)
        texts = texts.to(device=device, non_blocking=True)
        if args.to_float_on_device:                                                 # train.py:191-195
            m = getattr(model, "module", model).visual
            images = _Normalize(args.image_mean or m.image_mean, args.image_std or m.image_std)(images.float().div(255))
        optimizer.zero_grad()                                                       # train.py:200
        if A == 1:
            with autocast():                                                        # train.py:203-213
                model_out = model(images, texts)
                losses = loss(**model_out, output_dict=True)
                total_loss = sum(losses.values())
            total_loss.backward()                                                   # train.py:215
        else:
            with torch.no_grad():                                                   # train.py:218-230
                with autocast():
                    model_out = model(images, texts)
                    model_out.pop("logit_scale")
                    for key, val in model_out.items():
                        accum_features.setdefault(key, []).append(val)
                accum_images.append(images)
                accum_texts.append(texts)
            if ((i + 1) % A) > 0:                                                   # train.py:233-237
                continue
            optimizer.zero_grad()                                                   # train.py:242
            for j in range(A):                                                      # train.py:243-256
                images, texts = accum_images[j], accum_texts[j]
                with autocast():
                    model_out = model(images, texts)
                    logit_scale = model_out.pop("logit_scale")
                    inputs = {}
                    for key, val in accum_features.items():
                        accumulated = accum_features[key]
                        inputs[key] = torch.cat(accumulated[:j] + [model_out[key]] + accumulated[j + 1:])
                    losses = loss(**inputs, logit_scale=logit_scale, output_dict=True)
                    del inputs
                    total_loss = sum(losses.values())
                total_loss.backward()
        if args.grad_clip_norm is not None:                                         # train.py:276-278
            torch.nn.utils.clip_grad_norm_(model.parameters(), args.grad_clip_norm, norm_type=2.0)
        optimizer.step()
        if A > 1:
            accum_images, accum_texts, accum_features = [], [], {}                  # train.py:281-282
        with torch.no_grad():                                                       # train.py:285-286
            getattr(model, "module", model).logit_scale.clamp_(0, math.log(100))
