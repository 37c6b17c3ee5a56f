"""ORACLE - TEST INFRASTRUCTURE ONLY.  Generate tests/golden/train_loop_ref.npz with the REAL reference trainer.

Run in the build container (needs /root/reference):  python oracle/make_train_golden.py

The reference's own `training.train.train_one_epoch` (clipa_torch/training/train.py:158-314) drives the reference's own
`open_clip.model.CLIP` + `open_clip.loss.ClipLoss` on the CPU: `--precision amp_bf16` (the precision of every reference
GPU script, e.g. scripts/exp/gpu/vit_l16/i37_t8_pretrain.sh:18; on a CUDA-less host `torch.cuda.amp.autocast` disables
itself, so the arithmetic is fp32 - the accurate anchor), `--to-float-on-device`, `--grad-clip-norm 1`, AdamW groups of
main.py:311-326, linear warm-up.  Two runs on the toy `cls_erf` model (same config / seeded weights as
tests/golden/cls_erf.npz): accum_freq 1 x 3 steps and accum_freq 2 x 2 steps.  Stored: the loss of every loss call, the
learning rates, exp-free logit_scale after each run, and - per parameter - the update's L2 norm plus the full updated
tensor for a handful of parameters.  tests/test_trainer_gpu.py replays the same loop on the engine.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import clip_oracle as O  # noqa: E402
from oracle import trainer_harness as H  # noqa: E402
from oracle.make_golden import CASES  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "train_loop_ref.npz")
KEEP = ("logit_scale", "visual.proj", "text_projection", "visual.class_embedding", "positional_embedding",
        "visual.transformer.resblocks.0.attn.in_proj_weight", "visual.transformer.resblocks.1.mlp.c_fc.weight",
        "transformer.resblocks.1.mlp.c_proj.weight", "visual.ln_post.weight", "transformer.resblocks.0.attn.out_proj.bias")


def run(train, ref_model, ref_loss, spec, accum_freq, nsteps):
    cfg, B, S, seed = spec["cfg"], spec["B"], spec["S"], spec["seed"]
    torch.manual_seed(0)
    model = ref_model.CLIP(**cfg, output_dict=True).float()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd0 = O.make_state_dict(shapes, seed)
    model.load_state_dict(sd0, strict=True)
    model.visual.image_mean, model.visual.image_std = H.IMAGE_MEAN, H.IMAGE_STD     # what factory.py:249-250 does
    tc = cfg["text_cfg"]
    batches = H.synthetic_batches(nsteps * accum_freq, B, S, tc["context_length"], tc["vocab_size"], seed + 100)
    args = H.make_args("cpu", batch_size=B, accum_freq=accum_freq)
    opt = H.make_optimizer(model)
    lrs = []
    sched = H.make_scheduler(opt)
    loss = H.LossRecorder(ref_loss.ClipLoss(local_loss=False, gather_with_grad=False, cache_labels=True, rank=0, world_size=1))
    train.train_one_epoch(model, {"train": H.DataInfo(H.ListLoader(batches))}, loss, 0, opt, None,
                          lambda s: lrs.append(sched(s)), None, args)
    sd1 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    names = sorted(k for k in sd1 if k in dict(model.named_parameters()))
    dn = np.array([float((sd1[k] - sd0[k]).double().norm()) for k in names])
    return dict(losses=np.array(loss.values), lrs=np.array(lrs), names=np.array(names), delta_norms=dn,
                kept={k: sd1[k].numpy() for k in KEEP if k in sd1})


def main():
    torch.set_num_threads(4)
    train = H.load_trainer("reference")
    import open_clip
    ref_model, ref_loss = sys.modules["open_clip.model"], sys.modules["open_clip.loss"]
    assert open_clip.CLIP is ref_model.CLIP
    spec = CASES["cls_erf"]
    arrays = {"cfg": json.dumps(spec["cfg"]), "seed": spec["seed"], "B": spec["B"], "S": spec["S"], "keep": np.array(KEEP)}
    for tag, accum, nsteps in (("a1", 1, 3), ("a2", 2, 2)):
        r = run(train, ref_model, ref_loss, spec, accum, nsteps)
        arrays[f"{tag}_losses"], arrays[f"{tag}_lrs"] = r["losses"], r["lrs"]
        arrays[f"{tag}_names"], arrays[f"{tag}_delta_norms"] = r["names"], r["delta_norms"]
        for k, v in r["kept"].items():
            arrays[f"{tag}_w::{k}"] = v
        print(tag, "losses", np.round(r["losses"], 5), "lrs", r["lrs"], "logit_scale", float(r["kept"]["logit_scale"]))
    np.savez_compressed(OUT, **arrays)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
