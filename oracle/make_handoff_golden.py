"""ORACLE - TEST INFRASTRUCTURE ONLY.  tests/golden/handoff_S16_84_to_224.npz: BASELINE config 5 (CLIPA's two-resolution
schedule) as ONE flow through the REAL reference, at ViT-S/16 width so that the fixture stays small:

  phase 1  ViT-S/16 @ 84 px, fixed sin-cos positional table, GAP pooling, text context 16 (the pre-training form:
           scripts/exp/gpu/vit_l16/i37_t8_pretrain.sh uses ViT-L-16-CL8-Syntax-GAP + --pos-embed sin_cos_2d); its weights are
           written the way training/main.py:436-468 writes a checkpoint ({"epoch", "name", "state_dict" with DDP's "module."
           prefix, ...});
  phase 2  the reference's load path for the fine-tune (open_clip/factory.py:110-118 load_checkpoint ->
           model.py:452-515 resize_pos_embed / resize_text_pos_embed) into a 224 px / context-32 model with a LEARNABLE
           positional table, then one forward + ClipLoss + backward of the real reference on a seeded batch.

Stored: both configs, the seed of the phase-1 weights, the phase-2 batch, the reference's resized positional tables, its
features / loss / logit scale and per-parameter gradient digests.  Run in the build container: python oracle/make_handoff_golden.py
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import clip_oracle as O  # noqa: E402
from oracle import ref_loader  # noqa: E402
from oracle.make_golden import grad_digest  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "handoff_S16_84_to_224.npz")
SEED, B = 31, 4


def configs():
    text = {"vocab_size": 49408, "width": 384, "heads": 6, "layers": 12}
    vis = {"layers": 12, "width": 384, "patch_size": 16, "global_average_pool": True}
    cfg84 = {"embed_dim": 384, "vision_cfg": dict(vis, image_size=84, pos_embed="sin_cos_2d"),
             "text_cfg": dict(text, context_length=16)}
    cfg224 = {"embed_dim": 384, "vision_cfg": dict(vis, image_size=224), "text_cfg": dict(text, context_length=32)}
    return cfg84, cfg224


def phase1_state_dict(model84, seed):
    """Deterministic phase-1 weights: numpy-seeded values for every trainable tensor, the model's own values for the frozen
    ones (the sin-cos table).  Shared with tests/test_model_gpu.py so that the engine starts from the same checkpoint."""
    sd0 = model84.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd0.items()}
    frozen = [k for k, p in model84.named_parameters() if not p.requires_grad]
    sd = O.make_state_dict(shapes, seed, frozen=frozen)
    for k in frozen:
        sd[k] = sd0[k].detach().clone().float()
    return sd


def write_checkpoint(sd, path):
    """training/main.py:436-468: what `--save-frequency` writes under DistributedDataParallel."""
    torch.save({"epoch": 1, "name": "handoff", "state_dict": {"module." + k: v for k, v in sd.items()}, "optimizer": {}}, path)


def main():
    torch.set_num_threads(4)
    ref_model, ref_loss, _ = ref_loader.load()
    cfg84, cfg224 = configs()
    torch.manual_seed(0)
    m84 = ref_model.CLIP(**cfg84).float()
    sd84 = phase1_state_dict(m84, SEED)
    m84.load_state_dict(sd84, strict=True)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "epoch_1.pt")
        write_checkpoint(sd84, path)
        # open_clip/factory.py:99-118 restated call by call on the real reference functions (factory.py itself imports the
        # hub / tokenizer / timm stack that is absent here)
        ck = torch.load(path, map_location="cpu", weights_only=False)
        sd = ck["state_dict"]
        sd = {k[7:]: v for k, v in sd.items()} if next(iter(sd)).startswith("module") else sd
    m224 = ref_model.CLIP(**cfg224, output_dict=True).float().eval()
    ref_model.resize_pos_embed(sd, m224)
    ref_model.resize_text_pos_embed(sd, m224)
    m224.load_state_dict(sd, strict=True)
    assert m224.visual.positional_embedding.requires_grad and m224.visual.positional_embedding.shape[0] == 197
    images_u8, texts = O.synthetic_batch(B, 224, 32, cfg224["text_cfg"]["vocab_size"], SEED + 1)
    out = m224(O.normalize_images(images_u8), texts)
    loss = ref_loss.ClipLoss(local_loss=False, gather_with_grad=False, cache_labels=True, rank=0, world_size=1)(
        **out, output_dict=True)["contrastive_loss"]
    loss.backward()
    grads = {k: p.grad for k, p in m224.named_parameters() if p.grad is not None}
    names, norms, sums, idxs, vals = grad_digest(grads, SEED + 1000)
    np.savez_compressed(
        OUT, cfg84=json.dumps(cfg84), cfg224=json.dumps(cfg224), seed=SEED, images_u8=images_u8.numpy(), texts=texts.numpy(),
        visual_pos_224=sd["visual.positional_embedding"].numpy(), text_pos_32=sd["positional_embedding"].numpy(),
        image_features=out["image_features"].detach().numpy(), text_features=out["text_features"].detach().numpy(),
        logit_scale=out["logit_scale"].detach().numpy(), loss=loss.detach().numpy(), grad_names=np.array(names),
        grad_norms=norms, grad_sums=sums, grad_sample_idx=idxs, grad_sample_vals=vals)
    print(f"handoff: loss={float(loss):.7f} grads={len(names)} visual pos {tuple(sd['visual.positional_embedding'].shape)} "
          f"text pos {tuple(sd['positional_embedding'].shape)}")


if __name__ == "__main__":
    main()
