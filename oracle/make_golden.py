"""ORACLE - TEST INFRASTRUCTURE ONLY.  Generate tests/golden/*.npz by running the REAL reference.

Run in the build container (needs /root/reference):  python oracle/make_golden.py
Each fixture holds: the open_clip-style model config, deterministic-weight seed, inputs (uint8 images,
token ids), the reference's state_dict key/shape list, and the reference outputs: normalised image /
text features, exp(logit_scale), logits_per_image, loss, and - to keep fixtures small - per-parameter
gradient digests (L2 norm, sum, 16 sampled entries).  The weights themselves are regenerated from the
seed by oracle.clip_oracle.make_state_dict (numpy MT19937, platform independent) and loaded into the
reference with load_state_dict, so they need not be stored.
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
from oracle import ref_loader  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
N_SAMPLES = 16

CASES = {
    # CLS pooling, erf GELU, learnable pos-embed, causal text (the ViT-{S,B,L}-16 family)
    "cls_erf": dict(B=8, S=48, seed=11, cfg={
        "embed_dim": 64,
        "vision_cfg": {"image_size": 48, "layers": 2, "width": 128, "patch_size": 16},
        "text_cfg": {"context_length": 16, "vocab_size": 512, "width": 128, "heads": 2, "layers": 2}}),
    # CLIPA pre-training flavour: GAP pooling, fixed sin-cos table, 40 px (trailing 8 px dropped), ctx 8
    "gap_sincos_tanh": dict(B=8, S=40, seed=12, cfg={
        "embed_dim": 64,
        "vision_cfg": {"image_size": 40, "layers": 2, "width": 128, "patch_size": 16, "global_average_pool": True,
                       "pos_embed": "sin_cos_2d", "gelu_approximate": "tanh"},
        "text_cfg": {"context_length": 8, "vocab_size": 512, "width": 64, "heads": 1, "layers": 2,
                     "gelu_approximate": "tanh"}}),
    # BigVision flavour: no ln_pre, GAP over patches only, last-token text pooling without causal mask, QuickGELU
    "bigvision_quick": dict(B=8, S=32, seed=13, cfg={
        "embed_dim": 64, "quick_gelu": True,
        "vision_cfg": {"image_size": 32, "layers": 1, "width": 64, "patch_size": 16, "global_average_pool": True,
                       "ln_pre": False, "pool_style": "big_vision_gap"},
        "text_cfg": {"context_length": 8, "vocab_size": 512, "width": 64, "heads": 1, "layers": 1,
                     "pool_style": "big_vision_last", "attention_mask": False}}),
    # ViT-H/14 flavour: 14 px patches (588-wide im2col rows, not a multiple of 8), head_width 80
    "h14_dh80": dict(B=8, S=42, seed=14, cfg={
        "embed_dim": 64,
        "vision_cfg": {"image_size": 42, "layers": 2, "width": 160, "head_width": 80, "patch_size": 14},
        "text_cfg": {"context_length": 16, "vocab_size": 512, "width": 128, "heads": 2, "layers": 2}}),
    # ViT-g/14 flavour (model_configs/ViT-g-14.json; CLIPA-v2's G/14 is ViT-bigG-14): head_width 88 = 5.5 sixteen-wide
    # reduction steps, fractional mlp_ratio 4.3637 (176 -> 768), 14 px patches
    "g14_dh88": dict(B=8, S=42, seed=16, cfg={
        "embed_dim": 64,
        "vision_cfg": {"image_size": 42, "layers": 2, "width": 176, "head_width": 88, "mlp_ratio": 4.3637, "patch_size": 14},
        "text_cfg": {"context_length": 16, "vocab_size": 512, "width": 128, "heads": 2, "layers": 2}}),
    # PatchDropout (transformer.py:53-83; the reference's ViT-H/14 fine-tune scripts pass --force-patch-dropout): training
    # mode, 16 patches of which 8 are kept per sample (drawn from the global CPU generator after manual_seed(drop_seed)),
    # GAP pooling over the kept tokens
    "patchdrop_gap": dict(B=8, S=64, seed=15, drop_seed=1234, cfg={
        "embed_dim": 64,
        "vision_cfg": {"image_size": 64, "layers": 2, "width": 128, "patch_size": 16, "global_average_pool": True,
                       "patch_dropout": 0.5},
        "text_cfg": {"context_length": 16, "vocab_size": 512, "width": 128, "heads": 2, "layers": 2}}),
}
# fixtures whose forward is stochastic in training mode: not part of the generic MODEL_CASES sweeps of the test-suite
STOCHASTIC_CASES = ("patchdrop_gap",)


# Full model dimensions (SURVEY 8c "planned oracle artefacts"): the reference's own model_configs/*.json at BASELINE
# shapes, small batch.  cfg is read from the reference's JSON at generation time and stored in the fixture.
FULL_CASES = {
    "full_L16_224": dict(json="ViT-L-16.json", B=4, S=224, seed=21),                 # BASELINE configs 3 / 5b
    "full_B16_224": dict(json="ViT-B-16.json", B=4, S=224, seed=22),                 # BASELINE config 2
    "full_H14_224": dict(json="ViT-H-14.json", B=2, S=224, seed=23),                 # BASELINE config 4 (head dim 80)
    "full_L16_84_gap": dict(json="ViT-L-16-CL32-GAP.json", B=4, S=84, seed=24, ctx=77,   # BASELINE config 5a: 26 tokens,
                            vision_extra={"pos_embed": "sin_cos_2d"}),                   # GAP, frozen sin-cos table
    # BASELINE config 1 at its stated dimensions: ViT-S/16 @ 112 px (50 tokens), text ctx 32, local batch 64
    "full_S16_112_t32": dict(json="ViT-S-16.json", B=64, S=112, seed=25, ctx=32),
}


def full_cfg(spec):
    path = os.path.join(ref_loader.REF_ROOT, "open_clip", "model_configs", spec["json"])
    cfg = json.load(open(path))
    cfg["vision_cfg"]["image_size"] = spec["S"]
    cfg["vision_cfg"].update(spec.get("vision_extra", {}))
    if "ctx" in spec:
        cfg["text_cfg"]["context_length"] = spec["ctx"]
    return cfg


def grad_digest(named_grads, seed):
    names = sorted(named_grads)
    rng = np.random.RandomState(seed)
    norms, sums, idxs, vals = [], [], [], []
    for n in names:
        g = named_grads[n].detach().double().reshape(-1)
        idx = rng.randint(0, g.numel(), size=N_SAMPLES)
        norms.append(float(g.norm()))
        sums.append(float(g.sum()))
        idxs.append(idx)
        vals.append(g[torch.from_numpy(idx)].numpy())
    return names, np.array(norms), np.array(sums), np.stack(idxs), np.stack(vals)


def run_case(name, spec, ref_model, ref_loss):
    cfg, B, S, seed = spec["cfg"], spec["B"], spec["S"], spec["seed"]
    torch.manual_seed(0)
    model = ref_model.CLIP(**cfg, output_dict=True).float().eval()
    ref_sd = model.state_dict()
    shapes = {k: tuple(v.shape) for k, v in ref_sd.items()}
    frozen = [k for k, p in model.named_parameters() if not p.requires_grad]
    sd = O.make_state_dict(shapes, seed, frozen=frozen)
    for k in frozen:
        sd[k] = ref_sd[k].clone()
    model.load_state_dict(sd, strict=True)
    images_u8, texts = O.synthetic_batch(B, S, cfg["text_cfg"]["context_length"], cfg["text_cfg"]["vocab_size"], seed)
    images = O.normalize_images(images_u8)          # train.py:191-197
    extra = {}
    if "drop_seed" in spec:
        # the first random draw of a training-mode forward is PatchDropout's torch.randn(batch, num_tokens) (transformer.py:79):
        # the same seed reproduces the kept indices for the fixture
        model.train()
        n_tok = (S // cfg["vision_cfg"]["patch_size"]) ** 2
        torch.manual_seed(spec["drop_seed"])
        keep = torch.randn(B, n_tok).topk(max(1, int(n_tok * (1 - cfg["vision_cfg"]["patch_dropout"]))), dim=-1).indices
        extra = {"patch_keep": keep.numpy(), "drop_seed": spec["drop_seed"]}
        torch.manual_seed(spec["drop_seed"])
    out = model(images, texts)
    loss_fn = ref_loss.ClipLoss(local_loss=False, gather_with_grad=False, cache_labels=True, rank=0, world_size=1)
    logits_i, _ = loss_fn.get_logits(out["image_features"], out["text_features"], out["logit_scale"])
    loss = loss_fn(**out, output_dict=True)["contrastive_loss"]
    loss.backward()
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    names, norms, sums, idxs, vals = grad_digest(grads, seed + 1000)
    np.savez_compressed(
        os.path.join(OUT, f"{name}.npz"),
        cfg=json.dumps(cfg), seed=seed, images_u8=images_u8.numpy(), texts=texts.numpy(),
        keys=np.array(list(shapes.keys())), shapes=json.dumps({k: list(v) for k, v in shapes.items()}),
        frozen=np.array(frozen if frozen else [""]),
        frozen_values=json.dumps({k: ref_sd[k].numpy().tolist() for k in frozen}),
        image_features=out["image_features"].detach().numpy(), text_features=out["text_features"].detach().numpy(),
        logit_scale=out["logit_scale"].detach().numpy(), logits_per_image=logits_i.detach().numpy(),
        loss=loss.detach().numpy(), grad_names=np.array(names), grad_norms=norms, grad_sums=sums,
        grad_sample_idx=idxs, grad_sample_vals=vals, **extra)
    print(f"{name}: loss={float(loss):.7f} params={len(shapes)} grads={len(names)}")


def _dist_worker(rank, world, port, B, E, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _, ref_loss, _ = ref_loader.load()
    rng = np.random.RandomState(77)
    img = torch.from_numpy(rng.standard_normal((world * B, E)).astype(np.float32))
    txt = torch.from_numpy(rng.standard_normal((world * B, E)).astype(np.float32))
    img, txt = O.l2_normalize(img), O.l2_normalize(txt)
    res = {}
    for local_loss, gwg in ((True, True), (True, False), (False, True), (False, False)):
        i = img[rank * B:(rank + 1) * B].clone().requires_grad_(True)
        t = txt[rank * B:(rank + 1) * B].clone().requires_grad_(True)
        s = torch.tensor(14.285714, requires_grad=True)
        fn = ref_loss.ClipLoss(local_loss=local_loss, gather_with_grad=gwg, cache_labels=True, rank=rank, world_size=world)
        loss = fn(i, t, s)
        loss.backward()
        res[f"{int(local_loss)}{int(gwg)}"] = (float(loss), i.grad.numpy(), t.grad.numpy(), float(s.grad))
    q.put((rank, img.numpy(), txt.numpy(), res))
    dist.barrier()
    dist.destroy_process_group()


def run_dist(world=2, B=8, E=16, port=29731):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dist_worker, args=(r, world, port, B, E, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join()
    got.sort(key=lambda x: x[0])
    arrays = {"img": got[0][1], "txt": got[0][2], "world": world, "B": B, "logit_scale": 14.285714}
    for rank, _, _, res in got:
        for key, (loss, gi, gt, gs) in res.items():
            arrays[f"loss_{key}_r{rank}"] = loss
            arrays[f"gi_{key}_r{rank}"] = gi
            arrays[f"gt_{key}_r{rank}"] = gt
            arrays[f"gs_{key}_r{rank}"] = gs
    np.savez_compressed(os.path.join(OUT, f"dist_loss_w{world}.npz"), **arrays)
    print(f"dist_loss_w{world}: losses", {k: v for k, v in arrays.items() if k.startswith("loss_")})


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    ref_model, ref_loss, _ = ref_loader.load()
    only = sys.argv[1:]                      # optional: regenerate just the named fixtures
    for name, spec in CASES.items():
        if not only or name in only:
            run_case(name, spec, ref_model, ref_loss)
    for name, spec in FULL_CASES.items():
        if not only or name in only or "full" in only:
            run_case(name, dict(spec, cfg=full_cfg(spec)), ref_model, ref_loss)
    if not only or "dist_loss_w2" in only:
        run_dist()
    if not only or "dist_loss_w8" in only:
        # the node size of BASELINE configs 3-5: eight ranks, an odd per-rank batch (labels offset by 3 * rank)
        run_dist(world=8, B=3, E=16, port=29733)


if __name__ == "__main__":
    main()
