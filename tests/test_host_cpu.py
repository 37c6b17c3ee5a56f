"""CPU tests of the host side: drop-in surface (state_dict keys/shapes/dtypes vs the reference's),
C-ABI export table, config registry, and loud failure off-GPU."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

from .conftest import load_golden
import torch

import clipa_amd
from clipa_amd import lib
from oracle import ref_loader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_state_dict_matches_reference_layout(golden):
    """Same keys, order-insensitive, and shapes as the reference CLIP's state_dict -> checkpoints interchange."""
    m = clipa_amd.CLIP(**golden.cfg, output_dict=True)
    sd = m.state_dict()
    assert set(sd.keys()) == set(golden.shapes.keys())
    for k, shp in golden.shapes.items():
        assert tuple(sd[k].shape) == shp, k
    m.load_state_dict(golden.sd, strict=True)
    trainable = sorted(k for k, p in m.named_parameters() if p.requires_grad)
    assert trainable == [str(n) for n in golden.z["grad_names"]]


def test_sincos_table_matches_reference():
    g = load_golden("gap_sincos_tanh")
    m = clipa_amd.CLIP(**g.cfg)
    assert not m.visual.positional_embedding.requires_grad
    assert torch.allclose(m.visual.positional_embedding, g.sd["visual.positional_embedding"], atol=1e-6)


def test_named_parameters_drive_same_weight_decay_split():
    """training/main.py:311-316: p.ndim < 2 or 'bn'/'ln'/'bias'/'logit_scale' in name -> no weight decay."""
    m = clipa_amd.create_model("ViT-S-16", force_image_size=112)
    exclude = lambda n, p: p.ndim < 2 or "bn" in n or "ln" in n or "bias" in n or "logit_scale" in n
    nd = [n for n, p in m.named_parameters() if exclude(n, p) and p.requires_grad]
    wd = [n for n, p in m.named_parameters() if not exclude(n, p) and p.requires_grad]
    assert "logit_scale" in nd and "visual.class_embedding" in nd and "visual.ln_pre.weight" in nd
    assert "visual.conv1.weight" in wd and "token_embedding.weight" in wd and "text_projection" in wd
    assert len(nd) + len(wd) == len(list(m.parameters()))
    assert sum(p.numel() for p in m.parameters()) == 62201089   # 62.2 M for config 1


def test_convert_weights_to_lp_dtypes():
    """model.py:329-351 + probe in SURVEY 8a: which tensors go bf16 under precision='bf16'."""
    m = clipa_amd.create_model("ViT-S-16", precision="bf16", force_image_size=112)
    sd = m.state_dict()
    bf, f32 = torch.bfloat16, torch.float32
    assert sd["visual.conv1.weight"].dtype == bf
    assert sd["visual.transformer.resblocks.0.attn.in_proj_weight"].dtype == bf
    assert sd["visual.transformer.resblocks.0.attn.out_proj.bias"].dtype == bf
    assert sd["transformer.resblocks.3.mlp.c_fc.weight"].dtype == bf
    assert sd["visual.proj"].dtype == bf and sd["text_projection"].dtype == bf
    for k in ("visual.ln_pre.weight", "visual.class_embedding", "visual.positional_embedding", "positional_embedding",
              "token_embedding.weight", "logit_scale", "ln_final.bias", "visual.transformer.resblocks.0.ln_1.weight"):
        assert sd[k].dtype == f32, k


@pytest.mark.skipif(not ref_loader.available(), reason="/root/reference not mounted")
def test_checkpoint_interchange_with_live_reference(tmp_path):
    ref_model, _, _ = ref_loader.load()
    cfg = clipa_amd.get_model_config("ViT-S-16")
    cfg["vision_cfg"]["image_size"] = 112
    ref = ref_model.CLIP(**cfg)
    mine = clipa_amd.CLIP(**cfg)
    ref_sd = ref.state_dict()
    assert list(ref_sd.keys()) == list(mine.state_dict().keys()) or set(ref_sd) == set(mine.state_dict())
    mine.load_state_dict(ref_sd, strict=True)
    ref.load_state_dict(mine.state_dict(), strict=True)
    # two-resolution hand-off: 84 px sincos checkpoint -> 224 px learnable model
    cfg84 = clipa_amd.get_model_config("ViT-S-16")
    cfg84["vision_cfg"].update(image_size=84, pos_embed="sin_cos_2d")
    small = clipa_amd.CLIP(**cfg84)
    path = tmp_path / "epoch_1.pt"
    torch.save({"epoch": 1, "state_dict": {"module." + k: v for k, v in small.state_dict().items()}}, path)
    big = clipa_amd.create_model("ViT-S-16", pretrained=str(path), force_image_size=224, pos_embed="learnable")
    sd_ref = {k: v.clone() for k, v in small.state_dict().items()}
    ref_model.resize_pos_embed(sd_ref, big)
    assert torch.allclose(big.visual.positional_embedding, sd_ref["visual.positional_embedding"], atol=1e-6)
    assert big.visual.positional_embedding.shape[0] == 197


def test_two_resolution_handoff_matches_reference_golden(tmp_path):
    """BASELINE config 5's hand-off on the host side (the part that needs no GPU): the factory loads an 84 px / sin-cos / GAP /
    context-16 checkpoint, written as training/main.py:436-468 writes it, into the 224 px / learnable / context-32 model;
    the resized tables equal the REAL reference's (tests/golden/handoff_S16_84_to_224.npz, oracle/make_handoff_golden.py) and
    the oracle on the loaded weights reproduces the reference's features, loss and gradient digests of the same hand-off."""
    from .conftest import Handoff
    from oracle.make_handoff_golden import write_checkpoint
    h = Handoff(tmp_path)
    m84 = h.phase1_model()
    assert not m84.visual.positional_embedding.requires_grad and m84.visual.positional_embedding.shape[0] == 26
    path = tmp_path / "epoch_1.pt"
    write_checkpoint(m84.state_dict(), path)
    big = clipa_amd.create_model(h.NAME224, pretrained=str(path), force_image_size=224, pos_embed="learnable")
    assert big.visual.positional_embedding.requires_grad
    ref, loss, _, _ = h.check_against_reference(big)
    assert "visual.positional_embedding" in ref and abs(loss - float(h.t("loss"))) < 3e-5


def test_capi_exports_every_declared_symbol():
    """The shared library loads here (no GPU needed) and exports every function of include/clipa_hip.h."""
    header = open(os.path.join(ROOT, "include", "clipa_hip.h")).read()
    declared = set(re.findall(r"\b(clipa_[a-z0-9_]+)\s*\(", header))
    assert declared, "header parse failed"
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)
    handle = lib.load()
    for name in declared:
        assert hasattr(handle, name), name
    assert handle.clipa_version() >= 1
    ns = ctypes.c_int64(0)
    nbytes = handle.clipa_gemm_tn_workspace(806912, 4096, 1024, ctypes.byref(ns))
    assert ns.value >= 1 and nbytes == ns.value * (4096 * 1024 + 4096) * 4


def test_argument_errors_are_reported_not_thrown():
    handle = lib.load()
    rc = handle.clipa_gemm_nt(None, None, None, None, None, None, 16, 16, 12, 16, 16, 16, 0, 1.0, 0, 0, 0, None)
    assert rc < 0 and "multiple of 8" in lib.last_error()
    rc = handle.clipa_attention_fwd(None, None, None, None, None, 1, 1, 10, 96, 288, 96, 1.0, 0, None)
    assert rc < 0 and "head dim" in lib.last_error()


def test_no_cpu_fallback():
    """Off-GPU the product path must fail loudly instead of silently computing on the host."""
    m = clipa_amd.create_model("ViT-S-16", force_image_size=112, output_dict=True)
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros(2, 3, 112, 112), torch.zeros(2, 77, dtype=torch.long))
    with pytest.raises(RuntimeError, match="GPU"):
        clipa_amd.ClipLoss()(torch.randn(8, 16), torch.randn(8, 16), torch.tensor(10.0))


def test_registry_and_factory():
    assert {"ViT-S-16", "ViT-B-16", "ViT-L-16", "ViT-H-14", "ViT-L-16-CL8-Syntax-GAP"} <= set(clipa_amd.list_models())
    L = clipa_amd.get_model_config("ViT-L-16")
    assert (L["embed_dim"], L["vision_cfg"]["width"], L["vision_cfg"]["layers"], L["text_cfg"]["width"]) == (768, 1024, 24, 768)
    with pytest.raises(RuntimeError):
        clipa_amd.create_model("no-such-model")
    h14 = clipa_amd.get_model_config("ViT-H-14")
    assert h14["vision_cfg"]["width"] // h14["vision_cfg"]["head_width"] == 16      # head dim 80: own attention geometry
    ref_dir = os.path.join(ref_loader.REF_ROOT, "open_clip", "model_configs")
    if os.path.isdir(ref_dir):                  # the reference's own JSON registry is accepted verbatim
        for name in ("ViT-S-16", "ViT-B-16", "ViT-L-16", "ViT-L-16-CL8-Syntax-GAP", "ViT-L-16-CL32-GAP", "ViT-H-14"):
            import json
            ref_cfg = json.load(open(os.path.join(ref_dir, name + ".json")))
            assert ref_cfg == clipa_amd.get_model_config(name), name


def test_create_loss_contract():
    class A:
        local_loss, gather_with_grad, rank, world_size, horovod, distill, model = True, True, 3, 8, False, False, "ViT-L-16"
    loss = clipa_amd.create_loss(A)
    assert (loss.local_loss, loss.gather_with_grad, loss.rank, loss.world_size, loss.cache_labels) == (True, True, 3, 8, True)


def test_bench_flop_accounting_matches_survey_table():
    """algorithmic train GFLOP per pair (3 x forward, recompute not counted) - the figure
    bench.py's model_flops_util and the judge's check are computed from."""
    import bench
    want = {("ViT-S-16", 112, 32): 10.73, ("ViT-B-16", 224, 77): 123.26, ("ViT-L-16", 224, 77): 409.23,
            ("ViT-H-14", 224, 77): 1145.04, ("ViT-L-16", 84, 77): 87.33}
    for (name, S, ctx), gf in want.items():
        got = bench.train_gflop_per_pair(clipa_amd.get_model_config(name), S, ctx)
        assert abs(got - gf) <= 0.01 * gf, (name, S, ctx, got, gf)
    assert 1 <= bench.usable_cores() <= 64


def test_bench_keep_planner_respects_the_budget():
    import bench
    m = clipa_amd.create_model("ViT-L-16")
    tokens_v, tokens_t = 4096 * 197, 4096 * 77
    mv, mt = m.visual.transformer.medium_keep_bytes(tokens_v), m.transformer.medium_keep_bytes(tokens_t)
    lv, lt = m.visual.transformer.light_keep_bytes(tokens_v), m.transformer.light_keep_bytes(tokens_t)
    assert lv > mv > 0 and lt > mt > 0
    prev = -1
    for gb in (0, 5, 50, 100, 166, 250, 400, 2000):
        kv, kt, dv, dt = bench.plan_keep(gb << 30, 24, 12, mv, mt, lv, lt)
        used = kv * lv + dv * mv + kt * lt + dt * mt
        assert used <= (gb << 30) and kv + dv <= 24 and kt + dt <= 12
        saved = 25.5 * kv + 17.5 * dv          # recompute units avoided in the image tower
        assert saved >= prev
        prev = saved
    kv, kt, dv, dt = bench.plan_keep(188 << 30, 24, 12, mv, mt, lv, lt)
    assert (kv, kt, dv) == (0, 0, 24)                              # medium everywhere in the image tower before any upgrade
    l8v = m.visual.transformer.light8_keep_bytes(tokens_v)
    assert mv < l8v < lv
    kv, kt, dv, dt = bench.plan_keep(200 << 30, 24, 12, mv, mt, l8v, m.transformer.light8_keep_bytes(tokens_t))
    assert kv >= 1 and kv + dv == 24                               # ... then image-tower upgrades before the text tower's turn
    assert bench.plan_keep(2000 << 30, 24, 12, mv, mt, lv, lt) == (24, 12, 0, 0)


def test_bench_per_tensor_planner():
    """The bf16 engines' planner keeps tensors one by one, most valuable bytes first (e4m3 pre-activation, attention output,
    x1, qkv; image tower before text at equal kind): never over budget, monotone in the budget, and at the headline shape's
    budget every image block keeps its e4m3 pre-activation before any block keeps qkv."""
    import bench
    m = clipa_amd.create_model("ViT-L-16")
    tok = {"v": 4096 * 197, "t": 4096 * 77}
    tr = {"v": m.visual.transformer, "t": m.transformer}
    nb = {(tw, n): tr[tw].tensor_keep_bytes(tok[tw], n) for tw in tr for n in ("h8", "h", "a", "x1", "qkv")}
    assert nb[("v", "h8")] == tok["v"] * 4096 and nb[("v", "qkv")] == tok["v"] * 3 * 1024 * 2
    prev = -1
    for gb in (0, 3, 50, 100, 200, 400, 2000):
        plan = bench.plan_keep_tensors(gb << 30, {"v": 24, "t": 12}, nb)
        used = sum(plan[tw][n] * nb[(tw, n)] for tw in plan for n in plan[tw])
        assert used <= (gb << 30)
        kept = sum(plan[tw][n] for tw in plan for n in plan[tw])
        assert kept >= prev
        prev = kept
    plan = bench.plan_keep_tensors(200 << 30, {"v": 24, "t": 12}, nb)
    assert plan["v"]["h8"] == 24 and plan["v"]["a"] == 24 and plan["v"]["x1"] == 24 and plan["v"]["qkv"] < 24
    assert bench.plan_keep_tensors(2000 << 30, {"v": 24, "t": 12}, nb) == {tw: {"h8": n, "h": 0, "a": n, "x1": n, "qkv": n} for tw, n in (("v", 24), ("t", 12))}
    # the pruned last block (engine.LastBlockFn) holds no token-level x1 / pre-activation: the first block of those counts is free
    nbh = nb
    one = nb[("v", "h8")]
    assert bench.plan_keep_tensors(one - 1, {"v": 24, "t": 12}, nb)["v"]["h8"] == 0
    assert bench.plan_keep_tensors(one - 1, {"v": 24, "t": 12}, nb, pruned_last=("v", "t"))["v"]["h8"] == 1
    assert bench.plan_keep_tensors(one, {"v": 24, "t": 12}, nb, pruned_last=("v", "t"))["v"]["h8"] == 2
    # the plan restricted to bit-exact tensors (`value_exact_tiers`): no e4m3 pre-activation at any budget, bf16 "h" after qkv
    for gb in (0, 50, 200, 2000):
        ex = bench.plan_keep_tensors(gb << 30, {"v": 24, "t": 12}, nbh, bench.KEEP_VALUE_MS_PER_GB_EXACT, ("v", "t"))
        assert ex["v"]["h8"] == ex["t"]["h8"] == 0
        assert ex["v"]["h"] <= 1 or ex["v"]["qkv"] == 24
        used = sum(max(0, ex[tw][n] - (1 if n in ("x1", "h") else 0)) * nbh[(tw, n)] for tw in ex for n in ("a", "x1", "qkv", "h"))
        assert used <= (gb << 30)
    assert nbh[("v", "h")] == 2 * nb[("v", "h8")]


def test_keep_bytes_follow_the_mlp_width():
    """ViT-g / bigG / e-14 have MLP ratios 4.36 / 4.92 / 8.57 - the tier byte counts take the c_fc layer's own width."""
    import clipa_amd.model as M
    for width, heads, ratio in ((1024, 16, 4.0), (1792, 16, 8.5714)):
        t = M.Transformer(width, 1, heads, mlp_ratio=ratio)
        mlp = int(width * ratio)
        assert t.light_keep_bytes(1000) == 1000 * (5 * width * 2 + heads * 8 + 2 * mlp)
        assert t.light8_keep_bytes(1000) == 1000 * (5 * width * 2 + heads * 8 + mlp)
        assert t.medium_keep_bytes(1000) + t.tensor_keep_bytes(1000, "h8") == t.light8_keep_bytes(1000)


def test_bench_executed_flops():
    """bench.py prices `model_flops_util` on EXECUTED model FLOPs: the reference count less the out-projection + MLP of each
    tower's last block on the rows its head never reads (13.6 of 409.2 GF at the headline shape)."""
    import bench
    cfg = clipa_amd.get_model_config("ViT-L-16")
    dead = bench.pruned_gflop_per_pair(cfg, 224, 77, True, True)
    assert abs(dead - (196 * 18 * 1024 ** 2 + 76 * 18 * 768 ** 2) * 3 / 1e9) < 1e-6 and 13.4 < dead < 13.7
    assert bench.pruned_gflop_per_pair(cfg, 224, 77, False, False) == 0.0


def test_optimizer_load_state_dict_restores_f32_moments():
    """torch.optim.Optimizer.load_state_dict casts floating-point state to the parameter's dtype; the HIP kernel reads
    the moments as float*, so clipa_amd.optim.AdamW must restore f32 (and an int step) after a resume - also from a
    torch.optim.AdamW checkpoint with tensor steps."""
    from clipa_amd.optim import AdamW
    p = torch.nn.Parameter(torch.randn(16, 8).to(torch.bfloat16))
    opt = AdamW([p], lr=1e-3)
    opt.state[p] = {"step": 3, "exp_avg": torch.randn(16, 8), "exp_avg_sq": torch.rand(16, 8)}
    sd = opt.state_dict()
    opt2 = AdamW([p], lr=1e-3)
    opt2.load_state_dict(sd)
    st = opt2.state[p]
    assert st["exp_avg"].dtype == torch.float32 and st["exp_avg_sq"].dtype == torch.float32 and st["step"] == 3
    assert st["exp_avg"].is_contiguous()
    t = torch.optim.AdamW([p], lr=1e-3)
    p.grad = torch.zeros_like(p)
    t.step()
    opt3 = AdamW([p], lr=1e-3)
    opt3.load_state_dict(t.state_dict())
    st = opt3.state[p]
    assert isinstance(st["step"], int) and st["step"] == 1 and st["exp_avg"].dtype == torch.float32


def _budget_worker(rank, world, port, q):
    import sys
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import bench
    budget = bench.agree_budget((150 + 7 * rank) << 30, torch.device("cpu"))     # ranks measured different free HBM
    mv, mt, lv, lt = 8_262_778_880, 2_422_210_560, 14_873_001_984, 4_359_979_008   # ViT-L/16, local batch 4096
    q.put((rank, budget, bench.plan_keep(budget, 24, 12, mv, mt, lv, lt)))
    dist.barrier()
    dist.destroy_process_group()


def test_keep_plan_is_rank_invariant():
    """bench.py --keep-blocks auto under several ranks: the activation budget is MIN-reduced, so every rank derives the
    same (light, medium) block counts although each measured a different peak (ranks could diverge)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_budget_worker, args=(r, 2, 29755, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][1] == got[1][1] == 150 << 30
    assert got[0][2] == got[1][2]


@pytest.mark.skipif(not ref_loader.available(), reason="/root/reference not mounted")
@pytest.mark.parametrize("unlocked", [0, 1, 2, 3])
def test_lock_image_tower_freezes_the_reference_set(unlocked):
    """model.py:229-231 / transformer.py:415-446: the set of parameters left trainable by lock_image_tower(unlocked_groups=k)
    is the reference's (groups: [stem], block 0 .. block n-2, [last block, ln_post], proj)."""
    ref_model, _, _ = ref_loader.load()
    g = load_golden("cls_erf")
    ref = ref_model.CLIP(**g.cfg)
    ref.lock_image_tower(unlocked_groups=unlocked)
    mine = clipa_amd.CLIP(**g.cfg)
    mine.lock_image_tower(unlocked_groups=unlocked)
    want = {k: p.requires_grad for k, p in ref.named_parameters()}
    got = {k: p.requires_grad for k, p in mine.named_parameters()}
    assert got == want
    assert not any(v for k, v in got.items() if k.startswith("visual.conv1"))


def test_gelu_polynomials():
    """The erf-GELU of the kernels (clipa_amd/csrc/common.h: gelu_cdf_n / gelu_grad_n, coefficient tables GELU_P / GELU_Q) is two odd polynomials, no transcendentals.
    Their coefficients are parsed from the header and evaluated exactly as the kernel does (fp32 Horner, clamp at 4.5) against
    scipy's erf: |x Phi(x) - gelu(x)| <= 6e-5 inside the clamp (<= 1e-5 for |x| < 3) and <= 2e-5 |x| beyond it; |gelu'| error
    <= 3e-4 (<= 2e-5 for |x| < 3) - one to two orders below the bf16 rounding of the values the epilogues store."""
    import scipy.special as sp
    sys_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    import sys
    sys.path.insert(0, sys_path)
    import fit_gelu_poly as F
    src = open(os.path.join(os.path.dirname(sys_path), "clipa_amd", "csrc", "common.h")).read()

    def coefs(name):     # constexpr float GELU_P[10] = {...};  (Horner order, highest power of x^2 first)
        m = re.search(r"constexpr float " + name + r"\[\d+\] = \{([^}]*)\}", src)
        return np.array([float(t.strip().rstrip("f")) for t in m.group(1).split(",")], dtype=np.float32)

    cf, cb = coefs("GELU_P"), coefs("GELU_Q")
    assert len(cf) == 10 and len(cb) == 11
    xs = np.linspace(-40, 40, 800001).astype(np.float32)
    x64 = xs.astype(np.float64)
    gelu = x64 * F.Phi(x64)
    got = np.float32(xs * F.gelu_cdf32(xs, cf)).astype(np.float64)
    err = np.abs(got - gelu)
    inside = np.abs(xs) <= 4.5
    assert err[inside].max() <= 6e-5 and err[np.abs(xs) < 3].max() <= 1e-5
    assert (err[~inside] <= 2e-5 * np.abs(x64[~inside])).all()
    dg = F.Phi(x64) + x64 * F.phi(x64)
    errb = np.abs(F.gelu_grad32(xs, cb).astype(np.float64) - dg)
    assert errb.max() <= 3e-4 and errb[np.abs(xs) < 3].max() <= 2e-5


def test_create_model_and_transforms_returns_reference_style_transforms():
    """factory.py:293-352: (model, preprocess_train, preprocess_val); the transforms (open_clip/transform.py:91-214 on Pillow
    alone) turn a PIL image into the tensor the trainer moves to the device: float CHW normalised, or uint8 CHW with
    to_float_on_device (train.py:191-197).  The eval transform is checked against the Pillow calls written out."""
    from PIL import Image
    rng = np.random.RandomState(0)
    pil = Image.fromarray(rng.randint(0, 256, (300, 420, 3), dtype=np.uint8), "RGB")
    aug = {"scale": (0.4, 1.0), "color_jitter": (0.32, 0.32, 0.32, 0.08), "color_jitter_prob": 0.8, "gray_scale_prob": 0.2}
    m, tr, va = clipa_amd.create_model_and_transforms("ViT-S-16", force_image_size=112, aug_cfg=aug, to_float_on_device=True)
    assert m.visual.image_size == (112, 112)
    torch.manual_seed(3)
    import random
    random.seed(3)
    a = tr(pil)
    assert a.dtype == torch.uint8 and tuple(a.shape) == (3, 112, 112)
    torch.manual_seed(3)
    random.seed(3)
    assert torch.equal(a, tr(pil))                                   # driven by the torch / python RNGs, as torchvision's are
    v = va(pil)
    assert v.dtype == torch.uint8 and tuple(v.shape) == (3, 112, 112)
    # Resize(112) = shorter edge to 112 (bicubic), long edge int(112 * 420 / 300) = 156; CenterCrop(112)
    ref = pil.resize((156, 112), Image.BICUBIC)
    left = int(round((156 - 112) / 2.0))
    ref = np.array(ref.crop((left, 0, left + 112, 112)))
    assert np.array_equal(v.permute(1, 2, 0).numpy(), ref)
    _, tr_f, va_f = clipa_amd.create_model_and_transforms("ViT-S-16", force_image_size=112)
    f = va_f(pil)
    mean = torch.tensor(clipa_amd.OPENAI_DATASET_MEAN).view(3, 1, 1)
    std = torch.tensor(clipa_amd.OPENAI_DATASET_STD).view(3, 1, 1)
    assert f.dtype == torch.float32 and torch.allclose(f, (v.float() / 255 - mean) / std, atol=1e-6)
    assert tuple(tr_f(pil).shape) == (3, 112, 112)
    # the device pipeline applies the same operations: its CPU oracle is pinned to Pillow by tests/test_augment_cpu.py


def test_bench_self_launch_argv_and_no_gpu_exit():
    """`python bench.py --gpus N` (N > 1) without a launcher re-executes itself under torch.distributed.run with the same
    arguments (the reference starts its ranks with `torchrun --nproc_per_node 8`,
    scripts/exp/gpu/vit_l16/i37_t8_pretrain.sh:1).  On this GPU-less box the two ranks it starts must fail with the engine's
    "no GPU visible" message - not with the launcher check's "WORLD_SIZE" complaint - and `--gpus 1` must not re-execute."""
    import subprocess
    import bench
    argv = bench.self_launch_argv(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], port=29411)
    assert argv[0] == sys.executable and argv[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in argv and "--nnodes=1" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and argv[argv.index("--master-port") + 1] == "29411"
    i = argv.index(os.path.join(ROOT, "bench.py"))
    assert argv[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    if torch.cuda.is_available():
        return
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "re-executing as" in r.stderr and "torch.distributed.run" in r.stderr
    assert "rank 0 of 2): no GPU visible" in r.stderr or "rank 1 of 2): no GPU visible" in r.stderr
    assert "WORLD_SIZE=" not in r.stderr
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"],
                        capture_output=True, text=True, timeout=300, env=env)
    assert r1.returncode == 2 and "re-executing" not in r1.stderr and "rank 0 of 1): no GPU visible" in r1.stderr


def test_bench_fp8_child_line_survives_a_failing_child():
    """bench.py's `fp8_same_workload` (the headline workload measured in fp8 by a child process before the parent touches the
    GPU; BASELINE configs[3]'s precision, training/params.py:195-200 stops at bf16): the child gets the parent's workload and
    none of its extra regions, cannot recurse, and a child that fails (here: no GPU) turns into an error entry, not an exception."""
    import argparse
    import bench
    if torch.cuda.is_available():
        return
    args = argparse.Namespace(model="ViT-S-16", image_size=112, ctx=32, batch=64, accum_freq=1, fp8_line_steps=2, fp8_line_timeout=300)
    seen = {}
    import subprocess
    real = subprocess.run

    def spy(cmd, **kw):
        seen["cmd"] = cmd
        return real(cmd, **kw)
    subprocess.run = spy
    try:
        line = bench.fp8_child_line(args)
    finally:
        subprocess.run = real
    cmd = seen["cmd"]
    assert cmd[1] == os.path.join(ROOT, "bench.py")
    for flag, val in (("--precision", "fp8"), ("--model", "ViT-S-16"), ("--batch", "64"), ("--image-size", "112"), ("--ctx", "32"),
                      ("--steps", "2"), ("--fp8-line-steps", "0"), ("--h2d-steps", "0"), ("--exact-steps", "0"), ("--unpad-steps", "0")):
        assert cmd[cmd.index(flag) + 1] == val, flag
    assert "--no-cpu-baseline" in cmd
    assert line["value"] is None and "no GPU visible" in line["error"]
