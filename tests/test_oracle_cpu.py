"""CPU tests (no GPU): the oracle restatement is pinned against the golden vectors generated from the
real reference (oracle/make_golden.py), and - when /root/reference is mounted - against the live
reference modules."""
import numpy as np
import pytest
import torch

from oracle import clip_oracle as O
from oracle import ref_loader


def _oracle_run(g, dtype=torch.float32):
    sd = {k: v.to(dtype).clone().requires_grad_(k not in g.frozen) for k, v in g.sd.items()}
    images = O.normalize_images(g.images_u8, dtype=dtype)
    i, t, s = O.clip_forward(sd, g.ocfg, images, g.texts)
    loss, logits = O.clip_loss(i, t, s)
    loss.backward()
    grads = {k: v.grad for k, v in sd.items() if v.grad is not None}
    return i, t, s, loss, logits, grads


def test_oracle_matches_reference_golden(golden):
    _check_oracle_against_golden(golden)


def test_oracle_matches_reference_golden_full_dims(golden_full):
    """The same pin at BASELINE model dimensions (24 / 12 / 32 layers, widths 1024 / 768 / 1280, 197 / 257 / 26 image
    tokens, 77 text tokens): every feature, the loss and a digest of every parameter gradient of the real reference."""
    _check_oracle_against_golden(golden_full, feat_atol=1e-5, logit_atol=2e-4, loss_atol=3e-5)


def _check_oracle_against_golden(g, feat_atol=3e-6, logit_atol=5e-5, loss_atol=1e-5):
    i, t, s, loss, logits, grads = _oracle_run(g)
    assert torch.allclose(i, g.t("image_features"), atol=feat_atol, rtol=1e-5), float((i - g.t("image_features")).abs().max())
    assert torch.allclose(t, g.t("text_features"), atol=feat_atol, rtol=1e-5), float((t - g.t("text_features")).abs().max())
    assert abs(float(s) - float(g.t("logit_scale"))) < 1e-4
    assert torch.allclose(logits, g.t("logits_per_image"), atol=logit_atol, rtol=1e-5)
    assert abs(float(loss) - float(g.t("loss"))) < loss_atol
    names = [str(n) for n in g.z["grad_names"]]
    assert sorted(grads) == names, "oracle must produce a gradient for exactly the reference's trainable set"
    for j, n in enumerate(names):
        gr = grads[n].double().reshape(-1)
        ref_norm = float(g.z["grad_norms"][j])
        assert abs(float(gr.norm()) - ref_norm) <= 2e-4 * ref_norm + 1e-7, n
        assert abs(float(gr.sum()) - float(g.z["grad_sums"][j])) <= 5e-4 * ref_norm + 1e-6, n
        idx = torch.from_numpy(g.z["grad_sample_idx"][j])
        assert np.allclose(gr[idx].numpy(), g.z["grad_sample_vals"][j], atol=2e-4 * ref_norm + 1e-7), n


def test_oracle_patch_dropout_matches_reference_golden():
    """PatchDropout (transformer.py:53-83, applied at :501-502) in training mode: the oracle with the reference's kept indices
    reproduces the real reference's features, logits, loss and every gradient digest; and the engine-side index draw
    (clipa_amd.model.patch_dropout_indices: torch.randn from the global CPU generator + top-k, as transformer.py:79-80)
    reproduces those indices from the fixture's seed."""
    from .conftest import load_golden
    import clipa_amd.model as M
    g = load_golden("patchdrop_gap")
    assert g.patch_keep is not None and tuple(g.patch_keep.shape) == (8, 8)
    torch.manual_seed(g.drop_seed)
    assert torch.equal(M.patch_dropout_indices(8, 16, 0.5), g.patch_keep)
    sd = {k: v.clone().requires_grad_(k not in g.frozen) for k, v in g.sd.items()}
    i, t, s = O.clip_forward(sd, g.ocfg, O.normalize_images(g.images_u8), g.texts, patch_keep=g.patch_keep)
    loss, logits = O.clip_loss(i, t, s)
    loss.backward()
    assert torch.allclose(i, g.t("image_features"), atol=3e-6, rtol=1e-5)
    assert torch.allclose(logits, g.t("logits_per_image"), atol=5e-5, rtol=1e-5)
    assert abs(float(loss) - float(g.t("loss"))) < 1e-5
    names = [str(n) for n in g.z["grad_names"]]
    grads = {k: v.grad for k, v in sd.items() if v.grad is not None}
    assert sorted(grads) == names
    for j, n in enumerate(names):
        gr = grads[n].double().reshape(-1)
        ref_norm = float(g.z["grad_norms"][j])
        assert abs(float(gr.norm()) - ref_norm) <= 2e-4 * ref_norm + 1e-7, n
        idx = torch.from_numpy(g.z["grad_sample_idx"][j])
        assert np.allclose(gr[idx].numpy(), g.z["grad_sample_vals"][j], atol=2e-4 * ref_norm + 1e-7), n
    # without the indices (eval mode) the oracle is the plain forward: different features
    i0, _, _ = O.clip_forward(g.sd, g.ocfg, O.normalize_images(g.images_u8), g.texts)
    assert (i0 - g.t("image_features")).abs().max() > 1e-3


def test_oracle_fp64_agrees_with_fp32(golden):
    i32, t32, _, l32, _, _ = _oracle_run(golden, torch.float32)
    i64, t64, _, l64, _, _ = _oracle_run(golden, torch.float64)
    assert torch.allclose(i32.double(), i64, atol=5e-6)
    assert abs(float(l32) - float(l64)) < 1e-5


def test_emulated_bf16_oracle_is_close_to_fp32(golden):
    """The bf16-rounding emulation (what the engine stores) must stay within bf16 noise of fp32."""
    g = golden
    images = O.normalize_images(g.images_u8)
    i, t, s = O.clip_forward(g.sd, g.ocfg, images, g.texts)
    ie, te, _ = O.clip_forward(g.sd, g.ocfg, images, g.texts, emulate_bf16=True)
    assert (i - ie).abs().max() < 3e-2 and (t - te).abs().max() < 3e-2


def test_dist_loss_golden():
    """Reference ClipLoss under a real 2-rank gloo group == single-process restatement on the concatenated
    batch (all four local_loss x gather_with_grad variants)."""
    z = np.load("tests/golden/dist_loss_w2.npz")
    W, B = int(z["world"]), int(z["B"])
    s0 = float(z["logit_scale"])
    for local_loss in (True, False):
        for gwg in (True, False):
            key = f"{int(local_loss)}{int(gwg)}"
            for r in range(W):
                img = torch.from_numpy(z["img"]).clone().requires_grad_(True)
                txt = torch.from_numpy(z["txt"]).clone().requires_grad_(True)
                s = torch.tensor(s0, requires_grad=True)
                if gwg:
                    # differentiable gather: grads of every rank's loss flow into our rows (sum over ranks)
                    total = sum(O.clip_loss_rank(img, txt, s, rr, W, local_loss) for rr in range(W))
                    own = O.clip_loss_rank(img, txt, s, r, W, local_loss)
                    total.backward()
                    gi, gt = img.grad[r * B:(r + 1) * B], txt.grad[r * B:(r + 1) * B]
                else:
                    # gathered copies carry no grad: only this rank's own rows (as local operand /
                    # re-inserted slice) are differentiated
                    det_i, det_t = img.detach().clone(), txt.detach().clone()
                    li = img[r * B:(r + 1) * B]
                    lt = txt[r * B:(r + 1) * B]
                    if local_loss:
                        logits_i = s * li @ det_t.T
                        logits_t = s * lt @ det_i.T
                        labels = torch.arange(B) + B * r
                    else:
                        ai = torch.cat([det_i[:r * B], li, det_i[(r + 1) * B:]])
                        at = torch.cat([det_t[:r * B], lt, det_t[(r + 1) * B:]])
                        logits_i = s * ai @ at.T
                        logits_t = logits_i.T
                        labels = torch.arange(W * B)
                    own = (torch.nn.functional.cross_entropy(logits_i, labels) +
                           torch.nn.functional.cross_entropy(logits_t, labels)) / 2
                    own.backward()
                    gi, gt = img.grad[r * B:(r + 1) * B], txt.grad[r * B:(r + 1) * B]
                assert abs(float(own) - float(z[f"loss_{key}_r{r}"])) < 2e-5, (key, r)
                assert np.allclose(gi.numpy(), z[f"gi_{key}_r{r}"], atol=2e-6), (key, r)
                assert np.allclose(gt.numpy(), z[f"gt_{key}_r{r}"], atol=2e-6), (key, r)


@pytest.mark.skipif(not ref_loader.available(), reason="/root/reference not mounted")
def test_oracle_matches_live_reference_bf16_weights():
    """Extra pin where the reference is importable: a fresh random init of the reference model."""
    ref_model, ref_loss, _ = ref_loader.load()
    cfg = {"embed_dim": 64,
           "vision_cfg": {"image_size": 64, "layers": 2, "width": 128, "patch_size": 16},
           "text_cfg": {"context_length": 12, "vocab_size": 300, "width": 128, "heads": 2, "layers": 2}}
    torch.manual_seed(5)
    m = ref_model.CLIP(**cfg, output_dict=True).float()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    img_u8, txt = O.synthetic_batch(6, 64, 12, 300, 3)
    images = O.normalize_images(img_u8)
    out = m(images, txt)
    i, t, s = O.clip_forward(sd, O.oracle_cfg(cfg), images, txt)
    assert torch.allclose(i, out["image_features"], atol=3e-6)
    assert torch.allclose(t, out["text_features"], atol=3e-6)
    ref = ref_loss.ClipLoss()(out["image_features"], out["text_features"], out["logit_scale"])
    mine, _ = O.clip_loss(i, t, s)
    assert abs(float(ref) - float(mine)) < 1e-5
