"""GPU parity tests, kernel by kernel, through the C ABI (clipa_amd.ops -> libclipa_hip.so) against the
CPU oracle maths (oracle/clip_oracle.py) / plain fp32 torch on the same seeded inputs.

Tolerances (stated per check): GEMM-type outputs are compared against an fp64-accumulated product of the
SAME bf16 operands, so the only differences are fp32 accumulation order and the final bf16 rounding
(<= 2^-8 relative); HBM-bound fp32 kernels are held to ~1e-5.
"""
import math

import numpy as np
import pytest
import torch

from oracle import clip_oracle as O

pytestmark = pytest.mark.gpu
bf16, f32 = torch.bfloat16, torch.float32
DEV = "cuda"


def ops():
    from clipa_amd import ops as _ops
    return _ops


def rnd(*shape, seed=0, scale=1.0, dtype=bf16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def check(name, got, ref, rtol, atol):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(got).all(), f"{name}: non-finite output"
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    if bad.any():
        i = int(torch.argmax((err - tol).reshape(-1)))
        idx = np.unravel_index(i, tuple(got.shape)) if got.dim() else ()
        raise AssertionError(f"{name}: {int(bad.sum())}/{got.numel()} outside tol; worst at {idx}: got "
                             f"{got.reshape(-1)[i].item():.6g} ref {ref.reshape(-1)[i].item():.6g} "
                             f"(max abs err {err.max().item():.3g})")


def ref_act(x, act):
    return O.activation(x, {0: "gelu_erf", 1: "gelu_tanh", 2: "quick_gelu"}[act])


# -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 264, 136), (1000, 768, 1024), (77, 2304, 768), (512, 512, 3072)])
def test_gemm_nt_plain_bias_f32out(M, N, K):
    a, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)     # asymmetric operands
    bias = rnd(N, seed=3, dtype=f32)
    ref = a.double() @ b.double().T * 0.5 + bias.double()
    out = ops().gemm_nt(a.to(DEV), b.to(DEV), bias.to(DEV), alpha=0.5)
    check("bf16 out", out, ref, 2 ** -7, 2e-3)
    out32 = ops().gemm_nt(a.to(DEV), b.to(DEV), bias.to(DEV), alpha=0.5, out_f32=True)
    check("f32 out", out32, ref, 1e-4, 2e-4 * math.sqrt(K))


def test_gemm_nt_strided_operands():
    M, N, K = 320, 264, 128
    big_a, big_b = rnd(M, 3 * K, seed=4).to(DEV), rnd(N, 2 * K, seed=5, scale=0.05).to(DEV)
    a, b = big_a[:, K:2 * K], big_b[:, K:]
    out = ops().gemm_nt(a, b, out_f32=True)
    check("strided", out, a.double().cpu() @ b.double().cpu().T, 1e-4, 3e-3)


@pytest.mark.parametrize("act", [0, 1, 2])
def test_gemm_nt_epilogues(act):
    M, N, K = 520, 384, 256
    a, b = rnd(M, K, seed=6), rnd(N, K, seed=7, scale=0.08)
    bias, aux = rnd(N, seed=8, dtype=f32), rnd(M, N, seed=9)
    v = (a.double() @ b.double().T + bias.double()).to(bf16).double()       # engine rounds the GEMM output to bf16
    A, B, BIAS, AUX = a.to(DEV), b.to(DEV), bias.to(DEV), aux.to(DEV)
    out, pre = ops().gemm_nt(A, B, BIAS, epi=ops().EPI_ACT, act=act, want_pre=True)
    check("pre-activation", pre, v, 2 ** -7, 2e-3)
    check("act", out, ref_act(pre.double().cpu(), act), 2 ** -7, 2e-3)
    out = ops().gemm_nt(A, B, BIAS, epi=ops().EPI_ADD, aux=AUX)
    check("residual add", out, v + aux.double(), 2 ** -7, 8e-3)
    x = aux.double().clone().requires_grad_(True)
    ref_act(x, act).sum().backward()
    out = ops().gemm_nt(A, B, BIAS, epi=ops().EPI_DACT, act=act, aux=AUX)
    check("act backward", out, v * x.grad, 2 ** -6, 6e-3)


def _debug_set(variant, abl):
    from clipa_amd import lib
    lib.debug_set(variant, abl)          # csrc/internal_hooks.h: not part of the C ABI, enabled per process through the environment


def _last_gemm():
    from clipa_amd import lib
    return lib.last_gemm()


NTA, NT2, TNA = 2, 1, 5                  # kernel families reported by lib.last_gemm()


@pytest.mark.parametrize("M,N,K", [(1024, 768, 512), (2304, 512, 256), (512, 1024, 1152)])
def test_gemm_nta_small_shapes_every_epilogue_bitwise_vs_nt2(M, N, K):
    """The four-wave hand-scheduled kernel (gemm_nta.hip) on SMALL whole-tile shapes - several tiles per persistent workgroup
    when there are fewer workgroups than tiles is exercised by the production-width tests; here 8-18 tiles, 4-18 K steps,
    every epilogue, against fp64 AND bit for bit against gemm_nt2 (the round-1/2 kernel it replaced), and the dispatch itself:
    a silent regression of whole-tile shapes to gemm_nt2 would otherwise pass every value test."""
    o = ops()
    a, b = rnd(M, K, seed=M + 1), rnd(N, K, seed=N + 2, scale=0.06)
    bias, aux = rnd(N, seed=3, dtype=f32), rnd(M, N, seed=4)
    A, B, BIAS, AUX = a.to(DEV), b.to(DEV), bias.to(DEV), aux.to(DEV)
    lin = a.double() @ b.double().T * 0.5 + bias.double()

    def run():
        act, pre = o.gemm_nt(A, B, BIAS, alpha=0.5, epi=o.EPI_ACT, act=0, want_pre=True)
        return (o.gemm_nt(A, B, BIAS, alpha=0.5), act, pre, o.gemm_nt(A, B, BIAS, alpha=0.5, epi=o.EPI_ACT, act=1),
                o.gemm_nt(A, B, BIAS, alpha=0.5, epi=o.EPI_ACT, act=2), o.gemm_nt(A, B, None, alpha=0.5, epi=o.EPI_ADD, aux=AUX),
                o.gemm_nt(A, B, BIAS, alpha=0.5, epi=o.EPI_DACT, act=0, aux=AUX))
    names = ("bias", "gelu", "pre", "gelu_tanh", "quick_gelu", "residual", "gelu_bwd")
    try:
        _debug_set(1, 0)
        old = run()
        assert _last_gemm() == NT2
        _debug_set(0, 0)
        new = run()
        assert _last_gemm() == NTA, "a whole-tile bf16 shape did not reach gemm_nta"
        again = run()
    finally:
        _debug_set(0, 0)
    check("bias", new[0], lin, 2 ** -7, 2e-3)
    check("pre", new[2], lin, 2 ** -7, 2e-3)
    check("gelu", new[1], ref_act(new[2].double().cpu(), 0), 2 ** -7, 2e-3)
    check("residual", new[5], (a.double() @ b.double().T * 0.5).to(bf16).double() + aux.double(), 2 ** -7, 1.6e-2)
    for name, x, y, z in zip(names, old, new, again):
        assert torch.equal(x, y), f"{name}: gemm_nta differs from gemm_nt2"
        assert torch.equal(y, z), f"{name}: second launch differs"


@pytest.mark.parametrize("M,N,K", [(1024, 768, 512), (512, 1024, 256), (1000, 520, 776)])
def test_gemm_nt_e4m3_pre_activation_epilogues(M, N, K):
    """The "light8" keep tier's GEMM flavours (CLIPA_EPI_ACT_PRE8 / CLIPA_EPI_DACT8, fused in gemm_nta on whole-tile shapes; the
    third shape is ragged and takes the GEMM + cast fallback): the activation output is the plain epilogue's bit for bit,
    the e4m3 copy of the pre-activation is the saturating round-to-nearest-even e4m3 of the fp32 value (checked against torch's
    float8_e4m3fn cast of the fp64 product: equal bytes on >= 99 % of the elements - 95 % for the fallback, which rounds the bf16
    value a second time - and never more than one e4m3 step apart), and GELU-backward from those bytes equals GELU-backward
    from their exact bf16 values."""
    o = ops()
    a, b = rnd(M, K, seed=M + 7), rnd(N, K, seed=N + 8, scale=0.2)
    bias = rnd(N, seed=5, dtype=f32) * 3
    A, B, BIAS = a.to(DEV), b.to(DEV), bias.to(DEV)
    lin = a.double() @ b.double().T + bias.double()
    act_ref = o.gemm_nt(A, B, BIAS, epi=o.EPI_ACT, act=0)
    act, pre8 = o.gemm_nt(A, B, BIAS, epi=o.EPI_ACT, act=0, want_pre="e4m3")
    whole = M % 256 == 0 and N % 256 == 0 and K % 128 == 0
    assert _last_gemm() == (NTA if whole else NT2)
    assert pre8.dtype == torch.uint8 and tuple(pre8.shape) == (M, N)
    assert torch.equal(act, act_ref)
    want = lin.clamp(-448, 448).float().to(torch.float8_e4m3fn)
    got = pre8.cpu().view(torch.float8_e4m3fn)
    same = (got.view(torch.uint8) == want.view(torch.uint8)).float().mean()
    assert float(same) > (0.99 if whole else 0.95), float(same)      # (the fallback rounds bf16(v): ~3 % double-rounding ties)
    gf, wf = got.float().double(), want.float().double()
    assert torch.isfinite(gf).all()
    step = torch.maximum(wf.abs(), torch.tensor(2.0 ** -6, dtype=torch.float64)) * 2.0 ** -3      # one e4m3 step at that magnitude
    assert ((gf - wf).abs() <= step * 1.001).all()
    assert float(gf.abs().max()) <= 448.0 and float(lin.abs().max()) > 8
    # saturation: a pre-activation beyond the format's range becomes +-448, never NaN
    big, big8 = o.gemm_nt(A, B, BIAS * 0 + 1000.0, epi=o.EPI_ACT, act=0, want_pre="e4m3", alpha=0.0)
    assert (big8.cpu().view(torch.float8_e4m3fn).float() == 448.0).all()
    # GELU backward from the bytes
    dy = rnd(M, N, seed=11).to(DEV)
    w2 = rnd(K, N, seed=12, scale=0.05).to(DEV)                      # dh[M, K'] = dy[M,N] @ w2[K',N]^T * gelu'(h8[M, K']) - shapes swapped
    h8 = o.cast_e4m3(rnd(M, K, seed=13, scale=2.0).to(DEV))
    hb = o.e4m3_to_bf16(h8)
    assert torch.equal(o.cast_e4m3(hb), h8)                           # exact round trip
    d8 = o.gemm_nt(dy, w2, epi=o.EPI_DACT, act=0, aux=h8)
    d16 = o.gemm_nt(dy, w2, epi=o.EPI_DACT, act=0, aux=hb)
    assert torch.equal(d8, d16)
    assert torch.equal(o.activation_fwd(h8, 0), o.activation_fwd(hb, 0))
    # ... and the same launch writing act(h8) beside its output (round 6: gemm_nta<DACT, PRE = 3, AUX8>; the c_proj weight
    # gradient's operand without an activation_fwd pass): both outputs bit for bit those of the two separate launches, every
    # activation, every finite e4m3 code present in the operand
    codes = torch.arange(256, dtype=torch.uint8)
    codes = codes[(codes & 0x7f) != 0x7f]
    h8[0, :codes.numel()] = codes.to(DEV)
    from clipa_amd import lib as _lib
    for act in (0, 1, 2):
        _lib.gemm_counts(reset=True)
        dh, g = o.gemm_nt(dy, w2, epi=o.EPI_DACT, act=act, aux=h8, want_act=True)
        assert _lib.gemm_counts()[15] == (1 if whole else 0)
        assert torch.equal(dh, o.gemm_nt(dy, w2, epi=o.EPI_DACT, act=act, aux=h8)), act
        assert torch.equal(g, o.activation_fwd(h8, act)), act
        dh2, g2 = o.gemm_nt(dy, w2, epi=o.EPI_DACT, act=act, aux=h8, want_act=True)
        assert torch.equal(dh, dh2) and torch.equal(g, g2)


def test_gemm_nt_ragged_shapes_every_epilogue():
    """Ragged M / N / K tails, several tiles per persistent workgroup, every epilogue; the second launch re-uses ring
    state; the one-output and two-output activation epilogues agree bit for bit (a block's recompute relies on it)."""
    o = ops()
    for (M, N, K) in [(300, 264, 136), (1000, 520, 776), (777, 1024, 768), (2048, 256, 4096), (9000, 1280, 96)]:
        a, b = rnd(M, K, seed=M), rnd(N, K, seed=N, scale=0.05)
        bias, aux = rnd(N, seed=3, dtype=f32), rnd(M, N, seed=4)
        ad, bd, biasd, auxd = a.to(DEV), b.to(DEV), bias.to(DEV), aux.to(DEV)
        lin = a.double() @ b.double().T + bias.double()
        prev = None
        for rep in range(2):
            got = [o.gemm_nt(ad, bd, biasd), o.gemm_nt(ad, bd, biasd, epi=o.EPI_ADD, aux=auxd)]
            got += list(o.gemm_nt(ad, bd, biasd, epi=o.EPI_ACT, act=0, want_pre=True))
            got.append(o.gemm_nt(ad, bd, biasd, epi=o.EPI_ACT, act=0))
            check("bias", got[0], lin, 2 ** -7, 2e-3)
            check("residual", got[1], lin + aux.double(), 2 ** -6, 2e-2)   # two bf16 roundings
            check("pre", got[3], lin, 2 ** -7, 2e-3)
            check("gelu", got[2], ref_act(got[3].double().cpu(), 0), 2 ** -7, 2e-3)
            assert torch.equal(got[2], got[4]), "activation epilogue with / without the pre-activation copy"
            if prev is not None:
                assert all(torch.equal(x, y) for x, y in zip(prev, got)), "second launch differs"
            prev = got


def test_gemm_nt_production_rows():
    """M = 806 912 (ViT-L/16 @ 224, local batch 4096): the buffer-offset arithmetic of the real launch shape, checked
    on sampled rows against fp64 (N = 256, K = 64 keeps the operands small)."""
    o = ops()
    M, N, K = 806912, 256, 64
    g = torch.Generator(device="cpu").manual_seed(5)
    a = torch.randn(M, K, generator=g).to(bf16)
    b = (torch.randn(N, K, generator=g) * 0.1).to(bf16)
    bias = torch.randn(N, generator=g)
    out = o.gemm_nt(a.to(DEV), b.to(DEV), bias.to(DEV)).cpu()
    rows = torch.cat([torch.arange(0, 300), torch.arange(M - 300, M), torch.randint(0, M, (4000,), generator=g),
                      torch.tensor([2 ** 18 - 1, 2 ** 18, 2 ** 19, 2 ** 19 + 255, 524288 + 131072])])
    ref = a[rows].double() @ b.double().T + bias.double()
    check("sampled rows", out[rows], ref, 2 ** -7, 2e-3)
    assert torch.isfinite(out.float()).all()


@pytest.mark.parametrize("N,K", [(4096, 1024), (1024, 4096)])
def test_gemm_nt_production_width_values(N, K):
    """VALUE-level check of the real launch shapes of the MLP GEMMs (M = 806 912 = ViT-L/16 @ 224 x local batch 4096;
    c_fc: N = 4096, K = 1024; c_proj: N = 1024, K = 4096) against fp64 on sampled rows, for every fused epilogue: bias,
    GELU (+ pre-activation copy), residual add, GELU-backward.  ~3 150 x 16 (x 4) persistent tiles of 256 x 256."""
    o = ops()
    M = 806912
    g = torch.Generator(device=DEV).manual_seed(100 + N)
    a = torch.randn(M, K, generator=g, device=DEV, dtype=f32).to(bf16)
    b = (torch.randn(N, K, generator=g, device=DEV, dtype=f32) * (1.0 / math.sqrt(K))).to(bf16)
    bias = torch.randn(N, generator=g, device=DEV, dtype=f32)
    aux = torch.randn(M, N, generator=g, device=DEV, dtype=bf16)
    gc = torch.Generator().manual_seed(7)
    rows = torch.cat([torch.arange(0, 260), torch.arange(M - 260, M), torch.randint(0, M, (1500,), generator=gc),
                      torch.tensor([2 ** 18 - 1, 2 ** 18, 2 ** 19, 2 ** 19 + 255, 524288 + 131072, 255, 256, 511, 512])]).to(DEV)
    lin = a[rows].double().cpu() @ b.double().cpu().T + bias.double().cpu()
    v = lin.to(bf16).double()                                  # the engine rounds the GEMM result to bf16 before the epilogue
    auxr = aux[rows].double().cpu()
    out = o.gemm_nt(a, b, bias)
    assert _last_gemm() == NTA, "the production MLP shape did not reach gemm_nta"
    check("bias", out[rows], lin, 2 ** -7, 2e-3)
    assert torch.isfinite(out.float()).all()
    del out
    act, pre = o.gemm_nt(a, b, bias, epi=o.EPI_ACT, act=0, want_pre=True)
    check("pre-activation", pre[rows], lin, 2 ** -7, 2e-3)
    check("gelu", act[rows], ref_act(pre[rows].double().cpu(), 0), 2 ** -7, 2e-3)
    act1 = o.gemm_nt(a, b, bias, epi=o.EPI_ACT, act=0)
    assert torch.equal(act1, act), "one-output and two-output activation epilogues differ"
    del act, act1, pre
    out = o.gemm_nt(a, b, bias, epi=o.EPI_ADD, aux=aux)
    check("residual add", out[rows], v + auxr, 2 ** -7, 1.6e-2)
    del out
    x = auxr.clone().requires_grad_(True)
    ref_act(x, 0).sum().backward()
    out = o.gemm_nt(a, b, bias, epi=o.EPI_DACT, act=0, aux=aux)
    check("gelu backward", out[rows], v * x.grad, 2 ** -6, 6e-3)


def test_gemm_tn_production_rows():
    """Reduction over M = 806 912 rows (the weight-gradient launch shape), R = C = 256, against fp64."""
    o = ops()
    M, R, C = 806912, 256, 256
    g = torch.Generator(device="cpu").manual_seed(6)
    p = (torch.randn(M, R, generator=g) * 0.05).to(bf16)
    q = (torch.randn(M, C, generator=g) * 0.05).to(bf16)
    w, cs = o.gemm_tn(p.to(DEV), q.to(DEV), f32, want_colsum=True)
    assert _last_gemm() == TNA, "the production weight-gradient shape did not reach gemm_tna"
    ref = p.double().T @ q.double()
    check("weight gradient", w, ref, 1e-3, 5e-3)
    check("column sums", cs, p.double().sum(0), 1e-4, 5e-3)


def test_gemm_tn_kernel_generations_agree():
    """The two weight-gradient kernels (ping-pong 32x32x16, 16x16x32) in both work orders (slice-per-XCD,
    tile-per-XCD) against fp64."""
    o = ops()
    p, q = rnd(70000, 520, seed=13).to(DEV), rnd(70000, 264, seed=14, scale=0.1).to(DEV)
    ref = p.double().cpu().T @ q.double().cpu()
    try:
        for abl in (1024 | 4096, 1024 | 8192, 2048 | 4096, 2048 | 8192):
            _debug_set(0, abl)
            w, c = o.gemm_tn(p, q, f32, want_colsum=True)
            check(f"weight gradient (abl {abl})", w, ref, 2e-4, 2e-2)
            check(f"column sums (abl {abl})", c, p.double().cpu().sum(0), 1e-5, 2e-2)
    finally:
        _debug_set(0, 0)


@pytest.mark.parametrize("M,R,C", [(64, 256, 256), (1000, 264, 136), (4100, 1024, 512), (130, 8, 2304), (8, 16, 16)])
def test_gemm_tn(M, R, C):
    p, q = rnd(M, R, seed=10), rnd(M, C, seed=11, scale=0.1)
    ref = p.double().T @ q.double()
    out = ops().gemm_tn(p.to(DEV), q.to(DEV), f32)
    check("f32", out, ref, 1e-4, 3e-4 * math.sqrt(M))
    out, cs = ops().gemm_tn(p.to(DEV), q.to(DEV), bf16, want_colsum=True)
    check("bf16", out, ref, 2 ** -7, 3e-4 * math.sqrt(M))
    check("fused column sums", cs, p.double().sum(0), 1e-5, 1e-4 * math.sqrt(M))


def test_gemm_tn_strided_and_many_slices():
    M, R, C = 70000, 136, 72
    big = rnd(M, R + C, seed=12, scale=0.2).to(DEV)
    p, q = big[:, :R], big[:, R:]
    out = ops().gemm_tn(p, q, f32)
    check("slices", out, p.double().cpu().T @ q.double().cpu(), 2e-4, 2e-2)


# -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D", [384, 768, 1024, 1280])
@pytest.mark.parametrize("xdt,ydt", [(bf16, bf16), (f32, bf16), (f32, f32)])
def test_layernorm(D, xdt, ydt):
    rows = 777
    x = (rnd(rows, D, seed=20, dtype=f32) * 1.5 + 0.3).to(xdt)
    w, b = 1 + 0.1 * rnd(D, seed=21, dtype=f32), 0.1 * rnd(D, seed=22, dtype=f32)
    dy, dres = rnd(rows, D, seed=23, dtype=ydt), rnd(rows, D, seed=24, dtype=xdt)
    xr = x.double().requires_grad_(True)
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = O.layer_norm(xr, wr, br)
    yr.backward(dy.double())
    y = ops().layernorm_fwd(x.to(DEV), w.to(DEV), b.to(DEV), 1e-5, out_dtype=ydt)
    tol = 2 ** -7 if ydt == bf16 else 2e-5
    check("fwd", y, yr, tol, tol)
    dx, dw, db = ops().layernorm_bwd(x.to(DEV), w.to(DEV), dy.to(DEV), dres.to(DEV), 1e-5)
    tolx = 2 ** -7 if xdt == bf16 else 3e-5
    check("dx", dx, xr.grad + dres.double(), tolx, tolx * 4)
    check("dgamma", dw, wr.grad, 1e-4, 2e-3)
    check("dbeta", db, br.grad, 1e-4, 2e-3)
    dx2, _, _ = ops().layernorm_bwd(x.to(DEV), w.to(DEV), dy.to(DEV), None, 1e-5)
    check("dx no-res", dx2, xr.grad, tolx, tolx * 4)


@pytest.mark.parametrize("rows,D", [(1, 256), (41, 256), (8190, 256), (300001, 256), (9001, 1280), (5003, 384), (4099, 768), (70001, 1024)])
def test_layernorm_row_counts(rows, D):
    """The launch shapes of layernorm.hip: forward = one block per 8 rows (odd tails), backward = one block per C adjacent rows
    (C = 4 ... 128 by row count) whose dgamma / dbeta partials go through the one- or two-pass reduction (<= / > 128 blocks);
    rows wider than 1024 elements: the persistent 1024-block grid (more than one row per wave at 9001 rows)."""
    x = (rnd(rows, D, seed=25) * 1.5 + 0.3).to(bf16)
    w, b = 1 + 0.1 * rnd(D, seed=26, dtype=f32), 0.1 * rnd(D, seed=27, dtype=f32)
    dy, dres = rnd(rows, D, seed=28), rnd(rows, D, seed=29)
    xr = x.double().requires_grad_(True)
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = O.layer_norm(xr, wr, br)
    yr.backward(dy.double())
    y = ops().layernorm_fwd(x.to(DEV), w.to(DEV), b.to(DEV), 1e-5, out_dtype=bf16)
    check("fwd", y, yr, 2 ** -7, 2 ** -7)
    dx, dw, db = ops().layernorm_bwd(x.to(DEV), w.to(DEV), dy.to(DEV), dres.to(DEV), 1e-5)
    check("dx", dx, xr.grad + dres.double(), 2 ** -7, 2 ** -5)
    scale = max(1.0, rows ** 0.5)              # the column sums grow like sqrt(rows); fp32 partials
    check("dgamma", dw, wr.grad, 1e-4, 2e-3 * scale)
    check("dbeta", db, br.grad, 1e-4, 2e-3 * scale)
    dx2, dw2, db2 = ops().layernorm_bwd(x.to(DEV), w.to(DEV), dy.to(DEV), dres.to(DEV), 1e-5)
    assert torch.equal(dx, dx2) and torch.equal(dw, dw2) and torch.equal(db, db2)      # fixed summation order
    # the emitting form (clipa_layernorm_bwd_y): same gradients, plus the forward's output bit for bit - a block that recomputes
    # its LayerNorm outputs in backward takes them from this pass (engine._block_backward)
    dx3, dw3, db3, y3 = ops().layernorm_bwd(x.to(DEV), w.to(DEV), dy.to(DEV), dres.to(DEV), 1e-5, beta=b.to(DEV))
    assert torch.equal(dx, dx3) and torch.equal(dw, dw3) and torch.equal(db, db3)
    assert y3.dtype == y.dtype and torch.equal(y3, y)


# -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dh", [64, 80])       # 80 = ViT-H/14 (head_width 80)
@pytest.mark.parametrize("B,H,L,causal", [(2, 3, 26, False), (3, 2, 50, False), (2, 2, 77, True), (2, 4, 197, False),
                                          (1, 2, 257, False), (2, 1, 8, True), (2, 2, 32, True), (1, 2, 288, True),
                                          (3, 5, 257, False), (2, 2, 258, False), (2, 2, 257, True)])
def test_attention(B, H, L, causal, dh):
    D = dh * H
    qkv = rnd(B * L, 3 * D, seed=30 + L, scale=1.2)
    dout = rnd(B * L, D, seed=31 + L)
    x = qkv.double().reshape(B, L, 3 * D).requires_grad_(True)
    o = O.attention(x, H, causal)
    o.backward(dout.double().reshape(B, L, D))
    got, stats = ops().attention_fwd(qkv.to(DEV), B, L, H, causal, want_stats=True)
    check("fwd", got.reshape(B, L, D), o, 2 ** -6, 8e-3)
    assert torch.equal(got, ops().attention_fwd(qkv.to(DEV), B, L, H, causal))
    dq = ops().attention_bwd(qkv.to(DEV), got, dout.to(DEV), stats, B, L, H, causal)
    check("dqkv", dq.reshape(B, L, 3 * D), x.grad, 2 ** -5, 2e-2)


@pytest.mark.parametrize("dh,B,H,L", [(88, 2, 2, 257), (88, 3, 2, 37), (104, 1, 3, 257), (104, 2, 2, 50), (112, 2, 2, 197),
                                      (112, 1, 2, 26), (88, 1, 2, 577), (104, 1, 1, 300)])
def test_attention_wide_heads(dh, B, H, L):
    """Head dims 88 / 104 / 112 (open_clip/model_configs ViT-g-14 / ViT-bigG-14 / ViT-e-14; CLIPA-v2's G/14 tower): 6 - 7
    k-steps of which the last is ragged for 88 and 104 (zero-filled in the LDS images AND in the fragments read from global
    memory - a neighbouring head's columns must never leak in: H >= 2 makes that visible), 3 - 4 output tiles with a partial
    last one; one-piece (L <= 288) and streamed (L = 300, 577) kernels, image towers only (no causal mask)."""
    D = dh * H
    qkv = rnd(B * L, 3 * D, seed=230 + L + dh, scale=1.2)
    dout = rnd(B * L, D, seed=231 + L + dh)
    x = qkv.double().reshape(B, L, 3 * D).requires_grad_(True)
    o = O.attention(x, H, False)
    o.backward(dout.double().reshape(B, L, D))
    got, stats = ops().attention_fwd(qkv.to(DEV), B, L, H, False, want_stats=True)
    check("fwd", got.reshape(B, L, D), o, 2 ** -6, 8e-3)
    dq = ops().attention_bwd(qkv.to(DEV), got, dout.to(DEV), stats, B, L, H, False)
    check("dqkv", dq.reshape(B, L, 3 * D), x.grad, 2 ** -5, 2e-2)
    with pytest.raises(RuntimeError, match="image towers only"):
        ops().attention_fwd(qkv.to(DEV), B, L, H, True)


@pytest.mark.parametrize("dh,H,ctx,causal", [(64, 3, 77, True), (64, 2, 32, True), (80, 2, 77, True), (64, 2, 197, False)])
def test_attention_packed_variable_length(dh, H, ctx, causal):
    """clipa_attention_*_varlen: sequences of different lengths packed back to back (the text tower on the tokens up to EOT),
    one launch per class of 32-row tile counts.  Every sequence against the oracle on that sequence alone, and - causal case -
    BIT FOR BIT against the fixed-length kernels run on the zero-padded batch (what the reference computes): the rows up to
    each sequence's end agree in the forward, and with the gradient zero beyond them, in the backward."""
    o = ops()
    g = torch.Generator().manual_seed(5 + ctx + dh)
    B = 23
    lens = torch.randint(1, ctx + 1, (B,), generator=g)
    lens[0], lens[1], lens[2] = ctx, 1, min(ctx, 33)
    D = dh * H
    vl = o.VarLen(lens, ctx, DEV)
    assert vl.rows % 256 == 0 and vl.T == int(lens.sum())
    qkv = torch.zeros(vl.rows, 3 * D, dtype=bf16)
    dout = torch.zeros(vl.rows, D, dtype=bf16)
    qkv[:vl.T] = rnd(vl.T, 3 * D, seed=41, scale=1.2)
    dout[:vl.T] = rnd(vl.T, D, seed=42)
    qkv_d, dout_d = qkv.to(DEV), dout.to(DEV)
    got, stats = o.attention_fwd_varlen(qkv_d, vl, H, causal, want_stats=True)
    assert torch.equal(got, o.attention_fwd_varlen(qkv_d, vl, H, causal))
    dq = o.attention_bwd_varlen(qkv_d, got, dout_d, stats, vl, H, causal)
    starts = (torch.cumsum(lens, 0) - lens).tolist()
    for b, (s0, n) in enumerate(zip(starts, lens.tolist())):
        x = qkv[s0:s0 + n].double().reshape(1, n, 3 * D).requires_grad_(True)
        ref = O.attention(x, H, causal)
        ref.backward(dout[s0:s0 + n].double().reshape(1, n, D))
        check(f"fwd seq {b} (len {n})", got[s0:s0 + n].reshape(1, n, D), ref, 2 ** -6, 8e-3)
        check(f"dqkv seq {b} (len {n})", dq[s0:s0 + n].reshape(1, n, 3 * D), x.grad, 2 ** -5, 2e-2)
    assert float(got[vl.T:].float().abs().max() if vl.rows > vl.T else 0) == 0 and float(dq[vl.T:].float().abs().max() if vl.rows > vl.T else 0) == 0
    if causal:
        pad_qkv = torch.zeros(B * ctx, 3 * D, dtype=bf16)
        pad_dout = torch.zeros(B * ctx, D, dtype=bf16)
        for b, (s0, n) in enumerate(zip(starts, lens.tolist())):
            pad_qkv[b * ctx:b * ctx + n] = qkv[s0:s0 + n]
            pad_dout[b * ctx:b * ctx + n] = dout[s0:s0 + n]
        pq, pd = pad_qkv.to(DEV), pad_dout.to(DEV)
        pout, pstats = o.attention_fwd(pq, B, ctx, H, True, want_stats=True)
        pdq = o.attention_bwd(pq, pout, pd, pstats, B, ctx, H, True)
        for b, (s0, n) in enumerate(zip(starts, lens.tolist())):
            assert torch.equal(pout[b * ctx:b * ctx + n], got[s0:s0 + n]), (b, n)
            assert torch.equal(pdq[b * ctx:b * ctx + n], dq[s0:s0 + n]), (b, n)


@pytest.mark.parametrize("dh", [64, 80])
@pytest.mark.parametrize("B,H,L,causal", [(2, 2, 577, False), (1, 3, 401, False), (1, 2, 300, True), (1, 1, 1024, False)])
def test_attention_long_sequences(B, H, L, causal, dh):
    """288 < L <= 1024 (the 336-px / 14-px-patch stage of CLIPA-v2: 577 tokens; ViT-L-16-320: 401): keys / queries
    stream through LDS in 256-row chunks, online softmax in the forward - same oracle, same tolerances."""
    D = dh * H
    qkv = rnd(B * L, 3 * D, seed=130 + L, scale=1.2)
    dout = rnd(B * L, D, seed=131 + L)
    x = qkv.double().reshape(B, L, 3 * D).requires_grad_(True)
    o = O.attention(x, H, causal)
    o.backward(dout.double().reshape(B, L, D))
    got, stats = ops().attention_fwd(qkv.to(DEV), B, L, H, causal, want_stats=True)
    check("fwd", got.reshape(B, L, D), o, 2 ** -6, 8e-3)
    dq = ops().attention_bwd(qkv.to(DEV), got, dout.to(DEV), stats, B, L, H, causal)
    check("dqkv", dq.reshape(B, L, 3 * D), x.grad, 2 ** -5, 2e-2)


def test_embedding_gradient_is_bit_reproducible():
    """The token-embedding gradient is a scatter-add over ~B*ctx rows, most of them hitting a few ids (SOT / EOT): it runs on
    fixed-point integer atomics, so repeated launches agree bit for bit (float atomics do not)."""
    o = ops()
    B, T, D, V = 512, 32, 256, 1000
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, 8, (B, T), generator=g)                # heavy collisions
    ids[:, 0] = V - 2
    dx = (torch.randn(B * T, D, generator=g) * torch.exp(torch.randn(B * T, 1, generator=g) * 4 - 8)).to(bf16)   # 1e-9 .. 30
    a, _ = o.embed_tokens_bwd(ids.to(DEV), dx.to(DEV), V)
    for _ in range(3):
        b, _ = o.embed_tokens_bwd(ids.to(DEV), dx.to(DEV), V)
        assert torch.equal(a, b)
    ref = torch.zeros(V, D, dtype=torch.float64).index_add_(0, ids.reshape(-1), dx.double())
    check("dtable", a, ref, 1e-6, 1e-10)
    dx[5, 3] = float("nan")                                           # a non-finite gradient poisons its table row, as a float sum would
    c, _ = o.embed_tokens_bwd(ids.to(DEV), dx.to(DEV), V)
    row = int(ids.reshape(-1)[5])
    assert torch.isnan(c[row]).all() and torch.isfinite(c[[r for r in range(8) if r != row]]).all()


@pytest.mark.parametrize("dh", [64, 80])
def test_attention_reads_packed_projection_in_place(dh):
    """q/k/v are column blocks of a wider buffer (row stride != 3D) - no head-major copy is made."""
    B, H, L = 2, 2, 50
    D = dh * H
    wide = rnd(B * L, 3 * D + 64, seed=40).to(DEV)
    qkv = wide[:, :3 * D]
    ref = ops().attention_fwd(qkv.contiguous(), B, L, H, False)
    got = ops().attention_fwd(qkv, B, L, H, False)
    assert torch.equal(ref, got)


def test_attention_large_logits_stable():
    B, H, L = 1, 1, 64
    qkv = rnd(B * L, 192, seed=41, scale=6.0)
    x = qkv.double().reshape(B, L, 192)
    got = ops().attention_fwd(qkv.to(DEV), B, L, H, True)
    check("spiky", got.reshape(B, L, 64), O.attention(x, H, True), 2 ** -5, 3e-2)


# -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("S,P", [(48, 16), (40, 16), (42, 14)])
@pytest.mark.parametrize("kind", ["u8", "f32", "bf16", "u8_nhwc"])
def test_patchify(S, P, kind):
    B = 3
    g = S // P
    img_u8, _ = O.synthetic_batch(B, S, 8, 100, seed=50)
    K = 3 * P * P
    Kp = (K + 7) // 8 * 8
    if kind.startswith("u8"):
        x = img_u8
        xin = x.to(DEV)
        if kind == "u8_nhwc":
            xin = xin.contiguous(memory_format=torch.channels_last)
        ref_img = O.normalize_images(x)
        got = ops().patchify(xin, P, Kp, O.OPENAI_DATASET_MEAN, O.OPENAI_DATASET_STD)
    else:
        dt = f32 if kind == "f32" else bf16
        ref_img = O.normalize_images(img_u8).to(dt)
        got = ops().patchify(ref_img.to(DEV), P, Kp)
        ref_img = ref_img.float()
    ref = ref_img[:, :, :g * P, :g * P].reshape(B, 3, g, P, g, P).permute(0, 2, 4, 3, 5, 1).reshape(B * g * g, K)
    check("patches", got[:, :K], ref, 2 ** -8, 1e-6)
    assert (got[:, K:] == 0).all()


def test_assemble_embed_pool_l2norm_colsum_cast():
    o = ops()
    B, L, D = 5, 10, 128
    patch = rnd((B * (L - 1)), D, seed=60)
    cls, pos = rnd(D, seed=61, dtype=f32), rnd(L, D, seed=62, dtype=f32)
    tok = o.assemble_tokens(patch.to(DEV), cls.to(DEV), pos.to(DEV), B, L)
    ref = torch.cat([cls.to(bf16).float().expand(B, 1, D), patch.float().reshape(B, L - 1, D)], 1) + pos.to(bf16).float()
    check("assemble", tok.reshape(B, L, D), ref, 2 ** -8, 1e-6)
    dtok = rnd(B * L, D, seed=63)
    dpatch, dcls, dpos = o.assemble_tokens_bwd(dtok.to(DEV), B, L)
    d3 = dtok.float().reshape(B, L, D)
    check("dpatch", dpatch, d3[:, 1:].reshape(-1, D), 0, 0)
    check("dcls", dcls, d3[:, 0].sum(0), 1e-5, 1e-5)
    check("dpos", dpos, d3.sum(0), 1e-5, 1e-5)

    V, T = 300, 12
    _, ids = O.synthetic_batch(B, 16, T, V, seed=64)
    table = rnd(V, D, seed=65, dtype=f32, scale=0.02)
    tpos = rnd(T, D, seed=66, dtype=f32, scale=0.01)
    emb = o.embed_tokens(ids.to(DEV), table.to(DEV), tpos.to(DEV))
    ref = table.to(bf16).float()[ids] + tpos.to(bf16).float()
    check("embed", emb.reshape(B, T, D), ref, 2 ** -8, 1e-7)
    dx = rnd(B * T, D, seed=67)
    dx[T - 3:T] = 0                      # all-zero rows are skipped, result unchanged
    dtable, dtpos = o.embed_tokens_bwd(ids.to(DEV), dx.to(DEV), V)
    ref_t = torch.zeros(V, D, dtype=torch.float64).index_add_(0, ids.reshape(-1), dx.double())
    check("dtable", dtable, ref_t, 1e-5, 1e-5)
    check("dtpos", dtpos, dx.double().reshape(B, T, D).sum(0), 1e-5, 1e-5)
    assert torch.equal(o.argmax_tokens(ids.to(DEV)).cpu().long(), ids.argmax(-1))

    x = rnd(B * L, D, seed=68)
    idx = torch.tensor([3, 0, 9, 5, 1], dtype=torch.int32)
    x3 = x.double().reshape(B, L, D)
    refs = {o.POOL_FIRST: x3[:, 0], o.POOL_LAST: x3[:, -1], o.POOL_INDEX: x3[torch.arange(B), idx.long()],
            o.POOL_MEAN_ALL: x3.mean(1), o.POOL_MEAN_PATCH: x3[:, 1:].mean(1)}
    dout = rnd(B, D, seed=69, dtype=f32)
    for mode, ref in refs.items():
        ii = idx.to(DEV) if mode == o.POOL_INDEX else None
        check(f"pool{mode}", o.pool_fwd(x.to(DEV), B, L, mode, ii), ref, 1e-6, 1e-6)
        xr = x3.clone().requires_grad_(True)
        pooled = {o.POOL_FIRST: xr[:, 0], o.POOL_LAST: xr[:, -1], o.POOL_INDEX: xr[torch.arange(B), idx.long()],
                  o.POOL_MEAN_ALL: xr.mean(1), o.POOL_MEAN_PATCH: xr[:, 1:].mean(1)}[mode]
        pooled.backward(dout.double())
        check(f"pool_bwd{mode}", o.pool_bwd(dout.to(DEV), B, L, mode, ii).reshape(B, L, D), xr.grad, 2 ** -8, 1e-7)

    feat = rnd(37, 96, seed=70, dtype=f32)
    fr = feat.double().requires_grad_(True)
    yr = O.l2_normalize(fr)
    dy = rnd(37, 96, seed=71, dtype=f32)
    yr.backward(dy.double())
    y, ybf, inv = o.l2norm_fwd(feat.to(DEV), want_bf16=True)
    check("l2norm", y, yr, 1e-6, 1e-6)
    check("l2norm bf16", ybf, yr, 2 ** -8, 1e-6)
    check("l2norm bwd", o.l2norm_bwd(y, inv, dy.to(DEV)), fr.grad, 1e-5, 1e-5)

    m = rnd(5000, 1032, seed=72)
    check("colsum", o.colsum(m.to(DEV)), m.double().sum(0), 1e-5, 1e-3)
    w = rnd(200, 136, seed=73, dtype=f32)
    check("transpose", o.transpose_bf16(w.to(DEV)), w.to(bf16).T, 0, 0)
    check("to_bf16", o.to_bf16(w.to(DEV)), w.to(bf16), 0, 0)
    check("to_f32", o.to_f32(w.to(bf16).to(DEV)), w.to(bf16).float(), 0, 0)


@pytest.mark.parametrize("R,N,E,label0", [(8, 8, 64, 0), (4, 4, 32, 0), (16, 64, 128, 48), (300, 4100, 768, 1000), (520, 520, 512, 0)])
def test_fused_similarity_cross_entropy(R, N, E, label0):
    """ops.simce = similarity GEMM + cross-entropy in one kernel pair (fp32 logits never written): loss rows, the bf16
    gradient w.r.t. the raw similarities and d loss / d s against fp64; N need not be a multiple of 8 or of the tile."""
    g = torch.Generator().manual_seed(R + N)
    rows = torch.nn.functional.normalize(torch.randn(R, E, generator=g), dim=1).to(bf16)
    cols = torch.nn.functional.normalize(torch.randn(N, E, generator=g), dim=1).to(bf16)
    n8 = (N + 7) // 8 * 8
    cols8 = torch.zeros(n8, E, dtype=bf16)
    cols8[:N] = cols
    s = 14.3
    raw = (rows.double() @ cols.double().T).requires_grad_(True)
    labels = torch.arange(R) + label0
    per = torch.nn.functional.cross_entropy(raw * s, labels, reduction="none")
    gs = 0.5 / R
    (per.sum() * gs).backward()
    scale = torch.tensor([s], device=DEV, dtype=f32)
    loss_rows, dl, ds = ops().simce(rows.to(DEV), cols8.to(DEV), N, label0, gs, scale=scale)
    check("loss rows", loss_rows, per.detach(), 1e-4, 2e-4)
    check("d loss / d raw", dl[:, :N], raw.grad, 2 ** -7, 1e-6)
    assert not dl[:, N:].float().abs().any(), "pad columns of the gradient must be zero"
    check("d loss / d s rows", ds, (raw.grad / s * raw.detach()).sum(1), 2e-3, 2e-5)
    only_loss, none_dl, _ = ops().simce(rows.to(DEV), cols8.to(DEV), N, label0, gs, scale=scale, want_grad=False)
    assert none_dl is None and torch.equal(only_loss, loss_rows)
    check("sum", ops().sum_scale(loss_rows, gs), per.detach().sum() * gs, 1e-4, 1e-5)


@pytest.mark.parametrize("pdt,gdt", [(f32, f32), (bf16, bf16)])
def test_adamw_multi_tensor_equals_per_tensor(pdt, gdt):
    """One multi-tensor call (more tensors than fit one launch, ragged sizes incl. 1 and > one chunk) gives the
    result of the per-tensor kernel (same formula; the two kernels may contract one FMA differently)."""
    sizes = [1, 7, 4096, 4097, 10007, 300 * 1024 + 5] + [33 + 17 * i for i in range(60)]
    ps = [rnd(n, seed=100 + i, dtype=f32).to(pdt).to(DEV) for i, n in enumerate(sizes)]
    gs = [rnd(n, seed=300 + i, dtype=f32, scale=0.1).to(gdt).to(DEV) for i, n in enumerate(sizes)]
    one = [t.clone() for t in ps]
    m1, v1 = [torch.zeros(n, device=DEV) for n in sizes], [torch.zeros(n, device=DEV) for n in sizes]
    m2, v2 = [torch.zeros(n, device=DEV) for n in sizes], [torch.zeros(n, device=DEV) for n in sizes]
    kw = dict(lr=1e-3, beta1=0.9, beta2=0.95, eps=1e-6, weight_decay=0.2)
    for step in (1, 2):
        for p, g, m, v in zip(one, gs, m1, v1):
            ops().adamw_(p, g, m, v, step=step, **kw)
        ops().adamw_multi_(ps, gs, m2, v2, step=step, **kw)
    ptol = 2 ** -7 if pdt == bf16 else 2e-6
    for a, b, ma, mb, va, vb in zip(one, ps, m1, m2, v1, v2):
        assert torch.allclose(a.float(), b.float(), rtol=ptol, atol=1e-7)
        assert torch.allclose(ma, mb, rtol=2e-6, atol=1e-12) and torch.allclose(va, vb, rtol=2e-6, atol=1e-12)


@pytest.mark.parametrize("pdt,gdt", [(f32, f32), (bf16, bf16), (f32, bf16)])
def test_adamw_matches_torch(pdt, gdt):
    n = 10007
    p0, g = rnd(n, seed=90, dtype=f32), rnd(n, seed=91, dtype=f32, scale=0.1)
    ref = torch.nn.Parameter(p0.to(pdt).double())
    opt = torch.optim.AdamW([ref], lr=1e-3, betas=(0.9, 0.95), eps=1e-6, weight_decay=0.2)
    p = p0.to(pdt).to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in (1, 2, 3):
        gstep = (g * step).to(gdt)
        ref.grad = gstep.double()
        opt.step()
        if pdt == bf16:
            ref.data = ref.data.to(bf16).double()
        ops().adamw_(p, gstep.to(DEV), m, v, lr=1e-3, beta1=0.9, beta2=0.95, eps=1e-6,
                     weight_decay=0.2, step=step)
    check("adamw", p, ref.data, 2 ** -7 if pdt == bf16 else 1e-5, 1e-6)


def test_embedding_reports_out_of_range_ids():
    """nn.Embedding raises on ids outside the table; the kernels count them (no silent clamp) and clipa_amd.ops turns
    the count into a RuntimeError."""
    o = ops()
    V, D, T = 64, 32, 8
    table, tpos = rnd(V, D, seed=1, dtype=f32).to(DEV), rnd(T, D, seed=2, dtype=f32).to(DEV)
    ids = torch.randint(0, V, (4, T))
    o.embed_tokens(ids.to(DEV), table, tpos)
    o.check_token_ids(wait=True)                      # clean batch: nothing to report
    bad = ids.clone()
    bad[1, 3] = V + 5
    bad[2, 0] = -1
    o.embed_tokens(bad.to(DEV), table, tpos)
    with pytest.raises(RuntimeError, match="2 token ids outside"):
        o.check_token_ids(wait=True)
    dx = rnd(4 * T, D, seed=3).to(DEV)
    dtable, _ = o.embed_tokens_bwd(bad.to(DEV), dx, V)
    with pytest.raises(RuntimeError, match="token ids outside"):
        o.check_token_ids(wait=True)
    assert torch.isfinite(dtable).all()               # the bad rows were skipped, nothing was scattered out of bounds


def test_optimizer_state_dict_round_trip_and_fused_tail():
    """(1) state_dict() -> load_state_dict() with bf16 parameters: torch casts floating state to the parameter dtype;
    the optimizer must hand the kernel f32 moments again (out-of-bounds writes otherwise).  (2) the fused
    clip_grad_norm_ + logit-scale clamp tail equals the torch recipe of train.py:270-286."""
    from clipa_amd.optim import AdamW
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(300, 40, device=DEV).to(bf16)), torch.nn.Parameter(torch.randn(77, device=DEV)),
          torch.nn.Parameter(torch.tensor(4.0, device=DEV))]
    ref = [torch.nn.Parameter(p.detach().float().clone()) for p in ps]
    kw = dict(lr=1e-2, betas=(0.9, 0.95), eps=1e-6, weight_decay=0.1)
    opt = AdamW(ps, grad_clip_norm=0.5, clamp=(ps[2], 0.0, 4.0), **kw)
    ropt = torch.optim.AdamW(ref, **kw)
    for step in range(4):
        gs = [torch.randn_like(p.float()) * (3.0 if step % 2 else 0.01) for p in ps]
        gs[2] = torch.tensor(-50.0, device=DEV)      # pushes the scalar up: the clamp at 4.0 must hold it
        for p, r, g in zip(ps, ref, gs):
            p.grad = g.to(p.dtype).clone()
            r.grad = g.to(p.dtype).float().clone()      # separate storage: clip_grad_norm_ scales r.grad in place
        torch.nn.utils.clip_grad_norm_(ref, 0.5)
        ropt.step()
        with torch.no_grad():
            ref[2].clamp_(0.0, 4.0)
        opt.step()
        if step == 1:                                 # resume in the middle (training/main.py:338-356)
            sd = opt.state_dict()
            opt = AdamW(ps, grad_clip_norm=0.5, clamp=(ps[2], 0.0, 4.0), **kw)
            opt.load_state_dict(sd)
            for st in opt.state.values():
                assert st["exp_avg"].dtype == f32 and st["exp_avg_sq"].dtype == f32 and isinstance(st["step"], int)
    check("bf16 matrix", ps[0], ref[0].data, 2 ** -6, 2e-3)
    check("f32 vector", ps[1], ref[1].data, 1e-4, 1e-5)
    assert float(ps[2]) <= 4.0 and abs(float(ps[2]) - float(ref[2])) < 1e-4
    want = torch.sqrt(sum((p.grad.float() ** 2).sum() for p in ps))
    assert abs(float(opt.last_grad_norm) - float(want)) < 1e-3 * float(want)
