"""Tensor-level wrappers over the C ABI: argument checking, output / workspace allocation with torch,
current-stream plumbing.  No arithmetic happens here - every op below is one or two HIP launches.
"""
import ctypes
import math

import threading

import torch

from . import lib

EPI_NONE, EPI_ACT, EPI_ADD, EPI_DACT = 0, 1, 2, 3
ACT_GELU_ERF, ACT_GELU_TANH, ACT_QUICK_GELU = 0, 1, 2
DT_U8, DT_BF16, DT_F32 = 0, 1, 2
POOL_FIRST, POOL_LAST, POOL_INDEX, POOL_MEAN_ALL, POOL_MEAN_PATCH = 0, 1, 2, 3, 4

bf16 = torch.bfloat16
f32 = torch.float32


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---- optional per-launch timing (bench.py roofline): HIP events on the launch stream ------------
_PROF = None


def profile_start(detail=False):
    global _PROF, _DETAIL
    _PROF = {}
    _DETAIL = bool(detail)


def profile_stop():
    """-> {kernel: {"launches", "ms", "work"}}; work = algorithmic FLOPs (GEMM/attention) of the launches."""
    global _PROF
    prof, _PROF = _PROF, None
    if not prof:
        return {}
    torch.cuda.synchronize()
    out = {}
    for name, recs in prof.items():
        out[name] = {"launches": len(recs), "ms": sum(r[0].elapsed_time(r[1]) for r in recs),
                     "work": float(sum(r[2] for r in recs)), "bytes": float(sum(r[3] for r in recs))}
    return out


_DETAIL = False      # profile_start(detail=True): additionally key the records by shape ("gemm_nt|M,N,K,epi")


class _Timed:
    __slots__ = ("name", "work", "nbytes", "a", "tag")

    def __init__(self, name, work, nbytes=0.0, tag=None):
        self.name, self.work, self.nbytes, self.tag = name, work, nbytes, tag

    def __enter__(self):
        if _PROF is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if _PROF is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            _PROF.setdefault(self.name, []).append((self.a, b, self.work, self.nbytes))
            if self.name == "gemm_nt" and self.tag is not None:       # the family split by epilogue ("gemm_nt#epi3,aux8+act")
                _PROF.setdefault("gemm_nt#" + self.tag[self.tag.index("epi"):], []).append((self.a, b, self.work, self.nbytes))
            if _DETAIL and self.tag is not None:
                _PROF.setdefault(f"{self.name}|{self.tag}", []).append((self.a, b, self.work, self.nbytes))
        return False


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _chk(t, dtype, name, dims=None):
    if not t.is_cuda:
        raise RuntimeError(f"clipa_amd.ops: {name} must live on the GPU (no CPU fallback); got {t.device}")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"clipa_amd.ops: {name} must be {dtype}, got {t.dtype}")
    if dims is not None and t.dim() != dims:
        raise RuntimeError(f"clipa_amd.ops: {name} must be {dims}-D, got shape {tuple(t.shape)}")
    if t.dim() > 0 and t.stride(-1) != 1 and t.shape[-1] != 1:
        raise RuntimeError(f"clipa_amd.ops: {name} must be contiguous in its last dim")


def _rowmajor(t):
    """Accept [rows, cols] views with unit column stride; return (tensor, ld)."""
    if t.stride(-1) != 1:
        t = t.contiguous()
    return t, t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


EPI_ACT_PRE8, EPI_DACT8 = 4, 5      # include/clipa_hip.h: e4m3 pre-activation copy / e4m3 second operand (whole-tile shapes)


def _whole_tiles(M, N, K):
    """Shapes the four-wave GEMM kernel takes (gemm_nta.hip: nta_eligible): the fused e4m3 epilogues exist only there."""
    return M % 256 == 0 and N % 256 == 0 and K % 128 == 0 and K >= 256


def cast_e4m3(x):
    """bf16 -> saturating OCP e4m3 bytes (uint8, same shape): the unfused form of gemm_nt(..., want_pre="e4m3")."""
    _chk(x, bf16, "x")
    x = x.contiguous()
    out = torch.empty(x.shape, device=x.device, dtype=u8)
    with _Timed("cast_e4m3", 0.0, 3.0 * x.numel()):
        lib.call("clipa_cast_bf16_to_e4m3", _p(x), _p(out), x.numel(), _stream())
    return out


def e4m3_to_bf16(x8):
    """e4m3 bytes (uint8) -> bf16 (exact)."""
    _chk(x8, u8, "x8")
    x8 = x8.contiguous()
    out = torch.empty(x8.shape, device=x8.device, dtype=bf16)
    with _Timed("cast_e4m3", 0.0, 3.0 * x8.numel()):
        lib.call("clipa_cast_e4m3_to_bf16", _p(x8), _p(out), x8.numel(), _stream())
    return out


def gemm_nt(a, b, bias=None, *, epi=EPI_NONE, act=ACT_GELU_ERF, aux=None, alpha=1.0, out_f32=False,
            want_pre=False, out=None, want_act=False):
    """C[M,N] = epi(alpha * a[M,K] @ b[N,K]^T + bias). a, b bf16; bias f32 [N].
    want_pre: True -> also the bf16 pre-activation; "e4m3" -> it as saturating e4m3 bytes (uint8 [M,N], the "light8" keep tier:
    fused into the epilogue on whole-tile shapes, GEMM + cast otherwise).  aux of EPI_DACT may be such a uint8 tensor;
    want_act (with it): -> (C, act(aux) as bf16 [M,N]) - what activation_fwd(aux, act) returns, written by the epilogue that
    reads the bytes anyway on whole-tile shapes (by activation_fwd otherwise)."""
    if want_act:
        if not (epi == EPI_DACT and aux is not None and aux.dtype == u8 and not want_pre and not out_f32):
            raise RuntimeError("gemm_nt: want_act goes with EPI_DACT on an e4m3 (uint8) second operand")
        if not _whole_tiles(a.shape[0], b.shape[0], a.shape[1]) or (out is not None and out.stride(0) != b.shape[0]):
            return gemm_nt(a, b, bias, epi=epi, act=act, aux=aux, alpha=alpha, out=out), activation_fwd(aux, act)
    if want_pre == "e4m3" and not (epi == EPI_ACT and not out_f32 and _whole_tiles(a.shape[0], b.shape[0], a.shape[1])):
        o, pre = gemm_nt(a, b, bias, epi=epi, act=act, aux=aux, alpha=alpha, want_pre=True, out=out)
        return o, cast_e4m3(pre)
    if aux is not None and aux.dtype == u8:
        if epi != EPI_DACT:
            raise RuntimeError("gemm_nt: an e4m3 (uint8) second operand goes with EPI_DACT only")
        if not _whole_tiles(a.shape[0], b.shape[0], a.shape[1]):
            aux = e4m3_to_bf16(aux)
    _chk(a, bf16, "a", 2)
    _chk(b, bf16, "b", 2)
    a, lda = _rowmajor(a)
    b, ldb = _rowmajor(b)
    M, K = a.shape
    N, Kb = b.shape
    if K != Kb:
        raise RuntimeError(f"gemm_nt: K mismatch {K} vs {Kb}")
    if bias is not None:
        _chk(bias, f32, "bias", 1)
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=f32 if out_f32 else bf16)
    ldc = out.stride(0) if M > 1 else N
    pre8 = want_pre == "e4m3"
    pre = torch.empty((M, N), device=a.device, dtype=u8 if pre8 else bf16) if (want_pre or want_act) else None
    if pre8 and ldc != N:
        raise RuntimeError("gemm_nt: want_pre='e4m3' needs a dense output (the copy shares its row stride)")
    ldaux, aux_sz = 0, 2
    if aux is not None and aux.dtype == u8:
        _chk(aux, u8, "aux", 2)
        aux, ldaux = _rowmajor(aux)
        epi, aux_sz = EPI_DACT8, 1
    elif aux is not None:
        _chk(aux, bf16, "aux", 2)
        aux, ldaux = _rowmajor(aux)
    if pre8:
        epi = EPI_ACT_PRE8
    osz = 4 if out_f32 else 2
    nbytes = 2.0 * (M * K + N * K) + osz * M * N + (float(aux_sz) * M * N if aux is not None else 0) + \
        ((1.0 if pre8 else 2.0) * M * N if want_pre else 0)
    if want_act:
        nbytes += 2.0 * M * N
    tag_epi = {EPI_ACT_PRE8: "1+pre8", EPI_DACT8: "3,aux8+act" if want_act else "3,aux8"}.get(epi, f"{epi}{'+pre' if want_pre else ''}")
    with _Timed("gemm_nt", 2.0 * M * N * K, nbytes, f"{M},{N},{K},epi{tag_epi}{',f32' if out_f32 else ''}"):
        lib.call("clipa_gemm_nt", _p(a), _p(b), _p(out), _p(pre), _p(bias), _p(aux), M, N, K, lda, ldb, ldc, ldaux,
                 float(alpha), epi, act, 1 if out_f32 else 0, _stream())
    return (out, pre) if (want_pre or want_act) else out


FMT_E4M3, FMT_E5M2 = 0, 1
u8 = torch.uint8


def quantize_rows(x, fmt=FMT_E4M3, want_colsum=False, want_rownorm=False):
    """Row-scaled fp8 operand of a bf16 matrix: -> (q uint8 [M,K] holding OCP e4m3 / e5m2 bytes, dq f32 [M]) with
    x[r,:] ~ dq[r] * fp8(q[r,:]).  want_colsum: also sum_r x[r,:] (f32 [K]; the bias gradient when x is a layer's dY);
    want_rownorm (with want_colsum): also ||x[r,:]||_2 (f32 [M]: the row bound of the product this gradient feeds, row_bound)."""
    _chk(x, bf16, "x", 2)
    x, ld = _rowmajor(x)
    M, K = x.shape
    q = torch.empty((M, K), device=x.device, dtype=u8)
    dq = torch.empty(M, device=x.device, dtype=f32)
    if want_colsum:
        cs = torch.empty(K, device=x.device, dtype=f32)
        rn = torch.empty(M, device=x.device, dtype=f32) if want_rownorm else None
        wsb = lib.query("clipa_quantize_rows_colsum_workspace", M, K)
        ws = torch.empty(max(wsb, 4) // 4, device=x.device, dtype=f32)
        with _Timed("quantize_rows", 0.0, 3.0 * M * K, f"{M},{K},+colsum"):
            lib.call("clipa_quantize_rows_colsum", _p(x), _p(q), _p(dq), _p(cs), _p(rn), M, K, ld, K, int(fmt), _p(ws), wsb, _stream())
        return (q, dq, cs, rn) if want_rownorm else (q, dq, cs)
    with _Timed("quantize_rows", 0.0, 3.0 * M * K, f"{M},{K}"):
        lib.call("clipa_quantize_rows", _p(x), _p(q), _p(dq), M, K, ld, K, int(fmt), _stream())
    return q, dq


def layernorm_fwd_q8(x, gamma, beta, eps=1e-5, want_bf16=False, want_rownorm=False):
    """LayerNorm of bf16 rows emitting the e4m3 operand of the next GEMM: -> (y bf16 | None, q uint8, dq f32 [rows]) (+ the rows'
    L2 norms, f32 [rows], with want_rownorm)."""
    _chk(x, bf16, "x")
    _chk(gamma, f32, "gamma", 1)
    _chk(beta, f32, "beta", 1)
    x = x.contiguous()
    D = x.shape[-1]
    rows = x.numel() // D
    y = torch.empty(x.shape, device=x.device, dtype=bf16) if want_bf16 else None
    q = torch.empty(x.shape, device=x.device, dtype=u8)
    dq = torch.empty(rows, device=x.device, dtype=f32)
    rn = torch.empty(rows, device=x.device, dtype=f32) if want_rownorm else None
    with _Timed("ln_fwd_q8", 0.0, float(rows) * D * (3 + (2 if want_bf16 else 0)), f"{rows},{D}"):
        lib.call("clipa_layernorm_fwd_q8n", _p(x), _p(gamma), _p(beta), _p(y), _p(q), _p(dq), _p(rn), rows, D, float(eps), _stream())
    return (y, q, dq, rn) if want_rownorm else (y, q, dq)


def _whole_tiles_f8(M, N, K):
    """Shapes the four-wave fp8 GEMM takes (gemm_f8a.hip: f8a_eligible): the fused e4m3 epilogues exist only there."""
    return M % 256 == 0 and N % 256 == 0 and K % 256 == 0 and K >= 512


def row_bound(rownorm, wnorm, bmax=None, factor=1.13):
    """Predicted row scales of a GEMM output: bound[m] = factor * rownorm[m] * wnorm[0] + bmax[0] >= every |output| of row m
    (Cauchy-Schwarz; factor 1.13 covers the e4m3 rounding of both operands) -> (scale = bound / 448, inv = 448 / bound), f32 [M]."""
    _chk(rownorm, f32, "rownorm", 1)
    _chk(wnorm, f32, "wnorm", 1)
    if bmax is not None:
        _chk(bmax, f32, "bmax", 1)
    M = rownorm.numel()
    scale = torch.empty(M, device=rownorm.device, dtype=f32)
    inv = torch.empty(M, device=rownorm.device, dtype=f32)
    lib.call("clipa_row_bound", _p(rownorm.contiguous()), _p(wnorm), _p(bmax), float(factor), _p(scale), _p(inv), M, _stream())
    return scale, inv


def rownorm_max(w):
    """f32 device scalar [1] = max_r ||w[r,:]||_2 of a bf16 matrix (a weight operand: once per optimizer step)."""
    _chk(w, bf16, "w", 2)
    w, ld = _rowmajor(w)
    out = torch.empty(1, device=w.device, dtype=f32)
    lib.call("clipa_rownorm_max", _p(w), w.shape[0], w.shape[1], ld, _p(out), _stream())
    return out


def absmax(v):
    """f32 device scalar [1] = max |v[i]| of an f32 vector (a bias)."""
    _chk(v, f32, "v", 1)
    out = torch.empty(1, device=v.device, dtype=f32)
    lib.call("clipa_absmax_f32", _p(v.contiguous()), v.numel(), _p(out), _stream())
    return out


def gemm_nt_f8_emit(a8, sa, b8, sb, aux8, t, *, act=ACT_GELU_ERF, fmt_a=FMT_E4M3):
    """EPI_DACT from the kept e4m3 pre-activation that also emits the activation operand of the same layer's fp8 weight gradient:
    -> (C bf16 [M,N] = gemm_nt_f8(a8, sa, b8, sb, epi=EPI_DACT, aux=aux8), X8 uint8 [M,N] = scale_quantize_rows(aux8, sa, t, act=act)),
    X8 written by the GEMM's epilogue on whole-tile shapes (clipa_gemm_nt_f8_emit), by the two separate launches otherwise."""
    M, K = a8.shape
    N = b8.shape[0]
    if not (_whole_tiles_f8(M, N, K) and sa is not None):
        return (gemm_nt_f8(a8, sa, b8, sb, None, epi=EPI_DACT, act=act, aux=aux8, fmt_a=fmt_a),
                scale_quantize_rows(aux8, sa, t, act=act))
    _chk(a8, u8, "a8", 2)
    _chk(b8, u8, "b8", 2)
    _chk(aux8, u8, "aux8", 2)
    _chk(sa, f32, "sa", 1)
    _chk(t, f32, "t")
    a8, lda = _rowmajor(a8)
    b8, ldb = _rowmajor(b8)
    aux8, ldaux = _rowmajor(aux8)
    if b8.shape[1] != K or tuple(aux8.shape) != (M, N) or sa.numel() != M or (sb is not None and sb.numel() != N):
        raise RuntimeError("gemm_nt_f8_emit: shape mismatch")
    out = torch.empty((M, N), device=a8.device, dtype=bf16)
    x8 = torch.empty((M, N), device=a8.device, dtype=u8)
    with _Timed("gemm_nt_f8", 2.0 * M * N * K, 1.0 * (M * K + N * K) + 4.0 * M * N, f"{M},{N},{K},epi3,aux8+emit"):
        lib.call("clipa_gemm_nt_f8_emit", _p(a8), _p(b8), _p(sa), _p(sb), _p(out), _p(x8), _p(aux8), _p(t), M, N, K, lda, ldb, N, ldaux,
                 act, int(fmt_a), _stream())
    return out, x8


def gemm_nt_f8(a8, sa, b8, sb, bias=None, *, epi=EPI_NONE, act=ACT_GELU_ERF, aux=None, alpha=1.0, want_pre=False,
               fmt_a=FMT_E4M3, fmt_b=FMT_E4M3, out_scale=None, want_colsum=False):
    """C[M,N] bf16 = epi(alpha * sa[m] * sb[n] * a8[M,K] @ b8[N,K]^T + bias); a8, b8 uint8 tensors of fp8 bytes.
    want_pre: True -> also the bf16 pre-activation; "e4m3" -> it as saturating e4m3 bytes (uint8 [M,N]: fused into the epilogue on
    whole-tile shapes, GEMM + cast otherwise).  aux of EPI_DACT may be such a uint8 tensor.
    out_scale (f32 [M]; round 6): the output leaves as e4m3 bytes, row m = the bf16 result times out_scale[m] (the operand of the
    next GEMM with de-quantisation scale 1 / out_scale: row_bound) - written by the epilogue itself for EPI_ACT and for EPI_DACT
    from an e4m3 operand on whole-tile shapes, GEMM + scaled quantiser otherwise; want_colsum (EPI_DACT): also the column sums of
    the unscaled outputs (f32 [N]).  -> q8 [, pre] [, colsum]."""
    whole = _whole_tiles_f8(a8.shape[0], b8.shape[0], a8.shape[1]) and fmt_b == FMT_E4M3
    if out_scale is not None:
        fused = whole and ((epi == EPI_ACT and not want_colsum) or (epi == EPI_DACT and aux is not None and aux.dtype == u8 and bias is None))
        if not fused:
            r = gemm_nt_f8(a8, sa, b8, sb, bias, epi=epi, act=act, aux=aux, alpha=alpha, want_pre=want_pre, fmt_a=fmt_a, fmt_b=fmt_b)
            out, pre = r if want_pre else (r, None)
            ones = torch.ones(1, device=out.device, dtype=f32)
            res = [scale_quantize_rows(out, out_scale, ones)]
            if want_pre:
                res.append(pre)
            if want_colsum:
                res.append(colsum(out))
            return res[0] if len(res) == 1 else tuple(res)
        return _gemm_nt_f8q(a8, sa, b8, sb, bias, epi, act, aux, alpha, want_pre, fmt_a, out_scale, want_colsum)
    if want_pre == "e4m3" and not (epi == EPI_ACT and whole):
        o, pre = gemm_nt_f8(a8, sa, b8, sb, bias, epi=epi, act=act, aux=aux, alpha=alpha, want_pre=True, fmt_a=fmt_a, fmt_b=fmt_b)
        return o, cast_e4m3(pre)
    if aux is not None and aux.dtype == u8:
        if epi != EPI_DACT:
            raise RuntimeError("gemm_nt_f8: an e4m3 (uint8) second operand goes with EPI_DACT only")
        if not whole:
            aux = e4m3_to_bf16(aux)
    _chk(a8, u8, "a8", 2)
    _chk(b8, u8, "b8", 2)
    a8, lda = _rowmajor(a8)
    b8, ldb = _rowmajor(b8)
    M, K = a8.shape
    N, Kb = b8.shape
    if K != Kb:
        raise RuntimeError(f"gemm_nt_f8: K mismatch {K} vs {Kb}")
    for name, t, n in (("sa", sa, M), ("sb", sb, N)):
        if t is not None:
            _chk(t, f32, name, 1)
            if t.numel() != n or not t.is_contiguous():
                raise RuntimeError(f"gemm_nt_f8: {name} must be a contiguous f32 vector of {n} elements")
    if bias is not None:
        _chk(bias, f32, "bias", 1)
    pre8 = want_pre == "e4m3"
    out = torch.empty((M, N), device=a8.device, dtype=bf16)
    pre = torch.empty((M, N), device=a8.device, dtype=u8 if pre8 else bf16) if want_pre else None
    ldaux, aux_sz = 0, 2
    if aux is not None and aux.dtype == u8:
        _chk(aux, u8, "aux", 2)
        aux, ldaux = _rowmajor(aux)
        epi, aux_sz = EPI_DACT8, 1
    elif aux is not None:
        _chk(aux, bf16, "aux", 2)
        aux, ldaux = _rowmajor(aux)
    if pre8:
        epi = EPI_ACT_PRE8
    nbytes = 1.0 * (M * K + N * K) + 2.0 * M * N + (float(aux_sz) * M * N if aux is not None else 0) + \
        ((1.0 if pre8 else 2.0) * M * N if want_pre else 0)
    tag_epi = {EPI_ACT_PRE8: "1+pre8", EPI_DACT8: "3,aux8"}.get(epi, f"{epi}{'+pre' if want_pre else ''}")
    with _Timed("gemm_nt_f8", 2.0 * M * N * K, nbytes, f"{M},{N},{K},epi{tag_epi}"):
        lib.call("clipa_gemm_nt_f8", _p(a8), _p(b8), _p(sa), _p(sb), _p(out), _p(pre), _p(bias), _p(aux), M, N, K, lda, ldb,
                 N, ldaux, float(alpha), epi, act, int(fmt_a), int(fmt_b), _stream())
    return (out, pre) if want_pre else out


def _gemm_nt_f8q(a8, sa, b8, sb, bias, epi, act, aux, alpha, want_pre, fmt_a, out_scale, want_colsum):
    """The fused form of gemm_nt_f8(out_scale=...) (clipa_gemm_nt_f8q; whole-tile shapes)."""
    _chk(a8, u8, "a8", 2)
    _chk(b8, u8, "b8", 2)
    _chk(out_scale, f32, "out_scale", 1)
    a8, lda = _rowmajor(a8)
    b8, ldb = _rowmajor(b8)
    M, K = a8.shape
    N = b8.shape[0]
    if out_scale.numel() != M:
        raise RuntimeError(f"gemm_nt_f8: out_scale has {out_scale.numel()} entries for {M} rows")
    for name, t, n in (("sa", sa, M), ("sb", sb, N)):
        if t is not None:
            _chk(t, f32, name, 1)
            if t.numel() != n or not t.is_contiguous():
                raise RuntimeError(f"gemm_nt_f8: {name} must be a contiguous f32 vector of {n} elements")
    q = torch.empty((M, N), device=a8.device, dtype=u8)
    pre8 = want_pre == "e4m3"
    pre = torch.empty((M, N), device=a8.device, dtype=u8 if pre8 else bf16) if want_pre else None
    ldaux, code, part = 0, epi, None
    if epi == EPI_DACT:
        _chk(aux, u8, "aux", 2)
        aux, ldaux = _rowmajor(aux)
        code = EPI_DACT8
        part = torch.empty((M // 128, N), device=a8.device, dtype=f32)
    elif pre8:
        code = EPI_ACT_PRE8
    nbytes = 1.0 * (M * K + N * K) + 1.0 * M * N + (1.0 * M * N if aux is not None else 0) + ((1.0 if pre8 else 2.0) * M * N if want_pre else 0)
    tag = {EPI_ACT_PRE8: "1+pre8", EPI_DACT8: "3,aux8"}.get(code, f"{epi}{'+pre' if want_pre else ''}")
    with _Timed("gemm_nt_f8", 2.0 * M * N * K, nbytes, f"{M},{N},{K},epi{tag},q8"):
        lib.call("clipa_gemm_nt_f8q", _p(a8), _p(b8), _p(sa), _p(sb), _p(q), _p(pre), _p(bias), _p(aux), _p(out_scale.contiguous()),
                 _p(part), M, N, K, lda, ldb, N, ldaux, float(alpha), code, act, int(fmt_a), _stream())
    res = [q]
    if want_pre:
        res.append(pre)
    if want_colsum:
        cs = torch.empty(N, device=a8.device, dtype=f32)
        lib.call("clipa_reduce_partial_rows", _p(part), _p(cs), M // 128, N, _stream())
        res.append(cs)
    return res[0] if len(res) == 1 else tuple(res)


def rowscale_max(a, b=None):
    """f32 device scalar [1] = max_m a[m] * b[m] (b None = 1): the tensor scale t of an fp8 weight-gradient operand."""
    _chk(a, f32, "a", 1)
    a = a.contiguous()
    if b is not None:
        _chk(b, f32, "b", 1)
        b = b.contiguous()
        if b.numel() != a.numel():
            raise RuntimeError(f"rowscale_max: {a.numel()} vs {b.numel()} rows")
    out = torch.empty(1, device=a.device, dtype=f32)
    lib.call("clipa_rowscale_max", _p(a), _p(b), a.numel(), _p(out), _stream())
    return out


def scale_quantize_rows(x, rowscale, t, act=-1):
    """q[m,:] = e4m3(act(x[m,:]) * rowscale[m] / t[0]) - the activation operand of an fp8 weight gradient (act -1 = none).
    x bf16, or uint8 e4m3 bytes (the kept pre-activation of the "h8" tier)."""
    in8 = x.dtype == u8
    _chk(x, u8 if in8 else bf16, "x", 2)
    _chk(rowscale, f32, "rowscale", 1)
    _chk(t, f32, "t", 1)
    x, ld = _rowmajor(x)
    M, K = x.shape
    if rowscale.numel() != M:
        raise RuntimeError(f"scale_quantize_rows: {rowscale.numel()} row scales for {M} rows")
    q = torch.empty((M, K), device=x.device, dtype=u8)
    with _Timed("scale_quantize_rows", 0.0, (2.0 if in8 else 3.0) * M * K, f"{M},{K},act{act}{',in8' if in8 else ''}"):
        lib.call("clipa_scale_quantize_rows_e4m3" if in8 else "clipa_scale_quantize_rows", _p(x), _p(rowscale.contiguous()), _p(t),
                 _p(q), M, K, ld, K, int(act), _stream())
    return q


def layernorm_fwd_q8s(x, gamma, beta, rowscale, t, eps=1e-5):
    """q[m,:] = e4m3(LayerNorm(x)[m,:] * rowscale[m] / t[0]): the LayerNorm output as the activation operand of an fp8 weight gradient."""
    _chk(x, bf16, "x")
    _chk(gamma, f32, "gamma", 1)
    _chk(beta, f32, "beta", 1)
    _chk(rowscale, f32, "rowscale", 1)
    _chk(t, f32, "t", 1)
    x = x.contiguous()
    D = x.shape[-1]
    rows = x.numel() // D
    if rowscale.numel() != rows:
        raise RuntimeError(f"layernorm_fwd_q8s: {rowscale.numel()} row scales for {rows} rows")
    q = torch.empty(x.shape, device=x.device, dtype=u8)
    with _Timed("ln_fwd_q8", 0.0, 3.0 * rows * D, f"{rows},{D},s"):
        lib.call("clipa_layernorm_fwd_q8s", _p(x), _p(gamma), _p(beta), _p(rowscale.contiguous()), _p(t), _p(q), rows, D, float(eps), _stream())
    return q


def gemm_tn_f8(p8, q8, t=None, alpha=1.0, fmt_p=FMT_E4M3, out_dtype=f32):
    """out[R,C] = alpha * t[0] * p8[M,R]^T @ q8[M,C]; p8 (fmt_p: e4m3 / e5m2), q8 (e4m3) uint8 tensors of fp8 bytes, t an
    optional f32 device scalar."""
    _chk(p8, u8, "p8", 2)
    _chk(q8, u8, "q8", 2)
    p8, ldp = _rowmajor(p8)
    q8, ldq = _rowmajor(q8)
    M, R = p8.shape
    M2, C = q8.shape
    if M != M2:
        raise RuntimeError(f"gemm_tn_f8: M mismatch {M} vs {M2}")
    if t is not None:
        _chk(t, f32, "t", 1)
    wsb = lib.query("clipa_gemm_tn_f8_workspace", M, R, C)
    ws = torch.empty(max(wsb, 4) // 4, device=p8.device, dtype=f32)
    out = torch.empty((R, C), device=p8.device, dtype=out_dtype)
    with _Timed("gemm_tn_f8", 2.0 * M * R * C, 1.0 * M * (R + C) + out.element_size() * R * C, f"{M},{R},{C}"):
        lib.call("clipa_gemm_tn_f8", _p(p8), _p(q8), _p(out), M, R, C, ldp, ldq, float(alpha), _p(t), int(fmt_p),
                 1 if out_dtype == bf16 else 0, _p(ws), wsb, _stream())
    return out


def gemm_tn(p, q, out_dtype=f32, want_colsum=False):
    """out[R,C] = p[M,R]^T @ q[M,C]; p, q bf16.  want_colsum: also return sum_m p[m,:] (f32 [R])."""
    _chk(p, bf16, "p", 2)
    _chk(q, bf16, "q", 2)
    p, ldp = _rowmajor(p)
    q, ldq = _rowmajor(q)
    M, R = p.shape
    M2, C = q.shape
    if M != M2:
        raise RuntimeError(f"gemm_tn: M mismatch {M} vs {M2}")
    ns = ctypes.c_int64(0)
    wsb = lib.query("clipa_gemm_tn_workspace", M, R, C, ctypes.byref(ns))
    ws = torch.empty(max(wsb, 4) // 4, device=p.device, dtype=f32)
    out = torch.empty((R, C), device=p.device, dtype=out_dtype)
    cs = torch.empty(R, device=p.device, dtype=f32) if want_colsum else None
    with _Timed("gemm_tn", 2.0 * M * R * C, 2.0 * M * (R + C) + out.element_size() * R * C, f"{M},{R},{C}"):
        lib.call("clipa_gemm_tn", _p(p), _p(q), _p(out), _p(cs), M, R, C, ldp, ldq, 1 if out_dtype == bf16 else 0, _p(ws),
                 wsb, _stream())
    return (out, cs) if want_colsum else out


def layernorm_fwd(x, gamma, beta, eps=1e-5, out_dtype=None):
    _chk(gamma, f32, "gamma", 1)
    _chk(beta, f32, "beta", 1)
    x = x.contiguous()
    D = x.shape[-1]
    rows = x.numel() // D
    out_dtype = out_dtype or x.dtype
    y = torch.empty(x.shape, device=x.device, dtype=out_dtype)
    with _Timed("ln_fwd", 0.0, float(rows) * D * (x.element_size() + y.element_size()), f"{rows},{D}"):
        lib.call("clipa_layernorm_fwd", _p(x), _p(gamma), _p(beta), _p(y), rows, D, float(eps),
                 int(x.dtype == f32), int(out_dtype == f32), _stream())
    return y


def layernorm_bwd(x, gamma, dy, dres=None, eps=1e-5, beta=None, q8_fmt=None, want_rownorm=False):
    """Returns dx (dtype of x, + dres if given), dgamma, dbeta (f32) - and, when `beta` is given, y = LayerNorm(x) (dtype of
    dy) as a fourth value: bit for bit what layernorm_fwd returns, written by the pass that has the rows in registers anyway.
    q8_fmt (FMT_E4M3 / FMT_E5M2; bf16 tensors): a last value (q, dq, colsum[, rownorm]) = quantize_rows(dx, q8_fmt, want_colsum=True
    [, want_rownorm=True]) - the fp8 operand of the linear layer this gradient reaches next, from the same pass."""
    x = x.contiguous()
    dy = dy.contiguous()
    D = x.shape[-1]
    rows = x.numel() // D
    if q8_fmt is not None:
        _chk(x, bf16, "x")
        _chk(dy, bf16, "dy")
        if dres is not None:
            dres = dres.contiguous()
            _chk(dres, bf16, "dres")
        wsb = lib.query("clipa_layernorm_bwd_q8_workspace", rows, D)
        ws = torch.empty(max(wsb, 4) // 4, device=x.device, dtype=f32)
        dx = torch.empty_like(x)
        dgamma = torch.empty(D, device=x.device, dtype=f32)
        dbeta = torch.empty(D, device=x.device, dtype=f32)
        q = torch.empty((rows, D), device=x.device, dtype=u8)
        dq = torch.empty(rows, device=x.device, dtype=f32)
        cs = torch.empty(D, device=x.device, dtype=f32)
        rn = torch.empty(rows, device=x.device, dtype=f32) if want_rownorm else None
        y = None
        if beta is not None:
            _chk(beta, f32, "beta", 1)
            y = torch.empty(x.shape, device=x.device, dtype=bf16)
        nbytes = float(rows) * D * (7 + (2 if dres is not None else 0) + (2 if y is not None else 0))
        with _Timed("ln_bwd", 0.0, nbytes, f"{rows},{D},+q8{',+y' if y is not None else ''}"):
            lib.call("clipa_layernorm_bwd_q8", _p(x), _p(gamma), _p(beta), _p(dy), _p(dres), _p(dx), _p(y), _p(q), _p(dq), _p(cs), _p(rn),
                     _p(dgamma), _p(dbeta), rows, D, float(eps), int(q8_fmt), _p(ws), wsb, _stream())
        q8 = (q, dq, cs, rn) if want_rownorm else (q, dq, cs)
        return (dx, dgamma, dbeta, y, q8) if y is not None else (dx, dgamma, dbeta, q8)
    wsb = lib.query("clipa_layernorm_bwd_workspace", rows, D)
    ws = torch.empty(max(wsb, 4) // 4, device=x.device, dtype=f32)
    dx = torch.empty_like(x)
    dgamma = torch.empty(D, device=x.device, dtype=f32)
    dbeta = torch.empty(D, device=x.device, dtype=f32)
    if dres is not None:
        dres = dres.contiguous()
        if dres.dtype != x.dtype:
            raise RuntimeError("layernorm_bwd: dres dtype must match x")
    nbytes = float(rows) * D * (2 * x.element_size() + dy.element_size() + (x.element_size() if dres is not None else 0))
    if beta is not None:
        _chk(beta, f32, "beta", 1)
        y = torch.empty(x.shape, device=x.device, dtype=dy.dtype)
        with _Timed("ln_bwd", 0.0, nbytes + float(rows) * D * y.element_size(), f"{rows},{D},+y"):
            lib.call("clipa_layernorm_bwd_y", _p(x), _p(gamma), _p(beta), _p(dy), _p(dres), _p(dx), _p(y), _p(dgamma), _p(dbeta),
                     rows, D, float(eps), int(x.dtype == f32), int(dy.dtype == f32), _p(ws), wsb, _stream())
        return dx, dgamma, dbeta, y
    with _Timed("ln_bwd", 0.0, nbytes, f"{rows},{D}"):
        lib.call("clipa_layernorm_bwd", _p(x), _p(gamma), _p(dy), _p(dres), _p(dx), _p(dgamma), _p(dbeta), rows, D,
                 float(eps), int(x.dtype == f32), int(dy.dtype == f32), _p(ws), wsb, _stream())
    return dx, dgamma, dbeta


def attention_fwd(qkv, B, L, H, causal, want_stats=False):
    """qkv [B*L, 3*H*64] bf16 (q | k | v column blocks) -> out [B*L, H*64] bf16 (+ softmax statistics
    f32 [B*H*L, 2] for the backward when want_stats)."""
    _chk(qkv, bf16, "qkv", 2)
    D = qkv.shape[1] // 3
    dh = D // H
    out = torch.empty((B * L, D), device=qkv.device, dtype=bf16)
    base = qkv.data_ptr()
    ld = qkv.stride(0)
    stats = torch.empty((B * H * L, 2), device=qkv.device, dtype=f32) if want_stats else None
    with _Timed("attention_fwd", 4.0 * B * H * L * L * dh * (0.5 if causal else 1.0), 0.0, f"B{B},H{H},L{L},dh{dh}"):
        lib.call("clipa_attention_fwd", ctypes.c_void_p(base), ctypes.c_void_p(base + 2 * D),
                 ctypes.c_void_p(base + 4 * D), _p(out), _p(stats), B, H, L, dh, ld, D, 1.0 / math.sqrt(dh), int(causal),
                 _stream())
    return (out, stats) if want_stats else out


def attention_bwd(qkv, out, dout, stats, B, L, H, causal):
    _chk(qkv, bf16, "qkv", 2)
    _chk(stats, f32, "stats", 2)
    _chk(out, bf16, "out", 2)
    _chk(dout, bf16, "dout", 2)
    dout = dout.contiguous()
    D = qkv.shape[1] // 3
    dh = D // H
    dqkv = torch.empty_like(qkv)
    base, dbase = qkv.data_ptr(), dqkv.data_ptr()
    with _Timed("attention_bwd", 10.0 * B * H * L * L * dh * (0.5 if causal else 1.0), 0.0, f"B{B},H{H},L{L},dh{dh}"):
        lib.call("clipa_attention_bwd", ctypes.c_void_p(base), ctypes.c_void_p(base + 2 * D),
                 ctypes.c_void_p(base + 4 * D), _p(out), _p(dout), _p(stats), ctypes.c_void_p(dbase), ctypes.c_void_p(dbase + 2 * D),
                 ctypes.c_void_p(dbase + 4 * D), B, H, L, dh, qkv.stride(0), D, dqkv.stride(0), 1.0 / math.sqrt(dh),
                 int(causal), _stream())
    return dqkv


class VarLen:
    """Packed variable-length sequences (the text tower on the tokens up to each caption's EOT): host-built index structure.
    lens: int64 CPU tensor [B] (>= 1 each).  Sequence b occupies rows [start[b], start[b] + lens[b]) of a [rows, D] token matrix
    whose row count is padded to a multiple of 256 (whole GEMM tiles; the pad rows are zero and belong to no sequence)."""

    def __init__(self, lens, ctx, device):
        lens = lens.to(torch.int64).cpu()
        self.B, self.ctx = int(lens.numel()), int(ctx)
        start = torch.cumsum(lens, 0) - lens
        self.T = int(lens.sum())
        self.rows = max(256, (self.T + 255) // 256 * 256)
        self.lens_host = lens
        put = (lambda t: t.pin_memory().to(device, non_blocking=True)) if torch.device(device).type == "cuda" else (lambda t: t)
        self.seq_start = put(start.to(torch.int32))
        self.seq_len = put(lens.to(torch.int32))
        # packed row -> row of the padded [B * ctx, D] matrix (-1 = pad row: gathers zeros, receives no gradient)
        src = torch.repeat_interleave(torch.arange(self.B) * self.ctx - start, lens) + torch.arange(self.T)
        self.src_rows = put(torch.cat([src, torch.full((self.rows - self.T,), -1, dtype=torch.int64)]))
        self.last_rows = put(start + lens - 1)                     # the EOT token of every sequence, in packed rows
        tiles = (lens + 31) // 32
        self.classes = []                                          # (32-row tiles, int32 device ids, count)
        for k in sorted(set(tiles.tolist())):
            ids = torch.nonzero(tiles == k).flatten().to(torch.int32)
            self.classes.append((int(k), put(ids), int(ids.numel())))
        self.sum_len2 = float((lens.double() ** 2).sum())          # attention work: sum of len^2


def attention_fwd_varlen(qkv, vl, H, causal, want_stats=False):
    """attention_fwd on packed sequences: qkv [vl.rows, 3D] -> out [vl.rows, D] (pad rows zero) (+ statistics [vl.rows * H, 2])."""
    _chk(qkv, bf16, "qkv", 2)
    D = qkv.shape[1] // 3
    dh = D // H
    out = torch.zeros((vl.rows, D), device=qkv.device, dtype=bf16)
    stats = torch.empty((vl.rows * H, 2), device=qkv.device, dtype=f32)
    base, ld = qkv.data_ptr(), qkv.stride(0)
    with _Timed("attention_fwd", 4.0 * H * vl.sum_len2 * dh * (0.5 if causal else 1.0), 0.0, f"varlen{vl.B},H{H},T{vl.T},dh{dh}"):
        for tiles, ids, n in vl.classes:
            lib.call("clipa_attention_fwd_varlen", ctypes.c_void_p(base), ctypes.c_void_p(base + 2 * D), ctypes.c_void_p(base + 4 * D),
                     _p(out), _p(stats), _p(vl.seq_start), _p(vl.seq_len), _p(ids), n, tiles, H, dh, ld, D, 1.0 / math.sqrt(dh),
                     int(causal), _stream())
    return (out, stats) if want_stats else out


def attention_bwd_varlen(qkv, out, dout, stats, vl, H, causal):
    _chk(qkv, bf16, "qkv", 2)
    _chk(stats, f32, "stats", 2)
    dout = dout.contiguous()
    D = qkv.shape[1] // 3
    dh = D // H
    dqkv = torch.zeros_like(qkv)                                   # pad rows belong to no sequence: their gradient is zero
    base, dbase = qkv.data_ptr(), dqkv.data_ptr()
    with _Timed("attention_bwd", 10.0 * H * vl.sum_len2 * dh * (0.5 if causal else 1.0), 0.0, f"varlen{vl.B},H{H},T{vl.T},dh{dh}"):
        for tiles, ids, n in vl.classes:
            lib.call("clipa_attention_bwd_varlen", ctypes.c_void_p(base), ctypes.c_void_p(base + 2 * D), ctypes.c_void_p(base + 4 * D),
                     _p(out), _p(dout), _p(stats), ctypes.c_void_p(dbase), ctypes.c_void_p(dbase + 2 * D), ctypes.c_void_p(dbase + 4 * D),
                     _p(vl.seq_start), _p(vl.seq_len), _p(ids), n, tiles, H, dh, qkv.stride(0), D, dqkv.stride(0), 1.0 / math.sqrt(dh),
                     int(causal), _stream())
    return dqkv


def patchify(img, P, Kp, mean=None, std=None):
    """img [B,3,S,S] (NCHW or channels_last memory), u8 / bf16 / f32 -> bf16 [B*(S//P)^2, Kp]."""
    if not img.is_cuda:
        raise RuntimeError("patchify: image must be on the GPU")
    B, C, S, S2 = img.shape
    if C != 3 or S != S2:
        raise RuntimeError(f"patchify: expected [B,3,S,S], got {tuple(img.shape)}")
    if img.is_contiguous():
        nhwc = 0
    elif img.is_contiguous(memory_format=torch.channels_last):
        nhwc = 1
    else:
        img, nhwc = img.contiguous(), 0
    dt = {torch.uint8: DT_U8, bf16: DT_BF16, f32: DT_F32}.get(img.dtype)
    if dt is None:
        raise RuntimeError(f"patchify: unsupported image dtype {img.dtype}")
    g = S // P
    out = torch.empty((B * g * g, Kp), device=img.device, dtype=bf16)
    normalize = mean is not None
    m3 = (ctypes.c_float * 3)(*([float(v) for v in mean] if normalize else [0, 0, 0]))
    s3 = (ctypes.c_float * 3)(*([float(v) for v in std] if normalize else [1, 1, 1]))
    with _Timed("patchify", 0.0, float(img.numel()) * img.element_size() + 2.0 * out.numel(), f"{B},{S},{P}"):
        lib.call("clipa_patchify", _p(img), _p(out), B, S, P, Kp, dt, nhwc, int(normalize), m3, s3, _stream())
    return out


def resized_crop_u8(src, boxes, size, gray_flags=None):
    """Device-side RandomResizedCrop (+ Grayscale of the flagged samples), bit-exact with Pillow's bicubic resize of the
    cropped image: src uint8 [B,Hs,Ws,3] (NHWC), boxes int32 [B,4] = (top, left, height, width) -> uint8 [B,size,size,3].
    Samples whose box leaves the image (or shrinks more than 11x) come back as zeros and raise at the next check."""
    _chk(src, u8, "src", 4)
    _chk(boxes, torch.int32, "boxes", 2)
    if src.shape[3] != 3 or not src.is_contiguous():
        raise RuntimeError(f"resized_crop_u8: src must be a contiguous uint8 [B,H,W,3] tensor, got {tuple(src.shape)}")
    B, Hs, Ws, _ = src.shape
    if tuple(boxes.shape) != (B, 4) or not boxes.is_contiguous():
        raise RuntimeError(f"resized_crop_u8: boxes must be contiguous int32 [{B},4]")
    if gray_flags is not None:
        _chk(gray_flags, u8, "gray_flags", 1)
        if gray_flags.numel() != B:
            raise RuntimeError("resized_crop_u8: gray_flags must have one byte per sample")
    check_token_ids()                                   # surfaces an earlier launch's rejected samples
    out = torch.empty((B, size, size, 3), device=src.device, dtype=u8)
    wsb = lib.query("clipa_resized_crop_workspace", B, Hs, size)
    ws = torch.empty(wsb, device=src.device, dtype=u8)
    err = _oob_counter(src.device)
    with _Timed("resized_crop", 0.0, 3.0 * B * (Hs * Ws + Hs * size * 2 + size * size)):
        lib.call("clipa_resized_crop_u8", _p(src), _p(boxes), _p(gray_flags), _p(out), B, Hs, Ws, size, _p(ws), wsb, _p(err),
                 _stream())
    _oob_submit(err, "resized_crop_u8", "samples rejected (crop box outside the image or a down-scale above 11x)")
    return out


def color_jitter_u8_(img, apply=None, order=None, factors=None, gray_flags=None):
    """In place on uint8 [B,S,S,3]: torchvision ColorJitter with per-sample op order (int32 [B,4] over 0 brightness, 1 contrast,
    2 saturation, 3 hue) and factors (f32 [B,4], indexed by op) for the samples with apply[b] != 0, then Grayscale(3) of the
    samples flagged in gray_flags - bit-exact with the Pillow code paths.  order=None: grayscale only."""
    _chk(img, u8, "img", 4)
    if img.shape[3] != 3 or img.shape[1] != img.shape[2] or not img.is_contiguous():
        raise RuntimeError(f"color_jitter_u8_: img must be a contiguous uint8 [B,S,S,3] tensor, got {tuple(img.shape)}")
    B, S = img.shape[0], img.shape[1]
    if order is not None:
        _chk(order, torch.int32, "order", 2)
        _chk(factors, f32, "factors", 2)
        if tuple(order.shape) != (B, 4) or tuple(factors.shape) != (B, 4) or not order.is_contiguous() or not factors.is_contiguous():
            raise RuntimeError("color_jitter_u8_: order int32 [B,4] and factors f32 [B,4], contiguous")
    for name, t in (("apply", apply), ("gray_flags", gray_flags)):
        if t is not None:
            _chk(t, u8, name, 1)
            if t.numel() != B:
                raise RuntimeError(f"color_jitter_u8_: {name} must have one byte per sample")
    ws = torch.empty(B, device=img.device, dtype=torch.int64)
    lib.call("clipa_color_jitter_u8", _p(img), _p(apply), _p(order), _p(factors), _p(gray_flags), B, S, _p(ws), B * 8, _stream())
    return img


def assemble_tokens(patch, cls, pos, B, L):
    D = patch.shape[1]
    tok = torch.empty((B * L, D), device=patch.device, dtype=bf16)
    lib.call("clipa_assemble_tokens", _p(patch), _p(cls), _p(pos), _p(tok), B, L, D, _stream())
    return tok


def assemble_tokens_bwd(dtok, B, L, need_pos=True):
    D = dtok.shape[1]
    dtok = dtok.contiguous()
    dpatch = torch.empty((B * (L - 1), D), device=dtok.device, dtype=bf16)
    dcls = torch.empty(D, device=dtok.device, dtype=f32)
    dpos = torch.empty((L, D), device=dtok.device, dtype=f32) if need_pos else None
    wsb = lib.query("clipa_assemble_tokens_bwd_workspace", B, L, D)
    ws = torch.empty(max(wsb, 4) // 4, device=dtok.device, dtype=f32)
    lib.call("clipa_assemble_tokens_bwd", _p(dtok), _p(dpatch), _p(dcls), _p(dpos), B, L, D, _p(ws), wsb, _stream())
    return dpatch, dcls, dpos


# Out-of-range token ids: nn.Embedding raises; the kernels count them into a device int32 instead of clamping silently.
# The count is fetched without stalling the stream (pinned buffer + event) and checked at the next embedding call or by
# `check_token_ids()`: a bad id surfaces as a RuntimeError at most one step late.
_OOB_PENDING = []
_OOB_LOCK = threading.Lock()      # appended to from the autograd thread (embed_tokens_bwd), drained from the main thread


def _oob_counter(device):
    return torch.zeros(1, device=device, dtype=torch.int32)


def _oob_submit(counter, what, msg="token ids outside [0, vocab) (nn.Embedding would raise)"):
    host = torch.empty(1, dtype=torch.int32, pin_memory=True)
    host.copy_(counter, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    with _OOB_LOCK:
        _OOB_PENDING.append((host, ev, (what, msg), counter))


def check_token_ids(wait=False):
    """Raise if an earlier embedding forward / backward met token ids outside the table (wait=True: block on all)."""
    with _OOB_LOCK:
        pending = list(_OOB_PENDING)
        _OOB_PENDING.clear()
    keep, bad = [], None
    for host, ev, what, counter in pending:
        if wait:
            ev.synchronize()
        if ev.query():
            if int(host[0]) != 0 and bad is None:
                bad = f"clipa_amd.ops.{what[0]}: {int(host[0])} {what[1]}"
        else:
            keep.append((host, ev, what, counter))
    if bad is not None:
        raise RuntimeError(bad)      # everything submitted so far is dropped with it
    if keep:
        with _OOB_LOCK:
            _OOB_PENDING[:0] = keep


def embed_tokens(ids, table, pos):
    _chk(ids, torch.int64, "ids", 2)
    _chk(pos, f32, "pos", 2)
    check_token_ids()
    ids = ids.contiguous()
    B, T = ids.shape
    V, D = table.shape
    out = torch.empty((B * T, D), device=ids.device, dtype=bf16)
    oob = _oob_counter(ids.device)
    lib.call("clipa_embed_tokens", _p(ids), _p(table.contiguous()), int(table.dtype == bf16), _p(pos.contiguous()), _p(out),
             B, T, D, V, _p(oob), _stream())
    _oob_submit(oob, "embed_tokens")
    return out


def embed_tokens_bwd(ids, dx, vocab, need_table=True, need_pos=True):
    ids = ids.contiguous()
    B, T = ids.shape
    D = dx.shape[1]
    dx = dx.contiguous()
    dtable = torch.empty((vocab, D), device=dx.device, dtype=f32) if need_table else None
    dpos = torch.empty((T, D), device=dx.device, dtype=f32) if need_pos else None
    oob = _oob_counter(ids.device)
    wsb = lib.query("clipa_embed_tokens_bwd_workspace", B, T, D, vocab, int(need_table), int(need_pos))
    ws = torch.empty(max(wsb, 8) // 8, device=dx.device, dtype=torch.int64)
    lib.call("clipa_embed_tokens_bwd", _p(ids), _p(dx), _p(dtable), _p(dpos), B, T, D, vocab, _p(oob), _p(ws), wsb, _stream())
    _oob_submit(oob, "embed_tokens_bwd")
    return dtable, dpos


def argmax_tokens(ids):
    ids = ids.contiguous()
    B, T = ids.shape
    out = torch.empty(B, device=ids.device, dtype=torch.int32)
    lib.call("clipa_argmax_tokens", _p(ids), _p(out), B, T, _stream())
    return out


def pool_fwd(x, B, L, mode, idx=None):
    D = x.shape[-1]
    out = torch.empty((B, D), device=x.device, dtype=f32)
    lib.call("clipa_pool_fwd", _p(x), _p(idx), _p(out), B, L, D, mode, _stream())
    return out


def pool_bwd(dout, B, L, mode, idx=None):
    D = dout.shape[-1]
    dx = torch.empty((B * L, D), device=dout.device, dtype=bf16)
    lib.call("clipa_pool_bwd", _p(dout.contiguous()), _p(idx), _p(dx), B, L, D, mode, _stream())
    return dx


def gather_rows(x, rows):
    """out[r] = x[rows[r]]: bf16 [n_src, D], int64 device row list -> bf16 [len(rows), D] (PatchDropout's token selection)."""
    _chk(x, bf16, "x", 2)
    _chk(rows, torch.int64, "rows", 1)
    x = x.contiguous()
    out = torch.empty((rows.numel(), x.shape[1]), device=x.device, dtype=bf16)
    with _Timed("gather_rows", 0.0, 4.0 * out.numel()):
        lib.call("clipa_gather_rows", _p(x), _p(rows), _p(out), rows.numel(), x.shape[0], x.shape[1], _stream())
    return out


def scatter_rows(dy, rows, n_dst):
    """dx [n_dst, D] = 0; dx[rows[r]] = dy[r] (distinct rows): the backward of gather_rows."""
    _chk(dy, bf16, "dy", 2)
    _chk(rows, torch.int64, "rows", 1)
    dy = dy.contiguous()
    dx = torch.empty((n_dst, dy.shape[1]), device=dy.device, dtype=bf16)
    with _Timed("scatter_rows", 0.0, 2.0 * (dx.numel() + 2 * dy.numel())):
        lib.call("clipa_scatter_rows", _p(dy), _p(rows), _p(dx), dy.shape[0], n_dst, dy.shape[1], _stream())
    return dx


def l2norm_fwd(x, eps=1e-12, want_bf16=False):
    x = x.contiguous()
    rows, E = x.shape
    y = torch.empty_like(x)
    ybf = torch.empty((rows, E), device=x.device, dtype=bf16) if want_bf16 else None
    inv = torch.empty(rows, device=x.device, dtype=f32)
    lib.call("clipa_l2norm_fwd", _p(x), _p(y), _p(ybf), _p(inv), rows, E, float(eps), _stream())
    return y, ybf, inv


def l2norm_bwd(y, inv, dy):
    rows, E = y.shape
    dx = torch.empty_like(y)
    lib.call("clipa_l2norm_bwd", _p(y), _p(inv), _p(dy.contiguous()), _p(dx), rows, E, _stream())
    return dx


def colsum(dy):
    _chk(dy, bf16, "dy", 2)
    dy, ld = _rowmajor(dy)
    M, N = dy.shape
    wsb = lib.query("clipa_colsum_workspace", M, N)
    ws = torch.empty(max(wsb, 4) // 4, device=dy.device, dtype=f32)
    out = torch.empty(N, device=dy.device, dtype=f32)
    lib.call("clipa_colsum", _p(dy), _p(out), M, N, ld, _p(ws), wsb, _stream())
    return out


def to_bf16(t):
    """Flat cast f32/bf16 -> new bf16 tensor (weights are re-cast once per optimizer step)."""
    t = t.contiguous()
    if t.dtype not in (f32, bf16):
        raise RuntimeError(f"to_bf16: unsupported dtype {t.dtype}")
    out = torch.empty(t.shape, device=t.device, dtype=bf16)
    lib.call("clipa_cast_to_bf16", _p(t), int(t.dtype == f32), _p(out), t.numel(), _stream())
    return out


def to_f32(t):
    t = t.contiguous()
    if t.dtype == f32:
        return t
    out = torch.empty(t.shape, device=t.device, dtype=f32)
    lib.call("clipa_cast_bf16_to_f32", _p(t), _p(out), t.numel(), _stream())
    return out


def transpose_bf16(t):
    """[R,C] f32/bf16 -> [C,R] bf16 (used for W^T operands)."""
    t, ldi = _rowmajor(t)
    R, C = t.shape
    out = torch.empty((C, R), device=t.device, dtype=bf16)
    lib.call("clipa_transpose_to_bf16", _p(t), int(t.dtype == f32), _p(out), R, C, ldi, R, _stream())
    return out


def activation_fwd(x, act):
    """bf16 act(x) (gelu erf / tanh / quick); x bf16, or e4m3 bytes (uint8: the "light8" keep tier's pre-activation)."""
    if x.dtype == u8:
        x = x.contiguous()
        out = torch.empty(x.shape, device=x.device, dtype=bf16)
        with _Timed("activation_fwd", 0.0, 3.0 * x.numel()):
            lib.call("clipa_activation_fwd_e4m3", _p(x), _p(out), x.numel(), act, _stream())
        return out
    _chk(x, bf16, "x")
    x = x.contiguous()
    out = torch.empty_like(x)
    with _Timed("activation_fwd", 0.0, 4.0 * x.numel()):
        lib.call("clipa_activation_fwd", _p(x), _p(out), x.numel(), act, _stream())
    return out


def simce(rows, cols, n_valid, label0, gscale, scale=None, want_grad=True):
    """Fused similarity + cross-entropy (InfoNCE): logits = s * rows @ cols[:n_valid]^T, labels label0 + row; s =
    scale[0] read on the device (None = 1).  rows [R,E], cols [>= n_valid, E] bf16.  Returns loss_rows f32 [R], the bf16
    d loss / d (rows @ cols^T) [R, n8] (n8 = n_valid rounded up to 8, pad columns zero) | None, and the per-row
    d loss / d s.  The fp32 logits never exist in HBM (the backward re-runs the similarity GEMM)."""
    _chk(rows, bf16, "rows", 2)
    _chk(cols, bf16, "cols", 2)
    rows, lda = _rowmajor(rows)
    cols, ldb = _rowmajor(cols)
    R, E = rows.shape
    if cols.shape[1] != E or cols.shape[0] < n_valid:
        raise RuntimeError(f"simce: cols {tuple(cols.shape)} does not cover {n_valid} x {E}")
    if scale is not None:
        _chk(scale, f32, "scale")
    n8 = (n_valid + 7) // 8 * 8
    wsb = lib.query("clipa_simce_workspace", R, n_valid)
    ws = torch.empty(max(wsb, 4) // 4, device=rows.device, dtype=f32)
    lse = torch.empty(R, device=rows.device, dtype=f32)
    loss_rows = torch.empty(R, device=rows.device, dtype=f32)
    with _Timed("simce", 2.0 * R * n_valid * E * (2 if want_grad else 1)):
        lib.call("clipa_simce_fwd", _p(rows), _p(cols), R, n_valid, E, lda, ldb, _p(scale), label0, _p(lse), _p(loss_rows), _p(ws),
                 wsb, _stream())
        if not want_grad:
            return loss_rows, None, None
        dl = torch.empty((R, n8), device=rows.device, dtype=bf16)
        dscale_rows = torch.empty(R, device=rows.device, dtype=f32)
        lib.call("clipa_simce_bwd", _p(rows), _p(cols), R, n_valid, E, lda, ldb, _p(scale), label0, float(gscale), _p(lse), _p(dl),
                 n8, _p(dscale_rows), _p(ws), wsb, _stream())
    return loss_rows, dl, dscale_rows


def sum_scale(x, scale, out=None, accumulate=False):
    x = x.contiguous()
    if out is None:
        out = torch.empty((), device=x.device, dtype=f32)
    lib.call("clipa_sum_scale", _p(x), _p(out), x.numel(), float(scale), int(accumulate), _stream())
    return out


def _chk_moments(p, m, v):
    for name, t in (("exp_avg", m), ("exp_avg_sq", v)):
        if t.dtype != f32 or not t.is_contiguous() or t.numel() != p.numel() or not t.is_cuda:
            raise RuntimeError(f"adamw: {name} must be a contiguous f32 GPU tensor with the parameter's numel (got {t.dtype}, "
                               f"{t.numel()} vs {p.numel()}) - the kernel reads and writes it as float*")


def adamw_(param, grad, exp_avg, exp_avg_sq, *, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    _chk_moments(param, exp_avg, exp_avg_sq)
    lib.call("clipa_adamw", _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), int(param.dtype == f32),
             int(grad.dtype == f32), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
             float(grad_scale), _stream())


def adamw_multi_(params, grads, exp_avgs, exp_avg_sqs, *, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0,
                 grad_scale_dev=None, clamp_index=-1, clamp=(0.0, 0.0)):
    """One AdamW update over a list of tensors that share dtypes, hyper-parameters and step count.  grad_scale_dev: f32
    device scalar multiplied into every gradient (clip coefficient); clamp_index: list position of the tensor that is
    clamped to `clamp` after its update."""
    n = len(params)
    if n == 0:
        return
    pf32, gf32 = params[0].dtype == f32, grads[0].dtype == f32
    for t, g, m, v in zip(params, grads, exp_avgs, exp_avg_sqs):
        if (t.dtype == f32) != pf32 or (g.dtype == f32) != gf32 or not t.is_cuda:
            raise RuntimeError("adamw_multi_: mixed dtypes in one call / tensors must live on the GPU")
        if t.dtype not in (f32, bf16) or g.dtype not in (f32, bf16) or g.numel() != t.numel() or not t.is_contiguous() \
                or not g.is_contiguous():
            raise RuntimeError("adamw_multi_: parameters / gradients must be contiguous f32 or bf16 tensors of equal numel")
        _chk_moments(t, m, v)
    if grad_scale_dev is not None:
        _chk(grad_scale_dev, f32, "grad_scale_dev")
    arr = ctypes.c_void_p * n
    cnt = (ctypes.c_int64 * n)(*[t.numel() for t in params])
    lib.call("clipa_adamw_multi", arr(*[t.data_ptr() for t in params]), arr(*[t.data_ptr() for t in grads]),
             arr(*[t.data_ptr() for t in exp_avgs]), arr(*[t.data_ptr() for t in exp_avg_sqs]), cnt, n, int(pf32),
             int(gf32), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
             float(grad_scale), _p(grad_scale_dev), int(clamp_index), float(clamp[0]), float(clamp[1]), _stream())


def grad_sqnorm(grads, buf=None):
    """buf[0] += sum of squares of the listed gradients (device f32 [3] = [sum of squares, norm, coef]; created zeroed)."""
    dev = grads[0].device
    if buf is None:
        buf = torch.zeros(3, device=dev, dtype=f32)
    for is32 in (True, False):
        sel = [g for g in grads if (g.dtype == f32) == is32]
        if not sel:
            continue
        for g in sel:
            if g.dtype not in (f32, bf16) or not g.is_contiguous() or not g.is_cuda:
                raise RuntimeError("grad_sqnorm: gradients must be contiguous f32 / bf16 GPU tensors")
        n = len(sel)
        arr = ctypes.c_void_p * n
        cnt = (ctypes.c_int64 * n)(*[g.numel() for g in sel])
        nblk = sum((g.numel() + 4095) // 4096 for g in sel)            # one partial per 4096-element block, summed in a fixed order
        part = torch.empty(max(nblk, 1), device=dev, dtype=f32)
        lib.call("clipa_grad_sqnorm_multi", arr(*[g.data_ptr() for g in sel]), cnt, n, int(is32), _p(buf), _p(part), nblk, _stream())
    return buf


def clip_coef(buf, max_norm):
    """buf[0] = sum of squares -> (total_norm, coef) device scalars (views of buf), coef = min(1, max_norm / (norm + 1e-6))."""
    lib.call("clipa_clip_coef", _p(buf), float(max_norm), ctypes.c_void_p(buf.data_ptr() + 4), ctypes.c_void_p(buf.data_ptr() + 8),
             _stream())
    return buf[1], buf[2]


def grad_clip_coef(grads, max_norm):
    """clip_grad_norm_ on the device: -> (total_norm, coef) f32 device scalars, coef = min(1, max_norm / (norm + 1e-6))."""
    return clip_coef(grad_sqnorm(grads), max_norm)


def reduce_shards(pieces, world, out=None, scale=None, out_dtype=None):
    """out[i] = scale * sum_w pieces[w*n + i] (scale defaults to 1/world): the local sum of a one-hop reduce-scatter."""
    if pieces.dtype not in (f32, bf16) or not pieces.is_contiguous() or not pieces.is_cuda:
        raise RuntimeError("reduce_shards: pieces must be a contiguous f32 / bf16 GPU tensor")
    n = pieces.numel() // world
    if n * world != pieces.numel():
        raise RuntimeError("reduce_shards: numel must be a multiple of world")
    if out is None:
        out = torch.empty(n, device=pieces.device, dtype=out_dtype or pieces.dtype)
    lib.call("clipa_reduce_shards", _p(pieces), _p(out), n, int(world), int(pieces.dtype == f32), int(out.dtype == f32),
             float(1.0 / world if scale is None else scale), _stream())
    return out
