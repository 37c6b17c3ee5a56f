// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of clipa_amd.
// wave = 64 lanes; MFMA 32x32x16 bf16; LDS-DMA (buffer_load ... lds); ds_read_b64_tr_b16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = 4 VGPRs (MFMA A/B operand)
typedef __attribute__((ext_vector_type(4))) short bf16x4;   // 4 bf16 = 2 VGPRs
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4;    // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// error codes of the C ABI
#define CLIPA_OK 0
#define CLIPA_ERR_ARG -1
#define CLIPA_ERR_LAUNCH -2

void clipa_set_error(const char* fmt, ...);
int clipa_check_launch(const char* what);

// ---- bf16 <-> f32 -------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(unsigned short h) {
  return __uint_as_float(((unsigned int)h) << 16);
}
// round-to-nearest-even via the gfx950 hardware converter (v_cvt_pk_bf16_f32: one VALU op per pair)
typedef __attribute__((ext_vector_type(2))) __bf16 hw_bf16x2;
__device__ __forceinline__ unsigned int pack2bf(float lo, float hi) {
  hw_bf16x2 v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned int, v);
}
__device__ __forceinline__ unsigned short f2bf(float f) {
  return __builtin_bit_cast(unsigned short, (__bf16)f);
}
__device__ __forceinline__ void unpack8(const u32x4 v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(v[i] << 16);
    f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack2bf(f[2 * i], f[2 * i + 1]);
  return v;
}

// ---- activations (transformer.py:37-40 QuickGELU, nn.GELU erf / tanh) ----------------------
enum { ACT_GELU_ERF = 0, ACT_GELU_TANH = 1, ACT_QUICK_GELU = 2 };

// erf-GELU on PAIRS of values, WITHOUT transcendentals: Phi(x) - 1/2 = erf(x / sqrt 2) / 2 is an odd function, fitted on
// |x| <= 4.5 as x * P(x^2) (P of degree 9) and gelu'(x) - 1/2 = erf(x / sqrt 2) / 2 + x phi(x) likewise (degree 10).
// Coefficients: tools/fit_gelu_poly.py.  Measured in fp32 (tests/test_host_cpu.py::test_gelu_polynomials):
//   x Phi(x): |error| <= 5.4e-5 on |x| <= 4.5 (8e-6 for |x| < 3); beyond 4.5 the argument is clamped and Phi is clamped to
//             [0, 1], which leaves a tail error of ~1.2e-5 |x| (gelu(-12) = -1.4e-4 where the exact value is ~ -1e-32);
//   gelu'(x): |error| <= 2.4e-4 at the clamp edge (1.3e-5 for |x| < 3); beyond it the value stays at ~ -2e-4 / 1.0002 where
//             the exact derivative is 0 / 1.
// Against the bf16 rounding of what the callers store this is invisible for outputs of magnitude >= 1e-2 (bf16 step >= 4e-5
// there) and it is NOT for the far negative tail, where the exact output is a tiny number and ours is ~1e-4 in magnitude: an
// absolute error of 1e-4 on activations that the next GEMM multiplies by weights of ~0.03 and sums with O(1) neighbours.  The
// parity suite bounds the effect end to end (features / loss / every gradient vs the fp32 oracle).  The epilogues of the fused
// GEMMs are VALU-bound (profiles/r03_gemm_nta_*.jsonl): ten packed FMAs per pair cost a third of the rcp + exp2 formulation
// they replace (Abramowitz-Stegun 7.1.26, 2 quarter-rate transcendentals per element).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 clamp2(f32x2 x, float lo, float hi) {
  f32x2 r;
  r.x = __builtin_amdgcn_fmed3f(x.x, lo, hi);
  r.y = __builtin_amdgcn_fmed3f(x.y, lo, hi);
  return r;
}
// NP pairs in LOCKSTEP: the Horner chain of one pair is ten dependent packed FMAs (and gfx950 wants a wait state between
// dependent v_pk_fma_f32), so the pairs of a 16-byte chunk advance together - four independent FMAs per coefficient.
constexpr float GELU_P[10] = {-1.400016797e-12f, 1.697253120e-10f, -9.193533687e-09f, 2.958848065e-07f, -6.365184678e-06f,
                              9.787074174e-05f, -1.122675229e-03f, 9.833175198e-03f, -6.633704203e-02f, 3.988837869e-01f};
constexpr float GELU_Q[11] = {1.054065974e-12f, -1.329809045e-10f, 7.512951167e-09f, -2.523850424e-07f, 5.655319288e-06f,
                              -8.997389735e-05f, 1.054214113e-03f, -9.228582120e-03f, 5.942921224e-02f, -2.656680481e-01f,
                              7.978225015e-01f};
template <int NP>
__device__ __forceinline__ void gelu_cdf_n(const f32x2 (&x)[NP], f32x2 (&phi)[NP]) {        // Phi(x)
  f32x2 xc[NP], t[NP], p[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) { xc[i] = clamp2(x[i], -4.5f, 4.5f); t[i] = xc[i] * xc[i]; p[i] = t[i] * GELU_P[0] + GELU_P[1]; }
#pragma unroll
  for (int k = 2; k < 10; ++k)
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = p[i] * t[i] + GELU_P[k];
#pragma unroll
  for (int i = 0; i < NP; ++i) phi[i] = clamp2(xc[i] * p[i] + 0.5f, 0.0f, 1.0f);
}
template <int NP>
__device__ __forceinline__ void gelu_grad_n(const f32x2 (&x)[NP], f32x2 (&g)[NP]) {         // Phi(x) + x phi(x)
  f32x2 xc[NP], t[NP], p[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) { xc[i] = clamp2(x[i], -4.5f, 4.5f); t[i] = xc[i] * xc[i]; p[i] = t[i] * GELU_Q[0] + GELU_Q[1]; }
#pragma unroll
  for (int k = 2; k < 11; ++k)
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = p[i] * t[i] + GELU_Q[k];
#pragma unroll
  for (int i = 0; i < NP; ++i) g[i] = xc[i] * p[i] + 0.5f;
}
__device__ __forceinline__ f32x2 gelu_cdf2(f32x2 x) {
  f32x2 a[1] = {x}, r[1];
  gelu_cdf_n<1>(a, r);
  return r[0];
}
__device__ __forceinline__ f32x2 gelu_grad2(f32x2 x) {
  f32x2 a[1] = {x}, r[1];
  gelu_grad_n<1>(a, r);
  return r[0];
}
template <int ACT>
__device__ __forceinline__ float act_fwd(float x);
template <int ACT>
__device__ __forceinline__ float act_bwd(float x);

template <int ACT>
__device__ __forceinline__ f32x2 act_fwd2(f32x2 x) {
  if (ACT == ACT_GELU_ERF) return x * gelu_cdf2(x);
  f32x2 r;
  r.x = act_fwd<ACT>(x.x);
  r.y = act_fwd<ACT>(x.y);
  return r;
}
template <int ACT>
__device__ __forceinline__ f32x2 act_bwd2(f32x2 x) {  // d act(x) / dx
  if (ACT == ACT_GELU_ERF) return gelu_grad2(x);
  f32x2 r;
  r.x = act_bwd<ACT>(x.x);
  r.y = act_bwd<ACT>(x.y);
  return r;
}
template <int ACT>
__device__ __forceinline__ float act_fwd(float x) {
  if (ACT == ACT_GELU_ERF) {
    return act_fwd2<ACT_GELU_ERF>(f32x2{x, x}).x;
  } else if (ACT == ACT_GELU_TANH) {
    float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    float e = __expf(2.0f * u);
    float th = 1.0f - 2.0f * __frcp_rn(e + 1.0f);
    return 0.5f * x * (1.0f + th);
  } else {
    return x * __frcp_rn(1.0f + __expf(-1.702f * x));
  }
}
template <int ACT>
__device__ __forceinline__ float act_bwd(float x) {  // d act(x) / dx
  if (ACT == ACT_GELU_ERF) {
    return act_bwd2<ACT_GELU_ERF>(f32x2{x, x}).x;
  } else if (ACT == ACT_GELU_TANH) {
    float x2 = x * x;
    float u = 0.7978845608028654f * (x + 0.044715f * x * x2);
    float e = __expf(2.0f * u);
    float th = 1.0f - 2.0f * __frcp_rn(e + 1.0f);
    float du = 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * x2);
    return 0.5f * (1.0f + th) + 0.5f * x * (1.0f - th * th) * du;
  } else {
    float s = __frcp_rn(1.0f + __expf(-1.702f * x));
    return s + x * 1.702f * s * (1.0f - s);
  }
}

// 8 fp32 values -> 8 e4m3 bytes (OCP, round to nearest even), SATURATING at +-448: a pre-activation beyond the format's range
// must not become NaN (e4m3 has no infinity).  And back (exact: every e4m3 value is a bf16 value).
__device__ __forceinline__ u32x2 e4m3x8_sat(const float* f) {
  float c[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_fmed3f(f[i], -448.0f, 448.0f);
  int w0 = 0, w1 = 0;
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], w0, false);
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], w0, true);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], w1, false);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], w1, true);
  u32x2 q;
  q[0] = (unsigned)w0;
  q[1] = (unsigned)w1;
  return q;
}
__device__ __forceinline__ void e4m3x8_to_f32(const u32x2 q, float* f) {
  typedef float f32x2v __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const f32x2v lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)q[h], false);
    const f32x2v hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)q[h], true);
    f[4 * h + 0] = lo.x; f[4 * h + 1] = lo.y; f[4 * h + 2] = hi.x; f[4 * h + 3] = hi.y;
  }
}

// ---- wave-level reductions (64 lanes) -------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- buffer resource (SRD) for bounds-checked loads: out-of-range lanes read 0 -------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

// XCD-aware bijective remap of a 1-D workgroup id: block b runs on XCD b%8; give every XCD a
// contiguous chunk of the logical tile sequence so neighbouring tiles share one L2.
__device__ __forceinline__ unsigned int xcd_remap(unsigned int bid, unsigned int nwg) {
  const unsigned int q = nwg >> 3, r = nwg & 7u;
  const unsigned int xcd = bid & 7u, idx = bid >> 3;
  const unsigned int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
