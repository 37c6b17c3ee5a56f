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

// erf-GELU on PAIRS of values: the polynomial part compiles to packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32,
// two elements per VALU slot); only rcp and exp2 stay per element.  q(x) = Phi(-|x|) = 0.5*erfc(|x|/sqrt 2) by
// Abramowitz-Stegun 7.1.26 / 26.2.17 (|err| <= 0.75e-7, far below bf16 resolution):
//   q = 0.5 * t*(a1 + t*(a2 + t*(a3 + t*(a4 + t*a5)))) * exp(-x^2/2),  t = 1/(1 + 0.2316419*|x|)
//   gelu(x)  = x*Phi(x)           = max(x, 0) - |x|*q
//   gelu'(x) = Phi(x) + x*phi(x)  = (x >= 0 ? 1 - q : q) + x * exp(-x^2/2)/sqrt(2 pi)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_tail2(f32x2 x, f32x2 ax, f32x2& e) {
  const f32x2 d = ax * 0.2316419f + 1.0f;
  f32x2 t;
  t.x = __builtin_amdgcn_rcpf(d.x);
  t.y = __builtin_amdgcn_rcpf(d.y);
  const f32x2 u = (x * x) * -0.72134752044448170f;        // -x^2/2 * log2(e)
  e.x = __builtin_amdgcn_exp2f(u.x);
  e.y = __builtin_amdgcn_exp2f(u.y);
  const f32x2 p = ((((0.5307027145f * t - 0.7265760135f) * t + 0.7107068705f) * t - 0.142248368f) * t + 0.127414796f) * t;
  return p * e;
}
template <int ACT>
__device__ __forceinline__ float act_fwd(float x);
template <int ACT>
__device__ __forceinline__ float act_bwd(float x);

template <int ACT>
__device__ __forceinline__ f32x2 act_fwd2(f32x2 x) {
  if (ACT == ACT_GELU_ERF) {
    const f32x2 ax = {fabsf(x.x), fabsf(x.y)};
    f32x2 e;
    const f32x2 q = gelu_tail2(x, ax, e);
    const f32x2 relu = {fmaxf(x.x, 0.f), fmaxf(x.y, 0.f)};
    return relu - ax * q;
  }
  f32x2 r;
  r.x = act_fwd<ACT>(x.x);
  r.y = act_fwd<ACT>(x.y);
  return r;
}
template <int ACT>
__device__ __forceinline__ f32x2 act_bwd2(f32x2 x) {  // d act(x) / dx
  if (ACT == ACT_GELU_ERF) {
    const f32x2 ax = {fabsf(x.x), fabsf(x.y)};
    f32x2 e;
    const f32x2 q = gelu_tail2(x, ax, e);
    f32x2 cdf;
    cdf.x = x.x >= 0.f ? 1.0f - q.x : q.x;
    cdf.y = x.y >= 0.f ? 1.0f - q.y : q.y;
    return cdf + x * (e * 0.3989422804014327f);
  }
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
