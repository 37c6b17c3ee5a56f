// Pieces shared by the row quantisers (quant.hip) and the kernels that emit a row-quantised operand themselves
// (layernorm.hip: the LayerNorm backward that hands the next layer its fp8 gradient operand).
#pragma once
#include "common.h"

namespace {

template <int FMT>
__device__ __forceinline__ u32x2 cvt8(const float* f, float s) {
  int w0 = 0, w1 = 0;
  if (FMT == 0) {
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * s, f[1] * s, w0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * s, f[3] * s, w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * s, f[5] * s, w1, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * s, f[7] * s, w1, true);
  } else {
    w0 = __builtin_amdgcn_cvt_pk_bf8_f32(f[0] * s, f[1] * s, w0, false);
    w0 = __builtin_amdgcn_cvt_pk_bf8_f32(f[2] * s, f[3] * s, w0, true);
    w1 = __builtin_amdgcn_cvt_pk_bf8_f32(f[4] * s, f[5] * s, w1, false);
    w1 = __builtin_amdgcn_cvt_pk_bf8_f32(f[6] * s, f[7] * s, w1, true);
  }
  u32x2 r;
  r[0] = (unsigned)w0;
  r[1] = (unsigned)w1;
  return r;
}

template <int FMT>
__device__ __forceinline__ void row_scales(float amax, float& s, float& dq) {
  const float FMAX = FMT == 0 ? 448.0f : 57344.0f;
  s = amax > 0.f ? FMAX / amax : 1.0f;
  dq = amax > 0.f ? amax / FMAX : 0.0f;      // an all-zero row (padding tokens): zero bytes, zero scale - the fp8 weight gradient's tensor scale is a maximum over these
}

}  // namespace
