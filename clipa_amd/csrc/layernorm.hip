// LayerNorm forward / backward for gfx950: one 64-lane wave per row, fp32 statistics.
// Replaces F.layer_norm in clipa_torch/open_clip/transformer.py:19-34 (LayerNorm / LayerNormFp32:
// biased variance, eps inside the sqrt, affine) and its autograd.
// HBM-bound: every row is read once with 16-byte loads and kept in registers for both passes.
#include "common.h"
#include "clipa_hip.h"

namespace {

// load / store 8 consecutive elements as float
template <bool F32>
__device__ __forceinline__ void ld8(const void* base, size_t elem, float* f) {
  if (F32) {
    const float4 a = *(const float4*)((const float*)base + elem);
    const float4 b = *(const float4*)((const float*)base + elem + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  } else {
    unpack8(*(const u32x4*)((const unsigned short*)base + elem), f);
  }
}
template <bool F32>
__device__ __forceinline__ void st8(void* base, size_t elem, const float* f) {
  if (F32) {
    *(float4*)((float*)base + elem) = make_float4(f[0], f[1], f[2], f[3]);
    *(float4*)((float*)base + elem + 4) = make_float4(f[4], f[5], f[6], f[7]);
  } else {
    *(u32x4*)((unsigned short*)base + elem) = pack8(f);
  }
}

// NCH = chunks of 8 elements per lane (D <= NCH*512)
template <int NCH, bool XF32, bool YF32>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const void* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, void* __restrict__ y,
                                                     long rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nw = (long)gridDim.x * 4;
  const int nchunks = D >> 3;
  float g[NCH][8], bt[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = lane + c * 64;
    if (ch < nchunks) { ld8<true>(gamma, (size_t)ch * 8, g[c]); ld8<true>(beta, (size_t)ch * 8, bt[c]); }
  }
  const float invD = 1.0f / (float)D;
  // TWO rows per wave and iteration: both rows' loads are in flight before the first reduction starts (one row of 2 KB per
  // wave does not cover the HBM latency at 32 waves per CU: 4.3 TB/s; the backward kernel, which has three loads per row in
  // flight, runs at the roof)
  for (long r = wid; r < rows; r += 2 * nw) {
    const long r2 = r + nw;
    const bool two = r2 < rows;
    float v[2][NCH][8];
    float s[2] = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunks) {
        ld8<XF32>(x, (size_t)r * D + (size_t)ch * 8, v[0][c]);
        if (two) ld8<XF32>(x, (size_t)r2 * D + (size_t)ch * 8, v[1][c]);
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunks && (k == 0 || two)) {
#pragma unroll
          for (int i = 0; i < 8; ++i) s[k] += v[k][c][i];
        }
      }
    const float mean[2] = {wave_sum(s[0]) * invD, wave_sum(s[1]) * invD};
    float ss[2] = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunks && (k == 0 || two)) {
#pragma unroll
          for (int i = 0; i < 8; ++i) { const float d = v[k][c][i] - mean[k]; ss[k] += d * d; }
        }
      }
    const float rstd[2] = {rsqrtf(wave_sum(ss[0]) * invD + eps), rsqrtf(wave_sum(ss[1]) * invD + eps)};
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunks && (k == 0 || two)) {
          float o[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = (v[k][c][i] - mean[k]) * rstd[k] * g[c][i] + bt[c][i];
          st8<YF32>(y, (size_t)(k ? r2 : r) * D + (size_t)ch * 8, o);
        }
      }
  }
}

// dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat)) [+ dres];  dxhat = dy * gamma
// dgamma / dbeta: per-block partial sums -> part[2][gridDim.x][D]
template <int NCH, bool XF32, bool YF32>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ x, const float* __restrict__ gamma,
                                                     const void* __restrict__ dy, const void* __restrict__ dres,
                                                     void* __restrict__ dx, float* __restrict__ part,
                                                     long rows, int D, float eps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const long wid = (long)blockIdx.x * 4 + wv;
  const long nw = (long)gridDim.x * 4;
  const int nchunks = D >> 3;
  float g[NCH][8], dg[NCH][8], db[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = lane + c * 64;
#pragma unroll
    for (int i = 0; i < 8; ++i) { dg[c][i] = 0.f; db[c][i] = 0.f; g[c][i] = 0.f; }
    if (ch < nchunks) ld8<true>(gamma, (size_t)ch * 8, g[c]);
  }
  const float invD = 1.0f / (float)D;
  for (long r = wid; r < rows; r += nw) {
    float v[NCH][8], d[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunks) {
        ld8<XF32>(x, (size_t)r * D + (size_t)ch * 8, v[c]);
        ld8<YF32>(dy, (size_t)r * D + (size_t)ch * 8, d[c]);
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[c][i];
      }
    }
    const float mean = wave_sum(s) * invD;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunks) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float t = v[c][i] - mean; ss += t * t; }
      }
    }
    const float rstd = rsqrtf(wave_sum(ss) * invD + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunks) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float xh = (v[c][i] - mean) * rstd;
          const float dyv = d[c][i];
          dg[c][i] += dyv * xh;
          db[c][i] += dyv;
          const float dxh = dyv * g[c][i];
          s1 += dxh;
          s2 += dxh * xh;
          v[c][i] = xh;      // keep xhat
          d[c][i] = dxh;     // keep dxhat
        }
      }
    }
    const float m1 = wave_sum(s1) * invD, m2 = wave_sum(s2) * invD;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunks) {
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = rstd * (d[c][i] - m1 - v[c][i] * m2);
        if (dres) {
          float rr[8];
          ld8<XF32>(dres, (size_t)r * D + (size_t)ch * 8, rr);
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] += rr[i];
        }
        st8<XF32>(dx, (size_t)r * D + (size_t)ch * 8, o);
      }
    }
  }
  // block reduction of dgamma / dbeta over the 4 waves through LDS, then one partial row per block
  float* sd = (float*)smem;   // [4][2][D]
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = lane + c * 64;
    if (ch < nchunks) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        sd[(wv * 2 + 0) * D + ch * 8 + i] = dg[c][i];
        sd[(wv * 2 + 1) * D + ch * 8 + i] = db[c][i];
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * D; i += 256) {
    const int which = i / D, col = i - which * D;
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) a += sd[(w * 2 + which) * D + col];
    part[((size_t)which * gridDim.x + blockIdx.x) * D + col] = a;
  }
}

// out[which][col] = sum_b part[which][b][col]
// 64 columns per block, 4 waves striding the partial rows (4 independent loads in flight per lane)
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int nblk, int D) {
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
  if (col < D) {
    int s = grp;
    for (; s + 4 < nblk; s += 8) {
      a0 += part[(size_t)s * D + col];
      a1 += part[(size_t)(s + 4) * D + col];
      b0 += part[((size_t)nblk + s) * D + col];
      b1 += part[((size_t)nblk + s + 4) * D + col];
    }
    for (; s < nblk; s += 4) {
      a0 += part[(size_t)s * D + col];
      b0 += part[((size_t)nblk + s) * D + col];
    }
  }
  red[0][grp][lane] = a0 + a1;
  red[1][grp][lane] = b0 + b1;
  __syncthreads();
  if (grp == 0 && col < D) {
    dgamma[col] = red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
    dbeta[col] = red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
  }
}

int ln_grid(long rows, long cap = 2048) {
  long g = (rows + 3) / 4;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

template <int NCH>
void launch_fwd(const void* x, const float* g, const float* b, void* y, long rows, int D, float eps,
                int xf32, int yf32, hipStream_t st) {
  const int grid = ln_grid(rows);
  if (xf32 && yf32) hipLaunchKernelGGL((ln_fwd_kernel<NCH, true, true>), dim3(grid), dim3(256), 0, st, x, g, b, y, rows, D, eps);
  else if (xf32) hipLaunchKernelGGL((ln_fwd_kernel<NCH, true, false>), dim3(grid), dim3(256), 0, st, x, g, b, y, rows, D, eps);
  else if (yf32) hipLaunchKernelGGL((ln_fwd_kernel<NCH, false, true>), dim3(grid), dim3(256), 0, st, x, g, b, y, rows, D, eps);
  else hipLaunchKernelGGL((ln_fwd_kernel<NCH, false, false>), dim3(grid), dim3(256), 0, st, x, g, b, y, rows, D, eps);
}
template <int NCH>
void launch_bwd(const void* x, const float* g, const void* dy, const void* dres, void* dx, float* part,
                long rows, int D, float eps, int xf32, int yf32, int grid, hipStream_t st) {
  const size_t lds = (size_t)8 * D * sizeof(float);
  if (xf32 && yf32) hipLaunchKernelGGL((ln_bwd_kernel<NCH, true, true>), dim3(grid), dim3(256), lds, st, x, g, dy, dres, dx, part, rows, D, eps);
  else if (xf32) hipLaunchKernelGGL((ln_bwd_kernel<NCH, true, false>), dim3(grid), dim3(256), lds, st, x, g, dy, dres, dx, part, rows, D, eps);
  else if (yf32) hipLaunchKernelGGL((ln_bwd_kernel<NCH, false, true>), dim3(grid), dim3(256), lds, st, x, g, dy, dres, dx, part, rows, D, eps);
  else hipLaunchKernelGGL((ln_bwd_kernel<NCH, false, false>), dim3(grid), dim3(256), lds, st, x, g, dy, dres, dx, part, rows, D, eps);
}

}  // namespace

extern "C" int clipa_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y,
                                   int64_t rows, int64_t D, float eps, int x_f32, int y_f32, void* stream) {
  if (rows <= 0) return CLIPA_OK;
  if (D % 8 != 0 || D <= 0 || D > 2048) { clipa_set_error("layernorm: D=%ld must be a multiple of 8 in (0, 2048]", (long)D); return CLIPA_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  if (D <= 512) launch_fwd<1>(x, gamma, beta, y, rows, (int)D, eps, x_f32, y_f32, st);
  else if (D <= 1024) launch_fwd<2>(x, gamma, beta, y, rows, (int)D, eps, x_f32, y_f32, st);
  else if (D <= 1536) launch_fwd<3>(x, gamma, beta, y, rows, (int)D, eps, x_f32, y_f32, st);
  else launch_fwd<4>(x, gamma, beta, y, rows, (int)D, eps, x_f32, y_f32, st);
  return clipa_check_launch("layernorm_fwd");
}

extern "C" int64_t clipa_layernorm_bwd_workspace(int64_t rows, int64_t D) {
  return (int64_t)2 * ln_grid(rows, 1024) * D * sizeof(float);
}

extern "C" int clipa_layernorm_bwd(const void* x, const float* gamma, const void* dy, const void* dres,
                                   void* dx, float* dgamma, float* dbeta, int64_t rows, int64_t D,
                                   float eps, int x_f32, int y_f32, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
  if (D % 8 != 0 || D <= 0 || D > 2048) { clipa_set_error("layernorm: D=%ld must be a multiple of 8 in (0, 2048]", (long)D); return CLIPA_ERR_ARG; }
  if (rows <= 0) return CLIPA_OK;
  if (!workspace || workspace_bytes < clipa_layernorm_bwd_workspace(rows, D)) { clipa_set_error("layernorm_bwd: workspace too small"); return CLIPA_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  const int grid = ln_grid(rows, 1024);
  float* part = (float*)workspace;
  if (D <= 512) launch_bwd<1>(x, gamma, dy, dres, dx, part, rows, (int)D, eps, x_f32, y_f32, grid, st);
  else if (D <= 1024) launch_bwd<2>(x, gamma, dy, dres, dx, part, rows, (int)D, eps, x_f32, y_f32, grid, st);
  else if (D <= 1536) launch_bwd<3>(x, gamma, dy, dres, dx, part, rows, (int)D, eps, x_f32, y_f32, grid, st);
  else launch_bwd<4>(x, gamma, dy, dres, dx, part, rows, (int)D, eps, x_f32, y_f32, grid, st);
  if (int rc = clipa_check_launch("layernorm_bwd")) return rc;
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((unsigned)((D + 63) / 64)), dim3(256), 0, st, part, dgamma, dbeta, grid, (int)D);
  return clipa_check_launch("layernorm_bwd_reduce");
}
