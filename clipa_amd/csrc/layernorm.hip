// LayerNorm forward / backward for gfx950: one 64-lane wave per row, fp32 statistics.
// Replaces F.layer_norm in clipa_torch/open_clip/transformer.py:19-34 (LayerNorm / LayerNormFp32:
// biased variance, eps inside the sqrt, affine) and its autograd.
// HBM-bound: every row is read once with 16-byte loads and kept in registers for both passes.
// Grid shapes and cache policy follow tools/probes/stream_ab.hip (profiles/r04_stream_kernel_shapes_ab_*.jsonl, rows = 806912,
// D = 1024): a ONE-SHOT grid (a block lives for 8 rows) streams at 6.1 TB/s where 2048 persistent blocks striding the rows
// reach 4.7, and non-temporal loads + stores of the row streams are worth another 4 %; the backward kernel, which has to carry
// dgamma / dbeta partials, gives every block 128 ADJACENT rows (5.4 TB/s including the partial reduction, 4.4 persistent)
// - for rows of at most 1024 elements; wider rows stay on the persistent grid (ln_bwd_chunk).
#include "common.h"
#include "quant_common.h"
#include "clipa_hip.h"

namespace {

// load / store 8 consecutive elements as float
// STREAM: the operand is a row stream that this kernel touches once (non-temporal); gamma / beta are not
template <bool F32, bool STREAM = true>
__device__ __forceinline__ void ld8(const void* base, size_t elem, float* f) {
  if (F32) {
    const f32x4* p = (const f32x4*)((const float*)base + elem);
    const f32x4 a = STREAM ? __builtin_nontemporal_load(p) : p[0];
    const f32x4 b = STREAM ? __builtin_nontemporal_load(p + 1) : p[1];
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  } else {
    const u32x4* p = (const u32x4*)((const unsigned short*)base + elem);
    unpack8(STREAM ? __builtin_nontemporal_load(p) : *p, f);
  }
}
template <bool F32>
__device__ __forceinline__ void st8(void* base, size_t elem, const float* f) {
  if (F32) {
    f32x4* p = (f32x4*)((float*)base + elem);
    __builtin_nontemporal_store(f32x4{f[0], f[1], f[2], f[3]}, p);
    __builtin_nontemporal_store(f32x4{f[4], f[5], f[6], f[7]}, p + 1);
  } else {
    __builtin_nontemporal_store(pack8(f), (u32x4*)((unsigned short*)base + elem));
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
    if (ch < nchunks) { ld8<true, false>(gamma, (size_t)ch * 8, g[c]); ld8<true, false>(beta, (size_t)ch * 8, bt[c]); }
  }
  const float invD = 1.0f / (float)D;
  // TWO rows per wave and iteration, both rows' loads in flight before the first reduction starts; the launcher sizes the
  // grid so that this loop runs ONCE (the loop remains for row counts beyond the grid limit)
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
          for (int i = 0; i < 8; ++i) { const float d = v[k][c][i] - mean[k]; ss[k] = __builtin_fmaf(d, d, ss[k]); }
        }
      }
    const float rstd[2] = {rsqrtf(__builtin_fmaf(wave_sum(ss[0]), invD, eps)), rsqrtf(__builtin_fmaf(wave_sum(ss[1]), invD, eps))};
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunks && (k == 0 || two)) {
          float o[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = __builtin_fmaf((v[k][c][i] - mean[k]) * rstd[k], g[c][i], bt[c][i]);
          st8<YF32>(y, (size_t)(k ? r2 : r) * D + (size_t)ch * 8, o);
        }
      }
  }
}

// dx = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat)) [+ dres];  dxhat = dy * gamma
// dgamma / dbeta: per-block partial sums -> part[2][gridDim.x][D]
// EMIT (round 5): additionally y = LayerNorm(x) - the forward's output, bit for bit (same statistics, same expression) - for the
// weight-gradient GEMM of the layer that followed the LayerNorm: a block that recomputes that operand in backward gets it from
// the pass that has the row and its statistics in registers anyway (one more 2 D-byte store per row) instead of from a second
// ln_fwd launch over the same rows (a 2 D-byte read and a 2 D-byte write per row).
// QFMT >= 0 (round 6, fp8 engine; bf16 rows only): dx is also the incoming gradient of the linear layer in front of this
// LayerNorm's block position, whose two fp8 products want it row-quantised with its column sums (= that layer's bias gradient):
// this pass has the finished row in registers, so it writes the e4m3 / e5m2 bytes, the row scale and per-block column-sum
// partials itself (q, dq bit for bit what clipa_quantize_rows makes of the bf16 dx; one more byte per element instead of a
// pass that reads 2 and writes 1).  part then holds [3][gridDim.x][D]: dgamma, dbeta, column sums of dx.
template <int NCH, bool XF32, bool YF32, bool EMIT = false, int QFMT = -1>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ x, const float* __restrict__ gamma,
                                                     const void* __restrict__ dy, const void* __restrict__ dres,
                                                     void* __restrict__ dx, float* __restrict__ part,
                                                     long rows, int D, float eps, int C,
                                                     const float* __restrict__ beta = nullptr, void* __restrict__ yout = nullptr,
                                                     char* __restrict__ q8 = nullptr, float* __restrict__ q8scale = nullptr,
                                                     float* __restrict__ q8norm = nullptr) {
  constexpr bool QOUT = QFMT >= 0;
  static_assert(!QOUT || (!XF32 && !YF32), "the quantised output exists for bf16 rows");
  // Every multiply-add of the statistics and of the outputs is WRITTEN as __builtin_fmaf, here and in ln_fwd_kernel (and in the
  // LayerNorm + quantise kernels of quant.hip): which products hipcc contracts under -ffp-contract=fast depends on use counts that
  // differ between instantiations (and `#pragma clang fp contract(off)` is ignored under that flag - ADVICE r5), while the EMIT,
  // plain and forward kernels must agree bit for bit on rstd, y, dx, dgamma, dbeta (a block's gradients may not depend on which
  // of its tensors were kept).  What is left to the compiler are single multiplications and additions.
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  // C > 0: block b owns the C adjacent rows [b C, (b + 1) C), its four waves take them interleaved (one 4-row window per step);
  // C == 0: the rows are strided over all waves of a persistent grid
  const long row_end = (C && rows > (long)(blockIdx.x + 1) * C) ? (long)(blockIdx.x + 1) * C : rows;
  const long row_first = C ? (long)blockIdx.x * C + wv : (long)blockIdx.x * 4 + wv;
  const long row_step = C ? 4 : (long)gridDim.x * 4;
  const int nchunks = D >> 3;
  float g[NCH][8], dg[NCH][8], db[NCH][8], cs[QOUT ? NCH : 1][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = lane + c * 64;
#pragma unroll
    for (int i = 0; i < 8; ++i) { dg[c][i] = 0.f; db[c][i] = 0.f; g[c][i] = 0.f; if (QOUT) cs[c][i] = 0.f; }
    if (ch < nchunks) ld8<true, false>(gamma, (size_t)ch * 8, g[c]);
  }
  // (EMIT: beta is re-read per row from L1 / L2 - 4 KB shared by every wave - rather than held in NCH * 8 more registers)
  const float invD = 1.0f / (float)D;
  // A/B knob (round 6, OFF): bf16 rows with the NEXT row's x / dy loads issued before this row's arithmetic and this row's
  // residual gradient requested at the top instead of behind the four reductions - what lifted the LayerNorm + quantise kernels
  // of quant.hip from 4.2 to 5.5 TB/s.  Here it costs 4-13 % (D = 1280: 1.30 -> 1.36 ms, 1024: 1.20 -> 1.25, 768: 0.44 ->
  // 0.49; profiles/r06_stream_kernels_ln_bwd_prefetch_ab.jsonl): 152 / 210 registers for two / three chunks per lane take a
  // wave per SIMD away.  Same values, same arithmetic either way.
#ifndef LN_BWD_PREFETCH
#define LN_BWD_PREFETCH 0
#endif
  constexpr bool PF = LN_BWD_PREFETCH && !XF32 && !YF32;
  u32x4 nx[PF ? NCH : 1], nd[PF ? NCH : 1], nr[PF ? NCH : 1];
  if constexpr (PF) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      nx[c] = nd[c] = u32x4{0, 0, 0, 0};
      if (ch < nchunks && row_first < row_end) {
        nx[c] = __builtin_nontemporal_load((const u32x4*)((const unsigned short*)x + (size_t)row_first * D + (size_t)ch * 8));
        nd[c] = __builtin_nontemporal_load((const u32x4*)((const unsigned short*)dy + (size_t)row_first * D + (size_t)ch * 8));
      }
    }
  }
  for (long r = row_first; r < row_end; r += row_step) {
    float v[NCH][8], d[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunks) {
        if constexpr (PF) {
          unpack8(nx[c], v[c]);
          unpack8(nd[c], d[c]);
          if (dres) nr[c] = __builtin_nontemporal_load((const u32x4*)((const unsigned short*)dres + (size_t)r * D + (size_t)ch * 8));
          if (r + row_step < row_end) {
            nx[c] = __builtin_nontemporal_load((const u32x4*)((const unsigned short*)x + (size_t)(r + row_step) * D + (size_t)ch * 8));
            nd[c] = __builtin_nontemporal_load((const u32x4*)((const unsigned short*)dy + (size_t)(r + row_step) * D + (size_t)ch * 8));
          }
        } else {
          ld8<XF32>(x, (size_t)r * D + (size_t)ch * 8, v[c]);
          ld8<YF32>(dy, (size_t)r * D + (size_t)ch * 8, d[c]);
        }
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
        for (int i = 0; i < 8; ++i) { const float t = v[c][i] - mean; ss = __builtin_fmaf(t, t, ss); }   // (ln_fwd_kernel's expression: same rstd)
      }
    }
    const float rstd = rsqrtf(__builtin_fmaf(wave_sum(ss), invD, eps));
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunks) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float xh = (v[c][i] - mean) * rstd;
          const float dyv = d[c][i];
          dg[c][i] = __builtin_fmaf(dyv, xh, dg[c][i]);
          db[c][i] += dyv;
          const float dxh = dyv * g[c][i];
          s1 += dxh;
          s2 = __builtin_fmaf(dxh, xh, s2);
          v[c][i] = xh;      // keep xhat
          d[c][i] = dxh;     // keep dxhat
        }
      }
    }
    const float m1 = wave_sum(s1) * invD, m2 = wave_sum(s2) * invD;
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunks) {
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = rstd * __builtin_fmaf(-v[c][i], m2, d[c][i] - m1);
        if (dres) {
          float rr[8];
          if constexpr (PF) unpack8(nr[c], rr);
          else ld8<XF32>(dres, (size_t)r * D + (size_t)ch * 8, rr);
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] += rr[i];
        }
        if constexpr (QOUT) {      // the stored bf16 values are what gets quantised: d (dxhat, dead from here) keeps them for the second sweep
          const u32x4 pk = pack8(o);
          __builtin_nontemporal_store(pk, (u32x4*)((unsigned short*)dx + (size_t)r * D + (size_t)ch * 8));
          unpack8(pk, d[c]);
#pragma unroll
          for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(d[c][i]));
        } else {
          st8<XF32>(dx, (size_t)r * D + (size_t)ch * 8, o);
        }
        if constexpr (EMIT) {      // v holds xhat = (x - mean) * rstd: y = xhat * gamma + beta, ln_fwd_kernel's expression (tested)
          float bt[8];
          ld8<true, false>(beta, (size_t)ch * 8, bt);
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = __builtin_fmaf(v[c][i], g[c][i], bt[i]);
          st8<YF32>(yout, (size_t)r * D + (size_t)ch * 8, o);
        }
      }
    }
    if constexpr (QOUT) {
      float qs, qd, ssq = 0.f;
      row_scales<QFMT>(wave_max(amax), qs, qd);
      if (lane == 0) q8scale[r] = qd;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunks) {
          __builtin_nontemporal_store(cvt8<QFMT>(d[c], qs), (u32x2*)(q8 + (size_t)r * D + (size_t)ch * 8));
#pragma unroll
          for (int i = 0; i < 8; ++i) { cs[c][i] += d[c][i]; ssq = __builtin_fmaf(d[c][i], d[c][i], ssq); }
        }
      }
      if (q8norm) {
        ssq = wave_sum(ssq);
        if (lane == 0) q8norm[r] = sqrtf(ssq);
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
  if constexpr (QOUT) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunks) {
#pragma unroll
        for (int i = 0; i < 8; ++i) sd[wv * D + ch * 8 + i] = cs[c][i];
      }
    }
    __syncthreads();
    for (int col = threadIdx.x; col < D; col += 256)
      part[((size_t)2 * gridDim.x + blockIdx.x) * D + col] = (sd[col] + sd[D + col]) + (sd[2 * D + col] + sd[3 * D + col]);
  }
}

// out_a[y][col] = sum over the partial rows b of slice y of part[a][b][col] for the NA arrays of a launch (dgamma, dbeta and -
// the quantising backward - the column sums of dx): 64 columns per block, 4 waves striding the slice (independent loads in
// flight per lane).  Many partial rows are reduced in two passes (gridDim.y slices -> [NA][slices][D], then one slice over
// those) so that the first pass fills the chip.  Fixed order: bit-reproducible.
template <int NA>
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ part, float* __restrict__ out0,
                                                            float* __restrict__ out1, float* __restrict__ out2, int nblk, int D, int per) {
  __shared__ float red[NA][4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const int lo = blockIdx.y * per, hi = lo + per < nblk ? lo + per : nblk;
  float a0[NA], a1[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) { a0[a] = 0.f; a1[a] = 0.f; }
  if (col < D) {
    int s = lo + grp;
    for (; s + 4 < hi; s += 8) {
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        a0[a] += part[((size_t)a * nblk + s) * D + col];
        a1[a] += part[((size_t)a * nblk + s + 4) * D + col];
      }
    }
    for (; s < hi; s += 4) {
#pragma unroll
      for (int a = 0; a < NA; ++a) a0[a] += part[((size_t)a * nblk + s) * D + col];
    }
  }
#pragma unroll
  for (int a = 0; a < NA; ++a) red[a][grp][lane] = a0[a] + a1[a];
  __syncthreads();
  if (grp == 0 && col < D) {
    float* outs[3] = {out0, out1, out2};
#pragma unroll
    for (int a = 0; a < NA; ++a)
      outs[a][(size_t)blockIdx.y * D + col] = red[a][0][lane] + red[a][1][lane] + red[a][2][lane] + red[a][3][lane];
  }
}

// forward: one block per 8 rows (4 waves x 2 rows), i.e. the row loop of ln_fwd_kernel runs once
int ln_fwd_grid(long rows) {
  long g = (rows + 7) / 8;
  if (g > 0x7fffffffL) g = 0x7fffffffL;
  if (g < 1) g = 1;
  return (int)g;
}
// backward: rows per block - 128 where that still leaves >= 2048 blocks, fewer (a multiple of 4) for short inputs.
// Rows wider than 1024 (three or four 16-byte chunks per lane, ViT-H/14's 1280) keep the persistent 1024-block grid of rounds
// 1-3 (chunk 0): at 526 336 x 1280 the 128-row blocks measured 3.5 % slower (profiles/r04_stream_kernels_old_vs_new_lib.jsonl).
#ifndef LN_BWD_PERSISTENT_BLOCKS
#define LN_BWD_PERSISTENT_BLOCKS 1024         // A/B knob (tools/stream8_bench.py --lib): blocks of the persistent grid (rows wider than 1024)
#endif
constexpr int LN_BWD_SLICES = 16;
#ifndef LN_BWD_CHUNK_MAXD
#define LN_BWD_CHUNK_MAXD 1024                // A/B knob: widest row that takes the adjacent-rows grid
#endif
int ln_bwd_chunk(long rows, long D) {
  if (D > LN_BWD_CHUNK_MAXD) return 0;
  long c = (rows / 2048 + 3) / 4 * 4;
  return (int)(c < 4 ? 4 : (c > 128 ? 128 : c));
}
long ln_bwd_grid(long rows, long D) {
  const long c = ln_bwd_chunk(rows, D);
  if (c) return (rows + c - 1) / c;
  const long g = (rows + 3) / 4;
  return g > LN_BWD_PERSISTENT_BLOCKS ? LN_BWD_PERSISTENT_BLOCKS : g;
}

template <int NCH>
void launch_fwd(const void* x, const float* g, const float* b, void* y, long rows, int D, float eps,
                int xf32, int yf32, hipStream_t st) {
  const int grid = ln_fwd_grid(rows);
  if (xf32 && yf32) hipLaunchKernelGGL((ln_fwd_kernel<NCH, true, true>), dim3(grid), dim3(256), 0, st, x, g, b, y, rows, D, eps);
  else if (xf32) hipLaunchKernelGGL((ln_fwd_kernel<NCH, true, false>), dim3(grid), dim3(256), 0, st, x, g, b, y, rows, D, eps);
  else if (yf32) hipLaunchKernelGGL((ln_fwd_kernel<NCH, false, true>), dim3(grid), dim3(256), 0, st, x, g, b, y, rows, D, eps);
  else hipLaunchKernelGGL((ln_fwd_kernel<NCH, false, false>), dim3(grid), dim3(256), 0, st, x, g, b, y, rows, D, eps);
}
template <int NCH>
void launch_bwd(const void* x, const float* g, const void* dy, const void* dres, void* dx, float* part,
                long rows, int D, float eps, int xf32, int yf32, int grid, int C, hipStream_t st,
                const float* beta = nullptr, void* yout = nullptr, int qfmt = -1, char* q8 = nullptr, float* q8scale = nullptr,
                float* q8norm = nullptr) {
  const size_t lds = (size_t)8 * D * sizeof(float);
  if (qfmt >= 0) {     // bf16 rows (checked by the caller)
#define LN_BWD_Q(E, F) hipLaunchKernelGGL((ln_bwd_kernel<NCH, false, false, E, F>), dim3(grid), dim3(256), lds, st, x, g, dy, dres, dx, part, rows, D, eps, C, beta, yout, q8, q8scale, q8norm)
    if (yout && qfmt == 0) LN_BWD_Q(true, 0);
    else if (yout) LN_BWD_Q(true, 1);
    else if (qfmt == 0) LN_BWD_Q(false, 0);
    else LN_BWD_Q(false, 1);
#undef LN_BWD_Q
    return;
  }
  if (yout) {     // the emitting form: the engine's bf16 token matrices and the f32 rows of the heads
    if (xf32 && yf32) hipLaunchKernelGGL((ln_bwd_kernel<NCH, true, true, true>), dim3(grid), dim3(256), lds, st, x, g, dy, dres, dx, part, rows, D, eps, C, beta, yout);
    else if (xf32) hipLaunchKernelGGL((ln_bwd_kernel<NCH, true, false, true>), dim3(grid), dim3(256), lds, st, x, g, dy, dres, dx, part, rows, D, eps, C, beta, yout);
    else if (yf32) hipLaunchKernelGGL((ln_bwd_kernel<NCH, false, true, true>), dim3(grid), dim3(256), lds, st, x, g, dy, dres, dx, part, rows, D, eps, C, beta, yout);
    else hipLaunchKernelGGL((ln_bwd_kernel<NCH, false, false, true>), dim3(grid), dim3(256), lds, st, x, g, dy, dres, dx, part, rows, D, eps, C, beta, yout);
    return;
  }
  if (xf32 && yf32) hipLaunchKernelGGL((ln_bwd_kernel<NCH, true, true>), dim3(grid), dim3(256), lds, st, x, g, dy, dres, dx, part, rows, D, eps, C);
  else if (xf32) hipLaunchKernelGGL((ln_bwd_kernel<NCH, true, false>), dim3(grid), dim3(256), lds, st, x, g, dy, dres, dx, part, rows, D, eps, C);
  else if (yf32) hipLaunchKernelGGL((ln_bwd_kernel<NCH, false, true>), dim3(grid), dim3(256), lds, st, x, g, dy, dres, dx, part, rows, D, eps, C);
  else hipLaunchKernelGGL((ln_bwd_kernel<NCH, false, false>), dim3(grid), dim3(256), lds, st, x, g, dy, dres, dx, part, rows, D, eps, C);
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
  return (int64_t)2 * (ln_bwd_grid(rows, D) + LN_BWD_SLICES) * D * sizeof(float);   // [2][blocks][D] + [2][slices][D]
}
extern "C" int64_t clipa_layernorm_bwd_q8_workspace(int64_t rows, int64_t D) {
  return (int64_t)3 * (ln_bwd_grid(rows, D) + LN_BWD_SLICES) * D * sizeof(float);   // [3][blocks][D] + [3][slices][D]
}

namespace {
int ln_bwd_impl(const void* x, const float* gamma, const float* beta, const void* dy, const void* dres, void* dx, void* y,
                float* dgamma, float* dbeta, int64_t rows, int64_t D, float eps, int x_f32, int y_f32, void* workspace,
                int64_t workspace_bytes, void* stream, int qfmt = -1, void* q8 = nullptr, float* q8scale = nullptr,
                float* colsum = nullptr, float* q8norm = nullptr);
}

extern "C" int clipa_layernorm_bwd(const void* x, const float* gamma, const void* dy, const void* dres,
                                   void* dx, float* dgamma, float* dbeta, int64_t rows, int64_t D,
                                   float eps, int x_f32, int y_f32, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
  return ln_bwd_impl(x, gamma, nullptr, dy, dres, dx, nullptr, dgamma, dbeta, rows, D, eps, x_f32, y_f32, workspace, workspace_bytes, stream);
}

extern "C" int clipa_layernorm_bwd_y(const void* x, const float* gamma, const float* beta, const void* dy, const void* dres,
                                     void* dx, void* y, float* dgamma, float* dbeta, int64_t rows, int64_t D,
                                     float eps, int x_f32, int y_f32, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
  if (!beta || !y) { clipa_set_error("layernorm_bwd_y: beta and y are required"); return CLIPA_ERR_ARG; }
  return ln_bwd_impl(x, gamma, beta, dy, dres, dx, y, dgamma, dbeta, rows, D, eps, x_f32, y_f32, workspace, workspace_bytes, stream);
}

namespace {
int ln_bwd_impl(const void* x, const float* gamma, const float* beta, const void* dy, const void* dres, void* dx, void* y,
                float* dgamma, float* dbeta, int64_t rows, int64_t D, float eps, int x_f32, int y_f32, void* workspace,
                int64_t workspace_bytes, void* stream, int qfmt, void* q8, float* q8scale, float* colsum, float* q8norm) {
  if (D % 8 != 0 || D <= 0 || D > 2048) { clipa_set_error("layernorm: D=%ld must be a multiple of 8 in (0, 2048]", (long)D); return CLIPA_ERR_ARG; }
  if (rows <= 0) return CLIPA_OK;
  const bool Q = qfmt >= 0;
  if (!workspace || workspace_bytes < (Q ? clipa_layernorm_bwd_q8_workspace(rows, D) : clipa_layernorm_bwd_workspace(rows, D))) { clipa_set_error("layernorm_bwd: workspace too small"); return CLIPA_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  const int grid = (int)ln_bwd_grid(rows, D), C = ln_bwd_chunk(rows, D);
  float* part = (float*)workspace;
  if (D <= 512) launch_bwd<1>(x, gamma, dy, dres, dx, part, rows, (int)D, eps, x_f32, y_f32, grid, C, st, beta, y, qfmt, (char*)q8, q8scale, q8norm);
  else if (D <= 1024) launch_bwd<2>(x, gamma, dy, dres, dx, part, rows, (int)D, eps, x_f32, y_f32, grid, C, st, beta, y, qfmt, (char*)q8, q8scale, q8norm);
  else if (D <= 1536) launch_bwd<3>(x, gamma, dy, dres, dx, part, rows, (int)D, eps, x_f32, y_f32, grid, C, st, beta, y, qfmt, (char*)q8, q8scale, q8norm);
  else launch_bwd<4>(x, gamma, dy, dres, dx, part, rows, (int)D, eps, x_f32, y_f32, grid, C, st, beta, y, qfmt, (char*)q8, q8scale, q8norm);
  if (int rc = clipa_check_launch("layernorm_bwd")) return rc;
  const unsigned cols = (unsigned)((D + 63) / 64);
  const int NA = Q ? 3 : 2;
#define LN_REDUCE(GRID, ...) do { if (Q) hipLaunchKernelGGL(ln_bwd_reduce_kernel<3>, GRID, dim3(256), 0, st, __VA_ARGS__); \
                                  else hipLaunchKernelGGL(ln_bwd_reduce_kernel<2>, GRID, dim3(256), 0, st, __VA_ARGS__); } while (0)
  if (grid <= 8 * LN_BWD_SLICES) {
    LN_REDUCE(dim3(cols), part, dgamma, dbeta, colsum, grid, (int)D, grid);
    return clipa_check_launch("layernorm_bwd_reduce");
  }
  float* part2 = part + (size_t)NA * grid * D;
  const int per = (grid + LN_BWD_SLICES - 1) / LN_BWD_SLICES;
  LN_REDUCE(dim3(cols, LN_BWD_SLICES), part, part2, part2 + (size_t)LN_BWD_SLICES * D, part2 + (size_t)2 * LN_BWD_SLICES * D, grid, (int)D, per);
  if (int rc = clipa_check_launch("layernorm_bwd_reduce")) return rc;
  LN_REDUCE(dim3(cols), part2, dgamma, dbeta, colsum, LN_BWD_SLICES, (int)D, LN_BWD_SLICES);
#undef LN_REDUCE
  return clipa_check_launch("layernorm_bwd_reduce2");
}
}  // namespace

extern "C" int clipa_layernorm_bwd_q8(const void* x, const float* gamma, const float* beta, const void* dy, const void* dres,
                                      void* dx, void* y, void* q, float* dq, float* colsum, float* rownorm, float* dgamma,
                                      float* dbeta, int64_t rows, int64_t D, float eps, int fmt, void* workspace,
                                      int64_t workspace_bytes, void* stream) {
  if (fmt != 0 && fmt != 1) { clipa_set_error("layernorm_bwd_q8: fmt must be 0 (e4m3) or 1 (e5m2)"); return CLIPA_ERR_ARG; }
  if (!q || !dq || !colsum) { clipa_set_error("layernorm_bwd_q8: q, dq and colsum are required"); return CLIPA_ERR_ARG; }
  if ((beta == nullptr) != (y == nullptr)) { clipa_set_error("layernorm_bwd_q8: beta and y go together"); return CLIPA_ERR_ARG; }
  return ln_bwd_impl(x, gamma, beta, dy, dres, dx, y, dgamma, dbeta, rows, D, eps, 0, 0, workspace, workspace_bytes, stream, fmt, q, dq, colsum, rownorm);
}
