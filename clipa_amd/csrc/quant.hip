// Row-scaled fp8 quantisation for the fp8 GEMM path (gemm_f8.hip) on gfx950: OCP e4m3 (max 448) / e5m2 (max 57344).
//
// No reference counterpart (clipa_torch has no fp8 mode, training/params.py:195-200); BASELINE.json configs[3] asks for
// "fp8 MFMA weights/activations".  Recipe: every ROW of a GEMM operand (a token of an activation / gradient matrix, an
// output channel of a weight) gets its own scale s = FMAX / max|row|, q = rne_fp8(x * s), and the GEMM epilogue
// multiplies by the de-quantisation factors 1/s of its two rows.  Scales come from the data itself (no amax history,
// no atomics), so a block's backward-time recompute reproduces its forward bit for bit.
// HBM-bound: one wave per row, the row stays in registers between the max and the convert (2 B read + 1 B written
// per element); the LayerNorm variant emits the fp8 operand of the following GEMM straight from the normalised row.
#include "common.h"
#include "stream.h"
#include "quant_common.h"
#include "clipa_hip.h"

#ifndef LN_Q8_PREFETCH
#define LN_Q8_PREFETCH 1                          // A/B knob (tools/stream8_bench.py --lib)
#endif
#ifndef LN_Q8_BLOCK_CAP
#define LN_Q8_BLOCK_CAP 8192                      // A/B knob (tools/stream_lib_ab.py)
#endif

namespace {

// NCH = 16-byte chunks (8 bf16) per lane: K <= NCH * 512
template <int NCH, int FMT>
__global__ __launch_bounds__(256) void quantize_rows_kernel(const char* __restrict__ x, long ldx, char* __restrict__ q, long ldq,
                                                            float* __restrict__ dq, long rows, int K) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunks = K >> 3;
  u32x4 v[NCH];
  float amax = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = lane + c * 64;
    v[c] = u32x4{0, 0, 0, 0};
    if (ch < nchunks) {
      v[c] = ld_stream<u32x4>(x + ((size_t)row * ldx + (size_t)ch * 8) * 2);
      float f[8];
      unpack8(v[c], f);
#pragma unroll
      for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(f[i]));
    }
  }
  amax = wave_max(amax);
  float s, d;
  row_scales<FMT>(amax, s, d);
  if (lane == 0) dq[row] = d;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = lane + c * 64;
    if (ch < nchunks) {
      float f[8];
      unpack8(v[c], f);
      st_stream<u32x2>(q + (size_t)row * ldq + (size_t)ch * 8, cvt8<FMT>(f, s));
    }
  }
}

// quantize_rows_kernel that also sums the COLUMNS of x (the bias gradient of the layer whose input gradient consumes q: the
// weight-gradient GEMM of the fp8 engine has no bf16 operand to take it from).  Persistent: a wave strides the rows and keeps
// its column sums in registers; a block adds its four waves through LDS and writes one partial row; colsum_reduce_kernel adds the
// partial rows in a fixed order (bit-reproducible, no atomics).
template <int NCH, int FMT>
__global__ __launch_bounds__(256) void quantize_rows_colsum_kernel(const char* __restrict__ x, long ldx, char* __restrict__ q, long ldq,
                                                                   float* __restrict__ dq, float* __restrict__ partial, float* __restrict__ rnorm,
                                                                   long rows, int K) {
  __shared__ float red[4][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long wid = (long)blockIdx.x * 4 + wave;
  const long nw = (long)gridDim.x * 4;
  const int nchunks = K >> 3;
  float cs[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) cs[c][i] = 0.f;
  // The next row's loads are issued before this row's arithmetic (two rows in flight per wave, as the LayerNorm + quantise
  // kernels) where that fits the register file: hipcc needs 206 / 254 / 290 registers for the 6 / 8 / 16-chunk instantiations
  // with the second row (two waves per SIMD: 3.5 TB/s at K = 3840 instead of 5.4), 98 and 183 for 4 and 10 chunks.
  constexpr bool PF = NCH <= 4 || NCH == 10;
  u32x4 nxt[PF ? NCH : 1];
  if constexpr (PF) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      nxt[c] = u32x4{0, 0, 0, 0};
      if (ch < nchunks && wid < rows) nxt[c] = ld_stream<u32x4>(x + ((size_t)wid * ldx + (size_t)ch * 8) * 2);
    }
  }
  for (long row = wid; row < rows; row += nw) {
    u32x4 v[NCH];
    float amax = 0.f, ssq = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      v[c] = u32x4{0, 0, 0, 0};
      if constexpr (PF) v[c] = nxt[c];
      if (ch < nchunks) {
        if constexpr (PF) { if (row + nw < rows) nxt[c] = ld_stream<u32x4>(x + ((size_t)(row + nw) * ldx + (size_t)ch * 8) * 2); }
        else v[c] = ld_stream<u32x4>(x + ((size_t)row * ldx + (size_t)ch * 8) * 2);
        float f[8];
        unpack8(v[c], f);
#pragma unroll
        for (int i = 0; i < 8; ++i) { amax = fmaxf(amax, fabsf(f[i])); cs[c][i] += f[i]; ssq = __builtin_fmaf(f[i], f[i], ssq); }
      }
    }
    amax = wave_max(amax);
    float s, d;
    row_scales<FMT>(amax, s, d);
    if (lane == 0) dq[row] = d;
    if (rnorm) {      // ||x[row,:]||_2 (engine._row_bound: the Cauchy-Schwarz bound of the input gradient this row produces)
      const float n2 = wave_sum(ssq);
      if (lane == 0) rnorm[row] = sqrtf(n2);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunks) {
        float f[8];
        unpack8(v[c], f);
        st_stream<u32x2>(q + (size_t)row * ldq + (size_t)ch * 8, cvt8<FMT>(f, s));
      }
    }
  }
  float* out = partial + (size_t)blockIdx.x * K;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) red[wave][lane * 8 + i] = cs[c][i];
    __syncthreads();
    for (int e = threadIdx.x; e < 512; e += 256) {
      const int col = c * 512 + e;                      // chunk (c * 64 + e / 8), element e % 8
      if (col < K) out[col] = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
    }
  }
}

// out[col] = sum_b partial[b][col], b ascending inside 16 interleaved groups, the groups combined in order
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out, int nb, int K) {
  __shared__ float4 acc[16][16];
  const int cg = threadIdx.x & 15, sg = threadIdx.x >> 4;      // 16 column quads x 16 row groups
  const int col = blockIdx.x * 64 + cg * 4;
  float4 a = {0.f, 0.f, 0.f, 0.f};
  if (col < K)
    for (int b = sg; b < nb; b += 16) {
      const float4 p = *(const float4*)(partial + (size_t)b * K + col);
      a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
    }
  acc[sg][cg] = a;
  __syncthreads();
  if (sg == 0 && col < K) {
    for (int g = 1; g < 16; ++g) { const float4 p = acc[g][cg]; a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w; }
    *(float4*)(out + col) = a;
  }
}

// LayerNorm (transformer.py:19-34) whose output feeds an fp8 GEMM: y = bf16(LN(x)) (optional), q = e4m3(y * s), dq = 1/s
template <int NCH>
__global__ __launch_bounds__(256) void ln_fwd_q8_kernel(const char* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, char* __restrict__ y,
                                                        char* __restrict__ q, float* __restrict__ dq, float* __restrict__ rnorm,
                                                        long rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nw = (long)gridDim.x * 4;
  const int nchunks = D >> 3;
  float g[NCH][8], bt[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = lane + c * 64;
#pragma unroll
    for (int i = 0; i < 8; ++i) { g[c][i] = 0.f; bt[c][i] = 0.f; }
    if (ch < nchunks) {
      const float4 a = *(const float4*)(gamma + ch * 8), b = *(const float4*)(gamma + ch * 8 + 4);
      const float4 e = *(const float4*)(beta + ch * 8), h = *(const float4*)(beta + ch * 8 + 4);
      g[c][0] = a.x; g[c][1] = a.y; g[c][2] = a.z; g[c][3] = a.w; g[c][4] = b.x; g[c][5] = b.y; g[c][6] = b.z; g[c][7] = b.w;
      bt[c][0] = e.x; bt[c][1] = e.y; bt[c][2] = e.z; bt[c][3] = e.w; bt[c][4] = h.x; bt[c][5] = h.y; bt[c][6] = h.z; bt[c][7] = h.w;
    }
  }
  const float invD = 1.0f / (float)D;
  // the next row's loads are issued before this row's arithmetic (LN_Q8_PREFETCH: a wave keeps two rows in flight - these
  // kernels are latency-bound at one row per wave and ~100 registers, tools/stream8_bench.py)
  u32x4 nxt[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = lane + c * 64;
    nxt[c] = u32x4{0, 0, 0, 0};
    if (ch < nchunks && wid < rows) nxt[c] = ld_stream<u32x4>(x + ((size_t)wid * D + (size_t)ch * 8) * 2);
  }
  for (long r = wid; r < rows; r += nw) {
    float v[NCH][8];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (!LN_Q8_PREFETCH && ch < nchunks && r != wid) nxt[c] = ld_stream<u32x4>(x + ((size_t)r * D + (size_t)ch * 8) * 2);
      unpack8(nxt[c], v[c]);
      if (ch < nchunks) {
#pragma unroll
        for (int i = 0; i < 8; ++i) sum += v[c][i];
        if (LN_Q8_PREFETCH && r + nw < rows) nxt[c] = ld_stream<u32x4>(x + ((size_t)(r + nw) * D + (size_t)ch * 8) * 2);
      }
    }
    const float mean = wave_sum(sum) * invD;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunks) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = v[c][i] - mean; ss = __builtin_fmaf(d, d, ss); }
      }
    }
    const float rstd = rsqrtf(__builtin_fmaf(wave_sum(ss), invD, eps));      // (layernorm.hip: the expressions of ln_fwd_kernel, fused multiply-adds written out)
    float amax = 0.f, ssq = 0.f;
    u32x4 yb[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      yb[c] = u32x4{0, 0, 0, 0};
      if (ch < nchunks) {
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = __builtin_fmaf((v[c][i] - mean) * rstd, g[c][i], bt[c][i]);
        yb[c] = pack8(o);                       // the bf16 rounding of the plain LayerNorm kernel: q is derived from it
        if (y) st_stream<u32x4>(y + ((size_t)r * D + (size_t)ch * 8) * 2, yb[c]);
        unpack8(yb[c], o);
#pragma unroll
        for (int i = 0; i < 8; ++i) { amax = fmaxf(amax, fabsf(o[i])); ssq = __builtin_fmaf(o[i], o[i], ssq); }
      }
    }
    amax = wave_max(amax);
    float s, d;
    row_scales<0>(amax, s, d);
    if (lane == 0) dq[r] = d;
    if (rnorm) {      // ||y[r,:]||_2: with the next layer's largest weight-row norm a Cauchy-Schwarz bound of that layer's outputs (engine._row_bound)
      const float n2 = wave_sum(ssq);
      if (lane == 0) rnorm[r] = sqrtf(n2);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunks) {
        float o[8];
        unpack8(yb[c], o);
        st_stream<u32x2>(q + (size_t)r * D + (size_t)ch * 8, cvt8<0>(o, s));
      }
    }
  }
}

template <int FMT>
int launch_quant(const void* x, void* q, float* dq, int64_t rows, int64_t K, int64_t ldx, int64_t ldq, hipStream_t st) {
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  const char* xp = (const char*)x;
  char* qp = (char*)q;
  if (K <= 512) hipLaunchKernelGGL((quantize_rows_kernel<1, FMT>), grid, block, 0, st, xp, (long)ldx, qp, (long)ldq, dq, (long)rows, (int)K);
  else if (K <= 1024) hipLaunchKernelGGL((quantize_rows_kernel<2, FMT>), grid, block, 0, st, xp, (long)ldx, qp, (long)ldq, dq, (long)rows, (int)K);
  else if (K <= 2048) hipLaunchKernelGGL((quantize_rows_kernel<4, FMT>), grid, block, 0, st, xp, (long)ldx, qp, (long)ldq, dq, (long)rows, (int)K);
  else if (K <= 4096) hipLaunchKernelGGL((quantize_rows_kernel<8, FMT>), grid, block, 0, st, xp, (long)ldx, qp, (long)ldq, dq, (long)rows, (int)K);
  else hipLaunchKernelGGL((quantize_rows_kernel<16, FMT>), grid, block, 0, st, xp, (long)ldx, qp, (long)ldq, dq, (long)rows, (int)K);
  return clipa_check_launch("quantize_rows");
}

}  // namespace

extern "C" int clipa_quantize_rows(const void* x, void* q, float* dq, int64_t rows, int64_t K, int64_t ldx, int64_t ldq,
                                   int fmt, void* stream) {
  if (rows <= 0) return CLIPA_OK;
  if (K <= 0 || K % 8 != 0 || K > 8192) { clipa_set_error("quantize_rows: K=%ld must be a multiple of 8 in (0, 8192]", (long)K); return CLIPA_ERR_ARG; }
  if (ldx % 8 != 0 || ldq % 8 != 0 || ldx < K || ldq < K) { clipa_set_error("quantize_rows: ldx, ldq must be multiples of 8 and >= K"); return CLIPA_ERR_ARG; }
  if (fmt != 0 && fmt != 1) { clipa_set_error("quantize_rows: fmt is 0 (e4m3) or 1 (e5m2)"); return CLIPA_ERR_ARG; }
  if ((rows + 3) / 4 > 0x7fffffffL) { clipa_set_error("quantize_rows: too many rows"); return CLIPA_ERR_ARG; }
  return fmt == 0 ? launch_quant<0>(x, q, dq, rows, K, ldx, ldq, (hipStream_t)stream)
                  : launch_quant<1>(x, q, dq, rows, K, ldx, ldq, (hipStream_t)stream);
}

#ifndef QUANT_COLSUM_BLOCKS
#define QUANT_COLSUM_BLOCKS 2048                  // persistent blocks of quantize_rows_colsum (A/B: profiles/r06_stream_kernels_fp8_step_h14.jsonl)
#endif
namespace {
long quant_colsum_blocks(long rows) {
  long b = (rows + 3) / 4;
  if (b > QUANT_COLSUM_BLOCKS) b = QUANT_COLSUM_BLOCKS;
  return b < 1 ? 1 : b;
}
template <int FMT>
int launch_quant_colsum(const void* x, void* q, float* dq, float* colsum, float* rownorm, float* partial, int64_t rows, int64_t K, int64_t ldx,
                        int64_t ldq, hipStream_t st) {
  const long nb = quant_colsum_blocks(rows);
  const dim3 grid((unsigned)nb), block(256);
  const char* xp = (const char*)x;
  char* qp = (char*)q;
#define QC_LAUNCH(N) hipLaunchKernelGGL((quantize_rows_colsum_kernel<N, FMT>), grid, block, 0, st, xp, (long)ldx, qp, (long)ldq, dq, partial, rownorm, (long)rows, (int)K)
  if (K <= 512) QC_LAUNCH(1);
  else if (K <= 1024) QC_LAUNCH(2);
  else if (K <= 1536) QC_LAUNCH(3);
  else if (K <= 2048) QC_LAUNCH(4);
  else if (K <= 3072) QC_LAUNCH(6);
  else if (K <= 4096) QC_LAUNCH(8);
  else if (K <= 5120) QC_LAUNCH(10);
  else QC_LAUNCH(16);
#undef QC_LAUNCH
  if (int rc = clipa_check_launch("quantize_rows_colsum")) return rc;
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)((K + 63) / 64)), dim3(256), 0, st, (const float*)partial, colsum, (int)nb, (int)K);
  return clipa_check_launch("quantize_rows_colsum_reduce");
}
}  // namespace

extern "C" int clipa_reduce_partial_rows(const float* partial, float* out, int64_t nrows, int64_t K, void* stream) {
  if (K <= 0 || K % 4 != 0) { clipa_set_error("reduce_partial_rows: K must be a positive multiple of 4"); return CLIPA_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  if (nrows <= 0) { (void)hipMemsetAsync(out, 0, K * sizeof(float), st); return CLIPA_OK; }
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)((K + 63) / 64)), dim3(256), 0, st, partial, out, (int)nrows, (int)K);
  return clipa_check_launch("reduce_partial_rows");
}

extern "C" int64_t clipa_quantize_rows_colsum_workspace(int64_t rows, int64_t K) {
  return quant_colsum_blocks(rows) * K * (int64_t)sizeof(float);
}

extern "C" int clipa_quantize_rows_colsum(const void* x, void* q, float* dq, float* colsum, float* rownorm, int64_t rows, int64_t K,
                                          int64_t ldx, int64_t ldq, int fmt, void* workspace, int64_t workspace_bytes, void* stream) {
  if (K <= 0 || K % 8 != 0 || K > 8192) { clipa_set_error("quantize_rows_colsum: K=%ld must be a multiple of 8 in (0, 8192]", (long)K); return CLIPA_ERR_ARG; }
  if (ldx % 8 != 0 || ldq % 8 != 0 || ldx < K || ldq < K) { clipa_set_error("quantize_rows_colsum: ldx, ldq must be multiples of 8 and >= K"); return CLIPA_ERR_ARG; }
  if (fmt != 0 && fmt != 1) { clipa_set_error("quantize_rows_colsum: fmt is 0 (e4m3) or 1 (e5m2)"); return CLIPA_ERR_ARG; }
  if (K % 4 != 0 || !colsum) { clipa_set_error("quantize_rows_colsum: colsum is required"); return CLIPA_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  if (rows <= 0) { (void)hipMemsetAsync(colsum, 0, K * sizeof(float), st); return CLIPA_OK; }
  if (!workspace || workspace_bytes < clipa_quantize_rows_colsum_workspace(rows, K)) { clipa_set_error("quantize_rows_colsum: workspace too small"); return CLIPA_ERR_ARG; }
  return fmt == 0 ? launch_quant_colsum<0>(x, q, dq, colsum, rownorm, (float*)workspace, rows, K, ldx, ldq, st)
                  : launch_quant_colsum<1>(x, q, dq, colsum, rownorm, (float*)workspace, rows, K, ldx, ldq, st);
}

extern "C" int clipa_layernorm_fwd_q8n(const void* x, const float* gamma, const float* beta, void* y, void* q, float* dq,
                                       float* rownorm, int64_t rows, int64_t D, float eps, void* stream);
extern "C" int clipa_layernorm_fwd_q8(const void* x, const float* gamma, const float* beta, void* y, void* q, float* dq,
                                      int64_t rows, int64_t D, float eps, void* stream) {
  return clipa_layernorm_fwd_q8n(x, gamma, beta, y, q, dq, nullptr, rows, D, eps, stream);
}
extern "C" int clipa_layernorm_fwd_q8n(const void* x, const float* gamma, const float* beta, void* y, void* q, float* dq,
                                       float* rownorm, int64_t rows, int64_t D, float eps, void* stream) {
  if (rows <= 0) return CLIPA_OK;
  if (D <= 0 || D % 8 != 0 || D > 2048) { clipa_set_error("layernorm_fwd_q8: D=%ld must be a multiple of 8 in (0, 2048]", (long)D); return CLIPA_ERR_ARG; }
  // 8192 blocks striding the rows: at 806 912 x 1024 / 526 336 x 1280 / 157 696 x 1280 the persistent 2048 blocks of rounds
  // 2-3 take 0.827 / 0.733 / 0.221 ms, 8192 blocks 0.679 / 0.706 / 0.218, a one-shot grid 0.722 / 0.862 / 0.256 (gamma and
  // beta are re-read by every wave: 4x the row bytes) - profiles/r04_stream_kernels_old_vs_new_lib.jsonl
  long blocks = (rows + 3) / 4;
  if (blocks > LN_Q8_BLOCK_CAP) blocks = LN_Q8_BLOCK_CAP;
  const dim3 grid((unsigned)blocks), block(256);
  hipStream_t st = (hipStream_t)stream;
  const char* xp = (const char*)x;
  if (D <= 512) hipLaunchKernelGGL((ln_fwd_q8_kernel<1>), grid, block, 0, st, xp, gamma, beta, (char*)y, (char*)q, dq, rownorm, (long)rows, (int)D, eps);
  else if (D <= 1024) hipLaunchKernelGGL((ln_fwd_q8_kernel<2>), grid, block, 0, st, xp, gamma, beta, (char*)y, (char*)q, dq, rownorm, (long)rows, (int)D, eps);
  else if (D <= 1536) hipLaunchKernelGGL((ln_fwd_q8_kernel<3>), grid, block, 0, st, xp, gamma, beta, (char*)y, (char*)q, dq, rownorm, (long)rows, (int)D, eps);
  else hipLaunchKernelGGL((ln_fwd_q8_kernel<4>), grid, block, 0, st, xp, gamma, beta, (char*)y, (char*)q, dq, rownorm, (long)rows, (int)D, eps);
  return clipa_check_launch("layernorm_fwd_q8");
}

// ---- operands of the fp8 weight-gradient GEMM (gemm_tn8.hip) --------------------------------------------------------------
// dW = sum_m dY[m,:]^T X[m,:] reduces over tokens, so a per-token scale cannot leave the product: the activation operand absorbs
// the gradient's row scale ds[m] before it is quantised, Q8[m,:] = e4m3(ds[m] * X[m,:] / t), with one scalar
// t = max_m ds[m] * sx[m] (sx = the activation's own row scale, max|X[m,:]| / 448: |Q8| <= 448 without saturation) that stays on
// the device.  No amax history: recompute reproduces the same bytes.
namespace {

// out[0] = max_m a[m] * b[m] (b may be NULL = 1); a, b >= 0, so the float order is the order of the bit patterns
__global__ void rowscale_max_kernel(const float* __restrict__ a, const float* __restrict__ b, long n, unsigned* __restrict__ out) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) m = fmaxf(m, b ? a[i] * b[i] : a[i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

__device__ __forceinline__ float inv_or_zero(float t) { return t > 0.f ? 1.0f / t : 0.f; }

// ACT: -1 = none, else the MLP activation of common.h applied to x first.  IN8: x holds e4m3 bytes (the kept pre-activation
// of the "h8" tier) instead of bf16
template <int ACT, bool IN8>
__global__ void scale_quantize_rows_kernel(const char* __restrict__ x, long ldx, const float* __restrict__ rowscale,
                                           const float* __restrict__ t_dev, char* __restrict__ q, long ldq, long rows, int nch) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * nch) return;
  const long r = i / nch;
  const int ch = (int)(i - r * nch);
  const float s = rowscale[r] * inv_or_zero(t_dev[0]);
  float f[8];
  if (IN8) e4m3x8_to_f32(ld_stream<u32x2>(x + (size_t)r * ldx + (size_t)ch * 8), f);
  else unpack8(ld_stream<u32x4>(x + ((size_t)r * ldx + (size_t)ch * 8) * 2), f);
  if (ACT >= 0) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      const f32x2 v = {f[j], f[j + 1]};
      const f32x2 a = act_fwd2<ACT < 0 ? 0 : ACT>(v);
      f[j] = a.x;
      f[j + 1] = a.y;
    }
    // the forward's GEMM epilogue rounds the activation to bf16 before anything reads it: the same values here
    unpack8(pack8(f), f);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] *= s;
  st_stream<u32x2>(q + (size_t)r * ldq + (size_t)ch * 8, e4m3x8_sat(f));
}

// LayerNorm (transformer.py:19-34) emitting q = e4m3(bf16(LN(x)) * rowscale[r] / t): ln_fwd_q8_kernel with an external scale
template <int NCH>
__global__ __launch_bounds__(256) void ln_fwd_q8s_kernel(const char* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ rowscale,
                                                         const float* __restrict__ t_dev, char* __restrict__ q, long rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nw = (long)gridDim.x * 4;
  const int nchunks = D >> 3;
  float g[NCH][8], bt[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = lane + c * 64;
#pragma unroll
    for (int i = 0; i < 8; ++i) { g[c][i] = 0.f; bt[c][i] = 0.f; }
    if (ch < nchunks) {
      const float4 a = *(const float4*)(gamma + ch * 8), b = *(const float4*)(gamma + ch * 8 + 4);
      const float4 e = *(const float4*)(beta + ch * 8), h = *(const float4*)(beta + ch * 8 + 4);
      g[c][0] = a.x; g[c][1] = a.y; g[c][2] = a.z; g[c][3] = a.w; g[c][4] = b.x; g[c][5] = b.y; g[c][6] = b.z; g[c][7] = b.w;
      bt[c][0] = e.x; bt[c][1] = e.y; bt[c][2] = e.z; bt[c][3] = e.w; bt[c][4] = h.x; bt[c][5] = h.y; bt[c][6] = h.z; bt[c][7] = h.w;
    }
  }
  const float invD = 1.0f / (float)D;
  const float it = inv_or_zero(t_dev[0]);
  u32x4 nxt[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = lane + c * 64;
    nxt[c] = u32x4{0, 0, 0, 0};
    if (ch < nchunks && wid < rows) nxt[c] = ld_stream<u32x4>(x + ((size_t)wid * D + (size_t)ch * 8) * 2);
  }
  for (long r = wid; r < rows; r += nw) {
    float v[NCH][8];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (!LN_Q8_PREFETCH && ch < nchunks && r != wid) nxt[c] = ld_stream<u32x4>(x + ((size_t)r * D + (size_t)ch * 8) * 2);
      unpack8(nxt[c], v[c]);
      if (ch < nchunks) {
#pragma unroll
        for (int i = 0; i < 8; ++i) sum += v[c][i];
        if (LN_Q8_PREFETCH && r + nw < rows) nxt[c] = ld_stream<u32x4>(x + ((size_t)(r + nw) * D + (size_t)ch * 8) * 2);
      }
    }
    const float mean = wave_sum(sum) * invD;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunks) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = v[c][i] - mean; ss = __builtin_fmaf(d, d, ss); }
      }
    }
    const float rstd = rsqrtf(__builtin_fmaf(wave_sum(ss), invD, eps));      // (layernorm.hip: the expressions of ln_fwd_kernel, fused multiply-adds written out)
    const float s = rowscale[r] * it;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunks) {
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = __builtin_fmaf((v[c][i] - mean) * rstd, g[c][i], bt[c][i]);
        unpack8(pack8(o), o);                   // the bf16 rounding of the LayerNorm output the forward GEMM saw
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] *= s;
        st_stream<u32x2>(q + (size_t)r * D + (size_t)ch * 8, e4m3x8_sat(o));
      }
    }
  }
}

}  // namespace

// ---- predicted row scales (producer-fused quantisation, gemm_f8a.hip OUTQ) ---------------------------------------------------
// |sum_k a[m,k] w[n,k] + b[n]| <= ||a[m,:]||_2 * max_n ||w[n,:]||_2 + max_n |b[n]|: a bound of every output of row m that is known
// BEFORE the GEMM runs, from the row norm its producer returns and two scalars of the weights (once per optimizer step).
namespace {
// out[0] = max_r ||w[r,:]||_2 (w bf16 [rows, K]); one wave per row
__global__ __launch_bounds__(256) void rownorm_max_kernel(const unsigned short* __restrict__ w, long ld, long rows, int K, unsigned* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long nw = (long)gridDim.x * 4;
  float best = 0.f;
  for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += nw) {
    float ss = 0.f;
    for (int c = lane * 8; c < K; c += 512) {
      float f[8];
      unpack8(*(const u32x4*)(w + (size_t)r * ld + c), f);
#pragma unroll
      for (int i = 0; i < 8; ++i) ss = __builtin_fmaf(f[i], f[i], ss);
    }
    best = fmaxf(best, sqrtf(wave_sum(ss)));
  }
  if (lane == 0) atomicMax(out, __float_as_uint(best));
}
__global__ void absmax_kernel(const float* __restrict__ v, long n, unsigned* __restrict__ out) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(v[i]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}
// bound[m] = factor * rn[m] * wn[0] + bn[0];  scale[m] = bound / 448 (the de-quantisation scale of the row: 0 for a zero row),
// inv[m] = 448 / bound (what the producing epilogue multiplies by; 0 for a zero row)
__global__ void row_bound_kernel(const float* __restrict__ rn, const float* __restrict__ wn, const float* __restrict__ bn, float factor,
                                 float* __restrict__ scale, float* __restrict__ inv, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float b = __builtin_fmaf(factor * rn[i], wn[0], bn ? bn[0] : 0.f);
  scale[i] = b > 0.f ? b * (1.0f / 448.0f) : 0.f;
  inv[i] = b > 0.f ? 448.0f / b : 0.f;
}
}  // namespace

extern "C" int clipa_rownorm_max(const void* w, int64_t rows, int64_t K, int64_t ld, float* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (K <= 0 || K % 8 != 0 || ld % 8 != 0 || ld < K) { clipa_set_error("rownorm_max: K, ld must be multiples of 8, ld >= K"); return CLIPA_ERR_ARG; }
  if (hipMemsetAsync(out, 0, sizeof(float), st) != hipSuccess) { clipa_set_error("rownorm_max: hipMemsetAsync failed"); return CLIPA_ERR_LAUNCH; }
  if (rows <= 0) return CLIPA_OK;
  long blocks = (rows + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(rownorm_max_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const unsigned short*)w, (long)ld, (long)rows, (int)K, (unsigned*)out);
  return clipa_check_launch("rownorm_max");
}
extern "C" int clipa_absmax_f32(const float* v, int64_t n, float* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(out, 0, sizeof(float), st) != hipSuccess) { clipa_set_error("absmax_f32: hipMemsetAsync failed"); return CLIPA_ERR_LAUNCH; }
  if (n <= 0) return CLIPA_OK;
  long blocks = (n + 1023) / 1024;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, v, (long)n, (unsigned*)out);
  return clipa_check_launch("absmax_f32");
}
extern "C" int clipa_row_bound(const float* rownorm, const float* wnorm_dev, const float* bmax_dev, float factor, float* scale, float* inv,
                               int64_t n, void* stream) {
  if (n <= 0) return CLIPA_OK;
  if (!rownorm || !wnorm_dev || !scale || !inv) { clipa_set_error("row_bound: rownorm, wnorm, scale and inv are required"); return CLIPA_ERR_ARG; }
  hipLaunchKernelGGL(row_bound_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rownorm, wnorm_dev, bmax_dev, factor, scale, inv, (long)n);
  return clipa_check_launch("row_bound");
}

extern "C" int clipa_rowscale_max(const float* a, const float* b, int64_t n, float* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(out, 0, sizeof(float), st) != hipSuccess) { clipa_set_error("rowscale_max: hipMemsetAsync failed"); return CLIPA_ERR_LAUNCH; }
  if (n <= 0) return CLIPA_OK;
  long blocks = (n + 1023) / 1024;
  if (blocks > 256) blocks = 256;
  hipLaunchKernelGGL(rowscale_max_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, b, (long)n, (unsigned*)out);
  return clipa_check_launch("rowscale_max");
}

namespace {
// The e4m3-input activation variant through a table: an e4m3 byte has 256 values, so bf16(act(x)) is a 256-entry table in
// LDS (1 KiB, built once per block by its 256 threads) and an element costs one LDS gather, a multiply and its share of a
// conversion instead of ~14 VALU instructions of polynomial and rounding - the arithmetic version runs at 3.1 TB/s of its 2 bytes
// per element (VALU-bound), the table version at the memory system's rate.  Same values bit for bit (the table holds what the
// arithmetic kernel computes per element).  16 elements per thread (16-byte accesses), ROWS rows per block.
template <int ACT, int ROWS>
__global__ __launch_bounds__(256) void scale_quantize_rows_lut_kernel(const char* __restrict__ x, long ldx, const float* __restrict__ rowscale,
                                                                      const float* __restrict__ t_dev, char* __restrict__ q, long ldq,
                                                                      long rows, int nch16) {
  __shared__ float lut[256];
  {
    float f[8];
    u32x2 code;
    code[0] = threadIdx.x;        // byte 0 of the first word = this thread's e4m3 code
    code[1] = 0u;
    e4m3x8_to_f32(code, f);
    const f32x2 v = {f[0], f[0]};
    const f32x2 a = act_fwd2<ACT>(v);
    lut[threadIdx.x] = bf2f(f2bf(a.x));
  }
  __syncthreads();
  const float it = inv_or_zero(t_dev[0]);
  const long r0 = (long)blockIdx.x * ROWS;
#pragma unroll
  for (int rr = 0; rr < ROWS; ++rr) {
    const long r = r0 + rr;
    if (r >= rows) break;
    const float s = rowscale[r] * it;
    for (int ch = threadIdx.x; ch < nch16; ch += 256) {
      const u32x4 in = ld_stream<u32x4>(x + (size_t)r * ldx + (size_t)ch * 16);
      u32x4 out;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const unsigned c = in[w];
        const float f0 = lut[c & 255u] * s, f1 = lut[(c >> 8) & 255u] * s, f2 = lut[(c >> 16) & 255u] * s, f3 = lut[c >> 24] * s;
        int o = 0;
        o = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(f0, -448.f, 448.f), __builtin_amdgcn_fmed3f(f1, -448.f, 448.f), o, false);
        o = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(f2, -448.f, 448.f), __builtin_amdgcn_fmed3f(f3, -448.f, 448.f), o, true);
        out[w] = (unsigned)o;
      }
      st_stream<u32x4>(q + (size_t)r * ldq + (size_t)ch * 16, out);
    }
  }
}

template <bool IN8>
int scale_quantize_launch(const char* what, const void* x, const float* rowscale, const float* t_dev, void* q, int64_t rows, int64_t K,
                          int64_t ldx, int64_t ldq, int act, void* stream) {
  if (rows <= 0) return CLIPA_OK;
  if (K <= 0 || K % 8 != 0) { clipa_set_error("%s: K=%ld must be a positive multiple of 8", what, (long)K); return CLIPA_ERR_ARG; }
  if (ldx % 8 != 0 || ldq % 8 != 0 || ldx < K || ldq < K) { clipa_set_error("%s: ldx, ldq must be multiples of 8 and >= K", what); return CLIPA_ERR_ARG; }
  if (act < -1 || act > ACT_QUICK_GELU) { clipa_set_error("%s: unknown activation %d", what, act); return CLIPA_ERR_ARG; }
  if (!rowscale || !t_dev) { clipa_set_error("%s: rowscale and t are required", what); return CLIPA_ERR_ARG; }
  const long nch = K / 8, total = rows * nch;
  const long blocks = (total + 255) / 256;
  if (blocks > 0x7fffffffL) { clipa_set_error("%s: too many elements", what); return CLIPA_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
#ifndef SQ_LUT
#define SQ_LUT 1                                   // A/B knob: 0 keeps the arithmetic kernel for e4m3 inputs
#endif
  if (IN8 && SQ_LUT && act >= 0 && K % 16 == 0 && ldx % 16 == 0 && ldq % 16 == 0 && (((size_t)x | (size_t)q) & 15) == 0 && (rows + 3) / 4 <= 0x7fffffffL) {
    const dim3 g4((unsigned)((rows + 3) / 4)), b4(256);
#define SQ_LUT_LAUNCH(A) hipLaunchKernelGGL((scale_quantize_rows_lut_kernel<A, 4>), g4, b4, 0, st, (const char*)x, (long)ldx, rowscale, t_dev, (char*)q, (long)ldq, (long)rows, (int)(K / 16))
    if (act == ACT_GELU_ERF) SQ_LUT_LAUNCH(ACT_GELU_ERF);
    else if (act == ACT_GELU_TANH) SQ_LUT_LAUNCH(ACT_GELU_TANH);
    else SQ_LUT_LAUNCH(ACT_QUICK_GELU);
#undef SQ_LUT_LAUNCH
    return clipa_check_launch(what);
  }
  const dim3 grid((unsigned)blocks), block(256);
#define SQ_LAUNCH(A) hipLaunchKernelGGL((scale_quantize_rows_kernel<A, IN8>), grid, block, 0, st, (const char*)x, (long)ldx, rowscale, t_dev, (char*)q, (long)ldq, (long)rows, (int)nch)
  if (act < 0) SQ_LAUNCH(-1);
  else if (act == ACT_GELU_ERF) SQ_LAUNCH(ACT_GELU_ERF);
  else if (act == ACT_GELU_TANH) SQ_LAUNCH(ACT_GELU_TANH);
  else SQ_LAUNCH(ACT_QUICK_GELU);
#undef SQ_LAUNCH
  return clipa_check_launch(what);
}
}  // namespace

extern "C" int clipa_scale_quantize_rows(const void* x, const float* rowscale, const float* t_dev, void* q, int64_t rows, int64_t K,
                                         int64_t ldx, int64_t ldq, int act, void* stream) {
  return scale_quantize_launch<false>("scale_quantize_rows", x, rowscale, t_dev, q, rows, K, ldx, ldq, act, stream);
}
extern "C" int clipa_scale_quantize_rows_e4m3(const void* x8, const float* rowscale, const float* t_dev, void* q, int64_t rows,
                                              int64_t K, int64_t ldx, int64_t ldq, int act, void* stream) {
  return scale_quantize_launch<true>("scale_quantize_rows_e4m3", x8, rowscale, t_dev, q, rows, K, ldx, ldq, act, stream);
}

extern "C" int clipa_layernorm_fwd_q8s(const void* x, const float* gamma, const float* beta, const float* rowscale, const float* t_dev,
                                       void* q, int64_t rows, int64_t D, float eps, void* stream) {
  if (rows <= 0) return CLIPA_OK;
  if (D <= 0 || D % 8 != 0 || D > 2048) { clipa_set_error("layernorm_fwd_q8s: D=%ld must be a multiple of 8 in (0, 2048]", (long)D); return CLIPA_ERR_ARG; }
  if (!rowscale || !t_dev) { clipa_set_error("layernorm_fwd_q8s: rowscale and t are required"); return CLIPA_ERR_ARG; }
  long blocks = (rows + 3) / 4;
  if (blocks > LN_Q8_BLOCK_CAP) blocks = LN_Q8_BLOCK_CAP;
  const dim3 grid((unsigned)blocks), block(256);
  hipStream_t st = (hipStream_t)stream;
  const char* xp = (const char*)x;
  if (D <= 512) hipLaunchKernelGGL((ln_fwd_q8s_kernel<1>), grid, block, 0, st, xp, gamma, beta, rowscale, t_dev, (char*)q, (long)rows, (int)D, eps);
  else if (D <= 1024) hipLaunchKernelGGL((ln_fwd_q8s_kernel<2>), grid, block, 0, st, xp, gamma, beta, rowscale, t_dev, (char*)q, (long)rows, (int)D, eps);
  else if (D <= 1536) hipLaunchKernelGGL((ln_fwd_q8s_kernel<3>), grid, block, 0, st, xp, gamma, beta, rowscale, t_dev, (char*)q, (long)rows, (int)D, eps);
  else hipLaunchKernelGGL((ln_fwd_q8s_kernel<4>), grid, block, 0, st, xp, gamma, beta, rowscale, t_dev, (char*)q, (long)rows, (int)D, eps);
  return clipa_check_launch("layernorm_fwd_q8s");
}

// ---- e4m3 pre-activations (the "light8" keep tier; gemm_nta's CLIPA_EPI_ACT_PRE8 / CLIPA_EPI_DACT8 write and read the same bytes) ----
namespace {
__global__ void bf16_to_e4m3_kernel(const unsigned short* __restrict__ in, unsigned char* __restrict__ out, long n8) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float f[8];
    unpack8(ld_stream<u32x4>(in + 8 * i), f);
    st_stream<u32x2>(out + 8 * i, e4m3x8_sat(f));
  }
}
// 8 elements per thread (16 per thread - one 16-byte load, two 16-byte stores 32 bytes apart - measured 11 % slower,
// profiles/r04_stream_kernels_old_vs_new_lib.jsonl)
template <bool ACTIVATE>
__global__ void e4m3_to_bf16_kernel(const unsigned char* __restrict__ in, unsigned short* __restrict__ out, long n8, int act) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float f[8];
    e4m3x8_to_f32(ld_stream<u32x2>(in + 8 * i), f);
    if (ACTIVATE) {
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const f32x2 x = {f[j], f[j + 1]};
        const f32x2 r = act == ACT_GELU_ERF ? act_fwd2<ACT_GELU_ERF>(x) : (act == ACT_GELU_TANH ? act_fwd2<ACT_GELU_TANH>(x) : act_fwd2<ACT_QUICK_GELU>(x));
        f[j] = r.x;
        f[j + 1] = r.y;
      }
    }
    st_stream<u32x4>(out + 8 * i, pack8(f));
  }
}
// A table-driven variant (256-entry bf16 table in LDS, 16 gathers per 16-byte load) was measured in round 6 and is NOT kept:
// 2.30-2.52 ms against 1.81 ms for the kernel above on 806912 x 4096 bytes (profiles/r06_stream_kernels_activation_e4m3_lut_ab.jsonl).
// With two output bytes per input byte the polynomial kernel already streams (5.4 TB/s) and the LDS gathers become the limit;
// the table only pays where the per-element arithmetic is heavier (the e4m3 GELU emission and the gemm_nta DACT8 epilogue).
// one-shot: a thread per 8 elements (a persistent grid striding the array streams a third slower, tools/probes/stream_ab.hip)
unsigned cast_grid(long n8) { const long b = (n8 + 255) / 256; return (unsigned)(b < 1 ? 1 : (b > 0x7fffffffL ? 0x7fffffffL : b)); }
int cast_args(const char* what, const void* in, const void* out, int64_t n) {
  if (n % 8 != 0 || ((size_t)in & 7) || ((size_t)out & 7)) { clipa_set_error("%s: n must be a multiple of 8 and the buffers 8-byte aligned", what); return CLIPA_ERR_ARG; }
  return 0;
}
}  // namespace

extern "C" int clipa_cast_bf16_to_e4m3(const void* in, void* out, int64_t n, void* stream) {
  if (n <= 0) return CLIPA_OK;
  if (int rc = cast_args("cast_bf16_to_e4m3", in, out, n)) return rc;
  hipLaunchKernelGGL(bf16_to_e4m3_kernel, dim3(cast_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)in, (unsigned char*)out, (long)(n / 8));
  return clipa_check_launch("cast_bf16_to_e4m3");
}
extern "C" int clipa_cast_e4m3_to_bf16(const void* in, void* out, int64_t n, void* stream) {
  if (n <= 0) return CLIPA_OK;
  if (int rc = cast_args("cast_e4m3_to_bf16", in, out, n)) return rc;
  hipLaunchKernelGGL(e4m3_to_bf16_kernel<false>, dim3(cast_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)in, (unsigned short*)out, (long)(n / 8), 0);
  return clipa_check_launch("cast_e4m3_to_bf16");
}
extern "C" int clipa_activation_fwd_e4m3(const void* x8, void* out, int64_t n, int act, void* stream) {
  if (n <= 0) return CLIPA_OK;
  if (int rc = cast_args("activation_fwd_e4m3", x8, out, n)) return rc;
  if (act < ACT_GELU_ERF || act > ACT_QUICK_GELU) { clipa_set_error("activation_fwd_e4m3: unknown activation %d", act); return CLIPA_ERR_ARG; }
  hipLaunchKernelGGL(e4m3_to_bf16_kernel<true>, dim3(cast_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)x8, (unsigned short*)out, (long)(n / 8), act);
  return clipa_check_launch("activation_fwd_e4m3");
}
