// A/B of the HBM-streaming kernel shapes (copy, the MLP re-materialisation, LayerNorm forward) on one MI355X:
// how many 16-byte loads a lane has in flight, non-temporal loads / stores, persistent vs one-shot grids.
//   hipcc -O3 --offload-arch=gfx950 -I clipa_amd/csrc -I include tools/probes/stream_ab.hip -o tools/probes/stream_ab
//   ./tools/probes/stream_ab [rows=806912] [D=1024]
// Prints one JSON line per variant (GB/s of algorithmic bytes, median of 7 timed groups of 5 launches).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "common.h"

void clipa_set_error(const char*, ...) {}
int clipa_check_launch(const char*) { return 0; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NT> __device__ __forceinline__ u32x4 ldv(const u32x4* p) {
  if (NT & 1) return __builtin_nontemporal_load(p);
  return *p;
}
template <int NT> __device__ __forceinline__ void stv(u32x4* p, u32x4 v) {
  if (NT & 2) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// ---- copy / activation: U loads per lane in flight ----------------------------------------------------------------
template <int U, int NT, int ACT>
__global__ __launch_bounds__(256) void ew_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, long n) {
  const long stride = (long)gridDim.x * 256;
  for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += stride * U) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const long i = i0 + u * stride; if (i < n) v[u] = ldv<NT>(in + i); }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + u * stride;
      if (i < n) {
        if (ACT) {
          float f[8];
          unpack8(v[u], f);
#pragma unroll
          for (int k = 0; k < 8; k += 2) { const f32x2 r = act_fwd2<ACT_GELU_ERF>(f32x2{f[k], f[k + 1]}); f[k] = r.x; f[k + 1] = r.y; }
          v[u] = pack8(f);
        }
        stv<NT>(out + i, v[u]);
      }
    }
  }
}

// ---- LayerNorm forward, bf16 -> bf16, D = 1024 (2 chunks of 8 per lane), R rows per wave and iteration --------------
template <int R, int NT>
__global__ __launch_bounds__(256) void ln_kernel(const unsigned short* __restrict__ x, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, unsigned short* __restrict__ y, long rows, float eps, int C) {
  constexpr int D = 1024, NCH = 2;
  const int lane = threadIdx.x & 63;
  // C == 0: rows strided over all waves of a persistent grid; C > 0: block b owns rows [b C, (b + 1) C), its 4 waves interleaved
  const long wid = C ? (long)blockIdx.x * C + (threadIdx.x >> 6) : (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nw = C ? 4 : (long)gridDim.x * 4;
  if (C) rows = rows < (long)(blockIdx.x + 1) * C ? rows : (long)(blockIdx.x + 1) * C;
  float g[NCH][8], bt[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = lane + c * 64;
#pragma unroll
    for (int i = 0; i < 8; ++i) { g[c][i] = gamma[ch * 8 + i]; bt[c][i] = beta[ch * 8 + i]; }
  }
  const float invD = 1.0f / D;
  for (long r0 = wid; r0 < rows; r0 += R * nw) {
    u32x4 raw[R][NCH];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const long r = r0 + k * nw;
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        if (r < rows) raw[k][c] = ldv<NT>((const u32x4*)(x + (size_t)r * D + (size_t)(lane + c * 64) * 8));
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const long r = r0 + k * nw;
      if (r >= rows) break;
      float v[NCH][8];
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        unpack8(raw[k][c], v[c]);
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[c][i];
      }
      const float mean = wave_sum(s) * invD;
      float ss = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = v[c][i] - mean; ss += d * d; }
      const float rstd = rsqrtf(wave_sum(ss) * invD + eps);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (v[c][i] - mean) * rstd * g[c][i] + bt[c][i];
        stv<NT>((u32x4*)(y + (size_t)r * D + (size_t)(lane + c * 64) * 8), pack8(o));
      }
    }
  }
}


// ---- LayerNorm backward, bf16, D = 1024, with the residual gradient: 3 reads + 1 write per row, dgamma / dbeta partials per
// block (the product kernel's maths; same C mapping switch as above) ------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void lnb_kernel(const unsigned short* __restrict__ x, const float* __restrict__ gamma,
                                                  const unsigned short* __restrict__ dy, const unsigned short* __restrict__ dres,
                                                  unsigned short* __restrict__ dx, float* __restrict__ part, long rows, float eps, int C) {
  constexpr int D = 1024, NCH = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long wid = C ? (long)blockIdx.x * C + wv : (long)blockIdx.x * 4 + wv;
  const long nw = C ? 4 : (long)gridDim.x * 4;
  if (C) rows = rows < (long)(blockIdx.x + 1) * C ? rows : (long)(blockIdx.x + 1) * C;
  float g[NCH][8], dg[NCH][8], db[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) { dg[c][i] = 0.f; db[c][i] = 0.f; g[c][i] = gamma[(lane + c * 64) * 8 + i]; }
  const float invD = 1.0f / D;
  for (long r = wid; r < rows; r += nw) {
    float v[NCH][8], d[NCH][8];
    u32x4 rr[NCH];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const size_t o = (size_t)r * D + (size_t)(lane + c * 64) * 8;
      const u32x4 a = ldv<NT>((const u32x4*)(x + o)), b = ldv<NT>((const u32x4*)(dy + o));
      rr[c] = ldv<NT>((const u32x4*)(dres + o));
      unpack8(a, v[c]); unpack8(b, d[c]);
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[c][i];
    }
    const float mean = wave_sum(s) * invD;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float t = v[c][i] - mean; ss += t * t; }
    const float rstd = rsqrtf(wave_sum(ss) * invD + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float xh = (v[c][i] - mean) * rstd, dyv = d[c][i];
        dg[c][i] += dyv * xh; db[c][i] += dyv;
        const float dxh = dyv * g[c][i];
        s1 += dxh; s2 += dxh * xh;
        v[c][i] = xh; d[c][i] = dxh;
      }
    const float m1 = wave_sum(s1) * invD, m2 = wave_sum(s2) * invD;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      float o[8], q[8];
      unpack8(rr[c], q);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = rstd * (d[c][i] - m1 - v[c][i] * m2) + q[i];
      stv<NT>((u32x4*)(dx + (size_t)r * D + (size_t)(lane + c * 64) * 8), pack8(o));
    }
  }
  float* sd = (float*)smem;   // [4][2][D]
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      sd[(wv * 2 + 0) * D + (lane + c * 64) * 8 + i] = dg[c][i];
      sd[(wv * 2 + 1) * D + (lane + c * 64) * 8 + i] = db[c][i];
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
// 64 columns per block, 4 waves striding the partial rows (the product's reduce kernel); SPLIT > 1: blockIdx.y takes a slice
// of the partial rows and a second pass adds the SPLIT results
__global__ __launch_bounds__(256) void lnb_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int nblk, int D, int per) {
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const int lo = blockIdx.y * per, hi = lo + per < nblk ? lo + per : nblk;
  float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
  int s = lo + grp;
  for (; s + 4 < hi; s += 8) {
    a0 += part[(size_t)s * D + col]; a1 += part[(size_t)(s + 4) * D + col];
    b0 += part[((size_t)nblk + s) * D + col]; b1 += part[((size_t)nblk + s + 4) * D + col];
  }
  for (; s < hi; s += 4) { a0 += part[(size_t)s * D + col]; b0 += part[((size_t)nblk + s) * D + col]; }
  red[0][grp][lane] = a0 + a1; red[1][grp][lane] = b0 + b1;
  __syncthreads();
  if (grp == 0) {
    out[((size_t)blockIdx.y * 2 + 0) * D + col] = red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
    out[((size_t)blockIdx.y * 2 + 1) * D + col] = red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
  }
}

static hipEvent_t e0, e1;
template <class F> static double time_ms(F&& f) {
  for (int i = 0; i < 3; ++i) f();
  CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int g = 0; g < 7; ++g) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    t.push_back(ms / 5);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main(int argc, char** argv) {
  const long rows = argc > 1 ? atol(argv[1]) : 806912;
  const int D = 1024;
  const long n = rows * D / 8;              // 16-byte chunks
  const double bytes = 2.0 * rows * D * 2;  // read + write, bf16
  void *a, *b; float *g, *bt;
  CK(hipMalloc(&a, rows * D * 2)); CK(hipMalloc(&b, rows * D * 2));
  CK(hipMalloc(&g, D * 4)); CK(hipMalloc(&bt, D * 4));
  CK(hipMemset(a, 0x3c, rows * D * 2)); CK(hipMemset(g, 0, D * 4)); CK(hipMemset(bt, 0, D * 4));
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define EW(U, NT, ACT, CAP) do { \
    long grid = std::min<long>((n + 256L * U - 1) / (256L * U), CAP); \
    double ms = time_ms([&] { hipLaunchKernelGGL((ew_kernel<U, NT, ACT>), dim3(grid), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, n); }); \
    printf("{\"kernel\": \"%s\", \"U\": %d, \"nt\": %d, \"grid\": %ld, \"ms\": %.4f, \"gbps\": %.0f}\n", ACT ? "gelu" : "copy", U, NT, grid, ms, bytes / ms / 1e6); \
    fflush(stdout); } while (0)
#define LN(R, NT, CAP) do { \
    long grid = std::min<long>((rows + 4L * R - 1) / (4L * R), CAP); \
    double ms = time_ms([&] { hipLaunchKernelGGL((ln_kernel<R, NT>), dim3(grid), dim3(256), 0, 0, (const unsigned short*)a, g, bt, (unsigned short*)b, rows, 1e-5f, 0); }); \
    printf("{\"kernel\": \"ln_fwd\", \"R\": %d, \"nt\": %d, \"grid\": %ld, \"ms\": %.4f, \"gbps\": %.0f}\n", R, NT, grid, ms, bytes / ms / 1e6); \
    fflush(stdout); } while (0)
  const long BIG = 1L << 30;
  const int phase = argc > 2 ? atoi(argv[2]) : 2;
  if (phase == 2) {
    void *c, *d; float *part, *red;
    CK(hipMalloc(&c, rows * D * 2)); CK(hipMalloc(&d, rows * D * 2));
    CK(hipMemset(c, 0x3c, rows * D * 2)); CK(hipMemset(d, 0x3c, rows * D * 2));
    const long maxblk = (rows + 7) / 8;
    CK(hipMalloc(&part, (size_t)2 * maxblk * D * 4)); CK(hipMalloc(&red, (size_t)2 * 64 * D * 4));
#define LNC(R, NT, C) do { \
    long grid = (rows + (C) - 1) / (C); \
    double ms = time_ms([&] { hipLaunchKernelGGL((ln_kernel<R, NT>), dim3(grid), dim3(256), 0, 0, (const unsigned short*)a, g, bt, (unsigned short*)b, rows, 1e-5f, C); }); \
    printf("{\"kernel\": \"ln_fwd\", \"R\": %d, \"nt\": %d, \"rows_per_block\": %d, \"grid\": %ld, \"ms\": %.4f, \"gbps\": %.0f}\n", R, NT, C, grid, ms, bytes / ms / 1e6); \
    fflush(stdout); } while (0)
#define LNB(NT, C, GRID, SPLIT) do { \
    long grid = (C) ? (rows + (C) - 1) / (C) : (GRID); \
    int per = (int)((grid + (SPLIT) - 1) / (SPLIT)); \
    double ms = time_ms([&] { \
      hipLaunchKernelGGL((lnb_kernel<NT>), dim3(grid), dim3(256), 8 * D * 4, 0, (const unsigned short*)a, g, (const unsigned short*)c, (const unsigned short*)d, (unsigned short*)b, part, rows, 1e-5f, C); \
      hipLaunchKernelGGL(lnb_reduce_kernel, dim3(D / 64, SPLIT), dim3(256), 0, 0, part, red, (int)grid, D, per); \
      if ((SPLIT) > 1) hipLaunchKernelGGL(lnb_reduce_kernel, dim3(D / 64, 1), dim3(256), 0, 0, red, red + 2 * 64 * D - 2 * D, (int)(SPLIT), D, (int)(SPLIT)); }); \
    printf("{\"kernel\": \"ln_bwd+reduce\", \"nt\": %d, \"rows_per_block\": %d, \"grid\": %ld, \"split\": %d, \"ms\": %.4f, \"gbps\": %.0f}\n", NT, C, grid, SPLIT, ms, 2 * bytes / ms / 1e6); \
    fflush(stdout); } while (0)
    for (int rep = 0; rep < 2; ++rep) {
      EW(1, 0, 1, BIG); EW(2, 0, 1, BIG); EW(1, 3, 1, BIG); EW(2, 3, 1, BIG); EW(4, 3, 1, BIG); EW(1, 2, 1, BIG); EW(1, 3, 0, BIG); EW(1, 2, 0, BIG);
      LNC(2, 0, 8); LNC(2, 3, 8); LNC(2, 3, 16); LNC(2, 3, 32); LNC(2, 3, 64); LNC(1, 3, 4); LNC(1, 3, 8); LNC(1, 3, 16); LNC(2, 2, 8); LNC(4, 3, 16); LN(2, 3, BIG);
      LNB(0, 0, 1024, 1); LNB(3, 0, 1024, 1); LNB(0, 0, 1280, 1); LNB(0, 0, 2048, 1); LNB(3, 0, 2048, 1); LNB(0, 0, 4096, 1);
      LNB(0, 64, 0, 16); LNB(3, 64, 0, 16); LNB(3, 128, 0, 16); LNB(3, 256, 0, 8); LNB(3, 32, 0, 32); LNB(0, 128, 0, 16); LNB(3, 512, 0, 4);
    }
    return 0;
  }
  for (int rep = 0; rep < 2; ++rep) {
    EW(1, 0, 0, 8192); EW(2, 0, 0, 8192); EW(4, 0, 0, 8192); EW(4, 0, 0, 2048); EW(1, 0, 0, BIG); EW(4, 0, 0, BIG);
    EW(1, 1, 0, 8192); EW(1, 2, 0, 8192); EW(1, 3, 0, 8192); EW(4, 3, 0, 8192); EW(4, 1, 0, 8192); EW(4, 3, 0, BIG);
    EW(1, 0, 1, 8192); EW(2, 0, 1, 8192); EW(4, 0, 1, 8192); EW(4, 3, 1, 8192); EW(2, 3, 1, 8192); EW(1, 3, 1, 8192); EW(4, 3, 1, BIG); EW(4, 0, 1, 2048);
    LN(2, 0, 2048); LN(1, 0, 2048); LN(4, 0, 2048); LN(2, 0, 1024); LN(2, 0, 4096); LN(2, 0, BIG); LN(1, 0, BIG); LN(4, 0, BIG);
    LN(2, 1, 2048); LN(2, 2, 2048); LN(2, 3, 2048); LN(4, 3, 2048); LN(1, 3, BIG); LN(2, 3, BIG); LN(4, 3, 1024);
  }
  return 0;
}
