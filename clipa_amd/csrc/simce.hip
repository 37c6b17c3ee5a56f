// Fused similarity + cross-entropy for the InfoNCE loss (K17 + K18 of SURVEY 2.5; clipa_torch/open_clip/loss.py:128-155):
//   logits = s * A . B^T   (A = the rows' embeddings [R, E], B = all gathered embeddings [N, E], bf16, s on the device)
//   loss_r = logsumexp_j logits[r, j] - logits[r, label0 + r]
// The [R, N] fp32 logits never reach HBM: the forward GEMM's epilogue reduces every 256 x 256 tile to per-row
// (max, sum exp) partials (8 bytes per row and tile) and a second tiny kernel merges them into the log-sum-exp; the
// backward re-runs the GEMM and turns each tile straight into the bf16 gradient  s * gscale * (softmax - onehot)  that
// the four gradient GEMMs consume, plus per-row partials of  d loss / d s.  (At W = 8: 2 x 512 MB of fp32 logits and
// three passes over them are replaced by 2 x 4 MB of partials; the bf16 gradient matrix is still materialised.)
//
// Tile / ring / fragments: the 256x256x64, 8-wave, v_mfma_f32_32x32x16_bf16 structure of gemm_nt2<f32> with one
// output tile per workgroup (the loss is 0.07 % of the step's FLOPs: no persistence, no tile remap).
#include "gemm_common.h"

namespace clipa_gemm {
namespace {

struct CEArgs {
  const char* A; const char* B;
  int R, N, K;
  long lda, ldb;
  const float* scale;
  long label0;
  float gscale;
  float* part;          // fwd: [tilesN][R][2] (max, sum exp) of the scaled logits.   bwd: [tilesN][R] partial d loss / d s
  float* lab;           // fwd: [R] raw similarity at the label column
  const float* lse;     // bwd: [R] log-sum-exp of the scaled logits
  unsigned short* dl;   // bwd: bf16 [R, ldd] = s * gscale * (softmax - onehot), columns >= N zero
  long ldd;
};

template <bool BWD>
__global__ __launch_bounds__(NTHREADS) void simce_kernel(CEArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wave >> 2, wn = wave & 3;   // wave tile: 128 (m) x 64 (n)
  const int tilesN = (p.N + BN - 1) / BN;
  const int tm = blockIdx.x / tilesN, tn = blockIdx.x - tm * tilesN;
  const int m0 = tm * BM, n0 = tn * BN;
  const int rowsA = min(BM, p.R - m0), rowsB = min(BN, p.N - n0);
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A + (size_t)m0 * p.lda * 2, (unsigned)(rowsA * p.lda * 2));
  const __amdgpu_buffer_rsrc_t rsB = make_rsrc(p.B + (size_t)n0 * p.ldb * 2, (unsigned)(rowsB * p.ldb * 2));

  unsigned voffA[4], voffB[4];
  int kel[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (j * 8 + wave) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    voffA[j] = (unsigned)(row * p.lda * 2 + chunk * 16);
    voffB[j] = (unsigned)(row * p.ldb * 2 + chunk * 16);
    kel[j] = chunk * 8;
  }
  f32x16 acc[2][4];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;
  auto stage = [&](int buf, int k0) {
    char* sA = smem + buf * STAGE_BYTES;
    char* sB = sA + IMG_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pc = j * 8 + wave;
      const unsigned oob = (k0 + kel[j] >= p.K) ? 0x80000000u : 0u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(sA + pc * 1024), 16, voffA[j] | oob, k0 * 2, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(sB + pc * 1024), 16, voffB[j] | oob, k0 * 2, 0, 0);
    }
  };
  const int sw = (l31 >> 1) & 7;
  const int rowoffA = (wm * 128 + l31) * 128;
  const int rowoffB = (wn * 64 + l31) * 128;
  const int nkt = (p.K + BK - 1) / BK;
  stage(0, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nkt) stage((kt + 1) & 1, (kt + 1) * BK);
    const char* sA = smem + (kt & 1) * STAGE_BYTES;
    const char* sB = sA + IMG_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int coff = ((2 * ks + hi) ^ sw) << 4;
      bf16x8 fa[4], fb[2];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) fa[mi] = *(const bf16x8*)(sA + rowoffA + mi * 4096 + coff);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) fb[ni] = *(const bf16x8*)(sB + rowoffB + ni * 4096 + coff);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ni], fa[mi], acc[ni][mi], 0, 0, 0);
    }
  }
  __syncthreads();                                   // the ring is dead: its first bytes become the cross-wave scratch
  float* red = (float*)smem;                         // [4 wn][256 rows][2]

  // D[n][m] fragment: lane holds row m = wm*128 + mi*32 + l31 and columns n = wn*64 + ni*32 + 8*(r>>2) + 4*hi + (r&3)
  const float s = p.scale ? p.scale[0] : 1.0f;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int rl = wm * 128 + mi * 32 + l31;          // row within the tile
    const int m = m0 + rl;
    const long label = p.label0 + m;
    if (!BWD) {
      float mx = -3.0e38f;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + wn * 64 + ni * 32 + 8 * (r >> 2) + 4 * hi + (r & 3);
          if (n < p.N) mx = fmaxf(mx, acc[ni][mi][r] * s);
          if ((long)n == label && m < p.R) p.lab[m] = acc[ni][mi][r];
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + wn * 64 + ni * 32 + 8 * (r >> 2) + 4 * hi + (r & 3);
          if (n < p.N) sum += __expf(acc[ni][mi][r] * s - mx);
        }
      sum += __shfl_xor(sum, 32, 64);
      if (hi == 0) { red[(wn * 256 + rl) * 2] = mx; red[(wn * 256 + rl) * 2 + 1] = sum; }
    } else {
      const float lse = m < p.R ? p.lse[m] : 0.f;
      float ds = 0.f;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wn * 64 + ni * 32 + 8 * q + 4 * hi;
          float g[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x = acc[ni][mi][4 * q + e];
            float pr = (n + e < p.N) ? __expf(x * s - lse) : 0.f;
            if ((long)(n + e) == label) pr -= 1.0f;
            const float gg = pr * p.gscale;           // d loss / d logit
            ds += gg * x;                             // d loss / d s
            g[e] = gg * s;                            // d loss / d raw
          }
          if (m < p.R && n < p.ldd) {
            u32x2 w;
            w[0] = pack2bf(g[0], g[1]);
            w[1] = pack2bf(g[2], g[3]);
            *(u32x2*)(p.dl + (size_t)m * p.ldd + n) = w;
          }
        }
      ds += __shfl_xor(ds, 32, 64);
      if (hi == 0) red[wn * 256 + rl] = ds;
    }
  }
  __syncthreads();
  if (tid < 256 && m0 + tid < p.R) {
    if (!BWD) {
      float mx = -3.0e38f;
#pragma unroll
      for (int w = 0; w < 4; ++w) mx = fmaxf(mx, red[(w * 256 + tid) * 2]);
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) sum += red[(w * 256 + tid) * 2 + 1] * __expf(red[(w * 256 + tid) * 2] - mx);
      float* o = p.part + ((size_t)tn * p.R + m0 + tid) * 2;
      o[0] = mx;
      o[1] = sum;
    } else {
      p.part[(size_t)tn * p.R + m0 + tid] = red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid];
    }
  }
}

// fwd: merge the per-tile (max, sum) partials -> lse, loss row.   bwd: sum the per-tile partials of d loss / d s.
template <bool BWD>
__global__ void simce_merge_kernel(const float* __restrict__ part, int tilesN, long R, const float* __restrict__ lab,
                                   const float* __restrict__ scale, float* __restrict__ lse, float* __restrict__ out_rows) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  if (!BWD) {
    float mx = -3.0e38f;
    for (int t = 0; t < tilesN; ++t) mx = fmaxf(mx, part[((size_t)t * R + r) * 2]);
    float sum = 0.f;
    for (int t = 0; t < tilesN; ++t) sum += part[((size_t)t * R + r) * 2 + 1] * __expf(part[((size_t)t * R + r) * 2] - mx);
    const float l = mx + logf(sum);
    lse[r] = l;
    out_rows[r] = l - lab[r] * (scale ? scale[0] : 1.0f);
  } else {
    float a = 0.f;
    for (int t = 0; t < tilesN; ++t) a += part[(size_t)t * R + r];
    out_rows[r] = a;
  }
}

std::once_flag g_ce_once[MAX_DEVICES];
int g_ce_rc[MAX_DEVICES];
int ensure_ce_attrs(int dev) {
  std::call_once(g_ce_once[dev], [dev]() {
    g_ce_rc[dev] = 0;
    const void* ks[2] = {(const void*)simce_kernel<false>, (const void*)simce_kernel<true>};
    for (int i = 0; i < 2; ++i) {
      const hipError_t e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
      if (e != hipSuccess) { clipa_set_error("hipFuncSetAttribute(simce): %s", hipGetErrorString(e)); g_ce_rc[dev] = CLIPA_ERR_LAUNCH; }
    }
  });
  return g_ce_rc[dev];
}

int ce_check(int64_t R, int64_t N, int64_t E, int64_t lda, int64_t ldb, int64_t label0) {
  if (E <= 0 || E % 8 != 0 || lda % 8 != 0 || ldb % 8 != 0) { clipa_set_error("simce: E, lda, ldb must be multiples of 8"); return CLIPA_ERR_ARG; }
  if (label0 < 0 || label0 + R > N) { clipa_set_error("simce: labels [%ld, %ld) outside [0, %ld)", (long)label0, (long)(label0 + R), (long)N); return CLIPA_ERR_ARG; }
  if (256 * lda * 2 >= (1L << 30) || 256 * ldb * 2 >= (1L << 30)) { clipa_set_error("simce: leading dimension too large"); return CLIPA_ERR_ARG; }
  return 0;
}

}  // namespace
}  // namespace clipa_gemm

using namespace clipa_gemm;

extern "C" int64_t clipa_simce_workspace(int64_t R, int64_t N) {
  const int64_t tilesN = (N + BN - 1) / BN;
  return (tilesN * R * 2 + R) * (int64_t)sizeof(float);       // per-tile partials + the label similarities
}

extern "C" int clipa_simce_fwd(const void* rows, const void* cols, int64_t R, int64_t N, int64_t E, int64_t lda,
                               int64_t ldb, const float* scale, int64_t label0, float* lse, float* loss_rows,
                               void* workspace, int64_t workspace_bytes, void* stream) {
  if (R <= 0) return CLIPA_OK;
  if (int rc = ce_check(R, N, E, lda, ldb, label0)) return rc;
  if (!workspace || workspace_bytes < clipa_simce_workspace(R, N)) { clipa_set_error("simce_fwd: workspace too small"); return CLIPA_ERR_ARG; }
  int dev = 0;
  if (int rc = current_device(&dev)) return rc;
  if (int rc = ensure_ce_attrs(dev)) return rc;
  const int64_t tilesN = (N + BN - 1) / BN, tilesM = (R + BM - 1) / BM;
  CEArgs a = {};
  a.A = (const char*)rows; a.B = (const char*)cols; a.R = (int)R; a.N = (int)N; a.K = (int)E; a.lda = lda; a.ldb = ldb;
  a.scale = scale; a.label0 = label0; a.part = (float*)workspace; a.lab = (float*)workspace + tilesN * R * 2;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(simce_kernel<false>, dim3((unsigned)(tilesM * tilesN)), dim3(NTHREADS), 2 * STAGE_BYTES, st, a);
  if (int rc = clipa_check_launch("simce_fwd")) return rc;
  hipLaunchKernelGGL(simce_merge_kernel<false>, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, a.part, (int)tilesN, (long)R,
                     a.lab, scale, lse, loss_rows);
  return clipa_check_launch("simce_merge");
}

extern "C" int clipa_simce_bwd(const void* rows, const void* cols, int64_t R, int64_t N, int64_t E, int64_t lda,
                               int64_t ldb, const float* scale, int64_t label0, float gscale, const float* lse,
                               void* dlogits_bf16, int64_t ldd, float* dscale_rows, void* workspace,
                               int64_t workspace_bytes, void* stream) {
  if (R <= 0) return CLIPA_OK;
  if (int rc = ce_check(R, N, E, lda, ldb, label0)) return rc;
  const int64_t N8 = (N + 7) & ~(int64_t)7;
  if (!dlogits_bf16 || ldd % 8 != 0 || ldd < N8) { clipa_set_error("simce_bwd: dlogits needs ldd %% 8 == 0 and ldd >= N rounded up to 8"); return CLIPA_ERR_ARG; }
  if (!workspace || workspace_bytes < clipa_simce_workspace(R, N)) { clipa_set_error("simce_bwd: workspace too small"); return CLIPA_ERR_ARG; }
  int dev = 0;
  if (int rc = current_device(&dev)) return rc;
  if (int rc = ensure_ce_attrs(dev)) return rc;
  const int64_t tilesN = (N + BN - 1) / BN, tilesM = (R + BM - 1) / BM;
  CEArgs a = {};
  a.A = (const char*)rows; a.B = (const char*)cols; a.R = (int)R; a.N = (int)N; a.K = (int)E; a.lda = lda; a.ldb = ldb;
  a.scale = scale; a.label0 = label0; a.gscale = gscale; a.part = (float*)workspace; a.lse = lse;
  a.dl = (unsigned short*)dlogits_bf16; a.ldd = ldd;    // columns [N, ldd) inside the last tile are written as zeros
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(simce_kernel<true>, dim3((unsigned)(tilesM * tilesN)), dim3(NTHREADS), 2 * STAGE_BYTES, st, a);
  if (int rc = clipa_check_launch("simce_bwd")) return rc;
  hipLaunchKernelGGL(simce_merge_kernel<true>, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, a.part, (int)tilesN, (long)R,
                     (const float*)nullptr, scale, (float*)nullptr, dscale_rows);
  return clipa_check_launch("simce_merge");
}
