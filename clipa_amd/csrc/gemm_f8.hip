// fp8 MFMA GEMM for gfx950 (MI355X):  C[M,N] = epi(alpha * sa[m] * sb[n] * A8[M,K] . B8[N,K]^T + bias[N])
//
// BASELINE.json configs[3] ("fp8 MFMA weights/activations", ViT-H/14): the reference has no fp8 mode (its precision
// menu is clipa_torch/training/params.py:195-200), so this is the engine's own recipe for the same call sites as
// gemm_nt.hip - the four linear layers of a residual block, forward and input gradient (clipa_torch/open_clip/
// transformer.py:209,217-219,234): operands are OCP e4m3 / e5m2 bytes with one f32 de-quantisation scale per ROW
// (per token for activations / gradients, per output channel for weights: clipa_quantize_rows), products on
// v_mfma_f32_16x16x128_f8f6f4 (twice the bf16 MFMA rate), fp32 accumulation, the bf16 kernel's fused epilogues.
//
// Geometry = gemm_nt2_kernel<bf16, M16> with the K tile counted in BYTES: an operand image is still 256 rows x 128 B
// (now 128 fp8 k-values), so the LDS-DMA pattern, the (row >> 1) & 7 chunk swizzle, the 2-slot ring and the
// LDS-window epilogue are unchanged, and one MFMA consumes a whole image row: 32 MFMAs per K tile per wave instead
// of 64 at equal bytes moved - the same timeline as the bf16 kernel with twice the FLOPs in it.
// A lane's 32 operand bytes are chunks g and 4 + g of its row (g = lane >> 4), i.e. the two conflict-free 16-B reads
// of the bf16 kernel; which k-values a lane holds is immaterial as long as both operands use the same rule.
#include "gemm_common.h"
#include "gemm_f8a.h"

namespace clipa_gemm {
namespace {

constexpr int F8_CBUF_OFF = 2 * STAGE_BYTES;            // 131072: epilogue window behind the ring
constexpr int F8_LDS_BYTES = F8_CBUF_OFF + 64 * 512;    // 163840

typedef int i32x8 __attribute__((ext_vector_type(8)));

struct F8Args {
  const char* A; const char* B; char* C; char* C2; const float* bias; const char* aux;
  const float* sa; const float* sb;     // de-quantisation scales: per row of A [M], per row of B [N]; NULL = 1
  int M, N, K;
  long lda, ldb, ldc, ldaux;            // element strides (A, B: bytes)
  float alpha;
  int epi, act, abl;
  int gm;                               // A panels per tile group (nt_group_size)
};

__device__ __forceinline__ i32x8 frag8(const char* rowp, int g4, int sw16) {
  const u32x4 lo = *(const u32x4*)(rowp + ((g4 ^ sw16) << 4));
  const u32x4 hi = *(const u32x4*)(rowp + (((4 + g4) ^ sw16) << 4));
  i32x8 f;
  f[0] = (int)lo[0]; f[1] = (int)lo[1]; f[2] = (int)lo[2]; f[3] = (int)lo[3];
  f[4] = (int)hi[0]; f[5] = (int)hi[1]; f[6] = (int)hi[2]; f[7] = (int)hi[3];
  return f;
}

// FMT_A / FMT_B: 0 = e4m3, 1 = e5m2 of the A (activation / gradient) and B (weight) operand
// EPI / PRE: compile-time epilogue selector and "also store the pre-activation" (see window_epilogue in gemm_common.h)
template <int FMT_A, int FMT_B, int EPI, bool PRE>
__global__ __launch_bounds__(NTHREADS) void gemm_nt_f8_kernel(F8Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l15 = lane & 15, g4 = lane >> 4, sw16 = (l15 >> 1) & 7;

  const int tilesN = (p.N + BN - 1) / BN;
  const int tilesM = (p.M + BM - 1) / BM;
  const unsigned ntiles = (unsigned)(tilesM * tilesN);
  const unsigned G = gridDim.x, xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
  const unsigned gx = (G - xcd + 7u) >> 3;            // workgroups on this XCD
  const unsigned q8 = ntiles >> 3, r8 = ntiles & 7u;
  const unsigned base = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const unsigned len = q8 + (xcd < r8 ? 1u : 0u);

  // LEAN (activation-backward epilogue, the register-tightest instantiation): the 12 per-lane DMA offsets are recomputed
  // at every stage() from an opaque copy of the lane id instead of living in registers across the tile loop
  constexpr bool LEAN = EPI == CLIPA_EPI_DACT;
  const int nkt = (p.K + 127) / 128;

  auto tile_origin = [&](unsigned t, int& m0, int& n0) {
    const int GM = p.gm;
    const int per = GM * tilesN;
    const int g = (int)t / per, r = (int)t - g * per;
    const int gm = min(GM, tilesM - g * GM);
    const int tn = r / gm, mm = r - tn * gm;
    m0 = (g * GM + mm) * BM;
    n0 = tn * BN;
  };
  auto stage = [&](int buf, int m0, int n0, int k0) {
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A + (size_t)m0 * p.lda, (unsigned)(min(BM, p.M - m0) * p.lda));
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(p.B + (size_t)n0 * p.ldb, (unsigned)(min(BN, p.N - n0) * p.ldb));
    char* sA = smem + buf * STAGE_BYTES;
    char* sB = sA + IMG_BYTES;
    int l = lane;
    if constexpr (LEAN) asm volatile("" : "+v"(l));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pc = j * 8 + wave;
      const int row = pc * 8 + (l >> 3);
      const int chunk = (l & 7) ^ ((row >> 1) & 7);
      const unsigned oob = (k0 + chunk * 16 >= p.K) ? 0x80000000u : 0u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(sA + pc * 1024), 16, (unsigned)(row * p.lda + chunk * 16) | oob, k0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(sB + pc * 1024), 16, (unsigned)(row * p.ldb + chunk * 16) | oob, k0, 0, 0);
    }
  };

  if (idx >= len) return;
  unsigned it = idx;
  int m0, n0;
  tile_origin(base + it, m0, n0);
  stage(0, m0, n0, 0);
  unsigned gk = 0;
  RING_WAIT_ALL();
  for (;;) {
    const bool has_next = it + gx < len;
    int m1 = 0, n1 = 0;
    if (has_next) tile_origin(base + it + gx, m1, n1);

    f32x4v acc16[4][8];      // [n block of 16][m block of 16]
#pragma unroll
    for (int bj = 0; bj < 4; ++bj)
#pragma unroll
      for (int ai = 0; ai < 8; ++ai) acc16[bj][ai] = f32x4v{0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < nkt; ++kt, ++gk) {
      if (kt + 1 < nkt) stage((gk + 1) & 1, m0, n0, (kt + 1) * 128);
      else {
        // the tile's row scales, bias and column scales (256 floats each) ride the last K step as three 1 KiB LDS-DMA
        // into the (idle) epilogue window, one per wave 0..2, issued ahead of the next tile's operands
        const float* vsrc = wave == 0 ? p.sa : wave == 1 ? p.bias : wave == 2 ? p.sb : nullptr;
        if (vsrc) {
          const int o0 = wave == 0 ? m0 : n0, ext = wave == 0 ? p.M : p.N;
          const __amdgpu_buffer_rsrc_t rsV = make_rsrc(vsrc + o0, (unsigned)(max(0, min(256, ext - o0)) * 4));
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, LDS_PTR(smem + F8_CBUF_OFF + wave * 1024), 16, (unsigned)(lane * 16), 0, 0, 0);
        }
        if (has_next) stage((gk + 1) & 1, m1, n1, 0);
      }
      const char* sA = smem + (gk & 1) * STAGE_BYTES;
      const char* sB = sA + IMG_BYTES;
      const char* pa = sA + (wm * 128 + l15) * 128;
      const char* pb = sB + (wn * 64 + l15) * 128;
      // 8 sub-steps per K tile: A block u against the four B blocks (4 MFMAs of 16x16x128); A fragments double-buffered
      i32x8 gb[4], ga[2];
#pragma unroll
      for (int bj = 0; bj < 4; ++bj) gb[bj] = frag8(pb + bj * 2048, g4, sw16);
      ga[0] = frag8(pa, g4, sw16);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (u < 7) ga[(u + 1) & 1] = frag8(pa + (u + 1) * 2048, g4, sw16);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int bj = 0; bj < 4; ++bj)
          acc16[bj][u] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(gb[bj], ga[u & 1], acc16[bj][u], FMT_B, FMT_A, 0, 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
      if (kt + 1 < nkt) RING_WAIT_ALL();
    }

    // ---- epilogue of tile (m0, n0); the ring keeps filling for the next tile meanwhile ----
    if (p.abl & 2) {   // ablation: keep the accumulators live, write nothing
      float t = 0.f;
#pragma unroll
      for (int bj = 0; bj < 4; ++bj)
#pragma unroll
        for (int ai = 0; ai < 8; ++ai) t += acc16[bj][ai][0] + acc16[bj][ai][1] + acc16[bj][ai][2] + acc16[bj][ai][3];
      if (t == 1.2345e-30f) ((float*)p.C)[tid] = t;
    } else {
      char* cb = smem + F8_CBUF_OFF;
      // the tile's vectors (row scales | bias | column scales, 1 KiB each; out-of-range entries were DMA'd as zeros and are
      // never stored) are parked in the ring slot the last K step has finished with and read back by the packing waves
      char* park = smem + ((gk + 1) & 1) * STAGE_BYTES;   // gk & 1 holds the next tile's first K step
      if (p.sa || p.bias || p.sb) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        park_vectors(cb, park, tid, 3072);
      }
      float4 bias4[4], sb4[4];    // the 4 consecutive features of n block bj this lane holds
      float sam[4];               // alpha * row scale of the pass's 4 m blocks' row this lane holds
      WinOut o;
      o.C = p.C; o.C2 = p.C2; o.aux = p.aux; o.ldc = p.ldc; o.ldaux = p.ldaux;
      o.M = p.M; o.N = p.N; o.m0 = m0; o.n0 = n0; o.act = p.act; o.abl = p.abl;
      window_epilogue<EPI, PRE, true>(cb, o, tid, wm, wn,
        [&](int pass) {
          if (p.sa) lds_read4_f1(sam, park + (wm * 128 + (pass & 1) * 64 + l15) * 4);
          if (p.bias) lds_read4_f4(bias4, park + 1024 + (wn * 64 + 4 * g4) * 4);
          if (p.sb) lds_read4_f4(sb4, park + 2048 + (wn * 64 + 4 * g4) * 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            sam[i] = p.sa ? sam[i] * p.alpha : p.alpha;
            if (!p.bias) bias4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!p.sb) sb4[i] = make_float4(1.f, 1.f, 1.f, 1.f);
          }
        },
        [&](int ai, int bj) {
          const float s = sam[ai & 3];
          const float4 b4 = bias4[bj], q4 = sb4[bj];
          u32x2 w;
          w[0] = pack2bf(acc16[bj][ai][0] * (s * q4.x) + b4.x, acc16[bj][ai][1] * (s * q4.y) + b4.y);
          w[1] = pack2bf(acc16[bj][ai][2] * (s * q4.z) + b4.z, acc16[bj][ai][3] * (s * q4.w) + b4.w);
          return w;
        });
    }
    if (!has_next) break;
    RING_WAIT_AFTER_EPILOGUE(win_stores(PRE));   // the next tile's first K step has landed; this tile's stores need not have
    it += gx;
    m0 = m1;
    n0 = n1;
  }
}

std::once_flag g_f8_once[MAX_DEVICES];
int g_f8_rc[MAX_DEVICES];

int ensure_f8_attrs(int dev) {
  std::call_once(g_f8_once[dev], [dev]() {
    int rc = 0;
#define F8_KS(E, P2) (const void*)gemm_nt_f8_kernel<0, 0, E, P2>, (const void*)gemm_nt_f8_kernel<1, 0, E, P2>, \
                     (const void*)gemm_nt_f8_kernel<0, 1, E, P2>, (const void*)gemm_nt_f8_kernel<1, 1, E, P2>
    const void* ks[20] = {F8_KS(CLIPA_EPI_NONE, false), F8_KS(CLIPA_EPI_ACT, false), F8_KS(CLIPA_EPI_ACT, true),
                          F8_KS(CLIPA_EPI_ADD, false), F8_KS(CLIPA_EPI_DACT, false)};
#undef F8_KS
    for (int i = 0; i < 20; ++i) {
      const hipError_t e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, F8_LDS_BYTES);
      if (e != hipSuccess) { clipa_set_error("hipFuncSetAttribute(gemm_nt_f8): %s", hipGetErrorString(e)); rc = CLIPA_ERR_LAUNCH; }
    }
    g_f8_rc[dev] = rc;
  });
  return g_f8_rc[dev];
}

}  // namespace
}  // namespace clipa_gemm

using namespace clipa_gemm;

namespace {
int gemm_nt_f8_impl(const void* A8, const void* B8, const float* scale_a, const float* scale_b, void* C, void* C2, const float* bias,
                    const void* aux, const float* scale_out, float* colsum_partial, int outq, int64_t M, int64_t N, int64_t K,
                    int64_t lda, int64_t ldb, int64_t ldc, int64_t ldaux, float alpha, int epi, int act, int fmt_a, int fmt_b,
                    void* stream, const float* emit_t = nullptr);
}

extern "C" int clipa_gemm_nt_f8(const void* A8, const void* B8, const float* scale_a, const float* scale_b, void* C,
                                void* C2, const float* bias, const void* aux, int64_t M, int64_t N, int64_t K,
                                int64_t lda, int64_t ldb, int64_t ldc, int64_t ldaux, float alpha, int epi, int act,
                                int fmt_a, int fmt_b, void* stream) {
  return gemm_nt_f8_impl(A8, B8, scale_a, scale_b, C, C2, bias, aux, nullptr, nullptr, 0, M, N, K, lda, ldb, ldc, ldaux, alpha, epi,
                         act, fmt_a, fmt_b, stream);
}

// The producer-quantised form (round 6): C8 = e4m3(epi(...) rounded to bf16, row m times scale_out[m]) - the fp8 operand of the NEXT
// GEMM written by this one's epilogue, with a row scale the caller predicted (engine._row_bound) - and, for CLIPA_EPI_DACT8, the
// partial column sums of the unscaled outputs ([M / 128][N] floats; clipa_reduce_partial_rows adds them up: the bias gradient).
extern "C" int clipa_gemm_nt_f8q(const void* A8, const void* B8, const float* scale_a, const float* scale_b, void* C8, void* C2,
                                 const float* bias, const void* aux, const float* scale_out, float* colsum_partial, int64_t M,
                                 int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldaux, float alpha, int epi,
                                 int act, int fmt_a, void* stream) {
  if (M <= 0 || N <= 0) return CLIPA_OK;
  if (!(epi == CLIPA_EPI_ACT || epi == CLIPA_EPI_ACT_PRE8 || epi == CLIPA_EPI_DACT8)) { clipa_set_error("gemm_nt_f8q: epilogue %d has no quantised-output form (CLIPA_EPI_ACT, _ACT_PRE8, _DACT8)", epi); return CLIPA_ERR_ARG; }
  if (!scale_out) { clipa_set_error("gemm_nt_f8q: scale_out is required"); return CLIPA_ERR_ARG; }
  if (epi == CLIPA_EPI_DACT8 && (!scale_a || !scale_b)) { clipa_set_error("gemm_nt_f8q: CLIPA_EPI_DACT8 needs both scale vectors"); return CLIPA_ERR_ARG; }
  if (epi == CLIPA_EPI_DACT8 && bias) { clipa_set_error("gemm_nt_f8q: CLIPA_EPI_DACT8 takes no bias (an input-gradient product)"); return CLIPA_ERR_ARG; }
  if (epi == CLIPA_EPI_DACT8 && !colsum_partial) { clipa_set_error("gemm_nt_f8q: CLIPA_EPI_DACT8 needs colsum_partial ([M / 128][N] floats)"); return CLIPA_ERR_ARG; }
  if (!f8a_eligible(M, N, K, 0)) { clipa_set_error("gemm_nt_f8q: whole-tile shapes only (M, N %% 256 == 0, K %% 256 == 0, K >= 512)"); return CLIPA_ERR_ARG; }
  return gemm_nt_f8_impl(A8, B8, scale_a, scale_b, C8, C2, bias, aux, scale_out, colsum_partial, 1, M, N, K, lda, ldb, ldc, ldaux, alpha,
                         epi, act, fmt_a, 0, stream);
}

// The GELU-backward input-gradient product from the kept e4m3 pre-activation (CLIPA_EPI_DACT8) that ALSO writes the activation
// operand of the same layer's fp8 weight gradient: X8[m, n] = e4m3(act(aux8[m, n]) * scale_a[m] / t_dev[0]), uint8 [M, ldc] - the
// bytes clipa_scale_quantize_rows_e4m3(aux8, scale_a, t_dev, act) writes (scale_a = the row scales of the gradient operand A8,
// t = clipa_rowscale_max of them and the activation's forward scales), from the epilogue that reads aux8 anyway.  Whole tiles only.
extern "C" int clipa_gemm_nt_f8_emit(const void* A8, const void* B8, const float* scale_a, const float* scale_b, void* C, void* X8,
                                     const void* aux8, const float* t_dev, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                                     int64_t ldc, int64_t ldaux, int act, int fmt_a, void* stream) {
  if (M <= 0 || N <= 0) return CLIPA_OK;
  if (!scale_a || !X8 || !aux8 || !t_dev) { clipa_set_error("gemm_nt_f8_emit: scale_a, X8, aux8 and t are required"); return CLIPA_ERR_ARG; }
  if (!f8a_eligible(M, N, K, 0)) { clipa_set_error("gemm_nt_f8_emit: whole-tile shapes only (M, N %% 256 == 0, K %% 256 == 0, K >= 512)"); return CLIPA_ERR_ARG; }
  return gemm_nt_f8_impl(A8, B8, scale_a, scale_b, C, X8, nullptr, aux8, nullptr, nullptr, 0, M, N, K, lda, ldb, ldc, ldaux, 1.0f,
                         CLIPA_EPI_DACT8, act, fmt_a, 0, stream, t_dev);
}

namespace {
int gemm_nt_f8_impl(const void* A8, const void* B8, const float* scale_a, const float* scale_b, void* C, void* C2, const float* bias,
                    const void* aux, const float* scale_out, float* colsum_partial, int outq, int64_t M, int64_t N, int64_t K,
                    int64_t lda, int64_t ldb, int64_t ldc, int64_t ldaux, float alpha, int epi, int act, int fmt_a, int fmt_b,
                    void* stream, const float* emit_t) {
  if (M <= 0 || N <= 0) return CLIPA_OK;
  if (K <= 0 || K % 16 != 0) { clipa_set_error("gemm_nt_f8: K=%ld must be a positive multiple of 16", (long)K); return CLIPA_ERR_ARG; }
  if (lda % 16 != 0 || ldb % 16 != 0) { clipa_set_error("gemm_nt_f8: lda, ldb must be multiples of 16 bytes"); return CLIPA_ERR_ARG; }
  if (N % 8 != 0 || ldc % 8 != 0) { clipa_set_error("gemm_nt_f8: N, ldc must be multiples of 8"); return CLIPA_ERR_ARG; }
  if (epi < CLIPA_EPI_NONE || epi > CLIPA_EPI_DACT8) { clipa_set_error("gemm_nt_f8: unknown epilogue %d", epi); return CLIPA_ERR_ARG; }
  // e4m3 second output / operand (the "h8" kept tensor of the engine): epilogues of the four-wave kernel only
  const int pre8 = epi == CLIPA_EPI_ACT_PRE8, aux8 = epi == CLIPA_EPI_DACT8;
  if (pre8) epi = CLIPA_EPI_ACT;
  if (aux8) epi = CLIPA_EPI_DACT;
  if ((pre8 || aux8) && !f8a_eligible(M, N, K, fmt_b)) { clipa_set_error("gemm_nt_f8: CLIPA_EPI_ACT_PRE8 / CLIPA_EPI_DACT8 need whole-tile shapes (M, N %% 256 == 0, K %% 256 == 0, K >= 512, e4m3 weights)"); return CLIPA_ERR_ARG; }
  if (pre8 && !C2) { clipa_set_error("gemm_nt_f8: CLIPA_EPI_ACT_PRE8 needs C2"); return CLIPA_ERR_ARG; }
  if ((epi == CLIPA_EPI_ADD || epi == CLIPA_EPI_DACT) && (!aux || ldaux % 8 != 0)) { clipa_set_error("gemm_nt_f8: epilogue %d needs aux with ldaux%%8==0", epi); return CLIPA_ERR_ARG; }
  if (C2 && epi != CLIPA_EPI_ACT && !emit_t) { clipa_set_error("gemm_nt_f8: C2 (pre-activation copy) goes with CLIPA_EPI_ACT only"); return CLIPA_ERR_ARG; }
  if ((fmt_a != 0 && fmt_a != 1) || (fmt_b != 0 && fmt_b != 1)) { clipa_set_error("gemm_nt_f8: formats are 0 (e4m3) or 1 (e5m2)"); return CLIPA_ERR_ARG; }
  if (256 * lda >= (1L << 30) || 256 * ldb >= (1L << 30) || 256 * ldc * 2 >= (1L << 30) || 256 * ldaux * 2 >= (1L << 30)) { clipa_set_error("gemm_nt_f8: leading dimension too large"); return CLIPA_ERR_ARG; }
  int dev = 0;
  if (int rc = current_device(&dev)) return rc;
  if (int rc = ensure_f8_attrs(dev)) return rc;
  const int num_cu = gemm_num_cu(dev);
  F8Args a;
  a.A = (const char*)A8; a.B = (const char*)B8; a.C = (char*)C; a.C2 = (char*)C2; a.bias = bias; a.aux = (const char*)aux;
  a.sa = scale_a; a.sb = scale_b;
  a.M = (int)M; a.N = (int)N; a.K = (int)K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldaux = ldaux;
  a.alpha = alpha; a.epi = epi; a.act = act; a.abl = g_abl.load(std::memory_order_relaxed);
  a.gm = nt_group_size((N + BN - 1) / BN, 256L * K);
  hipStream_t st = (hipStream_t)stream;
  // whole-tile shapes with e4m3 weights (every block GEMM of the BASELINE configurations) run on the four-wave kernel with the
  // hand-scheduled main loop (gemm_f8a.hip); bit-identical outputs.  clipa_internal_debug_set(1, .) keeps them on this file's kernel.
  if (outq || pre8 || aux8 || (g_nt_variant.load(std::memory_order_relaxed) != 1 && !(a.abl & 13) && f8a_eligible(M, N, K, fmt_b))) {
    F8AArgs b;
    b.A = a.A; b.B = a.B; b.C = a.C; b.C2 = a.C2; b.bias = a.bias; b.aux = a.aux; b.sa = a.sa; b.sb = a.sb;
    b.M = a.M; b.N = a.N; b.K = a.K; b.lda = a.lda; b.ldb = a.ldb; b.ldc = a.ldc; b.ldaux = a.ldaux;
    b.alpha = a.alpha; b.epi = a.epi; b.act = a.act; b.abl = a.abl; b.gm = a.gm; b.pre8 = pre8; b.aux8 = aux8;
    b.so = scale_out; b.cs_part = colsum_partial; b.outq = outq; b.emit_t = emit_t;
    return f8a_launch(b, fmt_a, dev, num_cu, st);
  }
  note_gemm(7);
  const long tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  const dim3 grid((unsigned)(tiles < num_cu ? tiles : num_cu)), block(NTHREADS);
#define LAUNCH_F8(E, P2)                                                                                                   \
  do {                                                                                                                     \
    if (fmt_a == 0 && fmt_b == 0) hipLaunchKernelGGL((gemm_nt_f8_kernel<0, 0, E, P2>), grid, block, F8_LDS_BYTES, st, a);      \
    else if (fmt_a == 1 && fmt_b == 0) hipLaunchKernelGGL((gemm_nt_f8_kernel<1, 0, E, P2>), grid, block, F8_LDS_BYTES, st, a); \
    else if (fmt_a == 0 && fmt_b == 1) hipLaunchKernelGGL((gemm_nt_f8_kernel<0, 1, E, P2>), grid, block, F8_LDS_BYTES, st, a); \
    else hipLaunchKernelGGL((gemm_nt_f8_kernel<1, 1, E, P2>), grid, block, F8_LDS_BYTES, st, a);                               \
  } while (0)
  if (epi == CLIPA_EPI_NONE) LAUNCH_F8(CLIPA_EPI_NONE, false);
  else if (epi == CLIPA_EPI_ACT && C2) LAUNCH_F8(CLIPA_EPI_ACT, true);
  else if (epi == CLIPA_EPI_ACT) LAUNCH_F8(CLIPA_EPI_ACT, false);
  else if (epi == CLIPA_EPI_ADD) LAUNCH_F8(CLIPA_EPI_ADD, false);
  else LAUNCH_F8(CLIPA_EPI_DACT, false);
#undef LAUNCH_F8
  return clipa_check_launch("gemm_nt_f8");
}
}  // namespace

