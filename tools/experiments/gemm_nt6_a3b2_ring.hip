// EXPERIMENT (round 2, not linked into libclipa_hip.so): gemm_nt2<bf16, M16> with an asymmetric operand ring - THREE 32 KiB slots for
// the A (activation) images and TWO for the B (weight) images, the epilogue window aliased onto the B slot the last K step has
// just finished with.  Why: every main-loop experiment of the round says that what a K step waits for is the landing time of the
// LDS-DMA issued one step earlier (profiles/r02_gemm_counted_waits.md section 4: K = 4096 loses 5 % when that DMA gets 1/8 of a step
// less).  A rows are streamed from HBM (first touch: the longest latency), B rows come from L2 / Infinity Cache.  With 160 KiB
// of LDS there is no room for a third full slot, but there is for a third A slot once the window stops owning 32 KiB: A of step
// k+2 and B of step k+1 are issued at the top of step k (B first), and the end-of-step wait is `vmcnt(4)` - A(k+2) stays in
// flight and gets two steps to land.
// RESULT (profiles/r02_gemm_a3b2_ring_ab.jsonl): bit-identical to the production kernel on the first run (six epilogues, four
// production shapes, two ragged ones) and within +-3 % of its speed everywhere (K = 4096: -1...-5 %, N = K = 1024: +1...+7 %): two
// steps of landing time for the A operand change nothing, so the A stream's HBM latency is not what the K step waits for either.
// The mirrored ring (three B slots, two A slots: clipa_gemm_nt7, profiles/r02_gemm_b3a2_ring_ab.jsonl) is bit-identical too and within
// -4...+3 %: neither operand's landing time alone is the constraint.
// Self-contained: tools/build_variant.sh nt6 tools/experiments/gemm_nt6_a3b2_ring.hip ; entry point clipa_gemm_nt6 (signature of
// clipa_gemm_nt, bf16 output, K > 64); tools/gemm_nt4_ab.py nt6 compares it with the production kernel.
#include "../gemm_common.h"
#include <mutex>

namespace clipa_gemm {
namespace {

constexpr int A_SLOT = IMG_BYTES;                  // 32 KiB: 256 rows x 128 B
constexpr int B_BASE = 3 * IMG_BYTES;              // B slots behind the three A slots
constexpr int LDS6_BYTES = 5 * IMG_BYTES;          // 160 KiB

// B3 = false: A has the three slots (clipa_gemm_nt6); B3 = true: B has them (clipa_gemm_nt7) - the weight matrix is most of the
// fabric traffic (re-streamed from Infinity Cache once per tile group), so its landing time is the other candidate.
template <int EPI, bool PRE, bool B3>
__global__ __launch_bounds__(NTHREADS) void gemm_nt6_kernel(NTArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  const int tilesN = (p.N + BN - 1) / BN;
  const int tilesM = (p.M + BM - 1) / BM;
  const unsigned ntiles = (unsigned)(tilesM * tilesN);
  const unsigned G = gridDim.x, xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
  const unsigned gx = (G - xcd + 7u) >> 3;
  const unsigned q8 = ntiles >> 3, r8 = ntiles & 7u;
  const unsigned base = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const unsigned len = q8 + (xcd < r8 ? 1u : 0u);

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
  const int nkt = (p.K + BK - 1) / BK;       // >= 2 (host)
  auto tile_origin = [&](unsigned t, int& m0, int& n0) {
    const int GM = p.gm;
    const int per = GM * tilesN;
    const int g = (int)t / per, r = (int)t - g * per;
    const int gm = min(GM, tilesM - g * GM);
    const int tn = r / gm, mm = r - tn * gm;
    m0 = (g * GM + mm) * BM;
    n0 = tn * BN;
  };
  // the operand with three slots lives at [0, 96 KiB), the other at [96, 160 KiB); `slot` indexes within the operand's region
  constexpr int A_OFF = B3 ? B_BASE : 0, B_OFF = B3 ? 0 : B_BASE;
  auto stage_a = [&](int slot, int m0, int k0) {
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A + (size_t)m0 * p.lda * 2, (unsigned)(min(BM, p.M - m0) * p.lda * 2));
    char* sA = smem + A_OFF + slot * A_SLOT;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned oob = (k0 + kel[j] >= p.K) ? 0x80000000u : 0u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(sA + (j * 8 + wave) * 1024), 16, voffA[j] | oob, k0 * 2, 0, 0);
    }
  };
  auto stage_b = [&](int slot, int n0, int k0) {
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(p.B + (size_t)n0 * p.ldb * 2, (unsigned)(min(BN, p.N - n0) * p.ldb * 2));
    char* sB = smem + B_OFF + slot * A_SLOT;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned oob = (k0 + kel[j] >= p.K) ? 0x80000000u : 0u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(sB + (j * 8 + wave) * 1024), 16, voffB[j] | oob, k0 * 2, 0, 0);
    }
  };

  if (idx >= len) return;
  unsigned it = idx;
  int m0, n0;
  tile_origin(base + it, m0, n0);
  // global K-step counter gk: A of step g lives in A slot g % 3 (sa), B in B slot g & 1
  unsigned gk = 0;
  int sa = 0;                                  // gk % 3
  auto stage_x = [&](int slot, int mt, int nt, int k0) { if (B3) stage_b(slot, nt, k0); else stage_a(slot, mt, k0); };   // three slots
  auto stage_y = [&](int slot, int mt, int nt, int k0) { if (B3) stage_a(slot, mt, k0); else stage_b(slot, nt, k0); };   // two slots
  stage_x(0, m0, n0, 0);
  stage_y(0, m0, n0, 0);
  stage_x(1, m0, n0, BK);
  RING_WAIT_ALL();
  for (;;) {
    const bool has_next = it + gx < len;
    int m1 = 0, n1 = 0;
    if (has_next) tile_origin(base + it + gx, m1, n1);

    f32x4v acc16[4][8];
#pragma unroll
    for (int bj = 0; bj < 4; ++bj)
#pragma unroll
      for (int ai = 0; ai < 8; ++ai) acc16[bj][ai] = f32x4v{0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < nkt; ++kt, ++gk) {
      const int sa1 = sa == 2 ? 0 : sa + 1, sa2 = sa1 == 2 ? 0 : sa1 + 1;
      // B of the next step first (waited for at the end of this step), then A of the step after next (left in flight)
      if (kt + 1 < nkt) stage_y((gk + 1) & 1, m0, n0, (kt + 1) * BK);
      else if (has_next) stage_y((gk + 1) & 1, m1, n1, 0);
      bool a_issued = false;
      if (kt + 2 < nkt) { stage_x(sa2, m0, n0, (kt + 2) * BK); a_issued = true; }
      else if (has_next) { stage_x(sa2, m1, n1, (kt + 2 - nkt) * BK); a_issued = true; }
      const char* sA = smem + A_OFF + (B3 ? (int)(gk & 1) : sa) * A_SLOT;
      const char* sB = smem + B_OFF + (B3 ? sa : (int)(gk & 1)) * A_SLOT;
      const int l15 = lane & 15, g4 = lane >> 4, sw16 = (l15 >> 1) & 7;
      const char* pa = sA + (wm * 128 + l15) * 128;
      const char* pb = sB + (wn * 64 + l15) * 128;
      bf16x8 ga[2][2], gb[2][4];
#pragma unroll
      for (int bj = 0; bj < 4; ++bj) gb[0][bj] = *(const bf16x8*)(pb + bj * 2048 + ((g4 ^ sw16) << 4));
#pragma unroll
      for (int a = 0; a < 2; ++a) ga[0][a] = *(const bf16x8*)(pa + a * 2048 + ((g4 ^ sw16) << 4));
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int kk = u >> 2, sb = u & 3;
        if (u < 7) {
          const int k1 = (u + 1) >> 2, s1 = (u + 1) & 3;
#pragma unroll
          for (int a = 0; a < 2; ++a) ga[(u + 1) & 1][a] = *(const bf16x8*)(pa + (2 * s1 + a) * 2048 + (((4 * k1 + g4) ^ sw16) << 4));
        }
        if (u == 1) {
#pragma unroll
          for (int bj = 0; bj < 4; ++bj) gb[1][bj] = *(const bf16x8*)(pb + bj * 2048 + (((4 + g4) ^ sw16) << 4));
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int bj = 0; bj < 4; ++bj)
#pragma unroll
          for (int a = 0; a < 2; ++a)
            acc16[bj][2 * sb + a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gb[kk][bj], ga[u & 1][a], acc16[bj][2 * sb + a], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
      sa = sa1;
      if (kt + 1 < nkt) {
        // next step's A (issued a step ago) and B (issued at the top of this step) have landed; this step's 4 A pieces need not have
        if (a_issued) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else RING_WAIT_ALL();
      }
    }

    // ---- epilogue: window = the B slot of the last K step, vectors parked in its A slot (sa was advanced: last step's = sa - 1) ----
    char* cb = smem + B_BASE + ((gk + 1) & 1) * A_SLOT;     // gk was advanced: two-slot region, slot gk & 1 holds the next tile's first step
    const int sa_last = sa == 0 ? 2 : sa - 1;
    char* park = smem + sa_last * A_SLOT;                   // three-slot region, the last step's slot
    const bool use_bias = p.bias && !(p.abl & 4);
    if (use_bias) {
      // no LDS is free during the last K step in this layout: the bias vector is fetched here (a short exposed L2 round trip)
      asm volatile("s_barrier" ::: "memory");               // every wave is done reading the last step's A slot
      if (wave == 0) {
        const __amdgpu_buffer_rsrc_t rsBias = make_rsrc(p.bias + n0, (unsigned)(max(0, min(BN, p.N - n0)) * 4));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsBias, LDS_PTR(park), 16, (unsigned)(lane * 16), 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float4 bias4[4];
    WinOut o;
    o.C = p.C; o.C2 = p.C2; o.aux = p.aux; o.ldc = p.ldc; o.ldaux = p.ldaux;
    o.M = p.M; o.N = p.N; o.m0 = m0; o.n0 = n0; o.act = p.act; o.abl = p.abl;
    window_epilogue<EPI, PRE, false>(cb, o, tid, wm, wn,
      [&](int) {
        if (use_bias) lds_read4_f4(bias4, park + (wn * 64 + 4 * (lane >> 4)) * 4);
        else {
#pragma unroll
          for (int bj = 0; bj < 4; ++bj) bias4[bj] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      },
      [&](int ai, int bj) {
        const float4 b4 = bias4[bj];
        u32x2 w;
        w[0] = pack2bf(acc16[bj][ai][0] * p.alpha + b4.x, acc16[bj][ai][1] * p.alpha + b4.y);
        w[1] = pack2bf(acc16[bj][ai][2] * p.alpha + b4.z, acc16[bj][ai][3] * p.alpha + b4.w);
        return w;
      });
    if (!has_next) break;
    // the next tile's first step (A: two steps old, B: one) has landed; its second A slab (4 pieces, issued after them) and this
    // tile's stores need not have.  (With a bias the vmcnt(0) above has already drained the slabs.)
    if (use_bias) RING_WAIT_AFTER_EPILOGUE(win_stores(PRE));
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(win_stores(PRE) + 4) : "memory");
    it += gx;
    m0 = m1;
    n0 = n1;
  }
}

std::once_flag g_nt6_once[2][MAX_DEVICES];
int g_nt6_rc[2][MAX_DEVICES];

}  // namespace
}  // namespace clipa_gemm

using namespace clipa_gemm;

template <bool B3>
static int launch_nt6(const void* A, const void* B, void* C, void* C2, const float* bias, const void* aux, int64_t M, int64_t N,
                              int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldaux, float alpha, int epi, int act, int out_f32,
                              void* stream) {
  if (M <= 0 || N <= 0) return CLIPA_OK;
  if (out_f32 || K <= BK || K % 8 || N % 8 || ldc % 8 || lda % 8 || ldb % 8 || epi < CLIPA_EPI_NONE || epi > CLIPA_EPI_DACT ||
      ((epi == CLIPA_EPI_ADD || epi == CLIPA_EPI_DACT) && (!aux || ldaux % 8)) || (C2 && epi != CLIPA_EPI_ACT)) {
    clipa_set_error("gemm_nt6 (experiment): unsupported arguments");
    return CLIPA_ERR_ARG;
  }
  int dev = 0;
  if (int rc = current_device(&dev)) return rc;
  std::call_once(g_nt6_once[B3][dev], [dev]() {
    int rc = 0;
    const void* v[5] = {(const void*)gemm_nt6_kernel<CLIPA_EPI_NONE, false, B3>, (const void*)gemm_nt6_kernel<CLIPA_EPI_ACT, false, B3>,
                        (const void*)gemm_nt6_kernel<CLIPA_EPI_ACT, true, B3>, (const void*)gemm_nt6_kernel<CLIPA_EPI_ADD, false, B3>,
                        (const void*)gemm_nt6_kernel<CLIPA_EPI_DACT, false, B3>};
    for (int i = 0; i < 5; ++i)
      if (hipFuncSetAttribute(v[i], hipFuncAttributeMaxDynamicSharedMemorySize, LDS6_BYTES) != hipSuccess) rc = CLIPA_ERR_LAUNCH;
    g_nt6_rc[B3][dev] = rc;
  });
  if (g_nt6_rc[B3][dev]) return g_nt6_rc[B3][dev];
  NTArgs a;
  a.A = (const char*)A; a.B = (const char*)B; a.C = (char*)C; a.C2 = (char*)C2; a.bias = bias; a.aux = (const char*)aux;
  a.M = (int)M; a.N = (int)N; a.K = (int)K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldaux = ldaux;
  a.alpha = alpha; a.epi = epi; a.act = act; a.abl = 0;
  a.gm = nt_group_size((N + BN - 1) / BN, 256L * K * 2);
  const long tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  const int num_cu = gemm_num_cu(dev);
  const unsigned grid = (unsigned)(tiles < num_cu ? tiles : num_cu);
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH_NT6(E, P2) hipLaunchKernelGGL((gemm_nt6_kernel<E, P2, B3>), dim3(grid), dim3(NTHREADS), LDS6_BYTES, st, a)
  if (epi == CLIPA_EPI_NONE) LAUNCH_NT6(CLIPA_EPI_NONE, false);
  else if (epi == CLIPA_EPI_ACT && C2) LAUNCH_NT6(CLIPA_EPI_ACT, true);
  else if (epi == CLIPA_EPI_ACT) LAUNCH_NT6(CLIPA_EPI_ACT, false);
  else if (epi == CLIPA_EPI_ADD) LAUNCH_NT6(CLIPA_EPI_ADD, false);
  else LAUNCH_NT6(CLIPA_EPI_DACT, false);
#undef LAUNCH_NT6
  return clipa_check_launch("gemm_nt6<bf16>");
}

#define NT6_ARGS const void* A, const void* B, void* C, void* C2, const float* bias, const void* aux, int64_t M, int64_t N, int64_t K, int64_t lda, \
                 int64_t ldb, int64_t ldc, int64_t ldaux, float alpha, int epi, int act, int out_f32, void* stream
extern "C" int clipa_gemm_nt6(NT6_ARGS) { return launch_nt6<false>(A, B, C, C2, bias, aux, M, N, K, lda, ldb, ldc, ldaux, alpha, epi, act, out_f32, stream); }
extern "C" int clipa_gemm_nt7(NT6_ARGS) { return launch_nt6<true>(A, B, C, C2, bias, aux, M, N, K, lda, ldb, ldc, ldaux, alpha, epi, act, out_f32, stream); }
