// EXPERIMENT (round 2, not linked into libclipa_hip.so): the production 256x256x64 tile, ring, swizzle and window epilogue on FOUR
// waves of up to 512 registers (one wave per SIMD) instead of eight of 256 - the structure of hipBLASLt's hand-written kernel
// for these shapes (Custom_..._MT256x256x64_MI16x16x1: 256 threads, 2 x 64 KiB LDS ring, accumulators in AGPRs), which runs the
// plain GEMMs at 1190-1440 TF/s where gemm_nt2 does 990-1160 (profiles/r02_vendor_gemm_comparison.jsonl).
//
// Result: bit-identical to gemm_nt2<bf16> on the first GPU run (tests/test_kernels_gpu.py -k gemm_nt with the variant switch, and
// tools/gemm_lib_ab-style comparison of whole outputs for all six epilogues on four shapes) and 0.72-0.83x its speed
// (profiles/r02_gemm_four_wave_variant_ab.jsonl, profiles/r02_gemm_counted_waits.md section 5).
//
// What it took to get hipcc to emit it without scratch (scratch loads / stores count on vmcnt and would break the hand-counted
// waits of window_epilogue):
//   * accumulators cleared by volatile `v_accvgpr_write_b32 a, 0` - a plain `= 0` is hoisted by the loop rotation above the
//     previous tile's epilogue, and all 256 live accumulators are then shuffled through the arch VGPRs;
//   * accumulators read back in the pack callback by volatile `v_accvgpr_read_b32` ("a" operands) - otherwise the scheduler
//     pulls all 256 reads to the top of the epilogue;
//   * `__builtin_amdgcn_sched_barrier(0)` around each 16-MFMA group - otherwise the fragment reads of the next sub-step are sunk
//     behind the MFMAs and the double-buffered fragments collapse into one register set (every sub-step then waits a full LDS
//     latency: this alone is 0.76x -> 0.80x);
//   * the 24 per-lane DMA offsets recomputed per stage() (LEAN, as in gemm_nt_f8's activation-backward kernel).  Keeping them in
//     registers AND issuing the 16 LDS-DMA of a step between the MFMA halves (so that they are not a ~800-cycle bubble of the
//     matrix pipe at the top of every K step: with one wave per SIMD nothing else covers them) pushed the allocator back into
//     scratch - the open end of this experiment.  The missing 20 % is instruction scheduling that needs assembly-level control
//     (or sched_group_barrier for every slot), not a different tiling.
//
// To build it again: (1) generalise window_epilogue (gemm_common.h) with template parameters NTHR (threads: chunk index
// c = j * NTHR + tid, NCH = 2048 / NTHR chunks per thread and pass, aux registers NCH (ROLL) or 2 NCH, batch wait 2 NCH - 1,
// rolling waits NCH - 1 + j / 2 (NCH - 1) / 2 (NCH - 1) - j) and NBJ (16-column blocks per wave: nl = wn * 16 NBJ + 16 bj + 4 g4),
// win_stores(pre, nthr) = min(63, 4 passes x NCH x (pre ? 2 : 1)); (2) paste the kernel below into gemm_nt.hip next to
// gemm_nt2_kernel, set the LDS attribute for its five instantiations and launch it with 256 threads.
#include "../gemm_common.h"

namespace clipa_gemm {
namespace {

constexpr int CBUF_OFF4 = 2 * STAGE_BYTES;
constexpr int NT4_THREADS = 256;

template <int EPI, bool PRE>
__global__ __launch_bounds__(NT4_THREADS) void gemm_nt4_kernel(NTArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;          // wave tile 128 (m) x 128 (n)

  const int tilesN = (p.N + BN - 1) / BN;
  const int tilesM = (p.M + BM - 1) / BM;
  const unsigned ntiles = (unsigned)(tilesM * tilesN);
  const unsigned G = gridDim.x, xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
  const unsigned gx = (G - xcd + 7u) >> 3;
  const unsigned q8 = ntiles >> 3, r8 = ntiles & 7u;
  const unsigned base = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const unsigned len = q8 + (xcd < r8 ? 1u : 0u);
  const int nkt = (p.K + BK - 1) / BK;
  auto tile_origin = [&](unsigned t, int& m0, int& n0) {
    const int GM = p.gm;
    const int per = GM * tilesN;
    const int g = (int)t / per, r = (int)t - g * per;
    const int gm = min(GM, tilesM - g * GM);
    const int tn = r / gm, mm = r - tn * gm;
    m0 = (g * GM + mm) * BM;
    n0 = tn * BN;
  };
  // DMA piece pc (1 KiB) = image rows 8pc .. 8pc+7; wave w moves pieces 4j + w of A and of B (offsets recomputed: see above)
  auto stage = [&](int buf, int m0, int n0, int k0) {
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A + (size_t)m0 * p.lda * 2, (unsigned)(min(BM, p.M - m0) * p.lda * 2));
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(p.B + (size_t)n0 * p.ldb * 2, (unsigned)(min(BN, p.N - n0) * p.ldb * 2));
    char* sA = smem + buf * STAGE_BYTES;
    char* sB = sA + IMG_BYTES;
    int l = lane;
    asm volatile("" : "+v"(l));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int pc = j * 4 + wave;
      const int row = pc * 8 + (l >> 3);
      const int chunk = (l & 7) ^ ((row >> 1) & 7);
      const unsigned oob = (k0 + chunk * 8 >= p.K) ? 0x80000000u : 0u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(sA + pc * 1024), 16, (unsigned)(row * p.lda * 2 + chunk * 16) | oob, k0 * 2, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(sB + pc * 1024), 16, (unsigned)(row * p.ldb * 2 + chunk * 16) | oob, k0 * 2, 0, 0);
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

    f32x4v acc16[8][8];      // [n block of 16][m block of 16]: 256 AGPRs
#pragma unroll
    for (int bj = 0; bj < 8; ++bj)
#pragma unroll
      for (int ai = 0; ai < 8; ++ai) {
        float z0, z1, z2, z3;
        asm volatile("v_accvgpr_write_b32 %0, 0\n\tv_accvgpr_write_b32 %1, 0\n\tv_accvgpr_write_b32 %2, 0\n\tv_accvgpr_write_b32 %3, 0"
                     : "=a"(z0), "=a"(z1), "=a"(z2), "=a"(z3));
        acc16[bj][ai] = f32x4v{z0, z1, z2, z3};
      }

    for (int kt = 0; kt < nkt; ++kt, ++gk) {
      if (kt + 1 < nkt) stage((gk + 1) & 1, m0, n0, (kt + 1) * BK);
      else {
        if (p.bias && wave == 0) {
          const __amdgpu_buffer_rsrc_t rsBias = make_rsrc(p.bias + n0, (unsigned)(max(0, min(BN, p.N - n0)) * 4));
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsBias, LDS_PTR(smem + CBUF_OFF4), 16, (unsigned)(lane * 16), 0, 0, 0);
        }
        if (has_next) stage((gk + 1) & 1, m1, n1, 0);
      }
      const char* sA = smem + (gk & 1) * STAGE_BYTES;
      const char* sB = sA + IMG_BYTES;
      // 8 sub-steps per K tile: (kk, s) = 32-wide k-step kk, A blocks 2s and 2s+1 against the eight B blocks of kk (16 MFMAs)
      const int l15 = lane & 15, g4 = lane >> 4, sw16 = (l15 >> 1) & 7;
      const char* pa = sA + (wm * 128 + l15) * 128;
      const char* pb = sB + (wn * 128 + l15) * 128;
      bf16x8 ga[2][2], gb[2][8];
#pragma unroll
      for (int bj = 0; bj < 8; ++bj) gb[0][bj] = *(const bf16x8*)(pb + bj * 2048 + ((g4 ^ sw16) << 4));
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
          for (int bj = 0; bj < 8; ++bj) gb[1][bj] = *(const bf16x8*)(pb + bj * 2048 + (((4 + g4) ^ sw16) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);      // the reads above are ISSUED before this sub-step's MFMAs
#pragma unroll
        for (int bj = 0; bj < 8; ++bj)
#pragma unroll
          for (int a = 0; a < 2; ++a)
            acc16[bj][2 * sb + a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gb[kk][bj], ga[u & 1][a], acc16[bj][2 * sb + a], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (kt + 1 < nkt) RING_WAIT_ALL();
    }

    // ---- epilogue of tile (m0, n0) through the shared window routine (generalised: 256 threads, 8 n-blocks per wave) ----
    char* cb = smem + CBUF_OFF4;
    const bool use_bias = p.bias && !(p.abl & 4);
    char* park = smem + ((gk + 1) & 1) * STAGE_BYTES;
    if (use_bias) {
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      park_vectors(cb, park, tid, 1024);
    }
    float4 bias4[8];
    WinOut o;
    o.C = p.C; o.C2 = p.C2; o.aux = p.aux; o.ldc = p.ldc; o.ldaux = p.ldaux;
    o.M = p.M; o.N = p.N; o.m0 = m0; o.n0 = n0; o.act = p.act; o.abl = p.abl;
    window_epilogue<EPI, PRE, /*ROLL*/ true, NT4_THREADS, /*NBJ*/ 8>(cb, o, tid, wm, wn,
      [&](int) {
        if (use_bias) {
          float4 lo[4], hi4[4];
          lds_read4_f4(lo, park + (wn * 128 + 4 * (lane >> 4)) * 4);
          lds_read4_f4(hi4, park + (wn * 128 + 64 + 4 * (lane >> 4)) * 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) { bias4[i] = lo[i]; bias4[4 + i] = hi4[i]; }
        } else {
#pragma unroll
          for (int bj = 0; bj < 8; ++bj) bias4[bj] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      },
      [&](int ai, int bj) {
        float c0, c1, c2, c3;
        asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
                     : "=v"(c0), "=v"(c1), "=v"(c2), "=v"(c3)
                     : "a"(acc16[bj][ai][0]), "a"(acc16[bj][ai][1]), "a"(acc16[bj][ai][2]), "a"(acc16[bj][ai][3]));
        const float4 b4 = bias4[bj];
        u32x2 w;
        w[0] = pack2bf(c0 * p.alpha + b4.x, c1 * p.alpha + b4.y);
        w[1] = pack2bf(c2 * p.alpha + b4.z, c3 * p.alpha + b4.w);
        return w;
      });
    if (!has_next) break;
    RING_WAIT_AFTER_EPILOGUE(win_stores(PRE, NT4_THREADS));
    it += gx;
    m0 = m1;
    n0 = n1;
  }
}

}  // namespace
}  // namespace clipa_gemm
