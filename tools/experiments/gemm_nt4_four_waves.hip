// EXPERIMENT (round 2, not linked into libclipa_hip.so): the production 256x256x64 tile, ring, swizzle and window epilogue on FOUR
// waves of up to 512 registers (one wave per SIMD) instead of eight of 256 - the structure of hipBLASLt's hand-written kernel
// for these shapes (Custom_..._MT256x256x64_MI16x16x1: 256 threads, 2 x 64 KiB LDS ring, accumulators in AGPRs), which runs the
// plain GEMMs at 1190-1440 TF/s where gemm_nt2 does 990-1160 (profiles/r02_vendor_gemm_comparison.jsonl).
//
// Result: bit-identical to gemm_nt2<bf16> on the first GPU run (tests/test_kernels_gpu.py -k gemm_nt with the variant switch, and
// tools/gemm_lib_ab-style comparison of whole outputs for all six epilogues on four shapes) and 0.72-0.83x its speed
// (profiles/r02_gemm_four_wave_variant_ab.jsonl, profiles/r02_gemm_counted_waits.md section 5).
//
// What it took to get hipcc to emit it without scratch (scratch loads / stores sit on vmcnt too: they cannot make a hand-counted
// wait of window_epilogue too short, only stricter, but they are memory round trips in the hot path):
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
// [first form of the experiment; this file now holds the second form, see HAND-ORDERED K STEP below]
// To build the first form again: (1) generalise window_epilogue (gemm_common.h) with template parameters NTHR (threads: chunk index
// c = j * NTHR + tid, NCH = 2048 / NTHR chunks per thread and pass, aux registers NCH (ROLL) or 2 NCH, batch wait 2 NCH - 1,
// rolling waits NCH - 1 + j / 2 (NCH - 1) / 2 (NCH - 1) - j) and NBJ (16-column blocks per wave: nl = wn * 16 NBJ + 16 bj + 4 g4),
// win_stores(pre, nthr) = min(63, 4 passes x NCH x (pre ? 2 : 1)); (2) paste the kernel below into gemm_nt.hip next to
// gemm_nt2_kernel, set the LDS attribute for its five instantiations and launch it with 256 threads.
#include "../gemm_common.h"
#include <mutex>

// RESULT of the second form (profiles/r02_gemm_four_wave_hand_ordered_ab.jsonl): bit-identical to production for all six epilogues
// on four production shapes and two ragged ones, and 0.58-0.73x its speed - SLOWER than the compiler-scheduled first form: with
// every statement fenced, hipcc waits for each of the first eight B fragments two MFMA slots after issuing its read (eight LDS
// latencies in a row at the top of every K step instead of one).  The next attempt should open the step with all nine reads of
// the first MFMA row back to back, and needs the operands of step k+1's first row in registers BEFORE the barrier that ends step
// k - i.e. a ring with a third slot, or the barrier moved ahead of the last MFMA row.
//
// HAND-ORDERED K STEP (second form): every statement of the K step is fenced with __builtin_amdgcn_sched_barrier(0), so hipcc
// emits the instructions in exactly the source order and only adds the waits.  All 32 fragments of a K step are single-buffered
// in 128 arch VGPRs (the 256 accumulators live in AGPRs, so there is room): the step opens with 3 reads, then every MFMA slot
// of the first k-half carries one fragment read (in the order the MFMAs need them, the second k-half's fragments behind the
// first's) or one LDS-DMA pair of the next K step, and the second k-half is a pure stream of 64 MFMAs.
// Self-contained: build with tools/build_variant.sh (extra source), entry point clipa_gemm_nt4 (same signature as clipa_gemm_nt,
// bf16 output only); tools/gemm_nt4_ab.py compares it with the production kernel bit for bit and times both.

namespace clipa_gemm {
namespace {

constexpr int CBUF_OFF4 = 2 * STAGE_BYTES;
constexpr int LDS4_BYTES = CBUF_OFF4 + 64 * 512;
constexpr int NT4_THREADS = 256;
constexpr int win_stores4(bool pre) { return pre ? 63 : 32; }   // 4 passes x 8 chunks (x 2); the counter holds at most 63

template <int EPI, bool PRE, bool ROLL, int NTHR, int NBJ, typename Prepare, typename Pack>
__device__ __forceinline__ void window_epilogue_g(char* cb, const WinOut& o, int tid, int wm, int wn, Prepare&& prepare, Pack&& pack) {
  constexpr bool HAS_AUX = EPI == CLIPA_EPI_ADD || EPI == CLIPA_EPI_DACT;
  constexpr int NCH = 2048 / NTHR;          // 16-byte chunks per thread and pass
  static_assert(!(PRE && HAS_AUX), "the pre-activation copy goes with the activation epilogue");
  // address arithmetic is redone per tile from an opaque copy of the thread id: hoisted out of the persistent tile loop
  // it would sit in ~40 registers across the main loop
  int tid_e = tid;
  asm volatile("" : "+v"(tid_e));
  const int lane_e = tid_e & 63, g4 = lane_e >> 4, l15 = lane_e & 15;
  // output descriptors: rows past M and (by the offset's top bit) columns past N fall outside and are dropped
  const int rows_t = min(BM, o.M - o.m0), cols_t = min(BN, o.N - o.n0);
  const unsigned c_bytes = (unsigned)(((size_t)(rows_t - 1) * o.ldc + cols_t) * 2);
  const __amdgpu_buffer_rsrc_t rsC = make_rsrc(o.C + ((size_t)o.m0 * o.ldc + o.n0) * 2, c_bytes);
  const __amdgpu_buffer_rsrc_t rsC2 = make_rsrc(PRE ? o.C2 + ((size_t)o.m0 * o.ldc + o.n0) * 2 : o.C, PRE ? c_bytes : 0u);
  // this thread's chunks of a pass: chunk c = j*512 + tid -> row c>>5 (0..63), 16-B column c&31.
  // aux (residual / pre-activation) chunks are fetched two passes at a time, ahead of their use.
  u32x4 av[ROLL ? NCH : 2 * NCH];
  auto fetch_one = [&](u32x4& dst, int pass, int j) {
    const int c = j * NTHR + tid_e;
    // clamped inside the matrix: the value of an out-of-range chunk is never stored
    const int m = min(o.m0 + pass * 64 + (c >> 5), o.M - 1), n = min(o.n0 + (c & 31) * 8, o.N - 8);
    const char* ap = o.aux + ((size_t)m * o.ldaux + n) * 2;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ap) : "memory");
  };
  auto fetch_aux = [&](int pass0) {
    if constexpr (HAS_AUX) {
#pragma unroll
      for (int i = 0; i < (ROLL ? NCH : 2 * NCH); ++i) fetch_one(av[i], pass0 + i / NCH, i % NCH);
    }
  };
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    if (ROLL ? pass == 0 : (pass & 1) == 0) fetch_aux(pass);
    WG_BARRIER_LDS();   // readers of the previous pass (pass 0: of the bias / scale vectors) are done with the window
    if (wm == (pass >> 1)) {
      prepare(pass);
      // 16x16 blocks: lane holds features nl..nl+3 (nl = 16 bj + 4 (lane>>4)) of row 16 a2 + (lane & 15)
#pragma unroll
      for (int a2 = 0; a2 < 4; ++a2) {
        const int ai = 4 * (pass & 1) + a2;
        const int row = a2 * 16 + l15;
#pragma unroll
        for (int bj = 0; bj < NBJ; ++bj) {
          const int nl = wn * (16 * NBJ) + bj * 16 + 4 * g4;
          const u32x2 w = pack(ai, bj);
          const unsigned wa = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(cb + row * 512 + ((((nl >> 3) ^ row) & 31) << 4) + (nl & 7) * 2);
          asm volatile("ds_write_b64 %0, %1" :: "v"(wa), "v"(w) : "memory");
        }
      }
    }
    WG_BARRIER_LDS();
    // the activation-backward epilogue takes the window two chunks at a time (8 fewer live registers: no scratch)
    constexpr int CG = EPI == CLIPA_EPI_NONE ? 4 : 2;
#pragma unroll
    for (int g = 0; g < NCH / CG; ++g) {
      u32x4 cv[CG];
      unsigned a[CG];
#pragma unroll
      for (int jj = 0; jj < CG; ++jj) {
        const int c = (g * CG + jj) * NTHR + tid_e;
        const int row = c >> 5, cc = c & 31;
        a[jj] = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(cb + row * 512 + (((cc ^ row) & 31) << 4));
      }
      if constexpr (CG == 4)
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(cv[0]), "=&v"(cv[1]), "=&v"(cv[CG - 2]), "=&v"(cv[CG - 1])
                     : "v"(a[0]), "v"(a[1]), "v"(a[CG - 2]), "v"(a[CG - 1])
                     : "memory");
      else
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(cv[0]), "=&v"(cv[1])
                     : "v"(a[0]), "v"(a[1])
                     : "memory");
#pragma unroll
      for (int jj = 0; jj < CG; ++jj) {
        const int j = g * CG + jj;
        const int c = j * NTHR + tid_e;
        const int row = c >> 5, cc = c & 31;
        unsigned vo = (unsigned)(((size_t)(pass * 64 + row) * o.ldc + cc * 8) * 2);
        if (cc * 8 >= cols_t || ((o.abl & 1) && cv[jj][0] != 0x12345u)) vo |= 0x80000000u;
        u32x4 v = cv[jj];
        if constexpr (PRE) __builtin_amdgcn_raw_buffer_store_b128(v, rsC2, (int)vo, 0, 0);
        if constexpr (EPI != CLIPA_EPI_NONE) {
          u32x4& aj = av[ROLL ? j : (pass & 1) * NCH + j];
          // younger operations than chunk j's aux load: batches of 2 NCH -> 2 NCH - 1; ROLL: NCH - 1 + j (pass 0), 2 (NCH - 1),
          // 2 (NCH - 1) - j (last pass)
          if constexpr (HAS_AUX) {
            const int younger = !ROLL ? 2 * NCH - 1 : pass == 0 ? NCH - 1 + j : pass == 3 ? 2 * (NCH - 1) - j : 2 * (NCH - 1);
            switch (younger) {   // pass and j are unrolled constants: one case survives
              case 3: asm volatile("s_waitcnt vmcnt(3)" : "+v"(aj) :: "memory"); break;
              case 4: asm volatile("s_waitcnt vmcnt(4)" : "+v"(aj) :: "memory"); break;
              case 5: asm volatile("s_waitcnt vmcnt(5)" : "+v"(aj) :: "memory"); break;
              case 6: asm volatile("s_waitcnt vmcnt(6)" : "+v"(aj) :: "memory"); break;
              case 7: asm volatile("s_waitcnt vmcnt(7)" : "+v"(aj) :: "memory"); break;
              case 8: asm volatile("s_waitcnt vmcnt(8)" : "+v"(aj) :: "memory"); break;
              case 9: asm volatile("s_waitcnt vmcnt(9)" : "+v"(aj) :: "memory"); break;
              case 10: asm volatile("s_waitcnt vmcnt(10)" : "+v"(aj) :: "memory"); break;
              case 11: asm volatile("s_waitcnt vmcnt(11)" : "+v"(aj) :: "memory"); break;
              case 12: asm volatile("s_waitcnt vmcnt(12)" : "+v"(aj) :: "memory"); break;
              case 13: asm volatile("s_waitcnt vmcnt(13)" : "+v"(aj) :: "memory"); break;
              case 14: asm volatile("s_waitcnt vmcnt(14)" : "+v"(aj) :: "memory"); break;
              default: asm volatile("s_waitcnt vmcnt(15)" : "+v"(aj) :: "memory"); break;
            }
          } else aj = u32x4{0, 0, 0, 0};
          if (o.act == ACT_GELU_ERF) v = epi_chunk<ACT_GELU_ERF>(EPI, v, aj);
          else if (o.act == ACT_GELU_TANH) v = epi_chunk<ACT_GELU_TANH>(EPI, v, aj);
          else v = epi_chunk<ACT_QUICK_GELU>(EPI, v, aj);
        }
        __builtin_amdgcn_raw_buffer_store_b128(v, rsC, (int)vo, 0, 0);
        if constexpr (HAS_AUX && ROLL) {
          if (pass < 3) fetch_one(av[j], pass + 1, j);
        }
      }
    }
  }
}


#define FENCE() __builtin_amdgcn_sched_barrier(0)

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
  // DMA piece pc (1 KiB) = image rows 8pc .. 8pc+7; wave w moves pieces 4j + w of A and of B
  unsigned voffA[8], voffB[8];
  int kel[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = (j * 4 + wave) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    voffA[j] = (unsigned)(row * p.lda * 2 + chunk * 16);
    voffB[j] = (unsigned)(row * p.ldb * 2 + chunk * 16);
    kel[j] = chunk * 8;
  }
  auto stage_piece = [&](const __amdgpu_buffer_rsrc_t rsA, const __amdgpu_buffer_rsrc_t rsB, int buf, int k0, int j) {
    char* sA = smem + buf * STAGE_BYTES;
    char* sB = sA + IMG_BYTES;
    const int pc = j * 4 + wave;
    const unsigned oob = (k0 + kel[j] >= p.K) ? 0x80000000u : 0u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(sA + pc * 1024), 16, voffA[j] | oob, k0 * 2, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(sB + pc * 1024), 16, voffB[j] | oob, k0 * 2, 0, 0);
  };

  if (idx >= len) return;
  unsigned it = idx;
  int m0, n0;
  tile_origin(base + it, m0, n0);
  {
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A + (size_t)m0 * p.lda * 2, (unsigned)(min(BM, p.M - m0) * p.lda * 2));
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(p.B + (size_t)n0 * p.ldb * 2, (unsigned)(min(BN, p.N - n0) * p.ldb * 2));
#pragma unroll
    for (int j = 0; j < 8; ++j) stage_piece(rsA, rsB, 0, 0, j);
  }
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
      const bool last = kt + 1 >= nkt;
      const bool do_stage = !last || has_next;
      const int sm = last ? m1 : m0, sn = last ? n1 : n0, sk = last ? 0 : (kt + 1) * BK;
      const __amdgpu_buffer_rsrc_t srA = make_rsrc(p.A + (size_t)sm * p.lda * 2, (unsigned)(min(BM, p.M - sm) * p.lda * 2));
      const __amdgpu_buffer_rsrc_t srB = make_rsrc(p.B + (size_t)sn * p.ldb * 2, (unsigned)(min(BN, p.N - sn) * p.ldb * 2));
      if (last && p.bias && wave == 0) {
        const __amdgpu_buffer_rsrc_t rsBias = make_rsrc(p.bias + n0, (unsigned)(max(0, min(BN, p.N - n0)) * 4));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsBias, LDS_PTR(smem + CBUF_OFF4), 16, (unsigned)(lane * 16), 0, 0, 0);
      }
      const char* sA = smem + (gk & 1) * STAGE_BYTES;
      const char* sB = sA + IMG_BYTES;
      const int l15 = lane & 15, g4 = lane >> 4, sw16 = (l15 >> 1) & 7;
      // fragment (k-half kk, block b) of an image: row 16 b + l15, 16-byte chunk (4 kk + g4) ^ swizzle
      const char* pa0 = sA + (wm * 128 + l15) * 128 + ((g4 ^ sw16) << 4);
      const char* pa1 = sA + (wm * 128 + l15) * 128 + (((4 + g4) ^ sw16) << 4);
      const char* pb0 = sB + (wn * 128 + l15) * 128 + ((g4 ^ sw16) << 4);
      const char* pb1 = sB + (wn * 128 + l15) * 128 + (((4 + g4) ^ sw16) << 4);
      bf16x8 fa[2][8], fb[2][8];
      // read queue, in the order the MFMAs below need the fragments: A0 B0 B1 | B2..B7 A1..A7 (k-half 0) | k-half 1 likewise
      auto rd = [&](int q) {
        const int kk = q >> 4, r = q & 15;
        const bool is_a = (r == 0) || (r >= 9);
        const int b = r == 0 ? 0 : (r <= 8 ? r - 1 : r - 8);
        if (is_a) fa[kk][b] = *(const bf16x8*)((kk ? pa1 : pa0) + b * 2048);
        else fb[kk][b] = *(const bf16x8*)((kk ? pb1 : pb0) + b * 2048);
      };
      FENCE();
      rd(0); rd(1); rd(2);
      FENCE();
      int q = 3;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int ai = 0; ai < 8; ++ai)
#pragma unroll
          for (int bj = 0; bj < 8; ++bj) {
            const int slot = kk * 64 + ai * 8 + bj;
            // slots 0..5 feed B2..B7 (needed by the very next MFMAs); afterwards one queue entry every second slot, and one
            // LDS-DMA pair of the next K step in each of the slots 9, 13, .., 37 in between
            if (q < 32 && (slot < 6 || (slot & 1) == 0)) { rd(q); ++q; FENCE(); }
            if (do_stage && slot >= 9 && slot <= 37 && ((slot - 9) & 3) == 0) { stage_piece(srA, srB, (gk + 1) & 1, sk, (slot - 9) >> 2); FENCE(); }
            acc16[bj][ai] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][bj], fa[kk][ai], acc16[bj][ai], 0, 0, 0);
            FENCE();
          }
      if (kt + 1 < nkt) RING_WAIT_ALL();
    }

    // ---- epilogue of tile (m0, n0) through the window routine (256 threads, 8 n-blocks per wave) ----
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
    window_epilogue_g<EPI, PRE, /*ROLL*/ true, NT4_THREADS, /*NBJ*/ 8>(cb, o, tid, wm, wn,
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
    RING_WAIT_AFTER_EPILOGUE(win_stores4(PRE));
    it += gx;
    m0 = m1;
    n0 = n1;
  }
}
#undef FENCE

std::once_flag g_nt4_once[MAX_DEVICES];
int g_nt4_rc[MAX_DEVICES];

}  // namespace
}  // namespace clipa_gemm

using namespace clipa_gemm;

// same arguments as clipa_gemm_nt (include/clipa_hip.h); bf16 output only
extern "C" int clipa_gemm_nt4(const void* A, const void* B, void* C, void* C2, const float* bias, const void* aux, int64_t M, int64_t N,
                              int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldaux, float alpha, int epi, int act, int out_f32,
                              void* stream) {
  if (M <= 0 || N <= 0) return CLIPA_OK;
  if (out_f32 || K <= 0 || K % 8 || N % 8 || ldc % 8 || lda % 8 || ldb % 8 || epi < CLIPA_EPI_NONE || epi > CLIPA_EPI_DACT ||
      ((epi == CLIPA_EPI_ADD || epi == CLIPA_EPI_DACT) && (!aux || ldaux % 8)) || (C2 && epi != CLIPA_EPI_ACT)) {
    clipa_set_error("gemm_nt4 (experiment): unsupported arguments");
    return CLIPA_ERR_ARG;
  }
  int dev = 0;
  if (int rc = current_device(&dev)) return rc;
  std::call_once(g_nt4_once[dev], [dev]() {
    int rc = 0;
    const void* v4[5] = {(const void*)gemm_nt4_kernel<CLIPA_EPI_NONE, false>, (const void*)gemm_nt4_kernel<CLIPA_EPI_ACT, false>,
                         (const void*)gemm_nt4_kernel<CLIPA_EPI_ACT, true>, (const void*)gemm_nt4_kernel<CLIPA_EPI_ADD, false>,
                         (const void*)gemm_nt4_kernel<CLIPA_EPI_DACT, false>};
    for (int i = 0; i < 5; ++i)
      if (hipFuncSetAttribute(v4[i], hipFuncAttributeMaxDynamicSharedMemorySize, LDS4_BYTES) != hipSuccess) rc = CLIPA_ERR_LAUNCH;
    g_nt4_rc[dev] = rc;
  });
  if (g_nt4_rc[dev]) return g_nt4_rc[dev];
  NTArgs a;
  a.A = (const char*)A; a.B = (const char*)B; a.C = (char*)C; a.C2 = (char*)C2; a.bias = bias; a.aux = (const char*)aux;
  a.M = (int)M; a.N = (int)N; a.K = (int)K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldaux = ldaux;
  a.alpha = alpha; a.epi = epi; a.act = act; a.abl = 0;
  a.gm = nt_group_size((N + BN - 1) / BN, 256L * K * 2);
  const long tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  const int num_cu = gemm_num_cu(dev);
  const unsigned grid = (unsigned)(tiles < num_cu ? tiles : num_cu);
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH_NT4(E, P2) hipLaunchKernelGGL((gemm_nt4_kernel<E, P2>), dim3(grid), dim3(NT4_THREADS), LDS4_BYTES, st, a)
  if (epi == CLIPA_EPI_NONE) LAUNCH_NT4(CLIPA_EPI_NONE, false);
  else if (epi == CLIPA_EPI_ACT && C2) LAUNCH_NT4(CLIPA_EPI_ACT, true);
  else if (epi == CLIPA_EPI_ACT) LAUNCH_NT4(CLIPA_EPI_ACT, false);
  else if (epi == CLIPA_EPI_ADD) LAUNCH_NT4(CLIPA_EPI_ADD, false);
  else LAUNCH_NT4(CLIPA_EPI_DACT, false);
#undef LAUNCH_NT4
  return clipa_check_launch("gemm_nt4<bf16>");
}
