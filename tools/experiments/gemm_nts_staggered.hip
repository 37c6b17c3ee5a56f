// EXPERIMENT (round 2, not linked into libclipa_hip.so): staggered-epilogue GEMM.  Correct on the first run (every epilogue
// against fp64, one- vs two-output activation epilogue bit-equal, relaunch bit-equal: tools/nts_check_experiment.py) and
// SLOWER than gemm_nt2_kernel on every production shape (0.57-0.82x, profiles/r02_gemm_staggered_epilogue.md).  What it
// claimed then (RETRACTED later in the round - profiles/r02_gemm_counted_waits.md: a CU pulls 67 B/clk through LDS-DMA from L2,
// and most of the epilogue cost was compiler-inserted vmcnt(0) drains): the main loop is bound by the CU's vector-memory pipe, not by the matrix pipe - 64 LDS-DMA pieces of 1 KiB
// per K step at ~50 cycles each = the measured 1.6 us per step (17-20 B/clk/CU) - so a group that has the SIMDs to itself
// still needs 75 % of a full step's bytes for 50 % of its FLOPs, and the stores go through the same pipe.  To build it:
// add this file to clipa_amd/build.py SOURCES, declare gemm_nts_eligible / gemm_nts_launch in gemm_common.h and route
// variant 21 / 22 of clipa_debug_set to gemm_nts_launch in clipa_gemm_nt (git history: the commit that added this file).
// bf16 MFMA GEMM with a STAGGERED epilogue for gfx950:  C[M,N] = epi(alpha * A[M,K] . B[N,K]^T + bias[N])
//
// Same call sites as gemm_nt.hip (clipa_torch/open_clip/transformer.py:209,217-219,234 and their input gradients).  Why a
// second kernel: in gemm_nt2_kernel all eight waves of a workgroup reach the epilogue together, a CU's vector-store path
// moves ~16 B/clk, and the matrix pipe idles for >= 4.3 us per 256x256 output (profiles/r02_gemm_epilogue_experiments.md:
// 25-56 % on top of the main loop at K = 1024).  Hiding it needs a second instruction stream that owns the matrix pipe
// meanwhile - without loading any operand twice:
//
//   * the workgroup's tile is split by rows into two GROUPS of four waves (one wave of each group per SIMD); a group
//     owns a 128 x 256 half tile (its own A half-slab per K step) and both share the B slab;
//   * a workgroup keeps ONE B n-tile for its whole life and walks down M, so the B slab stream (k = step mod nkt) is the
//     same for every tile, and a tile may start at ANY k and wrap around: group 1 runs half a period behind group 0;
//   * time is cut into steps of one barrier each; per tile a group spends nkt steps in the main loop and NTS_E steps in
//     the epilogue (16 rows per step through a double-buffered LDS window, coalesced 16-byte stores), and the two
//     epilogue windows never overlap: while one group stores, the other has each SIMD's matrix pipe to itself;
//   * every wave moves its share of every slab in every step (B, the A half-slab of each computing group, and the aux
//     rows of the residual / activation-backward epilogues, which reach the storing group through LDS) - except the two
//     waves of a draining group that issue its stores: a store is acknowledged later than a step ends, so those waves
//     never wait on vmcnt and their two sibling waves move their LDS-DMA pieces as well.
//
// A tile's K order is a rotation fixed by the schedule (tile index, group), so results are deterministic, and the one- and
// two-output activation epilogues share the schedule: a block's recompute reproduces its forward bit for bit.
#include "gemm_common.h"

namespace clipa_gemm {
namespace {

constexpr int S_SLOT = 65536;              // ring slot: A half-slab of group 0 | of group 1 | B slab
constexpr int S_AHALF = 16384;
constexpr int S_B = 2 * S_AHALF;
constexpr int S_WIN = 2 * S_SLOT;          // epilogue window: two halves of 16 rows x 512 B
constexpr int S_AUX = S_WIN + 16384;       // aux rows of the storing group, same shape
constexpr int S_LDS = S_AUX + 16384;       // 163840: all of the CU's LDS
constexpr int NTS_E = 9;                   // epilogue steps per half tile: 8 passes of 16 rows + the drain of the last

struct NTSArgs {
  NTArgs g;
  int nchunks;       // row chunks per n-tile (workgroup = one n-tile x one chunk of m-tiles)
  int chunk_tiles;   // m-tiles per chunk
};

// A group's position is (tile, r): r counts the steps of the current period (negative before the group starts).
// state: 0 main loop (s = K step within the tile), 1 epilogue (s = epilogue step), 2 idle
__device__ __forceinline__ void group_state(int tile, int r, int nkt, int ntiles, int& state, int& s) {
  state = 2; s = 0;
  if (r < 0 || tile >= ntiles) return;
  if (r < nkt) { state = 0; s = r; } else { state = 1; s = r - nkt; }
}

__global__ __launch_bounds__(NTHREADS) void gemm_nts_kernel(NTSArgs q) {
  const NTArgs& p = q.g;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mg = wave >> 2, wn = wave & 3;         // my group (rows 128 mg ..), my 64-column block
  const int l15 = lane & 15, g4 = lane >> 4, sw16 = (l15 >> 1) & 7;
  const int gtid = tid & 255;

  // ---- this workgroup's work: n-tile tn, m-tiles [mt0, mt0 + ntiles) ----------------------------------------------
  const int tilesM = (p.M + BM - 1) / BM;
  const unsigned unit = xcd_remap(blockIdx.x, gridDim.x);            // consecutive units (same n-tile) share an XCD's L2
  const int tn = (int)unit / q.nchunks, chunk = (int)unit - tn * q.nchunks;
  const int n0 = tn * BN;
  const int mt0 = chunk * q.chunk_tiles;
  const int ntiles = max(0, min(q.chunk_tiles, tilesM - mt0));
  const int nkt = (p.K + BK - 1) / BK;
  const int P = nkt + NTS_E, H = P / 2;
  const int T = ntiles > 0 ? H + ntiles * P : 0;

  // ---- DMA addressing ----------------------------------------------------------------------------------------------
  // A half-slab (128 rows x 128 B): 16 pieces of 8 rows, piece j*8 + wave; B slab (256 rows): 32 pieces, piece j*8 + wave
  // every wave moves its share of every slab in every step (a wave's issue cost per LDS-DMA is what limits a step:
  // concentrating the DMA on the four waves that compute made the first version of this kernel 28 % slower)
  unsigned voffA[2], voffB[4];
  int kelA[2], kelB[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int rowB = (j * 8 + wave) * 8 + (lane >> 3);
    const int chB = (lane & 7) ^ ((rowB >> 1) & 7);
    voffB[j] = (unsigned)(rowB * p.ldb * 2 + chB * 16);
    kelB[j] = chB * 8;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rowA = (j * 8 + wave) * 8 + (lane >> 3);            // 128-row half-slab: 16 pieces, 2 per wave
    const int chA = (lane & 7) ^ ((rowA >> 1) & 7);
    voffA[j] = (unsigned)(rowA * p.lda * 2 + chA * 16);
    kelA[j] = chA * 8;
  }
  const __amdgpu_buffer_rsrc_t rsB = make_rsrc(p.B + (size_t)n0 * p.ldb * 2, (unsigned)(max(0, min(BN, p.N - n0)) * p.ldb * 2));
  // dw = 0: my own pieces; dw = -2: the pieces of wave - 2 (16 rows up: same chunk swizzle, so voffset - 16 rows)
  auto issueB = [&](int slot, int k0, int dw) {
    char* sB = smem + slot * S_SLOT + S_B;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pc = j * 8 + wave + dw;
      const unsigned oob = (k0 + kelB[j] >= p.K) ? 0x80000000u : 0u;
      const unsigned v = voffB[j] + (unsigned)(dw * 8 * (int)p.ldb * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(sB + pc * 1024), 16, v | oob, k0 * 2, 0, 0);
    }
  };
  auto issueA = [&](int slot, int g, int tile, int k0, int dw) {  // group g's half-slab of its tile `tile`
    const long m0 = (long)(mt0 + tile) * BM + 128 * g;
    const long rows = min(128L, (long)p.M - m0);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.A + (size_t)m0 * p.lda * 2, (unsigned)(max(0L, rows) * p.lda * 2));
    char* sA = smem + slot * S_SLOT + g * S_AHALF;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pc = j * 8 + wave + dw;
      const unsigned oob = (k0 + kelA[j] >= p.K) ? 0x80000000u : 0u;
      const unsigned v = voffA[j] + (unsigned)(dw * 8 * (int)p.lda * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(sA + pc * 1024), 16, v | oob, k0 * 2, 0, 0);
    }
  };
  const bool has_aux = p.epi == CLIPA_EPI_ADD || p.epi == CLIPA_EPI_DACT;
  auto issueAux = [&](int g, int tile, int pass, int dw) {         // 16 aux rows of pass `pass` of group g's half tile -> LDS
    const long m0 = (long)(mt0 + tile) * BM + 128 * g + 16 * pass;
    const long rows = min(16L, (long)p.M - m0);
    const long cols_left = (long)p.N - n0;                         // columns >= N are never stored: their aux may be anything
    const long bytes = rows > 0 ? (rows - 1) * p.ldaux * 2 + min(256L, cols_left) * 2 : 0;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.aux + ((size_t)m0 * p.ldaux + n0) * 2, (unsigned)max(0L, bytes));
    char* dst = smem + S_AUX + (pass & 1) * 8192;
    const int pc = wave + dw;                                      // rows 2 pc, 2 pc + 1 of the pass: 512 B each
    const int row = 2 * pc + (lane >> 5);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(dst + pc * 1024), 16, (unsigned)(row * p.ldaux * 2 + (lane & 31) * 16), 0, 0, 0);
  };

  // bias of my 4 n blocks (the n-tile never changes)
  float4 bias4[4];
#pragma unroll
  for (int bj = 0; bj < 4; ++bj) {
    const int n = n0 + wn * 64 + bj * 16 + 4 * g4;
    bias4[bj] = (p.bias && n < p.N) ? *(const float4*)(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  }

  f32x4v acc16[4][8];      // [n block of 16][m block of 16] of my 128 x 64 wave tile
#pragma unroll
  for (int bj = 0; bj < 4; ++bj)
#pragma unroll
    for (int ai = 0; ai < 8; ++ai) acc16[bj][ai] = f32x4v{0.f, 0.f, 0.f, 0.f};

  if (T == 0) return;
  const int epi = p.epi, act = p.act;
  // prologue: slabs of step 0 (only group 0 can be in its main loop at step 0)
  issueB(0, 0, 0);
  issueA(0, 0, 0, 0, 0);
  int tl[2] = {0, 0}, rr[2] = {0, -H};       // (tile, step within the period) of the two groups
  int kb = 0;                                // K slab of the current step = t mod nkt
  bool issued = true;
  // Roles inside the group that is draining its tile (epilogue steps 1..8): its waves 0, 1 (of 4) only store - they never
  // wait on vmcnt, a store takes longer to be acknowledged than a step lasts - and its waves 2, 3 issue the LDS-DMA pieces
  // of their twins 0, 1 on top of their own.
  const bool store_wave = (wave & 2) == 0;

  for (int t = 0; t < T; ++t) {
    if (issued) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the LDS-DMA pieces I issued for this step
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int st[2], ss[2], nst[2], nss[2], ntl[2], nrr[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      group_state(tl[g], rr[g], nkt, ntiles, st[g], ss[g]);
      nrr[g] = rr[g] + 1; ntl[g] = tl[g];
      if (nrr[g] == P) { nrr[g] = 0; ntl[g] = tl[g] + 1; }
      group_state(ntl[g], nrr[g], nkt, ntiles, nst[g], nss[g]);
    }
    const int slot = t & 1, nslot = slot ^ 1;
    const int kn = kb + 1 == nkt ? 0 : kb + 1;

    // ---- DMA for step t + 1 ----
    const bool draining = st[mg] == 1 && ss[mg] >= 1;      // my group stores in this step
    issued = false;
    if (t + 1 < T && !(draining && store_wave)) {
      const int k1 = kn * BK;
      const int ndw = draining ? 2 : 1;                     // a draining group's DMA waves also move their twins' pieces
      for (int w = 0; w < ndw; ++w) {
        const int dw = w ? -2 : 0;
        if (nst[0] == 0 || nst[1] == 0) issueB(nslot, k1, dw);
        if (nst[0] == 0) issueA(nslot, 0, ntl[0], k1, dw);
        if (nst[1] == 0) issueA(nslot, 1, ntl[1], k1, dw);
        if (has_aux) {
#pragma unroll
          for (int g = 0; g < 2; ++g)
            if (nst[g] == 1 && nss[g] >= 1 && nss[g] <= 8) issueAux(g, ntl[g], nss[g] - 1, dw);
        }
      }
      issued = true;
    }
    const int cur_tile = tl[mg];

    if (st[mg] == 0) {
      // ---- main loop step: my group's 128 x 256 half tile, K slab (t mod nkt) ----
      if (ss[mg] == 0) {
#pragma unroll
        for (int bj = 0; bj < 4; ++bj)
#pragma unroll
          for (int ai = 0; ai < 8; ++ai) acc16[bj][ai] = f32x4v{0.f, 0.f, 0.f, 0.f};
      }
      const char* sA = smem + slot * S_SLOT + mg * S_AHALF;
      const char* sB = smem + slot * S_SLOT + S_B;
      const char* pa = sA + l15 * 128;
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
    } else if (st[mg] == 1) {
      // ---- epilogue step s: drain pass s - 1 from the window to global memory, fill the window with pass s ----
      const int s = ss[mg];
      const long mbase = (long)(mt0 + cur_tile) * BM + 128 * mg;
      char* win = smem + S_WIN;
      if (s >= 1 && store_wave) {
        const int qp = s - 1;
        const int stid = tid & 127;                         // thread index within the group's two store waves
#pragma nounroll
        for (int hh = 0; hh < 2; ++hh) {                    // two chunks at a time: 128 accumulator registers stay live beside this
          u32x4 cv[2], av[2];
          unsigned wa[2], xa[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int c = (2 * hh + j) * 128 + stid;
            const int row = c >> 5, cc = c & 31;
            wa[j] = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(win + (qp & 1) * 8192 + row * 512 + (((cc ^ row) & 31) << 4));
            xa[j] = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + S_AUX + (qp & 1) * 8192 + row * 512 + (cc << 4));
          }
          // inline-asm LDS reads: hipcc would put `s_waitcnt vmcnt(0)` in front of compiler-visible LDS reads while stores
          // of the previous pass are in flight
          asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(cv[0]), "=&v"(cv[1]) : "v"(wa[0]), "v"(wa[1]) : "memory");
          av[0] = u32x4{0, 0, 0, 0};
          av[1] = u32x4{0, 0, 0, 0};
          if (has_aux)
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(av[0]), "=&v"(av[1]) : "v"(xa[0]), "v"(xa[1]) : "memory");
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int c = (2 * hh + j) * 128 + stid;
            const int row = c >> 5, cc = c & 31;
            const long m = mbase + 16 * qp + row;
            const int n = n0 + cc * 8;
            if (m < p.M && n < p.N) {
              u32x4 v = cv[j];
              if (epi == CLIPA_EPI_ACT && p.C2) *(u32x4*)(p.C2 + ((size_t)m * p.ldc + n) * 2) = v;
              if (epi != CLIPA_EPI_NONE) {
                if (act == ACT_GELU_ERF) v = epi_chunk<ACT_GELU_ERF>(epi, v, av[j]);
                else if (act == ACT_GELU_TANH) v = epi_chunk<ACT_GELU_TANH>(epi, v, av[j]);
                else v = epi_chunk<ACT_QUICK_GELU>(epi, v, av[j]);
              }
              *(u32x4*)(p.C + ((size_t)m * p.ldc + n) * 2) = v;
            }
          }
        }
      }
      if (s <= 7) {
        char* half = win + (s & 1) * 8192;
        const int row = l15;
        auto fill = [&](const f32x4v (&a)[4][8], int ai) {
#pragma unroll
          for (int bj = 0; bj < 4; ++bj) {
            const int nl = wn * 64 + bj * 16 + 4 * g4;
            const float4 b4 = bias4[bj];
            u32x2 w;
            w[0] = pack2bf(a[bj][ai][0] * p.alpha + b4.x, a[bj][ai][1] * p.alpha + b4.y);
            w[1] = pack2bf(a[bj][ai][2] * p.alpha + b4.z, a[bj][ai][3] * p.alpha + b4.w);
            unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(half + row * 512 + ((((nl >> 3) ^ row) & 31) << 4) + (nl & 7) * 2);
            asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(w) : "memory");
          }
        };
        switch (s) {     // the m block is a register index: it must be a compile-time constant
          case 0: fill(acc16, 0); break;
          case 1: fill(acc16, 1); break;
          case 2: fill(acc16, 2); break;
          case 3: fill(acc16, 3); break;
          case 4: fill(acc16, 4); break;
          case 5: fill(acc16, 5); break;
          case 6: fill(acc16, 6); break;
          default: fill(acc16, 7); break;
        }
      }
    }
    tl[0] = ntl[0]; tl[1] = ntl[1];
    rr[0] = nrr[0]; rr[1] = nrr[1];
    kb = kn;
  }
}

std::once_flag g_nts_once[MAX_DEVICES];
int g_nts_rc[MAX_DEVICES];

}  // namespace

// Shapes the staggered kernel takes: bf16 output, enough K steps to cover an epilogue window (nkt >= 2 * NTS_E would be
// ideal; >= 12 = K 768 measured), enough m-tiles per workgroup to amortise the half-period start-up.
bool gemm_nts_eligible(int64_t M, int64_t N, int64_t K, int num_cu) {
  const long nkt = (K + BK - 1) / BK, tilesM = (M + BM - 1) / BM, tilesN = (N + BN - 1) / BN;
  if (nkt < 12 || tilesN > num_cu) return false;
  const long nchunks = num_cu / tilesN;
  return tilesM >= 8 * nchunks;
}

int gemm_nts_launch(const NTArgs& a, int dev, int num_cu, hipStream_t st) {
  std::call_once(g_nts_once[dev], [dev]() {
    const hipError_t e = hipFuncSetAttribute((const void*)gemm_nts_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS);
    g_nts_rc[dev] = 0;
    if (e != hipSuccess) { clipa_set_error("hipFuncSetAttribute(gemm_nts): %s", hipGetErrorString(e)); g_nts_rc[dev] = CLIPA_ERR_LAUNCH; }
  });
  if (g_nts_rc[dev]) return g_nts_rc[dev];
  NTSArgs q;
  q.g = a;
  const long tilesM = (a.M + BM - 1) / BM, tilesN = (a.N + BN - 1) / BN;
  q.nchunks = (int)(num_cu / tilesN);
  q.chunk_tiles = (int)((tilesM + q.nchunks - 1) / q.nchunks);
  const unsigned grid = (unsigned)(tilesN * q.nchunks);
  hipLaunchKernelGGL(gemm_nts_kernel, dim3(grid), dim3(NTHREADS), S_LDS, st, q);
  return clipa_check_launch("gemm_nts");
}

}  // namespace clipa_gemm
