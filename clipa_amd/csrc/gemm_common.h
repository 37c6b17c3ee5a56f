// Pieces shared by the bf16 MFMA GEMM translation units (gemm_nt.hip, gemm_tn.hip): tile constants, launch
// arguments, the hidden LDS-DMA helper, fused-epilogue maths and the per-device launch state.
#pragma once
#include "common.h"
#include "clipa_hip.h"
#include <atomic>
#include <mutex>

namespace clipa_gemm {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int NTHREADS = 512;
constexpr int IMG_BYTES = 256 * 64 * 2;       // one 256-row operand image at BK = 64: 32 KiB
constexpr int STAGE_BYTES = 2 * IMG_BYTES;    // A image + B image
constexpr int HS_BYTES = 32768;               // gemm_tn2: one 32-row slab (P + Q images)

struct NTArgs {
  const char* A; const char* B; char* C; char* C2; const float* bias; const char* aux;
  int M, N, K;
  long lda, ldb, ldc, ldaux;   // element strides
  float alpha;
  int epi, act;
  int abl;   // experiment flags (clipa_internal_debug_set): 1 no global stores, 2 no epilogue, 8 row-major tile order
  int gm;    // A panels per tile group (nt_group_size)
  int pre8;  // gemm_nta: C2 is an e4m3 copy of the pre-activation (1 byte per element, row stride ldc BYTES) - CLIPA_EPI_ACT_PRE8
  int aux8;  // gemm_nta: aux holds e4m3 bytes (row stride ldaux BYTES) - CLIPA_EPI_DACT8
};

// Tile order of the persistent NT kernels: an XCD walks groups of `gm` A panels, N-tile major inside a group.  More panels per
// group = fewer passes of the weight matrix through the fabric (it does not fit one XCD's 4 MiB L2 and is re-fetched once per
// group); the group's A panels must stay cache-resident while the N tiles go by.  Measured (r02_gemm_tile_group_size_ab.jsonl):
// 8 for wide outputs, 16 for narrow ones, capped at 8 MiB of A panels per group.  Round 5 (r05_gemm_nta_tile_group_size_sweep.jsonl,
// M = 806 912): with four N tiles (1024-wide outputs of long-K products, the input-gradient GEMMs of the in-projection and of
// c_fc) a group of EIGHT panels is exactly the 32 tiles an XCD runs side by side - 1285 -> 1347 TF/s at K = 3072 where the cap
// gave 5 panels (a group and a half per XCD round), 1390 -> 1401 at K = 4096 (cap: 4) - so that case takes 8 up to 16 MiB.
inline int nt_group_size(long tilesN, long panel_bytes) {
  const long pb = panel_bytes > 0 ? panel_bytes : 1;
  if (tilesN == 4 && pb >= (1L << 20) && 8 * pb <= (16L << 20)) return 8;      // (K >= 2048; K = 1024 keeps 16: 1047-1051 vs 1040)
  const long want = tilesN <= 12 ? 16 : 8;
  const long cap = (8L << 20) / pb;
  const long g = want < cap ? want : cap;
  return (int)(g < 2 ? 2 : g);
}

struct TNArgs {
  const char* P; const char* Q; float* O;
  int M, R, C;
  long ldp, ldq, ldo;
  int slice_rows;   // multiple of 64
  float* colsum;    // optional [S][R] partial column sums of P (the bias gradient rides the weight-gradient GEMM)
  int nslices;      // > 0: 1-D grid, XCD x owns the M slices x, x+8, ... (all tiles of a slice share one L2)
  int abl;          // experiment flags (clipa_internal_debug_set), gemm_tna only
};

typedef float f32x4v __attribute__((ext_vector_type(4)));

// LOCK: pairs of the 8-value chunk evaluated in lockstep (common.h: gelu_cdf_n) - 4 = the whole chunk, 2 = two halves where
// the kernel has no registers to spare (the fp8 GEMM's epilogues)
template <int ACT, int LOCK = 4>
__device__ __forceinline__ void epi_apply(int epi, float* v, const float* a) {
  if (ACT == ACT_GELU_ERF) {
#pragma unroll
    for (int h = 0; h < 4; h += LOCK) {
      f32x2 x[LOCK], r[LOCK];
      if (epi == CLIPA_EPI_ACT) {
#pragma unroll
        for (int i = 0; i < LOCK; ++i) x[i] = f32x2{v[2 * (h + i)], v[2 * (h + i) + 1]};
        gelu_cdf_n<LOCK>(x, r);
#pragma unroll
        for (int i = 0; i < LOCK; ++i) { const f32x2 y = x[i] * r[i]; v[2 * (h + i)] = y.x; v[2 * (h + i) + 1] = y.y; }
      } else {
#pragma unroll
        for (int i = 0; i < LOCK; ++i) x[i] = f32x2{a[2 * (h + i)], a[2 * (h + i) + 1]};
        gelu_grad_n<LOCK>(x, r);
#pragma unroll
        for (int i = 0; i < LOCK; ++i) { const f32x2 y = f32x2{v[2 * (h + i)], v[2 * (h + i) + 1]} * r[i]; v[2 * (h + i)] = y.x; v[2 * (h + i) + 1] = y.y; }
      }
    }
    return;
  }
  if (epi == CLIPA_EPI_ACT) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      const f32x2 r = act_fwd2<ACT>(f32x2{v[i], v[i + 1]});
      v[i] = r.x;
      v[i + 1] = r.y;
    }
  } else {  // CLIPA_EPI_DACT
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      const f32x2 r = f32x2{v[i], v[i + 1]} * act_bwd2<ACT>(f32x2{a[i], a[i + 1]});
      v[i] = r.x;
      v[i + 1] = r.y;
    }
  }
}

template <int ACT, int LOCK = 4>
__device__ __forceinline__ u32x4 epi_chunk(int epi, u32x4 v, u32x4 av) {
  float f[8], a[8];
  unpack8(v, f);
  unpack8(av, a);
  if (epi == CLIPA_EPI_ADD) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] += a[i];
  } else {
    epi_apply<ACT, LOCK>(epi, f, a);
  }
  return pack8(f);
}

// the same with the second operand already in fp32 (e4m3 pre-activations, CLIPA_EPI_DACT8)
template <int ACT, int LOCK = 4>
__device__ __forceinline__ u32x4 epi_chunk_f(int epi, u32x4 v, const float* a) {
  float f[8];
  unpack8(v, f);
  if (epi == CLIPA_EPI_ADD) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] += a[i];
  } else {
    epi_apply<ACT, LOCK>(epi, f, a);
  }
  return pack8(f);
}

// A buffer descriptor as four SGPR words + an LDS-DMA load issued from inline asm: hipcc does not see the
// load, so it does not put `s_waitcnt vmcnt(0)` in front of LDS reads; the kernel waits by hand (counted vmcnt).
__device__ __forceinline__ u32x4 make_srd(const void* base, unsigned bytes) {
  const unsigned long long b = (unsigned long long)base;
  u32x4 s;
  s[0] = (unsigned)b;
  s[1] = (unsigned)(b >> 32) & 0xffffu;   // stride 0
  s[2] = bytes;
  s[3] = 0x00020000u;
  return s;
}
__device__ __forceinline__ void dma16(const u32x4 srd, unsigned lds_addr, unsigned voff, unsigned soff) {
  unsigned keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
               "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(srd), "s"(soff) : "memory");
}
// Several LDS-DMA instructions from ONE asm block (one M0 save/restore): piece j lands at LDS byte address
// lds_addr + j*LSTEP and reads at voff[j] + soff.  Row offsets must sit in the VGPR offset: the scalar offset of a
// buffer instruction is EXCLUDED from the descriptor's bounds check, so only voffset-addressed rows beyond the
// tile's valid rows read zeros.  (An M0 write needs one wait state before the LDS-DMA that uses it.)
template <int LSTEP>
__device__ __forceinline__ void dma16_x4(const u32x4 srd, unsigned lds_addr, unsigned v0, unsigned v1, unsigned v2, unsigned v3,
                                         unsigned soff) {
  unsigned keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
               "buffer_load_dwordx4 %2, %6, %7 offen lds\n\ts_add_u32 m0, m0, %8\n\ts_nop 0\n\t"
               "buffer_load_dwordx4 %3, %6, %7 offen lds\n\ts_add_u32 m0, m0, %8\n\ts_nop 0\n\t"
               "buffer_load_dwordx4 %4, %6, %7 offen lds\n\ts_add_u32 m0, m0, %8\n\ts_nop 0\n\t"
               "buffer_load_dwordx4 %5, %6, %7 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_addr), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(srd), "s"(soff), "n"(LSTEP) : "memory", "scc");
}
template <int LSTEP>
__device__ __forceinline__ void dma16_x2(const u32x4 srd, unsigned lds_addr, unsigned v0, unsigned v1, unsigned soff) {
  unsigned keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
               "buffer_load_dwordx4 %2, %4, %5 offen lds\n\ts_add_u32 m0, m0, %6\n\ts_nop 0\n\t"
               "buffer_load_dwordx4 %3, %4, %5 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_addr), "v"(v0), "v"(v1), "s"(srd), "s"(soff), "n"(LSTEP) : "memory", "scc");
}
#define WG_BARRIER_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// ---------------------------------------------------------------------------------------------------------------
// Epilogue of a 256 x 256 bf16 output tile through a 32 KiB LDS window (gemm_nt2<bf16>, gemm_nt_f8): four passes of 64
// rows - the waves holding those rows pack them into the window (16x16 MFMA layout -> row-major, XOR-swizzled 16-byte
// chunks), every thread then takes four 16-byte chunks, applies the element-wise epilogue and stores them.
//
// Every LDS access and every aux load here is inline asm, and every global store an unconditional buffer store
// (out-of-range lanes are dropped by the descriptor): with an LDS-DMA in flight hipcc puts `s_waitcnt vmcnt(0)` in front
// of each LDS access and each use of a loaded register it can see, and a store inside a divergent branch makes its
// counter bookkeeping pessimistic - each of those waits drained the stores of the pass before (five round trips to HBM
// per tile, 9.5 us of a 35 us tile at K = 1024).  Counted by hand instead; loads and stores retire in issue order on the
// vector-memory counter.  Per wave and tile:
//   [next tile's first K step: 8 LDS-DMA]  [aux: 8 loads per 2 passes]  [stores: 4 (8 with PRE) per pass]
// so chunk j's aux load is always followed by 7 younger operations ((7 - k) loads + k stores), and after the last pass
// exactly WIN_STORES(PRE) stores are younger than the next tile's operands.
struct WinOut {
  char* C; char* C2; const char* aux;     // C2: pre-activation copy (PRE)
  long ldc, ldaux;
  int M, N, m0, n0;                        // matrix extent, tile origin
  int act, abl;
};
constexpr int win_stores(bool pre) { return pre ? 32 : 16; }

// prepare(pass): the packing waves load what pack(ai, bj) needs for this pass (bias / scale vectors parked in LDS: held in
// registers across the passes they would push the aux epilogues into scratch: a spill's loads and stores sit on the same counter -
// extra operations can only make a counted wait stricter, never too short, but each reload is a round trip in the epilogue).
// ROLL: aux chunks in 16 registers instead of 32 - chunk j's register is refilled for the next pass right after its use
// (gemm_nt_f8, whose scale arithmetic needs the other 16).
template <int EPI, bool PRE, bool ROLL, typename Prepare, typename Pack>
__device__ __forceinline__ void window_epilogue(char* cb, const WinOut& o, int tid, int wm, int wn, Prepare&& prepare, Pack&& pack) {
  constexpr bool HAS_AUX = EPI == CLIPA_EPI_ADD || EPI == CLIPA_EPI_DACT;
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
  u32x4 av[ROLL ? 4 : 8];
  auto fetch_one = [&](u32x4& dst, int pass, int j) {
    const int c = j * NTHREADS + tid_e;
    // clamped inside the matrix: the value of an out-of-range chunk is never stored
    const int m = min(o.m0 + pass * 64 + (c >> 5), o.M - 1), n = min(o.n0 + (c & 31) * 8, o.N - 8);
    const char* ap = o.aux + ((size_t)m * o.ldaux + n) * 2;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ap) : "memory");
  };
  auto fetch_aux = [&](int pass0) {
    if constexpr (HAS_AUX) {
#pragma unroll
      for (int i = 0; i < (ROLL ? 4 : 8); ++i) fetch_one(av[i], pass0 + (i >> 2), i & 3);
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
        for (int bj = 0; bj < 4; ++bj) {
          const int nl = wn * 64 + bj * 16 + 4 * g4;
          const u32x2 w = pack(ai, bj);
          const unsigned wa = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(cb + row * 512 + ((((nl >> 3) ^ row) & 31) << 4) + (nl & 7) * 2);
          asm volatile("ds_write_b64 %0, %1" :: "v"(wa), "v"(w) : "memory");
        }
      }
    }
    WG_BARRIER_LDS();
    // the activation-backward epilogue takes the window two chunks at a time (8 fewer live registers: no scratch)
    constexpr int CG = EPI == CLIPA_EPI_DACT ? 2 : 4;
#pragma unroll
    for (int g = 0; g < 4 / CG; ++g) {
      u32x4 cv[CG];
      unsigned a[CG];
#pragma unroll
      for (int jj = 0; jj < CG; ++jj) {
        const int c = (g * CG + jj) * NTHREADS + tid_e;
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
        const int c = j * NTHREADS + tid_e;
        const int row = c >> 5, cc = c & 31;
        unsigned vo = (unsigned)(((size_t)(pass * 64 + row) * o.ldc + cc * 8) * 2);
        if (cc * 8 >= cols_t || ((o.abl & 1) && cv[jj][0] != 0x12345u)) vo |= 0x80000000u;
        u32x4 v = cv[jj];
        if constexpr (PRE) __builtin_amdgcn_raw_buffer_store_b128(v, rsC2, (int)vo, 0, 0);
        if constexpr (EPI != CLIPA_EPI_NONE) {
          u32x4& aj = av[ROLL ? j : (pass & 1) * 4 + j];
          // operations younger than chunk j's aux load.  Batches of 8: (7 - k) loads + k stores = 7 whatever the chunk.
          // ROLL: pass 0: (3 - j) loads + j (store + refill); later: (3 - j) (store + refill) of the pass before + j
          // (store + refill) of this one = 6, minus the j refills the last pass no longer issues.
          if constexpr (HAS_AUX) {
            if constexpr (!ROLL) asm volatile("s_waitcnt vmcnt(7)" : "+v"(aj) :: "memory");
            else {
              const int younger = pass == 0 ? 3 + j : pass == 3 ? 6 - j : 6;
              switch (younger) {   // pass and j are unrolled constants: one case survives
                case 3: asm volatile("s_waitcnt vmcnt(3)" : "+v"(aj) :: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(4)" : "+v"(aj) :: "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(5)" : "+v"(aj) :: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(6)" : "+v"(aj) :: "memory"); break;
              }
            }
          } else aj = u32x4{0, 0, 0, 0};
          if (o.act == ACT_GELU_ERF) v = epi_chunk<ACT_GELU_ERF, ROLL ? 2 : 4>(EPI, v, aj);
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

// The tile's bias / scale vectors arrive in the window by LDS-DMA during the last K step; the first pass overwrites the
// window, so the first `bytes` of it are parked in the ring slot that last K step has just finished with (free until the
// next tile's second K step is staged).  Call after `s_waitcnt vmcnt(0)` + barrier; the epilogue's first barrier publishes it.
__device__ __forceinline__ void park_vectors(const char* window, char* free_slot, int tid, int bytes) {
  if (tid * 16 < bytes) {
    const unsigned src = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(window + tid * 16);
    const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(free_slot + tid * 16);
    u32x4 t;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tds_write_b128 %2, %0" : "=&v"(t) : "v"(src), "v"(dst) : "memory");
  }
}
__device__ __forceinline__ void lds_read4_f4(float4 (&v)[4], const char* base) {   // v[i] = 16 bytes at base + 64 i
  const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)base;
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %4 offset:128\n\t"
               "ds_read_b128 %3, %4 offset:192\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(a) : "memory");
}
__device__ __forceinline__ void lds_read4_f1(float (&v)[4], const char* base) {    // v[i] = 4 bytes at base + 64 i
  const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)base;
  asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:64\n\tds_read_b32 %2, %4 offset:128\n\t"
               "ds_read_b32 %3, %4 offset:192\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(a) : "memory");
}

// Ring protocol shared by the persistent NT kernels: a K step's operands are waited for (vmcnt + barrier) at the END of the
// step before it, so the wait in front of a tile's first step sits after the previous tile's epilogue and can leave that
// epilogue's stores in flight (RING_WAIT_AFTER_EPILOGUE: the 8 LDS-DMA of the step are older than the stores).
#define RING_WAIT_ALL() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define RING_WAIT_AFTER_EPILOGUE(NSTORES) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(NSTORES) : "memory")

// Per-device launch state: LDS opt-in attributes are set once per device (std::call_once), the CU count is read
// from the device the call runs on.  No process-wide mutable state besides the experiment knobs (atomics).
int gemm_num_cu(int dev);                 // multiProcessorCount of `dev`, cached
int current_device(int* dev);             // hipGetDevice with error reporting
extern std::atomic<int> g_nt_variant;     // clipa_internal_debug_set: gemm_nt kernel selection (0 = per-shape default)
extern std::atomic<int> g_abl;            // clipa_internal_debug_set: experiment flags
extern std::atomic<int> g_last_gemm;      // clipa_internal_last_gemm: 1 gemm_nt2, 2 gemm_nta, 3 gemm_tn2, 4 gemm_tn3, 5 gemm_tna
// clipa_internal_gemm_counts: launches per kernel family since the last reset (tests prove that a model-level step really
// dispatched the production kernels): slots 1..9 = the families of g_last_gemm, 10 = gemm_nta<ACT, e4m3 pre-activation copy>,
// 11 = gemm_nta<DACT, e4m3 operand>, 12 / 13 = the same two epilogues of gemm_f8a
constexpr int GEMM_COUNT_SLOTS = 16;
extern std::atomic<long> g_gemm_count[GEMM_COUNT_SLOTS];
inline void note_gemm(int family, int extra = 0) {
  g_last_gemm.store(family, std::memory_order_relaxed);
  g_gemm_count[family].fetch_add(1, std::memory_order_relaxed);
  if (extra) g_gemm_count[extra].fetch_add(1, std::memory_order_relaxed);
}

constexpr int MAX_DEVICES = 64;

// gemm_nta.hip: the four-wave / hand-scheduled kernel for whole-tile bf16 shapes (clipa_gemm_nt dispatches to it)
constexpr int NTA_DEFAULT_SCHEDULE = 4;   // profiles/r03_gemm_nta_schedules_0_7_mainloop_ablation.jsonl
bool nta_eligible(const NTArgs& a, int out_f32);
int nta_launch(const NTArgs& a, int dev, int num_cu, int sched, hipStream_t st);
// gemm_tna.hip: the same structure for the weight-gradient product (clipa_gemm_tn dispatches to it)
bool tna_eligible(const TNArgs& a);
int tna_launch(const TNArgs& a, int dev, dim3 grid, int sched, hipStream_t st);
// gemm_tn.hip: work order and split-M slice count of the weight-gradient kernels (shared with the fp8 one, gemm_tn8.hip)
bool tn_per_xcd(long M, long R, long C);
long tn_slices(long M, long R, long C, int num_cu, bool per_xcd);

}  // namespace clipa_gemm
