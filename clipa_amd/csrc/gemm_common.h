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
  int abl;   // experiment flags (clipa_debug_set): 1 no global stores, 2 no epilogue, 8 row-major tile order
};

struct TNArgs {
  const char* P; const char* Q; float* O;
  int M, R, C;
  long ldp, ldq, ldo;
  int slice_rows;   // multiple of 64
  float* colsum;    // optional [S][R] partial column sums of P (the bias gradient rides the weight-gradient GEMM)
  int nslices;      // > 0: 1-D grid, XCD x owns the M slices x, x+8, ... (all tiles of a slice share one L2)
};

typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int ACT>
__device__ __forceinline__ void epi_apply(int epi, float* v, const float* a) {
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

template <int ACT>
__device__ __forceinline__ u32x4 epi_chunk(int epi, u32x4 v, u32x4 av) {
  float f[8], a[8];
  unpack8(v, f);
  unpack8(av, a);
  if (epi == CLIPA_EPI_ADD) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] += a[i];
  } else {
    epi_apply<ACT>(epi, f, a);
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

// Per-device launch state: LDS opt-in attributes are set once per device (std::call_once), the CU count is read
// from the device the call runs on.  No process-wide mutable state besides the experiment knobs (atomics).
int gemm_num_cu(int dev);                 // multiProcessorCount of `dev`, cached
int current_device(int* dev);             // hipGetDevice with error reporting
extern std::atomic<int> g_nt_variant;     // clipa_debug_set: gemm_nt kernel selection (0 = per-shape default)
extern std::atomic<int> g_abl;            // clipa_debug_set: experiment flags

constexpr int MAX_DEVICES = 64;

}  // namespace clipa_gemm
