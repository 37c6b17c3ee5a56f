// bf16 MFMA GEMMs for gfx950 (MI355X).
//
//   gemm_nt : C[M,N] = epi(alpha * A[M,K] . B[N,K]^T + bias[N])        (forward, dgrad with W^T)
//   gemm_tn : O[R,C] = sum_m P[m,R] * Q[m,C]                           (weight gradients)
//
// Reference call sites these replace (clipa_torch/open_clip/transformer.py): nn.MultiheadAttention
// in/out projection :209,:234, mlp.c_fc/c_proj :217-219, conv1 as a patch GEMM :371,:491, the final
// projections :528-529 and model.py:254, and the logits GEMMs of loss.py:135-142 - plus the autograd
// transposes of each.
//
// Structure (both kernels): 256x256 output tile, K-step 64, 8 waves (512 threads), one workgroup per
// CU. Operand tiles are DMA'd HBM->LDS with `buffer_load_dwordx4 ... lds` (bounds-checked SRD, so
// ragged edges read zeros) into a double-buffered 2 x 64 KiB LDS ring; tile t+1 is in flight while
// tile t feeds v_mfma_f32_32x32x16_bf16.  The LDS image is lane-linear (DMA constraint), so bank
// conflicts are removed by permuting the 16-byte chunks on the *source* address and applying the
// same XOR on the read (tools/lds_bank_sim.py).  gemm_tn reads its fragments with
// ds_read_b64_tr_b16 (hardware transpose) because the reduction index is the slow axis of both
// operands.  Tile ids are remapped so each XCD's L2 sees a contiguous run of tiles.
//
// Kernel generations in this file (clipa_debug_set / CLIPA_GEMM_NT pick one for in-process A/B runs; each is
// covered by tests/test_kernels_gpu.py):
//   gemm_nt_kernel        v1  one output tile per workgroup
//   gemm_nt2_kernel       v2  persistent workgroups, DMA ring running across tiles; template flags: STAG = ping-pong
//                             main loop (BK 32, four half-slots), M16 = v_mfma_f32_16x16x32_bf16 main loop + epilogue
//                             -> <bf16, M16> is the PRODUCTION kernel for every bf16-output GEMM (variant 11);
//                             <f32> serves the logits GEMMs
//   gemm_nt3_kernel       v3  v2 + loader / storer wave roles (plain and fused epilogues)
//   gemm_nt5_kernel       v5  ping-pong + roles, plain / bias epilogue
//   gemm_tn_kernel        v1  split-M weight gradient, all waves in step
//   gemm_tn2_kernel       v2  ping-pong schedule                      } production: chosen per shape by the host
//   gemm_tn3_kernel       v3  v1 schedule on 16x16x32 MFMAs           }
// Why so many: under the 1400 W power cap the schedules that stall less do not run faster, the ones that move
// fewer register bytes per FLOP do - the older generations are kept as the measured baselines.
#include "common.h"
#include "clipa_hip.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int NTHREADS = 512;
constexpr int IMG_BYTES = 256 * 64 * 2;       // one operand image: 32 KiB
constexpr int STAGE_BYTES = 2 * IMG_BYTES;    // A image + B image
constexpr int EPI_STRIDE = 528;               // bytes per C-tile row in the epilogue image
constexpr int LDS_BYTES = 256 * EPI_STRIDE;   // 135168 B >= 2 stages (131072 B)

struct NTArgs {
  const char* A; const char* B; char* C; char* C2; const float* bias; const char* aux;
  int M, N, K;
  long lda, ldb, ldc, ldaux;   // element strides
  float alpha;
  int epi, act;
  int abl;   // ablation flags for kernel experiments (CLIPA_GEMM_ABL): 1 no global stores, 2 no epilogue, 4 no bias loads
};

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

template <bool OUT_F32>
__global__ __launch_bounds__(NTHREADS) void gemm_nt_kernel(NTArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wave >> 2, wn = wave & 3;   // wave tile: 128 (m) x 64 (n)

  const int tilesN = (p.N + BN - 1) / BN;
  const int tilesM = (p.M + BM - 1) / BM;
  const unsigned t = xcd_remap(blockIdx.x, (unsigned)(tilesM * tilesN));
  const int tm = t / tilesN, tn = t - tm * tilesN;
  const int m0 = tm * BM, n0 = tn * BN;

  const int rowsA = min(BM, p.M - m0), rowsB = min(BN, p.N - n0);
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A + (size_t)m0 * p.lda * 2, (unsigned)(rowsA * p.lda * 2));
  const __amdgpu_buffer_rsrc_t rsB = make_rsrc(p.B + (size_t)n0 * p.ldb * 2, (unsigned)(rowsB * p.ldb * 2));

  // DMA piece pc = j*8+wave covers image rows pc*8..pc*8+7 (128 B each); lane -> (row, phys chunk).
  unsigned voffA[4], voffB[4];
  int kel[4];   // first k element (within a K tile) this lane fetches in piece j
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
      // K tail (K % 64 != 0): push the lane's offset past num_records so the DMA writes zeros
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
    __syncthreads();   // tile kt has landed for every wave; everyone is done reading buffer (kt+1)&1
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

  // D[n][m] fragment: lane holds m = l31, n = 8*(r>>2) + 4*hi + (r&3).
  if (OUT_F32) {
    float* C = (float*)p.C;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wm * 128 + mi * 32 + l31;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wn * 64 + ni * 32 + 8 * q + 4 * hi;
          if (m < p.M && n < p.N) {
            float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias) b4 = *(const float4*)(p.bias + n);
            float4 o;
            o.x = acc[ni][mi][4 * q + 0] * p.alpha + b4.x;
            o.y = acc[ni][mi][4 * q + 1] * p.alpha + b4.y;
            o.z = acc[ni][mi][4 * q + 2] * p.alpha + b4.z;
            o.w = acc[ni][mi][4 * q + 3] * p.alpha + b4.w;
            *(float4*)(C + (size_t)m * p.ldc + n) = o;
          }
        }
      }
    return;
  }

  __syncthreads();   // staging buffers are dead; reuse LDS as the [256][256] bf16 C image
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int nl = wn * 64 + ni * 32 + 8 * q + 4 * hi;
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias && n0 + nl < p.N) b4 = *(const float4*)(p.bias + n0 + nl);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int ml = wm * 128 + mi * 32 + l31;
        u32x2 w;
        w[0] = pack2bf(acc[ni][mi][4 * q + 0] * p.alpha + b4.x, acc[ni][mi][4 * q + 1] * p.alpha + b4.y);
        w[1] = pack2bf(acc[ni][mi][4 * q + 2] * p.alpha + b4.z, acc[ni][mi][4 * q + 3] * p.alpha + b4.w);
        *(u32x2*)(smem + ml * EPI_STRIDE + nl * 2) = w;
      }
    }
  __syncthreads();

  const int epi = p.epi, act = p.act;
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int c = it * NTHREADS + tid;
    const int row = c >> 5, cc = c & 31;
    const int m = m0 + row, n = n0 + cc * 8;
    if (m < p.M && n < p.N) {
      u32x4 v = *(const u32x4*)(smem + row * EPI_STRIDE + cc * 16);
      if (epi != CLIPA_EPI_NONE) {
        float f[8], a[8];
        unpack8(v, f);
        if (epi == CLIPA_EPI_ADD || epi == CLIPA_EPI_DACT) {
          const u32x4 av = *(const u32x4*)(p.aux + ((size_t)m * p.ldaux + n) * 2);
          unpack8(av, a);
        }
        if (epi == CLIPA_EPI_ADD) {
#pragma unroll
          for (int i = 0; i < 8; ++i) f[i] += a[i];
        } else {
          if (epi == CLIPA_EPI_ACT && p.C2) *(u32x4*)(p.C2 + ((size_t)m * p.ldc + n) * 2) = v;
          if (act == ACT_GELU_ERF) epi_apply<ACT_GELU_ERF>(epi, f, a);
          else if (act == ACT_GELU_TANH) epi_apply<ACT_GELU_TANH>(epi, f, a);
          else epi_apply<ACT_QUICK_GELU>(epi, f, a);
        }
        v = pack8(f);
      }
      *(u32x4*)(p.C + ((size_t)m * p.ldc + n) * 2) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// gemm_nt v2: PERSISTENT tiles.  One workgroup per CU walks its share of the output tiles; the DMA
// ring never drains (the first K tile of the next output tile is fetched while the last K tile of
// the current one computes), MFMA operand fragments are register-prefetched one k-step ahead, and
// the epilogue goes through a dedicated 32 KiB LDS window (4 passes of 64 rows) placed after the
// 128 KiB ring, so its global stores retire underneath the next tile's main loop.
constexpr int CBUF_OFF = 2 * STAGE_BYTES;            // 131072
constexpr int LDS2_BYTES = CBUF_OFF + 64 * 512;      // 163840 = all 160 KiB of the CU

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
#define WG_BARRIER_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// STAG = true selects the STAGGERED main loop ("nt4"): K is walked in phases of 32 through a ring of four
// 32 KiB half-slots (A [256][32] + B [256][32] images, 64-byte rows, chunk ^= (row>>2)&3), three phases in
// flight (96 KiB instead of 64 KiB: the loop is latency x bandwidth bound on the L2->LDS path), waits are
// counted (`vmcnt(8)`: the two newest phases may still be in flight).  The two wave groups (wm = 0 / 1 = the
// two waves of every SIMD) are offset by one k-step: group 1 carries the fragments of a phase's second k-step
// across the barrier, so right after every barrier each SIMD has eight MFMAs to issue while the LDS reads of
// the new phase are still in flight.
constexpr int HS_BYTES = 32768;

// M16 = true (bf16 output, non-STAG only): the main loop issues v_mfma_f32_16x16x32_bf16 (wave tile = 8 x 4
// blocks of 16 x 16, fragment = 16 rows x 32 k: lane -> row l&15, chunk 4*kk + (l>>4); the image and its
// (row>>1)&7 swizzle are unchanged and stay conflict-free for that read).  A quarter of the accumulator registers
// are read and written per instruction: less register-file energy per FLOP, which is what counts under the power cap.
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <bool OUT_F32, bool SETPRIO, bool STAG = false, bool M16 = false>
__global__ __launch_bounds__(NTHREADS) void gemm_nt2_kernel(NTArgs p) {
  static_assert(!M16 || (!OUT_F32 && !STAG), "16x16x32 variant: bf16 output, first-generation ring only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wave >> 2, wn = wave & 3;

  const int tilesN = (p.N + BN - 1) / BN;
  const int tilesM = (p.M + BM - 1) / BM;
  const unsigned ntiles = (unsigned)(tilesM * tilesN);
  // XCD x (blocks with bid%8 == x) owns the contiguous tile range [base, base+len); its workgroups
  // take consecutive tiles, so the N-tiles of one A panel run side by side under one L2.
  const unsigned G = gridDim.x, xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
  const unsigned gx = (G - xcd + 7u) >> 3;            // workgroups on this XCD
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
  const int sw = (l31 >> 1) & 7;
  const int rowoffA = (wm * 128 + l31) * 128;
  const int rowoffB = (wn * 64 + l31) * 128;
  const int nkt = (p.K + BK - 1) / BK;

  // Tile order inside an XCD's range: groups of GM A-panels, N-tile major inside a group, so the ~32
  // tiles an XCD runs concurrently touch GM A-panels x (32/GM) B-tiles instead of 2 x 16 (fewer distinct
  // operand tiles per L2).  p.abl bit 3 selects the plain row-major order for A/B experiments.
  auto tile_origin = [&](unsigned t, int& m0, int& n0) {
    const int GM = (p.abl & 8) ? 1 : 4;
    const int per = GM * tilesN;
    const int g = (int)t / per, r = (int)t - g * per;
    const int gm = min(GM, tilesM - g * GM);
    const int tn = r / gm, mm = r - tn * gm;
    m0 = (g * GM + mm) * BM;
    n0 = tn * BN;
  };
  auto stage = [&](int buf, int m0, int n0, int k0) {
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A + (size_t)m0 * p.lda * 2, (unsigned)(min(BM, p.M - m0) * p.lda * 2));
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(p.B + (size_t)n0 * p.ldb * 2, (unsigned)(min(BN, p.N - n0) * p.ldb * 2));
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

  // ---- staggered variant: per-wave DMA pieces of a phase.  A piece = 16 rows x 64 B; wave w moves pieces
  // w and w+8 of the A image and of the B image (4 DMA instructions per wave per phase).
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int chunk4 = (lane & 3) ^ ((lane >> 4) & 3);        // source chunk: (row>>2)&3 == (lane>>4)&3 for every piece
  const int row4 = wave * 16 + (lane >> 2);
  const unsigned voff4A0 = (unsigned)(row4 * p.lda * 2 + chunk4 * 16), voff4A1 = voff4A0 + (unsigned)(128 * p.lda * 2);
  const unsigned voff4B0 = (unsigned)(row4 * p.ldb * 2 + chunk4 * 16), voff4B1 = voff4B0 + (unsigned)(128 * p.ldb * 2);
  const int sw4 = (l31 >> 2) & 3;
  const int np = (p.K + 31) / 32;              // host guarantees np >= 3 when STAG
  auto stage4 = [&](unsigned slot, const u32x4 rsA, const u32x4 rsB, int k0) {
    const unsigned oob = (k0 + chunk4 * 8 >= p.K) ? 0x80000000u : 0u;
    const unsigned d = lds0 + slot * HS_BYTES + wave * 1024;
    dma16(rsA, d, voff4A0 | oob, (unsigned)(k0 * 2));
    dma16(rsA, d + 8192, voff4A1 | oob, (unsigned)(k0 * 2));
    dma16(rsB, d + 16384, voff4B0 | oob, (unsigned)(k0 * 2));
    dma16(rsB, d + 16384 + 8192, voff4B1 | oob, (unsigned)(k0 * 2));
  };
  auto srdA = [&](int m) { return make_srd(p.A + (size_t)m * p.lda * 2, (unsigned)(min(BM, p.M - m) * p.lda * 2)); };
  auto srdB = [&](int n) { return make_srd(p.B + (size_t)n * p.ldb * 2, (unsigned)(min(BN, p.N - n) * p.ldb * 2)); };

  if (idx >= len) return;
  unsigned it = idx;
  int m0, n0;
  tile_origin(base + it, m0, n0);
  if constexpr (STAG) {
    const u32x4 a = srdA(m0), b = srdB(n0);
    stage4(0, a, b, 0);
    stage4(1, a, b, 32);
    stage4(2, a, b, 64);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WG_BARRIER_LDS();
  } else {
    stage(0, m0, n0, 0);
  }
  unsigned gk = 0;   // global K-tile (STAG: phase) counter: ring slot = gk & 1 (STAG: gk & 3)
  for (;;) {
    const bool has_next = it + gx < len;
    int m1 = 0, n1 = 0;
    if (has_next) tile_origin(base + it + gx, m1, n1);

    f32x16 acc[2][4];
    f32x4v acc16[4][8];      // M16: [n block of 16][m block of 16]
    if constexpr (M16) {
#pragma unroll
      for (int bj = 0; bj < 4; ++bj)
#pragma unroll
        for (int ai = 0; ai < 8; ++ai) acc16[bj][ai] = f32x4v{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;
    }

    if constexpr (STAG) {
      const u32x4 rAc = srdA(m0), rBc = srdB(n0), rAn = srdA(m1), rBn = srdB(n1);
      const int offA = (wm * 128 + l31) * 64, offB = (wn * 64 + l31) * 64;
      const int c0 = (hi ^ sw4) << 4, c1 = ((2 + hi) ^ sw4) << 4;
      // PING-PONG: every wave alternates a LOAD phase (12 ds_read_b128 = both k-steps of one slab into
      // registers, its 4 DMA pieces of the slab three ahead, the counted vmcnt) and a COMPUTE phase (16 MFMAs
      // from those registers), one s_barrier after each.  Group 1 (waves 4-7, the second wave of every SIMD)
      // runs one barrier behind group 0, so on each SIMD one wave's MFMA block always sits beside the other
      // wave's LDS/DMA phase: the LDS latency is never on the MFMA pipe's critical path.
      //   slab s is read by group 0 in phase 2s and by group 1 in phase 2s+1; its half-slot is refilled (slab
      //   s+4) from phase 2s+2 on, i.e. in LOAD(s+1) of either group; a wave waits for its own pieces of slab
      //   s+1 at the end of LOAD(s) (the two newer slabs = 8 instructions may stay in flight), two barriers
      //   before anyone reads them.
      bf16x8 fa[2][4], fb[2][2];
      if (wm == 1) WG_BARRIER_LDS();
      for (int sl = 0; sl < np; ++sl, ++gk) {
        // ---- LOAD
        const char* sA = smem + (gk & 3) * HS_BYTES + offA;
        const char* sB = smem + (gk & 3) * HS_BYTES + 16384 + offB;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) fa[0][mi] = *(const bf16x8*)(sA + mi * 2048 + c0);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) fb[0][ni] = *(const bf16x8*)(sB + ni * 2048 + c0);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) fa[1][mi] = *(const bf16x8*)(sA + mi * 2048 + c1);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) fb[1][ni] = *(const bf16x8*)(sB + ni * 2048 + c1);
        {
          const int q = sl + 3;                 // refill the half-slot both groups finished reading one phase ago
          if (q < np) stage4((gk + 3) & 3, rAc, rBc, q * 32);
          else if (has_next) stage4((gk + 3) & 3, rAn, rBn, (q - np) * 32);
        }
        // own pieces of slab sl+1 must have landed (slabs 0..2 of a tile are drained before the loop starts).
        // The two newer slabs may stay in flight; at the tail of the last tile nothing newer was issued.
        if (sl >= 2) {
          if (!has_next && sl + 3 >= np) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        WG_BARRIER_LDS();
        // ---- COMPUTE
        if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
              acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks][ni], fa[ks][mi], acc[ni][mi], 0, 0, 0);
        if (SETPRIO) __builtin_amdgcn_s_setprio(0);
        WG_BARRIER_LDS();
      }
      if (wm == 0) WG_BARRIER_LDS();          // group 1's last COMPUTE phase: both groups are aligned again
      // drain the next tile's first three slabs (issued by the last three LOAD phases) before the epilogue's
      // own loads and stores enter the queue; the epilogue's barriers make them visible to every wave
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else
    for (int kt = 0; kt < nkt; ++kt, ++gk) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (kt + 1 < nkt) stage((gk + 1) & 1, m0, n0, (kt + 1) * BK);
      else if (has_next) stage((gk + 1) & 1, m1, n1, 0);
      const char* sA = smem + (gk & 1) * STAGE_BYTES;
      const char* sB = sA + IMG_BYTES;
      if constexpr (M16) {
        // 8 sub-steps per K tile: (kk, s) = 32-wide k-step kk, A blocks 2s and 2s+1 against the four B blocks of
        // kk (8 MFMAs); A fragments double-buffered per sub-step, B fragments per k-step
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
          if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int bj = 0; bj < 4; ++bj)
#pragma unroll
            for (int a = 0; a < 2; ++a)
              acc16[bj][2 * sb + a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gb[kk][bj], ga[u & 1][a], acc16[bj][2 * sb + a], 0, 0, 0);
          if (SETPRIO) __builtin_amdgcn_s_setprio(0);
        }
        continue;
      }
      bf16x8 fa[2][4], fb[2][2];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) fa[0][mi] = *(const bf16x8*)(sA + rowoffA + mi * 4096 + ((hi ^ sw) << 4));
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) fb[0][ni] = *(const bf16x8*)(sB + rowoffB + ni * 4096 + ((hi ^ sw) << 4));
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks < 3) {
          const int coff = ((2 * (ks + 1) + hi) ^ sw) << 4;
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) fa[nxt][mi] = *(const bf16x8*)(sA + rowoffA + mi * 4096 + coff);
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) fb[nxt][ni] = *(const bf16x8*)(sB + rowoffB + ni * 4096 + coff);
        }
        if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][ni], fa[cur][mi], acc[ni][mi], 0, 0, 0);
        if (SETPRIO) __builtin_amdgcn_s_setprio(0);
      }
    }

    // ---- epilogue of tile (m0, n0); the ring keeps filling for the next tile meanwhile ----
    if (p.abl & 2) {   // ablation: keep the accumulators live, write nothing
      float t = 0.f;
      if constexpr (M16) {
#pragma unroll
        for (int bj = 0; bj < 4; ++bj)
#pragma unroll
          for (int ai = 0; ai < 8; ++ai) t += acc16[bj][ai][0] + acc16[bj][ai][1] + acc16[bj][ai][2] + acc16[bj][ai][3];
      } else {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) t += acc[ni][mi][r];
      }
      if (t == 1.2345e-30f) ((float*)p.C)[tid] = t;
    } else if (OUT_F32) {
      float* C = (float*)p.C;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const int m = m0 + wm * 128 + mi * 32 + l31;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int n = n0 + wn * 64 + ni * 32 + 8 * q + 4 * hi;
            if (m < p.M && n < p.N) {
              float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
              if (p.bias) b4 = *(const float4*)(p.bias + n);
              float4 o;
              o.x = acc[ni][mi][4 * q + 0] * p.alpha + b4.x;
              o.y = acc[ni][mi][4 * q + 1] * p.alpha + b4.y;
              o.z = acc[ni][mi][4 * q + 2] * p.alpha + b4.z;
              o.w = acc[ni][mi][4 * q + 3] * p.alpha + b4.w;
              *(float4*)(C + (size_t)m * p.ldc + n) = o;
            }
          }
        }
    } else {
      char* cb = smem + CBUF_OFF;
      const int epi = p.epi, act = p.act;
      float4 bias4[2][4];     // M16 uses [0][bj]: the 4 consecutive features of n block bj this lane holds
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (M16 && ni == 1) continue;
          const int n = M16 ? n0 + wn * 64 + q * 16 + 4 * (lane >> 4) : n0 + wn * 64 + ni * 32 + 8 * q + 4 * hi;
          bias4[ni][q] = (p.bias && n < p.N && !(p.abl & 4)) ? *(const float4*)(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      // this thread's chunks of a pass: chunk c = j*512 + tid -> row c>>5 (0..63), 16-B column c&31.
      // aux (residual / pre-activation) chunks are fetched two passes at a time, ahead of their use.
      u32x4 av[8];
      auto fetch_aux = [&](int pass0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          av[i] = u32x4{0, 0, 0, 0};
          const int c = (i & 3) * NTHREADS + tid;
          const int m = m0 + (pass0 + (i >> 2)) * 64 + (c >> 5), n = n0 + (c & 31) * 8;
          if ((epi == CLIPA_EPI_ADD || epi == CLIPA_EPI_DACT) && m < p.M && n < p.N)
            av[i] = *(const u32x4*)(p.aux + ((size_t)m * p.ldaux + n) * 2);
        }
      };
      // LDS-only barriers: a full __syncthreads() would also wait (vmcnt) for the previous pass's global stores
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        if ((pass & 1) == 0) fetch_aux(pass);
        LDS_BARRIER();   // readers of the previous pass are done with the window
        if (M16 && wm == (pass >> 1)) {
          // 16x16 blocks: lane holds features nl..nl+3 (nl = 16 bj + 4 (lane>>4)) of row 16 a2 + (lane & 15)
#pragma unroll
          for (int a2 = 0; a2 < 4; ++a2) {
            const int ai = 4 * (pass & 1) + a2;
            const int row = a2 * 16 + (lane & 15);
#pragma unroll
            for (int bj = 0; bj < 4; ++bj) {
              const int nl = wn * 64 + bj * 16 + 4 * (lane >> 4);
              const float4 b4 = bias4[0][bj];
              u32x2 w;
              w[0] = pack2bf(acc16[bj][ai][0] * p.alpha + b4.x, acc16[bj][ai][1] * p.alpha + b4.y);
              w[1] = pack2bf(acc16[bj][ai][2] * p.alpha + b4.z, acc16[bj][ai][3] * p.alpha + b4.w);
              *(u32x2*)(cb + row * 512 + ((((nl >> 3) ^ row) & 31) << 4) + (nl & 7) * 2) = w;
            }
          }
        } else if (wm == (pass >> 1)) {
#pragma unroll
          for (int mi2 = 0; mi2 < 2; ++mi2) {
            const int mi = 2 * (pass & 1) + mi2;
            const int row = mi2 * 32 + l31;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int nl = wn * 64 + ni * 32 + 8 * q + 4 * hi;
                const float4 b4 = bias4[ni][q];
                u32x2 w;
                w[0] = pack2bf(acc[ni][mi][4 * q + 0] * p.alpha + b4.x, acc[ni][mi][4 * q + 1] * p.alpha + b4.y);
                w[1] = pack2bf(acc[ni][mi][4 * q + 2] * p.alpha + b4.z, acc[ni][mi][4 * q + 3] * p.alpha + b4.w);
                *(u32x2*)(cb + row * 512 + ((((nl >> 3) ^ row) & 31) << 4) + (nl & 7) * 2) = w;
              }
          }
        }
        LDS_BARRIER();
        // The window is read with inline-asm ds_read_b128: with an LDS-DMA possibly in flight hipcc puts
        // `s_waitcnt vmcnt(0)` in front of every compiler-visible LDS read, which would serialise each
        // pass behind the global stores of the previous one.  (Hidden loads: waited for by hand, §5.7.)
        u32x4 cv[4];
        {
          unsigned a[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int c = j * NTHREADS + tid;
            const int row = c >> 5, cc = c & 31;
            a[j] = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(cb + row * 512 + (((cc ^ row) & 31) << 4));
          }
          asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\t"
                       "s_waitcnt lgkmcnt(0)"
                       : "=&v"(cv[0]), "=&v"(cv[1]), "=&v"(cv[2]), "=&v"(cv[3])
                       : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3])
                       : "memory");
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = j * NTHREADS + tid;
          const int row = c >> 5, cc = c & 31;
          const int m = m0 + pass * 64 + row, n = n0 + cc * 8;
          if (m < p.M && n < p.N && !((p.abl & 1) && cv[j][0] != 0x12345u)) {
            u32x4 v = cv[j];
            if (epi == CLIPA_EPI_ACT && p.C2) *(u32x4*)(p.C2 + ((size_t)m * p.ldc + n) * 2) = v;
            if (epi != CLIPA_EPI_NONE) {
              if (act == ACT_GELU_ERF) v = epi_chunk<ACT_GELU_ERF>(epi, v, av[(pass & 1) * 4 + j]);
              else if (act == ACT_GELU_TANH) v = epi_chunk<ACT_GELU_TANH>(epi, v, av[(pass & 1) * 4 + j]);
              else v = epi_chunk<ACT_QUICK_GELU>(epi, v, av[(pass & 1) * 4 + j]);
            }
            *(u32x4*)(p.C + ((size_t)m * p.ldc + n) * 2) = v;
          }
        }
      }
#undef LDS_BARRIER
    }
    if (!has_next) break;
    it += gx;
    m0 = m1;
    n0 = n1;
  }
}

// ------------------------------------------------------------------------------------------------
// gemm_nt v3: persistent tiles + WAVE ROLES.  vmcnt is per wave and counts stores as well as loads, so a
// wave that both feeds the DMA ring and writes C must drain its stores before every ring hand-off.  Here
// waves 0-3 ("loaders") issue every LDS-DMA (operand ring + the aux/residual window) and never store to
// global memory; waves 4-7 ("storers") issue every global store and never touch the DMA queue.  The
// loaders' `s_waitcnt vmcnt(0)` then covers DMA only, the storers' C tile drains underneath the next
// tile's main loop, and all eight waves run the same MFMA main loop.  The DMA is issued from inline asm
// (hidden from hipcc, which otherwise puts vmcnt(0) in front of every LDS read while a DMA may be pending).
template <int EPI, bool SETPRIO>
__global__ __launch_bounds__(NTHREADS) void gemm_nt3_kernel(NTArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool NEED_AUX = (EPI == CLIPA_EPI_ADD || EPI == CLIPA_EPI_DACT);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wave >> 2, wn = wave & 3;
  const bool loader = wave < 4;              // wave-uniform role
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  char* cb = smem + CBUF_OFF;

  const int tilesN = (p.N + BN - 1) / BN;
  const int tilesM = (p.M + BM - 1) / BM;
  const unsigned ntiles = (unsigned)(tilesM * tilesN);
  const unsigned G = gridDim.x, xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
  const unsigned gx = (G - xcd + 7u) >> 3;
  const unsigned q8 = ntiles >> 3, r8 = ntiles & 7u;
  const unsigned base = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const unsigned len = q8 + (xcd < r8 ? 1u : 0u);

  // loader wave lw moves pieces pc = j*4 + lw (j = 0..7) of each 32-piece operand image; a piece is 8 rows x
  // 128 B.  The row's chunk swizzle does not depend on j, so one voffset per operand + a scalar offset per j.
  const int lw = wave & 3;
  const int row0 = lw * 8 + (lane >> 3);
  const int chunk0 = (lane & 7) ^ ((row0 >> 1) & 7);
  const unsigned voffA0 = (unsigned)(row0 * p.lda * 2 + chunk0 * 16);
  const unsigned voffB0 = (unsigned)(row0 * p.ldb * 2 + chunk0 * 16);
  const int kel0 = chunk0 * 8;
  const int sw = (l31 >> 1) & 7;
  const int rowoffA = (wm * 128 + l31) * 128;
  const int rowoffB = (wn * 64 + l31) * 128;
  const int nkt = (p.K + BK - 1) / BK;      // host guarantees nkt >= 3 for this kernel

  auto tile_origin = [&](unsigned t, int& m0, int& n0) {
    const int GM = 4;
    const int per = GM * tilesN;
    const int g = (int)t / per, r = (int)t - g * per;
    const int gm = min(GM, tilesM - g * GM);
    const int tn = r / gm, mm = r - tn * gm;
    m0 = (g * GM + mm) * BM;
    n0 = tn * BN;
  };
  auto stage = [&](int buf, int m0, int n0, int k0) {   // loaders only
    const u32x4 rsA = make_srd(p.A + (size_t)m0 * p.lda * 2, (unsigned)(min(BM, p.M - m0) * p.lda * 2));
    const u32x4 rsB = make_srd(p.B + (size_t)n0 * p.ldb * 2, (unsigned)(min(BN, p.N - n0) * p.ldb * 2));
    const unsigned oob = (k0 + kel0 >= p.K) ? 0x80000000u : 0u;
    const unsigned sA = lds0 + buf * STAGE_BYTES + lw * 1024, sB = sA + IMG_BYTES;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      dma16(rsA, sA + j * 4096, voffA0 | oob, (unsigned)(k0 * 2 + j * 32 * p.lda * 2));
      dma16(rsB, sB + j * 4096, voffB0 | oob, (unsigned)(k0 * 2 + j * 32 * p.ldb * 2));
    }
  };
  auto stage_part = [&](int buf, int m0, int n0, int k0, int part) {   // a quarter of stage(): pieces j = 2*part, 2*part+1
    const u32x4 rsA = make_srd(p.A + (size_t)m0 * p.lda * 2, (unsigned)(min(BM, p.M - m0) * p.lda * 2));
    const u32x4 rsB = make_srd(p.B + (size_t)n0 * p.ldb * 2, (unsigned)(min(BN, p.N - n0) * p.ldb * 2));
    const unsigned oob = (k0 + kel0 >= p.K) ? 0x80000000u : 0u;
    const unsigned sA = lds0 + buf * STAGE_BYTES + lw * 1024, sB = sA + IMG_BYTES;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = 2 * part + jj;
      dma16(rsA, sA + j * 4096, voffA0 | oob, (unsigned)(k0 * 2 + j * 32 * p.lda * 2));
      dma16(rsB, sB + j * 4096, voffB0 | oob, (unsigned)(k0 * 2 + j * 32 * p.ldb * 2));
    }
  };
  // aux rows [64*pass, +64) of the tile -> a 32 KiB window, stored with the C window's chunk swizzle
  // (chunk c of row r lives at chunk position c ^ (r & 31)); 32 pieces of 2 rows, 8 per loader wave
  auto fetch_aux = [&](int m0, int n0, int pass, unsigned win) {
    const int rows_left = p.M - (m0 + pass * 64);
    const int cols_left = p.N - n0;
    const u32x4 rs = make_srd(p.aux + ((size_t)(m0 + pass * 64) * p.ldaux + n0) * 2,
                              rows_left > 0 ? (unsigned)((long)min(rows_left, 64) * p.ldaux * 2) : 0u);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int pc = j * 4 + lw;
      const int row = 2 * pc + (lane >> 5);
      const int c = (lane & 31) ^ (row & 31);                         // source chunk for this LDS position
      const unsigned oob = (c * 8 >= cols_left) ? 0x80000000u : 0u;
      dma16(rs, win + pc * 1024, (unsigned)((lane >> 5) * p.ldaux * 2 + c * 16) | oob, (unsigned)(pc * 2 * p.ldaux * 2));
    }
  };

  if (idx >= len) return;
  unsigned it = idx;
  int m0, n0;
  tile_origin(base + it, m0, n0);
  if (loader) stage(0, m0, n0, 0);
  unsigned gk = 0;
  for (;;) {
    const bool has_next = it + gx < len;
    int m1 = 0, n1 = 0;
    if (has_next) tile_origin(base + it + gx, m1, n1);

    f32x16 acc[2][4];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

    for (int kt = 0; kt < nkt; ++kt, ++gk) {
      if (loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my DMA pieces of K tile gk have landed
      WG_BARRIER_LDS();                                               // ... and so have everyone else's
      // where the next K tile comes from (this tile, or the first K tile of the next output tile)
      const bool pf = (kt + 1 < nkt) || has_next;
      const int pm = (kt + 1 < nkt) ? m0 : m1, pn = (kt + 1 < nkt) ? n0 : n1, pk = (kt + 1 < nkt) ? (kt + 1) * BK : 0;
      const bool spread = (p.abl & 16) != 0;   // issue the DMA pieces between the MFMA groups instead of up front
      if (loader) {
        if (pf && !spread) stage((gk + 1) & 1, pm, pn, pk);
        // Everything a tile needs besides its operands rides the loaders' DMA queue into the idle C window,
        // because a storer wave must never wait on a global load while its C stores drain (one vmcnt for
        // both): kt 0: the 256 bias floats (1 KiB); kt 2: the aux rows of epilogue pass 0.
        if (kt == 0 && wave == 0 && p.bias) dma16(make_srd(p.bias + n0, (unsigned)((p.N - n0) * 4)), lds0 + CBUF_OFF, (unsigned)(lane * 16), 0u);
        if (NEED_AUX && kt == 2) fetch_aux(m0, n0, 0, lds0 + CBUF_OFF);
      }
      if (kt == 1 && p.bias) {
        // fold the bias into the fp32 accumulators (out = alpha * (acc + bias / alpha)): no registers held
        // across the main loop, no global load in the epilogue
        const float ia = 1.0f / p.alpha;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 b4 = *(const float4*)(cb + (wn * 64 + ni * 32 + 8 * q + 4 * hi) * 4);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
              acc[ni][mi][4 * q + 0] += b4.x * ia;
              acc[ni][mi][4 * q + 1] += b4.y * ia;
              acc[ni][mi][4 * q + 2] += b4.z * ia;
              acc[ni][mi][4 * q + 3] += b4.w * ia;
            }
          }
      }
      const char* sA = smem + (gk & 1) * STAGE_BYTES;
      const char* sB = sA + IMG_BYTES;
      bf16x8 fa[2][4], fb[2][2];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) fa[0][mi] = *(const bf16x8*)(sA + rowoffA + mi * 4096 + ((hi ^ sw) << 4));
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) fb[0][ni] = *(const bf16x8*)(sB + rowoffB + ni * 4096 + ((hi ^ sw) << 4));
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks < 3) {
          const int coff = ((2 * (ks + 1) + hi) ^ sw) << 4;
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) fa[nxt][mi] = *(const bf16x8*)(sA + rowoffA + mi * 4096 + coff);
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) fb[nxt][ni] = *(const bf16x8*)(sB + rowoffB + ni * 4096 + coff);
        }
        if (loader && pf && spread) stage_part((gk + 1) & 1, pm, pn, pk, ks);
        if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][ni], fa[cur][mi], acc[ni][mi], 0, 0, 0);
        if (SETPRIO) __builtin_amdgcn_s_setprio(0);
      }
    }

    // ---- epilogue: 4 passes of 64 rows.  Phase 1 (the waves that own the rows; ALL epilogue maths, in MFMA
    // fragment layout): acc -> bf16, activation / residual / activation-backward against the aux fragment read
    // from LDS -> C window.  Phase 2 (storer waves only): copy window(s) -> global, 16 B per lane, full rows.
    // Windows: C window (32 KiB, after the ring) + two 32 KiB windows in the ring slot the last K tile was read
    // from (idle until the next tile's second K tile).  Aux rows: pass 0 in place in the C window (DMA'd during
    // the main loop), pass 1 -> win0, pass 2 -> win1, pass 3 -> win0; EPI_ACT puts the pre-activation tile of
    // pass p into win(p & 1).
    {
      const int act = p.act;
      const unsigned win0 = lds0 + ((gk + 1) & 1) * STAGE_BYTES, win1 = win0 + 32768;
      char* w0p = smem + ((gk + 1) & 1) * STAGE_BYTES;
      WG_BARRIER_LDS();                        // every wave is done reading the last K tile's ring slot
      if (NEED_AUX && loader) { fetch_aux(m0, n0, 1, win0); fetch_aux(m0, n0, 2, win1); }
      auto do_pass = [&](auto PASS) {
        constexpr int pass = decltype(PASS)::value;
        if (NEED_AUX && loader) {              // aux rows of this pass have landed (loads retire in order)
          if (pass == 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // younger: aux rows of passes 1, 2
          if (pass == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // younger: aux rows of pass 2
          if (pass >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (pass > 0 || NEED_AUX) WG_BARRIER_LDS();   // storers are done with the windows of pass-1 / aux visible
        if (NEED_AUX && loader && pass == 2) fetch_aux(m0, n0, 3, win0);
        if (wm == (pass >> 1)) {
          char* auxp = pass == 0 ? cb : (pass == 2 ? w0p + 32768 : w0p);      // where this pass's aux rows sit
          char* prep = w0p + (pass & 1) * 32768;                              // EPI_ACT: pre-activation window
#pragma unroll
          for (int mi2 = 0; mi2 < 2; ++mi2) {
            const int mi = 2 * (pass & 1) + mi2;
            const int row = mi2 * 32 + l31;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int nl = wn * 64 + ni * 32 + 8 * q + 4 * hi;
                const int off = row * 512 + ((((nl >> 3) ^ row) & 31) << 4) + (nl & 7) * 2;
                u32x2 w;
                w[0] = pack2bf(acc[ni][mi][4 * q + 0] * p.alpha, acc[ni][mi][4 * q + 1] * p.alpha);
                w[1] = pack2bf(acc[ni][mi][4 * q + 2] * p.alpha, acc[ni][mi][4 * q + 3] * p.alpha);
                if constexpr (EPI != CLIPA_EPI_NONE) {
                  float v[4] = {__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xffff0000u),
                                __uint_as_float(w[1] << 16), __uint_as_float(w[1] & 0xffff0000u)};
                  float a[4] = {0.f, 0.f, 0.f, 0.f};
                  if constexpr (NEED_AUX) {
                    const u32x2 av = *(const u32x2*)(auxp + off);
                    a[0] = __uint_as_float(av[0] << 16); a[1] = __uint_as_float(av[0] & 0xffff0000u);
                    a[2] = __uint_as_float(av[1] << 16); a[3] = __uint_as_float(av[1] & 0xffff0000u);
                  }
                  if constexpr (EPI == CLIPA_EPI_ACT) *(u32x2*)(prep + off) = w;
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                    if constexpr (EPI == CLIPA_EPI_ADD) v[e] += a[e];
                    else if constexpr (EPI == CLIPA_EPI_ACT)
                      v[e] = act == ACT_GELU_ERF ? act_fwd<ACT_GELU_ERF>(v[e]) : (act == ACT_GELU_TANH ? act_fwd<ACT_GELU_TANH>(v[e]) : act_fwd<ACT_QUICK_GELU>(v[e]));
                    else
                      v[e] *= act == ACT_GELU_ERF ? act_bwd<ACT_GELU_ERF>(a[e]) : (act == ACT_GELU_TANH ? act_bwd<ACT_GELU_TANH>(a[e]) : act_bwd<ACT_QUICK_GELU>(a[e]));
                  }
                  w[0] = pack2bf(v[0], v[1]);
                  w[1] = pack2bf(v[2], v[3]);
                }
                *(u32x2*)(cb + off) = w;
              }
          }
        }
        WG_BARRIER_LDS();
        if (!loader) {
          // storers: 256 threads x 8 chunks = 64 rows x 32 chunks; chunk c = j*256 + t -> row c>>5, column c&31
          const int t = tid - 256;
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            u32x4 cv[4], xv[4];
            unsigned a[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int c = (half * 4 + j) * 256 + t;
              const int row = c >> 5, cc = c & 31;
              a[j] = row * 512 + (((cc ^ row) & 31) << 4);
            }
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(cv[0]), "=&v"(cv[1]), "=&v"(cv[2]), "=&v"(cv[3])
                         : "v"(a[0] + lds0 + CBUF_OFF), "v"(a[1] + lds0 + CBUF_OFF), "v"(a[2] + lds0 + CBUF_OFF), "v"(a[3] + lds0 + CBUF_OFF)
                         : "memory");
            if constexpr (EPI == CLIPA_EPI_ACT) {
              const unsigned pw = win0 + (pass & 1) * 32768;
              asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\t"
                           "s_waitcnt lgkmcnt(0)"
                           : "=&v"(xv[0]), "=&v"(xv[1]), "=&v"(xv[2]), "=&v"(xv[3])
                           : "v"(a[0] + pw), "v"(a[1] + pw), "v"(a[2] + pw), "v"(a[3] + pw) : "memory");
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int c = (half * 4 + j) * 256 + t;
              const int row = c >> 5, cc = c & 31;
              const int m = m0 + pass * 64 + row, n = n0 + cc * 8;
              if (m < p.M && n < p.N) {
                if constexpr (EPI == CLIPA_EPI_ACT) {
                  if (p.C2) *(u32x4*)(p.C2 + ((size_t)m * p.ldc + n) * 2) = xv[j];
                }
                *(u32x4*)(p.C + ((size_t)m * p.ldc + n) * 2) = cv[j];
              }
            }
          }
        }
      };
      do_pass(std::integral_constant<int, 0>{});
      do_pass(std::integral_constant<int, 1>{});
      do_pass(std::integral_constant<int, 2>{});
      do_pass(std::integral_constant<int, 3>{});
    }
    if (!has_next) break;
    it += gx;
    m0 = m1;
    n0 = n1;
  }
}

// ------------------------------------------------------------------------------------------------
// gemm_nt v5: PING-PONG main loop + WAVE ROLES, plain / bias epilogue (the shapes v3 used to take).
//   * K is walked in slabs of 32 through four 32 KiB half-slots (A [256][32] + B [256][32], 64-byte rows,
//     chunk ^= (row>>2)&3).  Every wave alternates LOAD (12 ds_read_b128 = one slab's fragments) and COMPUTE
//     (16 MFMAs from registers) with an s_barrier after each; group 1 (waves 4-7 = the second wave of every
//     SIMD) runs one barrier behind group 0, so each SIMD always has one wave in its MFMA block beside the
//     other wave's LDS phase.
//   * Group 0 issues every LDS-DMA (all 32 pieces of slab s+3 in LOAD(s), counted `vmcnt(16)`) and never
//     stores; group 1 issues every global store and never waits on vmcnt.  Stores and loads share one
//     counter per wave, and a store retires only after its HBM round trip: in v2 that latency (not the
//     bandwidth - staggering the workgroups changed nothing) sat in front of the next tile's first DMA wait.
//   * Epilogue: owners convert their 64-row pass into one of two 32 KiB windows (the C window and the ring
//     half-slot the last slab was read from) while group 1 copies the previous pass to global; the bias
//     arrives by DMA in the idle C window and is folded into the accumulators before the last slab.
__global__ __launch_bounds__(NTHREADS) void gemm_nt5_kernel(NTArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wave >> 2, wn = wave & 3;
  const bool loader = wm == 0;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  char* cb = smem + CBUF_OFF;

  const int tilesN = (p.N + BN - 1) / BN;
  const int tilesM = (p.M + BM - 1) / BM;
  const unsigned ntiles = (unsigned)(tilesM * tilesN);
  const unsigned G = gridDim.x, xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
  const unsigned gx = (G - xcd + 7u) >> 3;
  const unsigned q8 = ntiles >> 3, r8 = ntiles & 7u;
  const unsigned base = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const unsigned len = q8 + (xcd < r8 ? 1u : 0u);
  auto tile_origin = [&](unsigned t, int& m0, int& n0) {
    const int GM = 4;
    const int per = GM * tilesN;
    const int g = (int)t / per, r = (int)t - g * per;
    const int gm = min(GM, tilesM - g * GM);
    const int tn = r / gm, mm = r - tn * gm;
    m0 = (g * GM + mm) * BM;
    n0 = tn * BN;
  };

  // loader wave lw moves pieces lw, lw+4, lw+8, lw+12 (16 rows x 64 B each) of the A image and of the B image
  const int lw = wave & 3;
  const int chunk4 = (lane & 3) ^ ((lane >> 4) & 3);
  const int row4 = lw * 16 + (lane >> 2);
  const unsigned voffA = (unsigned)(row4 * p.lda * 2 + chunk4 * 16), voffB = (unsigned)(row4 * p.ldb * 2 + chunk4 * 16);
  const unsigned stepA = (unsigned)(64 * p.lda * 2), stepB = (unsigned)(64 * p.ldb * 2);
  const int sw4 = (l31 >> 2) & 3;
  const int np = (p.K + 31) / 32;              // host guarantees np >= 4
  auto stage5 = [&](unsigned slot, const u32x4 rsA, const u32x4 rsB, int k0) {
    const unsigned oob = (k0 + chunk4 * 8 >= p.K) ? 0x80000000u : 0u;
    const unsigned d = lds0 + slot * HS_BYTES + lw * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) dma16(rsA, d + j * 4096, (voffA + j * stepA) | oob, (unsigned)(k0 * 2));
#pragma unroll
    for (int j = 0; j < 4; ++j) dma16(rsB, d + 16384 + j * 4096, (voffB + j * stepB) | oob, (unsigned)(k0 * 2));
  };
  auto srdA = [&](int m) { return make_srd(p.A + (size_t)m * p.lda * 2, (unsigned)(min(BM, p.M - m) * p.lda * 2)); };
  auto srdB = [&](int n) { return make_srd(p.B + (size_t)n * p.ldb * 2, (unsigned)(min(BN, p.N - n) * p.ldb * 2)); };

  if (idx >= len) return;
  unsigned it = idx;
  int m0, n0;
  tile_origin(base + it, m0, n0);
  if (loader) {
    const u32x4 a = srdA(m0), b = srdB(n0);
    stage5(0, a, b, 0);
    stage5(1, a, b, 32);
    stage5(2, a, b, 64);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  WG_BARRIER_LDS();
  const int offA = (wm * 128 + l31) * 64, offB = (wn * 64 + l31) * 64;
  const int c0 = (hi ^ sw4) << 4, c1 = ((2 + hi) ^ sw4) << 4;
  unsigned gk = 0;   // global slab counter: half-slot = gk & 3
  for (;;) {
    const bool has_next = it + gx < len;
    int m1 = 0, n1 = 0;
    if (has_next) tile_origin(base + it + gx, m1, n1);
    const u32x4 rAc = srdA(m0), rBc = srdB(n0), rAn = srdA(m1), rBn = srdB(n1);

    f32x16 acc[2][4];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

    bf16x8 fa[2][4], fb[2][2];
    if (wm == 1) WG_BARRIER_LDS();            // group 1 runs one barrier behind
    for (int sl = 0; sl < np; ++sl, ++gk) {
      // ---- LOAD: slab sl (landed two barriers ago) -> registers
      const char* sA = smem + (gk & 3) * HS_BYTES + offA;
      const char* sB = smem + (gk & 3) * HS_BYTES + 16384 + offB;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) fa[0][mi] = *(const bf16x8*)(sA + mi * 2048 + c0);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) fb[0][ni] = *(const bf16x8*)(sB + ni * 2048 + c0);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) fa[1][mi] = *(const bf16x8*)(sA + mi * 2048 + c1);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) fb[1][ni] = *(const bf16x8*)(sB + ni * 2048 + c1);
      if (loader) {
        // the bias row of this tile rides the DMA queue into the C window (idle until the epilogue; the copy of
        // the previous tile's last pass out of it ended before this tile's first barrier)
        if (sl == 1 && wave == 0 && p.bias)
          dma16(make_srd(p.bias + n0, (unsigned)((p.N - n0) * 4)), lds0 + CBUF_OFF, (unsigned)(lane * 16), 0u);
        // slab sl+3 into the half-slot of slab sl-1 (group 1 finished reading it one phase ago)
        const int q = sl + 3;
        if (q < np) stage5((gk + 3) & 3, rAc, rBc, q * 32);
        else if (has_next) stage5((gk + 3) & 3, rAn, rBn, (q - np) * 32);
        // slab sl+1 must have landed; the two newer slabs (16 instructions) may stay in flight.  At the tail of
        // the last tile nothing newer follows.
        if (!has_next && q >= np) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      }
      WG_BARRIER_LDS();
      // ---- COMPUTE
      if (sl == np - 1 && p.bias) {           // out = alpha * (acc + bias / alpha): no bias registers, no load in the epilogue
        const float ia = __builtin_amdgcn_rcpf(p.alpha);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 b4 = *(const float4*)(cb + (wn * 64 + ni * 32 + 8 * q + 4 * hi) * 4);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
              acc[ni][mi][4 * q + 0] += b4.x * ia;
              acc[ni][mi][4 * q + 1] += b4.y * ia;
              acc[ni][mi][4 * q + 2] += b4.z * ia;
              acc[ni][mi][4 * q + 3] += b4.w * ia;
            }
          }
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks][ni], fa[ks][mi], acc[ni][mi], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      WG_BARRIER_LDS();
    }

    // ---- epilogue.  Window of pass p: W1 (the half-slot the last slab was read from), C window, W1, C window -
    // the last pass sits in the C window, so the next tile's first DMA (into W1) cannot run into its copy.
    //   group 0:  write 0 | B0 | write 1 | B1 |                  | B2 |                  | B3 |
    //   group 1:  (MFMA)  | B0 | copy 0  | B1 | write 2, copy 1  | B2 | write 3, copy 2  | B3 | copy 3
    char* w1 = smem + ((gk + 3) & 3) * HS_BYTES;
    int te = tid;
    asm volatile("" : "+v"(te));     // keep the epilogue's address arithmetic out of the main loop's live range
    const int l31e = te & 31, hie = (te >> 5) & 1;
    auto write_pass = [&](int pass, char* win) {     // rows [64*pass, +64) of the tile, MFMA fragment layout -> window
#pragma unroll
      for (int mi2 = 0; mi2 < 2; ++mi2) {
        const int mi = 2 * (pass & 1) + mi2;
        const int row = mi2 * 32 + l31e;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int nl = wn * 64 + ni * 32 + 8 * q + 4 * hie;
            u32x2 w;
            w[0] = pack2bf(acc[ni][mi][4 * q + 0] * p.alpha, acc[ni][mi][4 * q + 1] * p.alpha);
            w[1] = pack2bf(acc[ni][mi][4 * q + 2] * p.alpha, acc[ni][mi][4 * q + 3] * p.alpha);
            *(u32x2*)(win + row * 512 + ((((nl >> 3) ^ row) & 31) << 4) + (nl & 7) * 2) = w;
          }
      }
    };
    auto copy_pass = [&](int pass, const char* win) {   // group 1: 256 threads x 8 chunks = 64 rows x 32 chunks of 16 B
      const int t = te - 256;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        u32x4 cv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = (half * 4 + j) * 256 + t;
          const int row = c >> 5, cc = c & 31;
          cv[j] = *(const u32x4*)(win + row * 512 + (((cc ^ row) & 31) << 4));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = (half * 4 + j) * 256 + t;
          const int row = c >> 5, cc = c & 31;
          const int m = m0 + pass * 64 + row, n = n0 + cc * 8;
          const int ms = (p.abl & 32) ? (m & 255) : m;   // ablation 32: every tile's rows alias 256 rows (stores stay in L2)
          if (m < p.M && n < p.N && !(p.abl & 1)) *(u32x4*)(p.C + ((size_t)ms * p.ldc + n) * 2) = cv[j];
        }
      }
    };
    if (p.abl & 2) {          // ablation: no epilogue (barrier pattern kept)
      float t = 0.f;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) t += acc[ni][mi][r];
      if (t == 1.2345e-30f) ((float*)p.C)[tid] = t;
      if (wm == 0) WG_BARRIER_LDS();
    } else if (wm == 0) {
      write_pass(0, w1);
      WG_BARRIER_LDS();
      write_pass(1, cb);
      WG_BARRIER_LDS();
      WG_BARRIER_LDS();
      WG_BARRIER_LDS();
    } else {
      copy_pass(0, w1);
      WG_BARRIER_LDS();
      write_pass(2, w1);
      copy_pass(1, cb);
      WG_BARRIER_LDS();
      write_pass(3, cb);
      copy_pass(2, w1);
      WG_BARRIER_LDS();
      copy_pass(3, cb);
    }
    if (!has_next) break;
    it += gx;
    m0 = m1;
    n0 = n1;
  }
}
#undef WG_BARRIER_LDS

// ------------------------------------------------------------------------------------------------
struct TNArgs {
  const char* P; const char* Q; float* O;
  int M, R, C;
  long ldp, ldq, ldo;
  int slice_rows;   // multiple of 64
  float* colsum;    // optional [S][R] partial column sums of P (the bias gradient rides the weight-gradient GEMM)
};

// LDS image [64 m][256 cols] bf16 (512-B rows, 32 chunks); chunk permutation per row:
__device__ __forceinline__ int tn_swz(int row) { return ((row & 3) << 2) ^ (((row >> 2) & 1) << 1); }

__global__ __launch_bounds__(NTHREADS) void gemm_tn_kernel(TNArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wr = wave >> 2, wc = wave & 3;   // wave tile: 128 (r) x 64 (c)

  const int tilesC = (p.C + 255) / 256;
  const int tilesR = (p.R + 255) / 256;
  const unsigned t = xcd_remap(blockIdx.x, (unsigned)(tilesR * tilesC));
  const int tr = t / tilesC, tc = t - tr * tilesC;
  const int r0 = tr * 256, c0 = tc * 256;
  const int slice = blockIdx.y;
  const long mbeg = (long)slice * p.slice_rows;
  const long mend = min((long)p.M, mbeg + p.slice_rows);
  float* O = p.O + (size_t)slice * p.R * p.ldo;

  // SRD base = first row of the slice, first column of the tile; rows >= M read zeros.
  const long rows_left = p.M - mbeg;
  auto nrec = [&](long ld, int col0) -> unsigned {
    long b = rows_left * ld * 2 - (long)col0 * 2;
    if (b < 0) b = 0;
    return (unsigned)(b > 0xffffffffL ? 0xffffffffL : b);
  };
  const __amdgpu_buffer_rsrc_t rsP = make_rsrc(p.P + ((size_t)mbeg * p.ldp + r0) * 2, nrec(p.ldp, r0));
  const __amdgpu_buffer_rsrc_t rsQ = make_rsrc(p.Q + ((size_t)mbeg * p.ldq + c0) * 2, nrec(p.ldq, c0));

  // DMA piece pc = j*8+wave covers image rows 2pc, 2pc+1 (512 B each).
  unsigned voffP[4], voffQ[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (j * 8 + wave) * 2 + (lane >> 5);
    const int chunk = (lane & 31) ^ tn_swz(row);
    voffP[j] = (unsigned)(row * p.ldp * 2 + chunk * 16);
    voffQ[j] = (unsigned)(row * p.ldq * 2 + chunk * 16);
  }

  f32x16 acc[4][2];
#pragma unroll
  for (int ri = 0; ri < 4; ++ri)
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ri][ci][r] = 0.f;

  auto stage = [&](int buf, long mrow) {   // mrow relative to mbeg
    char* sP = smem + buf * STAGE_BYTES;
    char* sQ = sP + IMG_BYTES;
    const unsigned soffP = (unsigned)(mrow * p.ldp * 2), soffQ = (unsigned)(mrow * p.ldq * 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pc = j * 8 + wave;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, LDS_PTR(sP + pc * 1024), 16, voffP[j], soffP, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, LDS_PTR(sQ + pc * 1024), 16, voffQ[j], soffQ, 0, 0);
    }
  };

  // transpose-read addressing: lane (hi, q, i): row = ms*16 + 8*hi + 4*half + (i>>2),
  // col = colbase + 16*q + 4*(i&3)  ->  returns (rows +0..3, col colbase+16q+i)
  const int q16 = (lane >> 4) & 1, i16 = lane & 15;
  const int rsub = 8 * hi + (i16 >> 2);          // + ms*16 + 4*half
  const int csub = 16 * q16 + 4 * (i16 & 3);     // + colbase (multiple of 32)

  const bool do_colsum = p.colsum != nullptr && tc == 0;   // one C-tile column of workgroups owns the sums
  const int cs_ch = tid & 31, cs_rg = tid >> 5;             // 32 column chunks x 16 row groups
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int nmt = (int)((mend - mbeg + 63) / 64);
  if (nmt > 0) stage(0, 0);
  for (int mt = 0; mt < nmt; ++mt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (mt + 1 < nmt) stage((mt + 1) & 1, (long)(mt + 1) * 64);
    const char* sP = smem + (mt & 1) * STAGE_BYTES;
    const char* sQ = sP + IMG_BYTES;
    if (do_colsum) {
      // column sums of the P tile (64 m-rows x 256 columns): thread = (16-B column chunk, group of 4 rows)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int row = cs_rg * 4 + rr;
        float f[8];
        unpack8(*(const u32x4*)(sP + row * 512 + ((cs_ch ^ tn_swz(row)) << 4)), f);
#pragma unroll
        for (int i = 0; i < 8; ++i) csum[i] += f[i];
      }
    }
    // fragments of k-step ms+1 are fetched (ds_read_b64_tr_b16) while the MFMAs of step ms issue
    bf16x8 fp[2][4], fq[2][2];
    auto load_frags = [&](int ms, bf16x8 (&dp)[4], bf16x8 (&dq)[2]) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int row = ms * 16 + 4 * half + rsub;
        const int swz = tn_swz(row);
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) {
          const int col = wr * 128 + ri * 32 + csub;
          const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) bf16x4*)(sP + row * 512 + (((col >> 3) ^ swz) << 4) + (col & 7) * 2));
          dp[ri][4 * half + 0] = v[0]; dp[ri][4 * half + 1] = v[1];
          dp[ri][4 * half + 2] = v[2]; dp[ri][4 * half + 3] = v[3];
        }
#pragma unroll
        for (int ci = 0; ci < 2; ++ci) {
          const int col = wc * 64 + ci * 32 + csub;
          const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) bf16x4*)(sQ + row * 512 + (((col >> 3) ^ swz) << 4) + (col & 7) * 2));
          dq[ci][4 * half + 0] = v[0]; dq[ci][4 * half + 1] = v[1];
          dq[ci][4 * half + 2] = v[2]; dq[ci][4 * half + 3] = v[3];
        }
      }
    };
    load_frags(0, fp[0], fq[0]);
#pragma unroll
    for (int ms = 0; ms < 4; ++ms) {
      const int cur = ms & 1, nxt = cur ^ 1;
      if (ms < 3) load_frags(ms + 1, fp[nxt], fq[nxt]);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ri = 0; ri < 4; ++ri)
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
          acc[ri][ci] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fp[cur][ri], fq[cur][ci], acc[ri][ci], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
  }

  // D[r][c]: lane holds c = l31, r = 8*(reg>>2) + 4*hi + (reg&3)
#pragma unroll
  for (int ri = 0; ri < 4; ++ri)
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
      const int c = c0 + wc * 64 + ci * 32 + l31;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = r0 + wr * 128 + ri * 32 + 8 * (reg >> 2) + 4 * hi + (reg & 3);
        if (r < p.R && c < p.C) O[(size_t)r * p.ldo + c] = acc[ri][ci][reg];
      }
    }
  if (do_colsum) {
    __syncthreads();                       // every wave is done with the ring: reuse it as [16][256] floats
    float* red = (float*)smem;
#pragma unroll
    for (int i = 0; i < 8; ++i) red[cs_rg * 256 + cs_ch * 8 + i] = csum[i];
    __syncthreads();
    if (tid < 256) {
      float a = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) a += red[g * 256 + tid];
      if (r0 + tid < p.R) p.colsum[(size_t)slice * p.R + r0 + tid] = a;
    }
  }
}

// gemm_tn v3 = the first-generation schedule (one barrier per 64 rows, all waves in step, register prefetch) on
// v_mfma_f32_16x16x32_bf16 with the sub-step interleave of gemm_nt2's 16x16x32 loop (A/B: ablation bit 1024).
__device__ __forceinline__ int tn_swz16(int row) { return ((row & 3) << 2) ^ (((row >> 3) & 1) << 1); }
__global__ __launch_bounds__(NTHREADS) void gemm_tn3_kernel(TNArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wr = wave >> 2, wc = wave & 3;   // wave tile: 128 (r) x 64 (c)

  const int tilesC = (p.C + 255) / 256;
  const int tilesR = (p.R + 255) / 256;
  const unsigned t = xcd_remap(blockIdx.x, (unsigned)(tilesR * tilesC));
  const int tr = t / tilesC, tc = t - tr * tilesC;
  const int r0 = tr * 256, c0 = tc * 256;
  const int slice = blockIdx.y;
  const long mbeg = (long)slice * p.slice_rows;
  const long mend = min((long)p.M, mbeg + p.slice_rows);
  float* O = p.O + (size_t)slice * p.R * p.ldo;

  // SRD base = first row of the slice, first column of the tile; rows >= M read zeros.
  const long rows_left = p.M - mbeg;
  auto nrec = [&](long ld, int col0) -> unsigned {
    long b = rows_left * ld * 2 - (long)col0 * 2;
    if (b < 0) b = 0;
    return (unsigned)(b > 0xffffffffL ? 0xffffffffL : b);
  };
  const __amdgpu_buffer_rsrc_t rsP = make_rsrc(p.P + ((size_t)mbeg * p.ldp + r0) * 2, nrec(p.ldp, r0));
  const __amdgpu_buffer_rsrc_t rsQ = make_rsrc(p.Q + ((size_t)mbeg * p.ldq + c0) * 2, nrec(p.ldq, c0));

  // DMA piece pc = j*8+wave covers image rows 2pc, 2pc+1 (512 B each).
  unsigned voffP[4], voffQ[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (j * 8 + wave) * 2 + (lane >> 5);
    const int chunk = (lane & 31) ^ tn_swz16(row);
    voffP[j] = (unsigned)(row * p.ldp * 2 + chunk * 16);
    voffQ[j] = (unsigned)(row * p.ldq * 2 + chunk * 16);
  }

  f32x4v acc[8][4];      // [r block of 16][c block of 16]
#pragma unroll
  for (int ri = 0; ri < 8; ++ri)
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) acc[ri][ci] = f32x4v{0.f, 0.f, 0.f, 0.f};

  auto stage = [&](int buf, long mrow) {   // mrow relative to mbeg
    char* sP = smem + buf * STAGE_BYTES;
    char* sQ = sP + IMG_BYTES;
    const unsigned soffP = (unsigned)(mrow * p.ldp * 2), soffQ = (unsigned)(mrow * p.ldq * 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pc = j * 8 + wave;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, LDS_PTR(sP + pc * 1024), 16, voffP[j], soffP, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, LDS_PTR(sQ + pc * 1024), 16, voffQ[j], soffQ, 0, 0);
    }
  };

  // transpose-read addressing: lane (hi, q, i): row = ms*16 + 8*hi + 4*half + (i>>2),
  // col = colbase + 16*q + 4*(i&3)  ->  returns (rows +0..3, col colbase+16q+i)
  // 16x16x32 fragment: lane (g = l>>4, i = l&15) -> rows kk*32 + 8g + 4*half + (i>>2), columns colbase + 4*(i&3) .. +3
  const int g4 = lane >> 4, i16 = lane & 15;
  const int rsub = 8 * g4 + (i16 >> 2);          // + kk*32 + 4*half
  const int csub = 4 * (i16 & 3);                // + colbase (multiple of 16)

  const bool do_colsum = p.colsum != nullptr && tc == 0;   // one C-tile column of workgroups owns the sums
  const int cs_ch = tid & 31, cs_rg = tid >> 5;             // 32 column chunks x 16 row groups
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int nmt = (int)((mend - mbeg + 63) / 64);
  if (nmt > 0) stage(0, 0);
  for (int mt = 0; mt < nmt; ++mt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (mt + 1 < nmt) stage((mt + 1) & 1, (long)(mt + 1) * 64);
    const char* sP = smem + (mt & 1) * STAGE_BYTES;
    const char* sQ = sP + IMG_BYTES;
    if (do_colsum) {
      // column sums of the P tile (64 m-rows x 256 columns): thread = (16-B column chunk, group of 4 rows)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int row = cs_rg * 4 + rr;
        float f[8];
        unpack8(*(const u32x4*)(sP + row * 512 + ((cs_ch ^ tn_swz16(row)) << 4)), f);
#pragma unroll
        for (int i = 0; i < 8; ++i) csum[i] += f[i];
      }
    }
    // 8 sub-steps per 64-row tile: (kk, s) = 32-row k-step kk, P blocks 2s and 2s+1 against the four Q blocks of kk
    // (8 MFMAs); P fragments double-buffered per sub-step, Q fragments per k-step
    auto frag = [&](const char* img, int kk, int colbase) {
      bf16x8 f;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int row = kk * 32 + 4 * half + rsub;
        const int col = colbase + csub;
        const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) bf16x4*)(img + row * 512 + (((col >> 3) ^ tn_swz16(row)) << 4) + (col & 7) * 2));
        f[4 * half + 0] = v[0]; f[4 * half + 1] = v[1]; f[4 * half + 2] = v[2]; f[4 * half + 3] = v[3];
      }
      return f;
    };
    bf16x8 gp[2][2], gq[2][4];
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) gq[0][ci] = frag(sQ, 0, wc * 64 + ci * 16);
#pragma unroll
    for (int a = 0; a < 2; ++a) gp[0][a] = frag(sP, 0, wr * 128 + a * 16);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int kk = u >> 2, sb = u & 3;
      if (u < 7) {
        const int k1 = (u + 1) >> 2, s1 = (u + 1) & 3;
#pragma unroll
        for (int a = 0; a < 2; ++a) gp[(u + 1) & 1][a] = frag(sP, k1, wr * 128 + (2 * s1 + a) * 16);
      }
      if (u == 1) {
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) gq[1][ci] = frag(sQ, 1, wc * 64 + ci * 16);
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
          acc[2 * sb + a][ci] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gp[u & 1][a], gq[kk][ci], acc[2 * sb + a][ci], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
  }

  // D[r][c]: lane holds c = cblock + (l & 15), r = rblock + 4*(l >> 4) + e
#pragma unroll
  for (int ri = 0; ri < 8; ++ri)
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      const int c = c0 + wc * 64 + ci * 16 + i16;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = r0 + wr * 128 + ri * 16 + 4 * g4 + e;
        if (r < p.R && c < p.C) O[(size_t)r * p.ldo + c] = acc[ri][ci][e];
      }
    }
  if (do_colsum) {
    __syncthreads();                       // every wave is done with the ring: reuse it as [16][256] floats
    float* red = (float*)smem;
#pragma unroll
    for (int i = 0; i < 8; ++i) red[cs_rg * 256 + cs_ch * 8 + i] = csum[i];
    __syncthreads();
    if (tid < 256) {
      float a = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) a += red[g * 256 + tid];
      if (r0 + tid < p.R) p.colsum[(size_t)slice * p.R + r0 + tid] = a;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// gemm_tn v2: the same product with the PING-PONG schedule of gemm_nt v5.  The reduction axis is walked in
// slabs of 32 rows (P [32][256] + Q [32][256] = one 32 KiB half-slot, ring of four, three slabs in flight,
// counted vmcnt(8)); every wave alternates LOAD (24 ds_read_b64_tr_b16 = both k-steps of a slab, its 4 DMA
// pieces of slab s+3) and COMPUTE (16 MFMAs from registers) with an s_barrier after each, and waves 4-7 run
// (a 16x16x32 port of this loop measured 3-6 % slower than 32x32x16 here, unlike in gemm_nt, and was dropped)
// one barrier behind waves 0-3, so each SIMD always has one wave in its MFMA block beside the other wave's
// LDS phase.  There are no stores in the loop, so every wave can feed the DMA ring.
#define TN_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
__global__ __launch_bounds__(NTHREADS) void gemm_tn2_kernel(TNArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wr = wave >> 2, wc = wave & 3;   // wave tile: 128 (r) x 64 (c)
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  const int tilesC = (p.C + 255) / 256;
  const int tilesR = (p.R + 255) / 256;
  const unsigned t = xcd_remap(blockIdx.x, (unsigned)(tilesR * tilesC));
  const int tr = t / tilesC, tc = t - tr * tilesC;
  const int r0 = tr * 256, c0 = tc * 256;
  const int slice = blockIdx.y;
  const long mbeg = (long)slice * p.slice_rows;
  const long mend = min((long)p.M, mbeg + p.slice_rows);
  float* O = p.O + (size_t)slice * p.R * p.ldo;

  const long rows_left = p.M - mbeg;
  auto nrec = [&](long ld, int col0) -> unsigned {
    long b = rows_left * ld * 2 - (long)col0 * 2;
    if (b < 0) b = 0;
    return (unsigned)(b > 0xffffffffL ? 0xffffffffL : b);
  };
  const u32x4 rsP = make_srd(p.P + ((size_t)mbeg * p.ldp + r0) * 2, nrec(p.ldp, r0));
  const u32x4 rsQ = make_srd(p.Q + ((size_t)mbeg * p.ldq + c0) * 2, nrec(p.ldq, c0));

  // DMA piece pc (1 KiB) = slab rows 2pc, 2pc+1 (512 B each); wave w moves pieces w and w+8 of P and of Q
  unsigned voffP[2], voffQ[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (j * 8 + wave) * 2 + (lane >> 5);
    const int chunk = (lane & 31) ^ tn_swz(row);
    voffP[j] = (unsigned)(row * p.ldp * 2 + chunk * 16);
    voffQ[j] = (unsigned)(row * p.ldq * 2 + chunk * 16);
  }
  auto stage = [&](unsigned slot, long mrow) {   // mrow relative to mbeg
    const unsigned d = lds0 + slot * HS_BYTES + wave * 1024;
    const unsigned soffP = (unsigned)(mrow * p.ldp * 2), soffQ = (unsigned)(mrow * p.ldq * 2);
    dma16(rsP, d, voffP[0], soffP);
    dma16(rsP, d + 8192, voffP[1], soffP);
    dma16(rsQ, d + 16384, voffQ[0], soffQ);
    dma16(rsQ, d + 16384 + 8192, voffQ[1], soffQ);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int ri = 0; ri < 4; ++ri)
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ri][ci][r] = 0.f;

  const int q16 = (lane >> 4) & 1, i16 = lane & 15;
  const int rsub = 8 * hi + (i16 >> 2);          // + ms*16 + 4*half
  const int csub = 16 * q16 + 4 * (i16 & 3);     // + colbase (multiple of 32)
  const bool do_colsum = p.colsum != nullptr && tc == 0;
  const int cs_ch = tid & 31, cs_rg = tid >> 5;   // 32 column chunks x 16 row groups (2 rows of a slab each)
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  const int nsl = (int)((mend - mbeg + 31) / 32);
  if (nsl <= 0) {
    // nothing to reduce in this slice: the slab is still defined (zeros) because the reduce kernel sums every slice
  } else {
    for (int s0 = 0; s0 < 3 && s0 < nsl; ++s0) stage(s0, (long)s0 * 32);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  TN_BARRIER();
  if (wr == 1) TN_BARRIER();                 // waves 4-7 run one barrier behind
  for (int sl = 0; sl < nsl; ++sl) {
    // ---- LOAD
    const char* sP = smem + (sl & 3) * HS_BYTES;
    const char* sQ = sP + 16384;
    bf16x8 fp[2][4], fq[2][2];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int row = ms * 16 + 4 * half + rsub;
        const int swz = tn_swz(row);
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) {
          const int col = wr * 128 + ri * 32 + csub;
          const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) bf16x4*)(sP + row * 512 + (((col >> 3) ^ swz) << 4) + (col & 7) * 2));
          fp[ms][ri][4 * half + 0] = v[0]; fp[ms][ri][4 * half + 1] = v[1];
          fp[ms][ri][4 * half + 2] = v[2]; fp[ms][ri][4 * half + 3] = v[3];
        }
#pragma unroll
        for (int ci = 0; ci < 2; ++ci) {
          const int col = wc * 64 + ci * 32 + csub;
          const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) bf16x4*)(sQ + row * 512 + (((col >> 3) ^ swz) << 4) + (col & 7) * 2));
          fq[ms][ci][4 * half + 0] = v[0]; fq[ms][ci][4 * half + 1] = v[1];
          fq[ms][ci][4 * half + 2] = v[2]; fq[ms][ci][4 * half + 3] = v[3];
        }
      }
    if (do_colsum) {
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int row = cs_rg * 2 + rr;
        float f[8];
        unpack8(*(const u32x4*)(sP + row * 512 + ((cs_ch ^ tn_swz(row)) << 4)), f);
#pragma unroll
        for (int i = 0; i < 8; ++i) csum[i] += f[i];
      }
    }
    if (sl + 3 < nsl) {
      stage((unsigned)((sl + 3) & 3), (long)(sl + 3) * 32);   // into the half-slot of slab sl-1
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // slab sl+1 landed; sl+2, sl+3 may be in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    TN_BARRIER();
    // ---- COMPUTE
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
      for (int ri = 0; ri < 4; ++ri)
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
          acc[ri][ci] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fp[ms][ri], fq[ms][ci], acc[ri][ci], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    TN_BARRIER();
  }
  if (wr == 0) TN_BARRIER();

#pragma unroll
  for (int ri = 0; ri < 4; ++ri)
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
      const int c = c0 + wc * 64 + ci * 32 + l31;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = r0 + wr * 128 + ri * 32 + 8 * (reg >> 2) + 4 * hi + (reg & 3);
        if (r < p.R && c < p.C) O[(size_t)r * p.ldo + c] = acc[ri][ci][reg];
      }
    }
  if (do_colsum) {
    __syncthreads();
    float* red = (float*)smem;
#pragma unroll
    for (int i = 0; i < 8; ++i) red[cs_rg * 256 + cs_ch * 8 + i] = csum[i];
    __syncthreads();
    if (tid < 256) {
      float a = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) a += red[g * 256 + tid];
      if (r0 + tid < p.R) p.colsum[(size_t)slice * p.R + r0 + tid] = a;
    }
  }
}
#undef TN_BARRIER

// out[i] = cast(sum_s slab[s][i]); out dtype bf16 or f32
template <bool OUT_BF16>
__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, void* __restrict__ out, long n, int S) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  float4 a = *(const float4*)(slabs + i);
  for (int s = 1; s < S; ++s) {
    const float4 b = *(const float4*)(slabs + (size_t)s * n + i);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  if (OUT_BF16) {
    u32x2 w; w[0] = pack2bf(a.x, a.y); w[1] = pack2bf(a.z, a.w);
    *(u32x2*)((char*)out + i * 2) = w;
  } else {
    *(float4*)((float*)out + i) = a;
  }
}

bool g_attr_done = false;
int g_num_cu = 256;
int g_nt_variant = 11;
int g_abl = 0;
int ensure_attrs() {
  if (g_attr_done) return 0;
  hipError_t e;
  e = hipFuncSetAttribute((const void*)gemm_nt_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  if (e != hipSuccess) { clipa_set_error("hipFuncSetAttribute(gemm_nt<bf16>): %s", hipGetErrorString(e)); return CLIPA_ERR_LAUNCH; }
  e = hipFuncSetAttribute((const void*)gemm_nt_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  if (e != hipSuccess) { clipa_set_error("hipFuncSetAttribute(gemm_nt<f32>): %s", hipGetErrorString(e)); return CLIPA_ERR_LAUNCH; }
  e = hipFuncSetAttribute((const void*)gemm_tn3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
  if (e != hipSuccess) { clipa_set_error("hipFuncSetAttribute(gemm_tn3): %s", hipGetErrorString(e)); return CLIPA_ERR_LAUNCH; }
  e = hipFuncSetAttribute((const void*)gemm_tn2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
  if (e != hipSuccess) { clipa_set_error("hipFuncSetAttribute(gemm_tn2): %s", hipGetErrorString(e)); return CLIPA_ERR_LAUNCH; }
  e = hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
  if (e != hipSuccess) { clipa_set_error("hipFuncSetAttribute(gemm_tn): %s", hipGetErrorString(e)); return CLIPA_ERR_LAUNCH; }
  const void* v2[6] = {(const void*)gemm_nt2_kernel<false, false>, (const void*)gemm_nt2_kernel<false, true>,
                       (const void*)gemm_nt2_kernel<true, false>, (const void*)gemm_nt2_kernel<true, true>,
                       (const void*)gemm_nt2_kernel<false, true, true>, (const void*)gemm_nt2_kernel<false, true, false, true>};
  for (int i = 0; i < 6; ++i) {
    e = hipFuncSetAttribute(v2[i], hipFuncAttributeMaxDynamicSharedMemorySize, LDS2_BYTES);
    if (e != hipSuccess) { clipa_set_error("hipFuncSetAttribute(gemm_nt2): %s", hipGetErrorString(e)); return CLIPA_ERR_LAUNCH; }
  }
  const void* v3[4] = {(const void*)gemm_nt3_kernel<CLIPA_EPI_NONE, true>, (const void*)gemm_nt3_kernel<CLIPA_EPI_ACT, true>,
                       (const void*)gemm_nt3_kernel<CLIPA_EPI_ADD, true>, (const void*)gemm_nt3_kernel<CLIPA_EPI_DACT, true>};
  for (int i = 0; i < 4; ++i) {
    e = hipFuncSetAttribute(v3[i], hipFuncAttributeMaxDynamicSharedMemorySize, LDS2_BYTES);
    if (e != hipSuccess) { clipa_set_error("hipFuncSetAttribute(gemm_nt3): %s", hipGetErrorString(e)); return CLIPA_ERR_LAUNCH; }
  }
  e = hipFuncSetAttribute((const void*)gemm_nt5_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2_BYTES);
  if (e != hipSuccess) { clipa_set_error("hipFuncSetAttribute(gemm_nt5): %s", hipGetErrorString(e)); return CLIPA_ERR_LAUNCH; }
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_num_cu = prop.multiProcessorCount;
  if (const char* v = getenv("CLIPA_GEMM_NT")) g_nt_variant = atoi(v);   // 1 = one tile per workgroup, 2 = persistent, 3 = persistent + setprio
  if (const char* v = getenv("CLIPA_GEMM_ABL")) g_abl = atoi(v);
  g_attr_done = true;
  return 0;
}

}  // namespace

extern "C" int clipa_debug_set(int gemm_nt_variant, int ablation_flags) {
  if (int rc = ensure_attrs()) return rc;
  g_nt_variant = gemm_nt_variant;
  g_abl = ablation_flags;
  return CLIPA_OK;
}

extern "C" int clipa_gemm_nt(const void* A, const void* B, void* C, void* C2, const float* bias,
                             const void* aux, int64_t M, int64_t N, int64_t K, int64_t lda,
                             int64_t ldb, int64_t ldc, int64_t ldaux, float alpha, int epi, int act,
                             int out_f32, void* stream) {
  if (M <= 0 || N <= 0) return CLIPA_OK;
  if (K <= 0 || K % 8 != 0) { clipa_set_error("gemm_nt: K=%ld must be a positive multiple of 8", (long)K); return CLIPA_ERR_ARG; }
  if (N % 8 != 0 || ldc % 8 != 0 || lda % 8 != 0 || ldb % 8 != 0) { clipa_set_error("gemm_nt: N, lda, ldb, ldc must be multiples of 8"); return CLIPA_ERR_ARG; }
  if ((epi == CLIPA_EPI_ADD || epi == CLIPA_EPI_DACT) && (!aux || ldaux % 8 != 0)) { clipa_set_error("gemm_nt: epilogue %d needs aux with ldaux%%8==0", epi); return CLIPA_ERR_ARG; }
  if (out_f32 && epi != CLIPA_EPI_NONE) { clipa_set_error("gemm_nt: f32 output supports epilogue NONE only"); return CLIPA_ERR_ARG; }
  if (256 * lda * 2 >= (1L << 30) || 256 * ldb * 2 >= (1L << 30)) { clipa_set_error("gemm_nt: leading dimension too large"); return CLIPA_ERR_ARG; }
  if (int rc = ensure_attrs()) return rc;
  NTArgs a;
  a.A = (const char*)A; a.B = (const char*)B; a.C = (char*)C; a.C2 = (char*)C2; a.bias = bias; a.aux = (const char*)aux;
  a.M = (int)M; a.N = (int)N; a.K = (int)K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldaux = ldaux;
  a.alpha = alpha; a.epi = epi; a.act = act; a.abl = g_abl;
  const long tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  hipStream_t st = (hipStream_t)stream;
  // 5 (default): loader/storer wave roles for plain (bias-only) epilogues - measured +11..15 % on the K = 1024
  // shapes - while fused-activation / residual epilogues stay on the all-waves epilogue of v2 (their epilogue
  // maths is VALU-bound and wants all eight waves; measured 10..30 % slower under the role split);
  // 6: role split for every epilogue (experiments).
  // 7: staggered two-group main loop ("nt4") for every bf16-output GEMM; 8: nt4 for the fused epilogues, role
  // split for plain ones.
  // 9: v5 (ping-pong + wave roles) for plain / bias epilogues, ping-pong v2 for the fused ones.
  if (g_nt_variant == 9 && epi == CLIPA_EPI_NONE && !out_f32 && K >= 128) {
    const unsigned grid = (unsigned)(tiles < g_num_cu ? tiles : g_num_cu);
    hipLaunchKernelGGL(gemm_nt5_kernel, dim3(grid), dim3(NTHREADS), LDS2_BYTES, st, a);
    return clipa_check_launch("gemm_nt5");
  }
  if ((g_nt_variant == 7 || g_nt_variant == 9 || (g_nt_variant == 8 && epi != CLIPA_EPI_NONE)) && !out_f32 && K >= 96) {
    const unsigned grid = (unsigned)(tiles < g_num_cu ? tiles : g_num_cu);
    hipLaunchKernelGGL((gemm_nt2_kernel<false, true, true>), dim3(grid), dim3(NTHREADS), LDS2_BYTES, st, a);
    return clipa_check_launch("gemm_nt4");
  }
  // 10: 16x16x32 MFMAs (v2 ring) for the fused epilogues, roles for plain; 11: 16x16x32 v2 for everything.
  if (g_nt_variant >= 5 && g_nt_variant != 11 && !out_f32 && K >= 3 * BK && (epi == CLIPA_EPI_NONE || g_nt_variant == 6)) {
    const unsigned grid = (unsigned)(tiles < g_num_cu ? tiles : g_num_cu);
    if (epi == CLIPA_EPI_ACT) hipLaunchKernelGGL((gemm_nt3_kernel<CLIPA_EPI_ACT, true>), dim3(grid), dim3(NTHREADS), LDS2_BYTES, st, a);
    else if (epi == CLIPA_EPI_ADD) hipLaunchKernelGGL((gemm_nt3_kernel<CLIPA_EPI_ADD, true>), dim3(grid), dim3(NTHREADS), LDS2_BYTES, st, a);
    else if (epi == CLIPA_EPI_DACT) hipLaunchKernelGGL((gemm_nt3_kernel<CLIPA_EPI_DACT, true>), dim3(grid), dim3(NTHREADS), LDS2_BYTES, st, a);
    else hipLaunchKernelGGL((gemm_nt3_kernel<CLIPA_EPI_NONE, true>), dim3(grid), dim3(NTHREADS), LDS2_BYTES, st, a);
    return clipa_check_launch("gemm_nt3");
  }
  if (g_nt_variant >= 2) {
    const unsigned grid = (unsigned)(tiles < g_num_cu ? tiles : g_num_cu);
    const bool prio = g_nt_variant == 3 || g_nt_variant == 5;
    if (!out_f32 && (g_nt_variant == 10 || g_nt_variant == 11)) hipLaunchKernelGGL((gemm_nt2_kernel<false, true, false, true>), dim3(grid), dim3(NTHREADS), LDS2_BYTES, st, a);
    else if (out_f32 && prio) hipLaunchKernelGGL((gemm_nt2_kernel<true, true>), dim3(grid), dim3(NTHREADS), LDS2_BYTES, st, a);
    else if (out_f32) hipLaunchKernelGGL((gemm_nt2_kernel<true, false>), dim3(grid), dim3(NTHREADS), LDS2_BYTES, st, a);
    else if (prio) hipLaunchKernelGGL((gemm_nt2_kernel<false, true>), dim3(grid), dim3(NTHREADS), LDS2_BYTES, st, a);
    else hipLaunchKernelGGL((gemm_nt2_kernel<false, false>), dim3(grid), dim3(NTHREADS), LDS2_BYTES, st, a);
    return clipa_check_launch("gemm_nt2");
  }
  if (out_f32) hipLaunchKernelGGL(gemm_nt_kernel<true>, dim3((unsigned)tiles), dim3(NTHREADS), LDS_BYTES, st, a);
  else hipLaunchKernelGGL(gemm_nt_kernel<false>, dim3((unsigned)tiles), dim3(NTHREADS), LDS_BYTES, st, a);
  return clipa_check_launch("gemm_nt");
}

extern "C" int64_t clipa_gemm_tn_workspace(int64_t M, int64_t R, int64_t C, int64_t* nslices) {
  const long tiles = ((R + 255) / 256) * ((C + 255) / 256);
  const long mt = (M + 63) / 64;
  // ~4 workgroups per CU, but never a thin last round: one workgroup per CU is resident (128 KiB LDS), so a grid of
  // 4.1 x #CUs costs five rounds.  Take the largest slice count <= 64 whose grid fills >= 97 % of its rounds.
  long S = (4L * g_num_cu) / tiles;
  if (S > 64) S = 64;
  if (S < 1) S = 1;
  for (long c = S; c >= 1; --c) {
    const long w = tiles * c, rounds = (w + g_num_cu - 1) / g_num_cu;
    if (w * 100 >= rounds * g_num_cu * 97 || c == 1) { if (w * 100 >= rounds * g_num_cu * 97) S = c; break; }
  }
  if (S > mt) S = mt;
  if (S < 1) S = 1;
  if (mt > 0) { const long per = (mt + S - 1) / S; S = (mt + per - 1) / per; }
  if (nslices) *nslices = S;
  return (S * R * C + S * R) * (int64_t)sizeof(float);   // split-M slabs + partial column sums
}

extern "C" int clipa_gemm_tn(const void* P, const void* Q, void* out, float* colsum_out, int64_t M, int64_t R,
                             int64_t C, int64_t ldp, int64_t ldq, int out_bf16, void* workspace,
                             int64_t workspace_bytes, void* stream) {
  if (R <= 0 || C <= 0) return CLIPA_OK;
  if (R % 8 != 0 || C % 8 != 0 || ldp % 8 != 0 || ldq % 8 != 0) { clipa_set_error("gemm_tn: R, C, ldp, ldq must be multiples of 8"); return CLIPA_ERR_ARG; }
  int64_t S = 1;
  const int64_t need = clipa_gemm_tn_workspace(M, R, C, &S);
  if (workspace_bytes < need || !workspace) { clipa_set_error("gemm_tn: workspace %ld < %ld bytes", (long)workspace_bytes, (long)need); return CLIPA_ERR_ARG; }
  const long mt = (M + 63) / 64;
  const long slice_rows = ((mt + S - 1) / S) * 64;
  if (slice_rows * ldp * 2 >= (1L << 32) - (1 << 24) || slice_rows * ldq * 2 >= (1L << 32) - (1 << 24)) { clipa_set_error("gemm_tn: slice too large for 32-bit buffer offsets"); return CLIPA_ERR_ARG; }
  if (int rc = ensure_attrs()) return rc;
  TNArgs a;
  a.P = (const char*)P; a.Q = (const char*)Q; a.O = (float*)workspace;
  a.M = (int)M; a.R = (int)R; a.C = (int)C; a.ldp = ldp; a.ldq = ldq; a.ldo = C; a.slice_rows = (int)slice_rows;
  a.colsum = colsum_out ? (float*)workspace + S * R * C : nullptr;
  const long tiles = ((R + 255) / 256) * ((C + 255) / 256);
  // ablation bit 9 (512) selects the first-generation kernel (one barrier per 64 rows, all waves in step)
  // Default: the 16x16x32 first-generation schedule (v3) for the image tower's wide-P products (in-proj and c_fc
  // weight gradients) and its 1024 x 1024 out-proj, the ping-pong kernel (v2) elsewhere (c_proj, the text tower) - per-shape winners of tools/tn_ab.py, all within
  // +-4 % except the text tower (v2 +18 %).  Ablation bits force one kernel: 512 v1, 1024 v3, 2048 v2.
  const bool use_v3 = (g_abl & 1024) || (!(g_abl & (512 | 2048)) && ((R >= 3072 && C >= 1024) || (R == 1024 && C == 1024)));
  if (use_v3) hipLaunchKernelGGL(gemm_tn3_kernel, dim3((unsigned)tiles, (unsigned)S), dim3(NTHREADS), 2 * STAGE_BYTES, (hipStream_t)stream, a);
  else if (g_abl & 512) hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)tiles, (unsigned)S), dim3(NTHREADS), 2 * STAGE_BYTES, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(gemm_tn2_kernel, dim3((unsigned)tiles, (unsigned)S), dim3(NTHREADS), 2 * STAGE_BYTES, (hipStream_t)stream, a);
  if (int rc = clipa_check_launch("gemm_tn")) return rc;
  const long n = R * C;
  const unsigned blocks = (unsigned)((n / 4 + 255) / 256);
  if (out_bf16) hipLaunchKernelGGL(reduce_slabs_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, out, n, (int)S);
  else hipLaunchKernelGGL(reduce_slabs_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, out, n, (int)S);
  if (colsum_out)   // bias gradient: sum the per-slice partial column sums of P
    hipLaunchKernelGGL(reduce_slabs_kernel<false>, dim3((unsigned)((R / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)a.colsum, (void*)colsum_out, (long)R, (int)S);
  return clipa_check_launch("gemm_tn_reduce");
}
