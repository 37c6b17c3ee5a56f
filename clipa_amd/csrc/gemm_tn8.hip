// gemm_tn8: the fp8 weight-gradient GEMM for gfx950 (MI355X) on v_mfma_f32_16x16x128_f8f6f4.
//
//   O[R,C] = alpha * sum_m P8[m,R] * Q8[m,C]        P8, Q8: fp8 bytes, the reduction index m is the slow axis of both
//
// Call sites: the weight gradients of the four linear layers of a residual block in fp8 mode - the autograd transposes of
// nn.Linear / packed in-proj / out-proj (clipa_torch/open_clip/transformer.py:209,217-219,234); the reference has no fp8 mode
// (training/params.py:195-200 stops at bf16), BASELINE.json configs[3] asks for "fp8 MFMA weights/activations".
//
// Scaling (clipa_amd/engine.py: _wgrad8).  The reduction runs over tokens, so a per-token scale cannot be factored out of the
// product.  P8 is the row-quantised gradient dq[m,:] = e(dY[m,:] / ds[m]) the input-gradient GEMM already consumes; the
// activation operand absorbs the gradient's row scale BEFORE it is quantised, Q8[m,:] = e4m3(ds[m] * X[m,:] / t), with ONE
// scalar t = max_m ds[m] * sx[m] (sx = the activation's own row scale from the forward pass, so |Q8| <= 448 exactly and nothing
// saturates): dW = sum_m dY[m,:]^T X[m,:] = t * sum_m dq[m,:]^T Q8[m,:].  t stays on the device (alpha_dev).
//
// gemm_tn8_kernel: whole 256 x 256 tiles, split-M slices of an even number (>= 4) of 128-row K steps - the structure of
// gemm_tna.hip (one wave per SIMD, 128 x 128 wave tile in a[0:255], the K loop ONE generated inline-asm statement,
// tools/gen_gemm_tn8.py -> gemm_tn8_asm.inc) with gemm_f8a.hip's step (64 MFMAs over 128 k-values from 128 registers of
// fragments, read-ahead following the registers as they die).  Fragments: `ds_read_b64_tr_b8` over [128 m][256 B] LDS images
// filled by LDS-DMA with a source-side chunk swizzle.  fp32 split-M slabs + a fixed-order reduce that applies alpha.
// gemm_tn8_generic_kernel: any shape (test-size models, ragged row counts): the same MFMA fed by byte gathers from global
// memory; also takes the row remainder of a whole-tile product as one more slab.
#include "gemm_common.h"
#include "gemm_tn8_asm.inc"
#include <utility>

namespace clipa_gemm {
namespace {

struct TN8Args {
  const char* P; const char* Q; float* O;
  int M, R, C;
  long ldp, ldq, ldo;      // ldp, ldq: bytes
  int slice_rows;          // multiple of 256 (fast kernel) / 128 (generic)
  int nslices;             // > 0: 1-D grid, XCD x owns the M slices x, x + 8, ...
  long m_first;            // generic kernel: first row it covers
  int abl;
};

constexpr int TN8_THREADS = 256;
constexpr int TN8_LDS = 2 * STAGE_BYTES;

template <int IDX>
__device__ __forceinline__ float t8acc_rd() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(IDX));
  return x;
}
// block I = 8 ri + ci of the wave's 8 x 8: lane holds O[rblock + 4 (lane >> 4) + e][cblock + (lane & 15)], e = 0..3
template <int I>
__device__ __forceinline__ void tn8_store_block(float* o, long ldo) {
  constexpr int RI = I >> 3, CI = I & 7;
  float* q = o + (size_t)(RI * 16) * ldo + CI * 16;
  q[0] = t8acc_rd<4 * I + 0>();
  q[ldo] = t8acc_rd<4 * I + 1>();
  q[2 * ldo] = t8acc_rd<4 * I + 2>();
  q[3 * ldo] = t8acc_rd<4 * I + 3>();
}
template <int... Is>
__device__ __forceinline__ void tn8_store_all(float* o, long ldo, std::integer_sequence<int, Is...>) {
  (tn8_store_block<Is>(o, ldo), ...);
}

#define TN8_OPERANDS                                                                                                        \
  : [skP] "=&s"(skP), [skQ] "=&s"(skQ), [cnt] "=&s"(cnt)                                                                    \
  : [vP] "v"(vP), [vQ] "v"(vQ), [voffP] "v"(voffP), [voffQ] "v"(voffQ), [curP] "s"(useP), [curQ] "s"(useQ),                 \
    [nulP] "s"(nul), [nulQ] "s"(nul), [sP16] "s"(sP16), [sQ16] "s"(sQ16), [sP128] "s"(sP128), [sQ128] "s"(sQ128),           \
    [ldsw] "s"(ldsw), [nloop] "s"(nloop)                                                                                    \
  : "memory", "scc", TN8_CLOBBERS

// FMT_P: 0 = e4m3, 1 = e5m2 gradient operand (the first MFMA source: cbsz); the activation operand is e4m3
template <int SCHED, int FMT_P>
__global__ __launch_bounds__(TN8_THREADS) void gemm_tn8_kernel(TN8Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;            // wave tile 128 (r) x 128 (c)
  const int g4 = lane >> 4, i16 = lane & 15;

  const int tilesC = p.C / 256, tilesR = p.R / 256;
  unsigned t;
  int slice;
  if (p.nslices > 0) {      // slice-per-XCD order (gemm_tn.hip: tn_per_xcd)
    const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3, nt = (unsigned)(tilesR * tilesC);
    slice = (int)(xcd + 8u * (j / nt));
    t = j % nt;
    if (slice >= p.nslices) return;
  } else {
    t = xcd_remap(blockIdx.x, (unsigned)(tilesR * tilesC));
    slice = blockIdx.y;
  }
  const int tr = t / tilesC, tc = t - tr * tilesC;
  const int r0 = tr * 256, c0 = tc * 256;
  const long mbeg = (long)slice * p.slice_rows;
  const long mend = min((long)p.M, mbeg + p.slice_rows);
  const int nmt = (int)((mend - mbeg) / 128);         // even, >= 4 (tn8_fast_rows)

  const unsigned rows = (unsigned)(mend - mbeg);
  const u32x4 curP = make_srd(p.P + (size_t)mbeg * p.ldp + r0, (unsigned)min((long)0xffffff00L, (long)rows * p.ldp));
  const u32x4 curQ = make_srd(p.Q + (size_t)mbeg * p.ldq + c0, (unsigned)min((long)0xffffff00L, (long)rows * p.ldq));
  const u32x4 nul = make_srd(p.P, 0u);
  // ablations (clipa_internal_debug_set flags; wrong results): 65536 = operands never fetched, 131072 = every K step re-reads
  // the slice's first 128 rows (operand bytes stay L2-resident)
  const int abl = p.abl;
  const u32x4 useP = (abl & 65536) ? nul : curP, useQ = (abl & 65536) ? nul : curQ;

  // per-lane constants (gen_gemm_tn8.py: register map).  Transposing read: lanes 2j, 2j + 1 of a 16-lane group supply the two
  // 8-byte halves of m-row 8 g + j (+ 32 i by instruction offset); chunk position = (8 w + block) ^ (row & 15)
  const unsigned smem_base = (unsigned)(size_t)LDS_PTR(smem);
  const int row0 = 8 * g4 + (i16 >> 1);
  const int f = row0 & 15;
  const unsigned lane_base = smem_base + (unsigned)(row0 * 256 + 8 * (i16 & 1));
  const unsigned vP = lane_base + (unsigned)((((wr * 8) ^ f)) << 4);
  const unsigned vQ = lane_base + (unsigned)(IMG_BYTES + ((((wc * 8) ^ f)) << 4));
  // LDS-DMA piece j of this wave = image rows 16 j + rb, rb = 4 wave + (lane >> 4); the 16-byte chunk position (lane & 15) of
  // the row holds source chunk (lane & 15) ^ (row & 15) = (lane & 15) ^ rb
  const int rb = 4 * wave + g4;
  const int ch = i16 ^ rb;
  const unsigned voffP = (unsigned)(rb * (int)p.ldp + ch * 16), voffQ = (unsigned)(rb * (int)p.ldq + ch * 16);
  const unsigned sP16 = (unsigned)(16 * p.ldp), sQ16 = (unsigned)(16 * p.ldq);            // bytes per 16 rows
  const unsigned sP128 = (abl & 131072) ? 0u : (unsigned)(128 * p.ldp), sQ128 = (abl & 131072) ? 0u : (unsigned)(128 * p.ldq);
  const unsigned ldsw = (unsigned)__builtin_amdgcn_readfirstlane((int)(smem_base + wave * 1024));
  const unsigned nloop = (unsigned)(nmt / 2 - 2);

  unsigned skP, skQ, cnt;
  if constexpr (SCHED == 0) {
    if constexpr (FMT_P == 1) asm volatile(TN8_ASM_0(" cbsz:1") TN8_OPERANDS);
    else asm volatile(TN8_ASM_0("") TN8_OPERANDS);
  } else if constexpr (SCHED == 1) {
    if constexpr (FMT_P == 1) asm volatile(TN8_ASM_1(" cbsz:1") TN8_OPERANDS);
    else asm volatile(TN8_ASM_1("") TN8_OPERANDS);
  } else {
    if constexpr (FMT_P == 1) asm volatile(TN8_ASM_2(" cbsz:1") TN8_OPERANDS);
    else asm volatile(TN8_ASM_2("") TN8_OPERANDS);
  }

  // fp32 tile of this slice: lane holds rows rblock + 4 g4 + e, column cblock + i16
  int tid_e = tid;
  asm volatile("" : "+v"(tid_e));
  const int g4e = (tid_e & 63) >> 4, i16e = tid_e & 15;
  float* O = p.O + (size_t)slice * p.R * p.ldo + (size_t)(r0 + wr * 128 + 4 * g4e) * p.ldo + (c0 + wc * 128 + i16e);
  tn8_store_all(O, p.ldo, std::make_integer_sequence<int, 64>{});
}

typedef int i32x8 __attribute__((ext_vector_type(8)));

// Any shape: a wave per 16 x 16 output block and slice; lane (g = lane >> 4, i = lane & 15) gathers bytes m = 32 g .. 32 g + 31
// of the step for its column i from global memory (rows / columns beyond the matrix read zero).  Test sizes and row
// remainders only - it makes no attempt at speed.
template <int FMT_P>
__global__ __launch_bounds__(256) void gemm_tn8_generic_kernel(TN8Args p) {
  const int lane = threadIdx.x & 63;
  const int g = lane >> 4, i = lane & 15;
  const long blocksC = (p.C + 15) / 16, blocksR = (p.R + 15) / 16;
  const long blk = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (blk >= blocksR * blocksC) return;
  const int rbk = (int)(blk / blocksC), cbk = (int)(blk - (long)rbk * blocksC);
  const int slice = blockIdx.y;
  const long mbeg = p.m_first + (long)slice * p.slice_rows;
  const long mend = min((long)p.M, mbeg + p.slice_rows);
  const int r = rbk * 16 + i, c = cbk * 16 + i;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long m0 = mbeg; m0 < mend; m0 += 128) {
    i32x8 a, b;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      unsigned wa = 0, wb = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long m = m0 + 32 * g + 4 * w + k;
        const bool in = m < mend;
        const unsigned pa = (in && r < p.R) ? (unsigned)(unsigned char)p.P[(size_t)m * p.ldp + r] : 0u;
        const unsigned qb = (in && c < p.C) ? (unsigned)(unsigned char)p.Q[(size_t)m * p.ldq + c] : 0u;
        wa |= pa << (8 * k);
        wb |= qb << (8 * k);
      }
      a[w] = (int)wa;
      b[w] = (int)wb;
    }
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, FMT_P, 0, 0, 0, 0, 0);
  }
  float* O = p.O + (size_t)slice * p.R * p.ldo;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int rr = rbk * 16 + 4 * g + e;
    if (rr < p.R && c < p.C) O[(size_t)rr * p.ldo + c] = acc[e];
  }
}

// out[i] = cast(alpha * sum_s slab[s][i]) in slab order; alpha = host factor x optional device scalar
template <bool OUT_BF16>
__global__ void reduce_slabs_alpha_kernel(const float* __restrict__ slabs, void* __restrict__ out, long n, int S, float alpha,
                                          const float* __restrict__ alpha_dev) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  const float al = alpha_dev ? alpha * alpha_dev[0] : alpha;
  float4 a = *(const float4*)(slabs + i);
  for (int s = 1; s < S; ++s) {
    const float4 b = *(const float4*)(slabs + (size_t)s * n + i);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  a.x *= al; a.y *= al; a.z *= al; a.w *= al;
  if (OUT_BF16) {
    u32x2 w; w[0] = pack2bf(a.x, a.y); w[1] = pack2bf(a.z, a.w);
    *(u32x2*)((char*)out + i * 2) = w;
  } else {
    *(float4*)((float*)out + i) = a;
  }
}

std::once_flag g_tn8_once[MAX_DEVICES];
int g_tn8_rc[MAX_DEVICES];

// schedule 2 (four column-major tail rows: one transposing read per MFMA slot throughout) wins 14 of the 16 production shapes by
// 1-5 % over schedule 1 (profiles/r06_gemm_tn8_schedules_and_work_orders.jsonl)
constexpr int TN8_DEFAULT_SCHEDULE = 2;

struct TN8Plan {
  long fast_rows;      // rows the four-wave kernel covers (0: none)
  long slice_rows;     // its slice (multiple of 256)
  long S_fast;         // its slabs
  bool per_xcd;
  long gen_slice;      // generic kernel: rows per slab (multiple of 128)
  long S_gen;          // its slabs
};

// Rows [0, fast_rows) in S_fast slices of an even number (>= 4) of 128-row steps on the four-wave kernel (whole 256 x 256 tiles
// only), the rest - and every other shape - on the generic kernel in at most 32 slabs.
TN8Plan tn8_plan(long M, long R, long C, int num_cu) {
  TN8Plan pl{0, 0, 0, false, 0, 0};
  if (R % 256 == 0 && C % 256 == 0 && M >= 512) {      // (tiles of whole 16-byte chunks: R, C % 256 == 0; the row strides are checked by the caller)
    pl.per_xcd = tn_per_xcd(M, R, C);
    const long S0 = tn_slices(M, R, C, num_cu, pl.per_xcd);
    const long mt = (M + 63) / 64;
    long sr = ((mt + S0 - 1) / S0) * 64;
    sr = (sr + 255) / 256 * 256;
    if (sr < 512) sr = 512;
    long S = M / sr;                                   // whole slices
    long rest = M - S * sr;
    long fast = S * sr;
    if (rest >= 512) {                                 // a shorter last slice, still an even number >= 4 of steps
      const long lastr = rest / 256 * 256;
      fast += lastr;
      S += 1;
    }
    pl.fast_rows = fast; pl.slice_rows = sr; pl.S_fast = S;
  }
  const long left = M - pl.fast_rows;
  if (left > 0) {
    long gs = ((left + 31) / 32 + 127) / 128 * 128;
    if (gs < 128) gs = 128;
    pl.gen_slice = gs;
    pl.S_gen = (left + gs - 1) / gs;
  }
  return pl;
}

template <int FMT_P>
int tn8_set_attrs() {
  int rc = 0;
  const void* k[3] = {(const void*)gemm_tn8_kernel<0, FMT_P>, (const void*)gemm_tn8_kernel<1, FMT_P>, (const void*)gemm_tn8_kernel<2, FMT_P>};
  for (int i = 0; i < 3; ++i) {
    const hipError_t e = hipFuncSetAttribute(k[i], hipFuncAttributeMaxDynamicSharedMemorySize, TN8_LDS);
    if (e != hipSuccess) { clipa_set_error("hipFuncSetAttribute(gemm_tn8): %s", hipGetErrorString(e)); rc = CLIPA_ERR_LAUNCH; }
  }
  return rc;
}

template <int FMT_P>
void tn8_launch_fast(const TN8Args& a, dim3 grid, int sched, hipStream_t st) {
  if (sched == 0) hipLaunchKernelGGL((gemm_tn8_kernel<0, FMT_P>), grid, dim3(TN8_THREADS), TN8_LDS, st, a);
  else if (sched == 2) hipLaunchKernelGGL((gemm_tn8_kernel<2, FMT_P>), grid, dim3(TN8_THREADS), TN8_LDS, st, a);
  else hipLaunchKernelGGL((gemm_tn8_kernel<1, FMT_P>), grid, dim3(TN8_THREADS), TN8_LDS, st, a);
}

}  // namespace
}  // namespace clipa_gemm

using namespace clipa_gemm;

extern "C" int64_t clipa_gemm_tn_f8_workspace(int64_t M, int64_t R, int64_t C) {
  int dev = 0;
  const int ncu = (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < MAX_DEVICES) ? gemm_num_cu(dev) : 256;
  const TN8Plan pl = tn8_plan(M, R, C, ncu);
  return (pl.S_fast + pl.S_gen) * R * C * (int64_t)sizeof(float);
}

extern "C" int clipa_gemm_tn_f8(const void* P8, const void* Q8, void* out, int64_t M, int64_t R, int64_t C, int64_t ldp,
                                int64_t ldq, float alpha, const float* alpha_dev, int fmt_p, int out_bf16, void* workspace,
                                int64_t workspace_bytes, void* stream) {
  if (R <= 0 || C <= 0) return CLIPA_OK;
  if (M <= 0) { clipa_set_error("gemm_tn_f8: M must be positive"); return CLIPA_ERR_ARG; }
  if (R % 8 != 0 || C % 8 != 0 || ldp < R || ldq < C) { clipa_set_error("gemm_tn_f8: R, C must be multiples of 8 and ldp >= R, ldq >= C"); return CLIPA_ERR_ARG; }
  if (fmt_p != 0 && fmt_p != 1) { clipa_set_error("gemm_tn_f8: fmt_p is 0 (e4m3) or 1 (e5m2)"); return CLIPA_ERR_ARG; }
  int dev = 0;
  if (int rc = current_device(&dev)) return rc;
  const int abl = g_abl.load(std::memory_order_relaxed);
  // the four-wave kernel moves 16-byte chunks: operands it cannot address that way take the byte-gather kernel (same workspace bound)
  const bool aligned = ldp % 16 == 0 && ldq % 16 == 0 && ((size_t)P8 & 15) == 0 && ((size_t)Q8 & 15) == 0;
  TN8Plan pl = tn8_plan(M, R, C, gemm_num_cu(dev));
  if (!aligned && pl.S_fast > 0) {
    pl.S_gen = pl.S_fast + pl.S_gen; pl.S_fast = 0; pl.fast_rows = 0;
    pl.gen_slice = ((M + pl.S_gen - 1) / pl.S_gen + 127) / 128 * 128;
    pl.S_gen = (M + pl.gen_slice - 1) / pl.gen_slice;
  }
  const int64_t S = pl.S_fast + pl.S_gen;
  const int64_t need = S * R * C * (int64_t)sizeof(float);
  if (workspace_bytes < need || !workspace) { clipa_set_error("gemm_tn_f8: workspace %ld < %ld bytes", (long)workspace_bytes, (long)need); return CLIPA_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  if (pl.S_fast > 0) {
    if (pl.slice_rows * ldp >= (1L << 32) - (1 << 24) || pl.slice_rows * ldq >= (1L << 32) - (1 << 24)) { clipa_set_error("gemm_tn_f8: slice too large for 32-bit buffer offsets"); return CLIPA_ERR_ARG; }
    std::call_once(g_tn8_once[dev], [dev]() { g_tn8_rc[dev] = tn8_set_attrs<0>() | tn8_set_attrs<1>(); });
    if (g_tn8_rc[dev]) return g_tn8_rc[dev];
    TN8Args a;
    a.P = (const char*)P8; a.Q = (const char*)Q8; a.O = (float*)workspace;
    a.M = (int)pl.fast_rows; a.R = (int)R; a.C = (int)C; a.ldp = ldp; a.ldq = ldq; a.ldo = C;
    a.slice_rows = (int)pl.slice_rows; a.nslices = pl.per_xcd ? (int)pl.S_fast : 0; a.m_first = 0; a.abl = abl;
    const long tiles = (R / 256) * (C / 256);
    const dim3 grid = pl.per_xcd ? dim3((unsigned)(tiles * ((pl.S_fast + 7) / 8 * 8)), 1) : dim3((unsigned)tiles, (unsigned)pl.S_fast);
    // experiment flags: bits 26..27 select the schedule (1 + value; 0 = default)
    const int sel = (abl >> 26) & 3;
    const int sched = sel ? sel - 1 : TN8_DEFAULT_SCHEDULE;
    note_gemm(8);
    if (fmt_p == 1) tn8_launch_fast<1>(a, grid, sched, st);
    else tn8_launch_fast<0>(a, grid, sched, st);
    if (int rc = clipa_check_launch("gemm_tn8")) return rc;
  }
  if (pl.S_gen > 0) {
    TN8Args a;
    a.P = (const char*)P8; a.Q = (const char*)Q8; a.O = (float*)workspace + pl.S_fast * R * C;
    a.M = (int)M; a.R = (int)R; a.C = (int)C; a.ldp = ldp; a.ldq = ldq; a.ldo = C;
    a.slice_rows = (int)pl.gen_slice; a.nslices = 0; a.m_first = pl.fast_rows; a.abl = abl;
    const long blocks = ((R + 15) / 16) * ((C + 15) / 16);
    const dim3 grid((unsigned)((blocks + 3) / 4), (unsigned)pl.S_gen);
    if (pl.S_fast == 0) note_gemm(9);
    if (fmt_p == 1) hipLaunchKernelGGL(gemm_tn8_generic_kernel<1>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(gemm_tn8_generic_kernel<0>, grid, dim3(256), 0, st, a);
    if (int rc = clipa_check_launch("gemm_tn8_generic")) return rc;
  }
  const long n = R * C;
  const unsigned blocks = (unsigned)((n / 4 + 255) / 256);
  if (out_bf16) hipLaunchKernelGGL(reduce_slabs_alpha_kernel<true>, dim3(blocks), dim3(256), 0, st, (const float*)workspace, out, n, (int)S, alpha, alpha_dev);
  else hipLaunchKernelGGL(reduce_slabs_alpha_kernel<false>, dim3(blocks), dim3(256), 0, st, (const float*)workspace, out, n, (int)S, alpha, alpha_dev);
  return clipa_check_launch("gemm_tn8_reduce");
}
