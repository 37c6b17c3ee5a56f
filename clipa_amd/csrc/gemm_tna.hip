// gemm_tna: the bf16 weight-gradient GEMM of gemm_tn.hip on FOUR waves of 512 registers with a hand-scheduled main loop.
//
//   O[slice][R,C] = sum_{m in slice} P[m,R] * Q[m,C]      R % 256 == 0, C % 256 == 0, every slice a multiple of 128 rows (>= 256)
//
// Same reference call sites as gemm_tn.hip (the autograd transposes of nn.Linear / in-proj / out-proj,
// clipa_torch/open_clip/transformer.py:209,217-219,234), same split-M fp32 slabs + reduce_slabs_kernel, same LDS image
// ([64 m][256] bf16, tn_swz16) and `ds_read_b64_tr_b16` fragments as gemm_tn3_kernel.  What differs is the execution structure
// (the one gemm_nta.hip introduced for the NT product): one wave per SIMD with a 128 x 128 wave tile (64 accumulator blocks in
// a[0:255]), all 32 fragments of a K step in registers so that the LDS slot is re-filled two steps ahead, and the whole K loop
// of the workgroup as ONE generated inline-asm statement (tools/gen_gemm_tna.py -> gemm_tna_asm.inc).  The weight-gradient
// product has no epilogue to speak of (one fp32 tile per ~800 K steps), so its rate is its main loop's - which in gemm_tn2/3
// is issue- and latency-bound (64 transposing reads per wave and step between barriers), not power-bound like gemm_nt's.
// The bias gradient (column sums of P = dY) rides the matrix pipe: P^T . ones, 16 extra MFMAs per step in the workgroups of
// tile column 0 only.
#include "gemm_common.h"
#include "gemm_tna_asm.inc"
#include <utility>

namespace clipa_gemm {
namespace {

constexpr int TNA_THREADS = 256;
constexpr int TNA_LDS = 2 * STAGE_BYTES;

template <int IDX>
__device__ __forceinline__ float tacc_rd() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(IDX));
  return x;
}

// block I = 8 ri + ci of the wave's 8 x 8: lane holds O[rblock + 4 (lane >> 4) + e][cblock + (lane & 15)], e = 0..3
template <int I>
__device__ __forceinline__ void tna_store_block(float* o, long ldo) {
  constexpr int RI = I >> 3, CI = I & 7;
  float* q = o + (size_t)(RI * 16) * ldo + CI * 16;
  q[0] = tacc_rd<4 * I + 0>();
  q[ldo] = tacc_rd<4 * I + 1>();
  q[2 * ldo] = tacc_rd<4 * I + 2>();
  q[3 * ldo] = tacc_rd<4 * I + 3>();
}
template <int... Is>
__device__ __forceinline__ void tna_store_all(float* o, long ldo, std::integer_sequence<int, Is...>) {
  (tna_store_block<Is>(o, ldo), ...);
}

#define TNA_INPUTS                                                                                                          \
  [vP] "v"(vP), [vQ] "v"(vQ), [vPe] "v"(vPe), [vPo] "v"(vPo), [vQe] "v"(vQe), [vQo] "v"(vQo), [curP] "s"(useP),             \
  [curQ] "s"(useQ), [nulP] "s"(nul), [nulQ] "s"(nul), [sP16] "s"(sP16), [sQ16] "s"(sQ16), [sP64] "s"(sP64),                 \
  [sQ64] "s"(sQ64), [ldsw] "s"(ldsw), [nloop] "s"(nloop)

template <int SCHED>
__global__ __launch_bounds__(TNA_THREADS) void gemm_tna_kernel(TNArgs p) {
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
  const int nmt = (int)((mend - mbeg) / 64);          // even, >= 4 (tna_eligible)

  // descriptors: base = first row of the slice, first column of the tile; the extent covers the slice's rows
  const unsigned rows = (unsigned)(mend - mbeg);
  const u32x4 curP = make_srd(p.P + ((size_t)mbeg * p.ldp + r0) * 2, (unsigned)min((long)0xffffff00L, (long)rows * p.ldp * 2));
  const u32x4 curQ = make_srd(p.Q + ((size_t)mbeg * p.ldq + c0) * 2, (unsigned)min((long)0xffffff00L, (long)rows * p.ldq * 2));
  const u32x4 nul = make_srd(p.P, 0u);
  // ablations (clipa_internal_debug_set flags; wrong results): 65536 = operands never fetched (the LDS-DMA write zeros: LDS traffic
  // without L2 traffic), 131072 = every K step re-reads the slice's first 64 rows (operand bytes stay L2-resident)
  const int abl = p.abl;
  const u32x4 useP = (abl & 65536) ? nul : curP, useQ = (abl & 65536) ? nul : curQ;

  // per-lane constants (gen_gemm_tna.py: register map)
  const unsigned smem_base = (unsigned)(size_t)LDS_PTR(smem);
  const int rsub = 8 * g4 + (i16 >> 2), csub = 4 * (i16 & 3);
  const unsigned swz4 = (unsigned)((((((i16 >> 2) & 3) << 2) ^ ((g4 & 1) << 1))) << 4);       // tn_swz16(row) << 4
  const unsigned lane_base = smem_base + (unsigned)(rsub * 512 + (csub & 7) * 2 + ((csub >> 3) << 4));
  const unsigned vP = (lane_base + (unsigned)(wr * 256)) ^ swz4;
  const unsigned vQ = (lane_base + (unsigned)(IMG_BYTES + wc * 256)) ^ swz4;
  // LDS-DMA piece j of this wave = image rows 8 j + rb, rb = 2 wave + (lane >> 5); the 16-byte chunk (lane & 31) of the row
  // holds source chunk (lane & 31) ^ tn_swz16(row) = (lane & 31) ^ ((rb & 3) << 2) ^ ((j & 1) << 1)
  const int rb = 2 * wave + (lane >> 5);
  const int ch = (lane & 31) ^ ((rb & 3) << 2);
  const unsigned vPe = (unsigned)(rb * (int)p.ldp * 2 + ch * 16), vPo = (unsigned)((rb + 8) * (int)p.ldp * 2 + (ch ^ 2) * 16);
  const unsigned vQe = (unsigned)(rb * (int)p.ldq * 2 + ch * 16), vQo = (unsigned)((rb + 8) * (int)p.ldq * 2 + (ch ^ 2) * 16);
  const unsigned sP16 = (unsigned)(32 * p.ldp), sQ16 = (unsigned)(32 * p.ldq);     // bytes per 16 rows
  const unsigned sP64 = (abl & 131072) ? 0u : (unsigned)(128 * p.ldp), sQ64 = (abl & 131072) ? 0u : (unsigned)(128 * p.ldq);   // bytes per K step (64 rows)
  const unsigned ldsw = (unsigned)__builtin_amdgcn_readfirstlane((int)(smem_base + wave * 1024));
  const unsigned nloop = (unsigned)(nmt / 2 - 2);

  const bool do_colsum = p.colsum != nullptr && tc == 0;
  unsigned skP, skQ, cnt;
  f32x4 cs[8];
  if (do_colsum) {
    if constexpr (SCHED == 1)
      asm volatile(TNA_ASM_1_1
                   : [skP] "=&s"(skP), [skQ] "=&s"(skQ), [cnt] "=&s"(cnt), [cs0] "=&v"(cs[0]), [cs1] "=&v"(cs[1]), [cs2] "=&v"(cs[2]),
                     [cs3] "=&v"(cs[3]), [cs4] "=&v"(cs[4]), [cs5] "=&v"(cs[5]), [cs6] "=&v"(cs[6]), [cs7] "=&v"(cs[7])
                   : TNA_INPUTS : "memory", "scc", TNA_CLOBBERS);
    else
      asm volatile(TNA_ASM_0_1
                   : [skP] "=&s"(skP), [skQ] "=&s"(skQ), [cnt] "=&s"(cnt), [cs0] "=&v"(cs[0]), [cs1] "=&v"(cs[1]), [cs2] "=&v"(cs[2]),
                     [cs3] "=&v"(cs[3]), [cs4] "=&v"(cs[4]), [cs5] "=&v"(cs[5]), [cs6] "=&v"(cs[6]), [cs7] "=&v"(cs[7])
                   : TNA_INPUTS : "memory", "scc", TNA_CLOBBERS);
  } else {
    if constexpr (SCHED == 1)
      asm volatile(TNA_ASM_1_0 : [skP] "=&s"(skP), [skQ] "=&s"(skQ), [cnt] "=&s"(cnt) : TNA_INPUTS : "memory", "scc", TNA_CLOBBERS);
    else
      asm volatile(TNA_ASM_0_0 : [skP] "=&s"(skP), [skQ] "=&s"(skQ), [cnt] "=&s"(cnt) : TNA_INPUTS : "memory", "scc", TNA_CLOBBERS);
  }

  // fp32 tile of this slice: lane holds rows rblock + 4 g4 + e, column cblock + i16
  float* O = p.O + (size_t)slice * p.R * p.ldo + (size_t)(r0 + wr * 128 + 4 * g4) * p.ldo + (c0 + wc * 128 + i16);
  tna_store_all(O, p.ldo, std::make_integer_sequence<int, 64>{});
  if (do_colsum && wc == 0 && i16 == 0) {
    // every column of P^T . ones holds the same sums: take column 0.  Block ri, element e -> P column r0 + 128 wr + 16 ri + 4 g4 + e
    float* c = p.colsum + (size_t)slice * p.R + r0 + wr * 128 + 4 * g4;
#pragma unroll
    for (int ri = 0; ri < 8; ++ri) {
      c[ri * 16 + 0] = cs[ri][0];
      c[ri * 16 + 1] = cs[ri][1];
      c[ri * 16 + 2] = cs[ri][2];
      c[ri * 16 + 3] = cs[ri][3];
    }
  }
}

std::once_flag g_tna_once[MAX_DEVICES];
int g_tna_rc[MAX_DEVICES];

}  // namespace

// Shapes the four-wave kernel takes: whole 256 x 256 tiles, slices of an even number (>= 4) of 64-row K steps.
bool tna_eligible(const TNArgs& a) {
  if (a.R % 256 || a.C % 256 || a.M % 128 || a.slice_rows % 128 || a.slice_rows < 256) return false;
  const long last = (long)a.M % a.slice_rows;          // rows of the last slice (0: full)
  return last == 0 || last >= 256;
}

int tna_launch(const TNArgs& a, int dev, dim3 grid, int sched, hipStream_t st) {
  std::call_once(g_tna_once[dev], [dev]() {
    g_tna_rc[dev] = 0;
    const void* ks[2] = {(const void*)gemm_tna_kernel<0>, (const void*)gemm_tna_kernel<1>};
    for (int i = 0; i < 2; ++i) {
      const hipError_t e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, TNA_LDS);
      if (e != hipSuccess) { clipa_set_error("hipFuncSetAttribute(gemm_tna): %s", hipGetErrorString(e)); g_tna_rc[dev] = CLIPA_ERR_LAUNCH; }
    }
  });
  if (g_tna_rc[dev]) return g_tna_rc[dev];
  note_gemm(5);
  if (sched == 1) hipLaunchKernelGGL(gemm_tna_kernel<1>, grid, dim3(TNA_THREADS), TNA_LDS, st, a);
  else hipLaunchKernelGGL(gemm_tna_kernel<0>, grid, dim3(TNA_THREADS), TNA_LDS, st, a);
  return clipa_check_launch("gemm_tna");
}

}  // namespace clipa_gemm
