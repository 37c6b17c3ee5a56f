// bf16 MFMA weight-gradient GEMM for gfx950 (MI355X):  O[R,C] = sum_m P[m,R] * Q[m,C]   (dW = dY^T . X)
//
// Reference call sites: the autograd transposes of nn.Linear / packed in-proj / out-proj / conv1 / `@ proj`
// (clipa_torch/open_clip/transformer.py:209,217-219,234,371,491,528-529; model.py:254) and the gathered-feature
// gradients of loss.py:135-139.
//
// 256x256 output tile, 8 waves, one workgroup per CU; both operands have the reduction index as their slow axis,
// so MFMA fragments come from `ds_read_b64_tr_b16` (hardware transpose) over a [m][256] LDS image filled by
// `buffer_load_dwordx4 ... lds` with a source-side chunk swizzle; split-M over ~4 workgroups per CU into fp32
// slabs + a reduce kernel that casts to the parameter dtype.  The bias gradient (column sums of dY) rides along.
//   gemm_tn2_kernel  ping-pong schedule on v_mfma_f32_32x32x16_bf16 (32-row slabs, ring of four, counted vmcnt)
//   gemm_tn3_kernel  one barrier per 64 rows on v_mfma_f32_16x16x32_bf16 (wide-P products)
// The host picks per shape (tools/tn_ab.py).
#include "gemm_common.h"

namespace clipa_gemm {
namespace {

// LDS image [64 m][256 cols] bf16 (512-B rows, 32 chunks); chunk permutation per row:
__device__ __forceinline__ int tn_swz(int row) { return ((row & 3) << 2) ^ (((row >> 2) & 1) << 1); }

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
  unsigned t;
  int slice;
  if (p.nslices > 0) {      // slice-per-XCD order: the workgroups that run side by side under one L2 read the same M rows
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
  unsigned t;
  int slice;
  if (p.nslices > 0) {      // slice-per-XCD order: the workgroups that run side by side under one L2 read the same M rows
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

std::once_flag g_tn_once[MAX_DEVICES];
int g_tn_rc[MAX_DEVICES];
int ensure_tn_attrs(int dev) {
  std::call_once(g_tn_once[dev], [dev]() {
    g_tn_rc[dev] = 0;
    const void* ks[2] = {(const void*)gemm_tn3_kernel, (const void*)gemm_tn2_kernel};
    for (int i = 0; i < 2; ++i) {
      const hipError_t e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
      if (e != hipSuccess) { clipa_set_error("hipFuncSetAttribute(gemm_tn): %s", hipGetErrorString(e)); g_tn_rc[dev] = CLIPA_ERR_LAUNCH; }
    }
  });
  return g_tn_rc[dev];
}

}  // namespace

// Work order.  Slice-per-XCD (an XCD's workgroups all read the same M rows: fabric reads drop from ~2.2x to ~1.2x the
// operand bytes) is +1..10 % on the square and tall products and -4..-8 % on the narrow-P / wide-Q one (the c_proj weight
// gradient: 1024 x 4096, 1280 x 5120, the text tower's 768 x 3072), which keeps the tile-per-XCD order with its fewer,
// longer slices (round 2 with gemm_tn2/3: tools/tn_ab.py, profiles/r02_tn_ab.jsonl; round 3 with gemm_tna at the production
// row counts: tools/probes/gemm_tn_order_sweep.hip, profiles/r03_gemm_tn_work_order_sweep.jsonl).  Experiment flags: 4096
// forces the slice order, 8192 the tile order.
bool tn_per_xcd(long M, long R, long C) {
  const int abl = g_abl.load(std::memory_order_relaxed);
  if (abl & 4096) return true;
  if (abl & 8192) return false;
  (void)M;
  return !(C >= 3 * R);
}

long tn_slices(long M, long R, long C, int num_cu, bool per_xcd) {
  const long tiles = ((R + 255) / 256) * ((C + 255) / 256);
  const long mt = (M + 63) / 64;
  if (per_xcd) {
    // S = 8k slices, k per XCD: an XCD (num_cu / 8 CUs, one resident workgroup each) runs k * tiles workgroups;
    // take the k <= 8 with ~3..6 rounds whose last round is fullest
    const long cu_x = num_cu / 8 > 0 ? num_cu / 8 : 1;
    long best = 1; double best_fill = -1.0;
    for (long k = 1; k <= 8; ++k) {
      const long w = k * tiles, rounds = (w + cu_x - 1) / cu_x;
      if (8 * k > mt) break;
      const double fill = (double)w / (double)(rounds * cu_x);
      const double score = fill - (rounds < 3 ? 0.15 * (3 - rounds) : 0.0) - (rounds > 8 ? 0.02 * (rounds - 8) : 0.0);
      if (score > best_fill + 1e-9) { best_fill = score; best = k; }
    }
    return 8 * best;
  }
  // ~4 workgroups per CU, but never a thin last round: one workgroup per CU is resident (128 KiB LDS), so a grid of
  // 4.1 x #CUs costs five rounds.  Take the largest slice count <= 64 whose grid fills >= 97 % of its rounds.
  long S = (4L * num_cu) / tiles;
  if (S > 64) S = 64;
  if (S < 1) S = 1;
  for (long c = S; c >= 1; --c) {
    const long w = tiles * c, rounds = (w + num_cu - 1) / num_cu;
    if (w * 100 >= rounds * num_cu * 97 || c == 1) { if (w * 100 >= rounds * num_cu * 97) S = c; break; }
  }
  if (S > mt) S = mt;
  if (S < 1) S = 1;
  if (mt > 0) { const long per = (mt + S - 1) / S; S = (mt + per - 1) / per; }
  return S;
}

}  // namespace clipa_gemm

using namespace clipa_gemm;

extern "C" int64_t clipa_gemm_tn_workspace(int64_t M, int64_t R, int64_t C, int64_t* nslices) {
  // a size query is pure arithmetic: without a visible device (build / CPU-side checks) price it for the 256 CUs
  // of an MI355X, which is what gemm_num_cu() reports on the real device
  int dev = 0;
  const int ncu = (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < MAX_DEVICES) ? gemm_num_cu(dev) : 256;
  const long S = tn_slices(M, R, C, ncu, tn_per_xcd(M, R, C));
  if (nslices) *nslices = S;
  return (S * R * C + S * R) * (int64_t)sizeof(float);   // split-M slabs + partial column sums
}

extern "C" int clipa_gemm_tn(const void* P, const void* Q, void* out, float* colsum_out, int64_t M, int64_t R,
                             int64_t C, int64_t ldp, int64_t ldq, int out_bf16, void* workspace,
                             int64_t workspace_bytes, void* stream) {
  if (R <= 0 || C <= 0) return CLIPA_OK;
  if (R % 8 != 0 || C % 8 != 0 || ldp % 8 != 0 || ldq % 8 != 0) { clipa_set_error("gemm_tn: R, C, ldp, ldq must be multiples of 8"); return CLIPA_ERR_ARG; }
  int dev = 0;
  if (int rc = current_device(&dev)) return rc;
  const int abl = g_abl.load(std::memory_order_relaxed);
  const bool per_xcd = tn_per_xcd(M, R, C);
  const int64_t S_plan = tn_slices(M, R, C, gemm_num_cu(dev), per_xcd);
  const int64_t need = (S_plan * R * C + S_plan * R) * (int64_t)sizeof(float);
  if (workspace_bytes < need || !workspace) { clipa_set_error("gemm_tn: workspace %ld < %ld bytes", (long)workspace_bytes, (long)need); return CLIPA_ERR_ARG; }
  const long mt = (M + 63) / 64;
  const long slice_rows = ((mt + S_plan - 1) / S_plan) * 64;
  if (slice_rows * ldp * 2 >= (1L << 32) - (1 << 24) || slice_rows * ldq * 2 >= (1L << 32) - (1 << 24)) { clipa_set_error("gemm_tn: slice too large for 32-bit buffer offsets"); return CLIPA_ERR_ARG; }
  if (int rc = ensure_tn_attrs(dev)) return rc;
  TNArgs a;
  a.P = (const char*)P; a.Q = (const char*)Q; a.O = (float*)workspace;
  a.M = (int)M; a.R = (int)R; a.C = (int)C; a.ldp = ldp; a.ldq = ldq; a.ldo = C; a.slice_rows = (int)slice_rows;
  a.colsum = colsum_out ? (float*)workspace + S_plan * R * C : nullptr;
  a.nslices = per_xcd ? (int)S_plan : 0;
  a.abl = abl;
  const long tiles = ((R + 255) / 256) * ((C + 255) / 256);
  // The 16x16x32 kernel (v3) for the image tower's wide-P products (in-proj and c_fc weight gradients) and its
  // 1024 x 1024 out-proj, the ping-pong kernel (v2) elsewhere (c_proj, the text tower) - per-shape winners of
  // tools/tn_ab.py, all within +-4 % except the text tower (v2 +18 %).  Experiment flags 1024 / 2048 force v3 / v2.
  const bool use_v3 = (abl & 1024) || (!(abl & 2048) && ((R >= 3072 && C >= 1024) || (R == 1024 && C == 1024)));
  // Whole-tile shapes with slices of an even number of K steps (every block GEMM of the BASELINE configurations) run on the
  // four-wave kernel with the hand-scheduled main loop (gemm_tna.hip): slices are rounded up to 128 rows, which may leave
  // fewer (never more) slices than the workspace was sized for.  Experiment flags: 16384 keeps gemm_tn2/3, 32768 = schedule 1.
  int64_t S_used = S_plan;
  bool launched = false;
  if (!(abl & (16384 | 1024 | 2048))) {
    TNArgs b = a;
    b.slice_rows = (int)((slice_rows + 127) / 128 * 128);
    const int64_t S4 = (M + b.slice_rows - 1) / b.slice_rows;
    if (per_xcd) b.nslices = (int)S4;
    if (S4 <= S_plan && tna_eligible(b)) {
      const dim3 g4 = per_xcd ? dim3((unsigned)(tiles * ((S4 + 7) / 8 * 8)), 1) : dim3((unsigned)tiles, (unsigned)S4);
      b.colsum = colsum_out ? (float*)workspace + S4 * R * C : nullptr;
      if (int rc = tna_launch(b, dev, g4, (abl & 32768) ? 1 : 0, (hipStream_t)stream)) return rc;
      a = b;
      S_used = S4;
      launched = true;
    }
  }
  if (!launched) {
    const dim3 grid = per_xcd ? dim3((unsigned)(tiles * S_plan), 1) : dim3((unsigned)tiles, (unsigned)S_plan);
    note_gemm(use_v3 ? 4 : 3);
    if (use_v3) hipLaunchKernelGGL(gemm_tn3_kernel, grid, dim3(NTHREADS), 2 * STAGE_BYTES, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(gemm_tn2_kernel, grid, dim3(NTHREADS), 2 * STAGE_BYTES, (hipStream_t)stream, a);
    if (int rc = clipa_check_launch("gemm_tn")) return rc;
  }
  const int64_t S = S_used;      // slabs actually written
  const long n = R * C;
  const unsigned blocks = (unsigned)((n / 4 + 255) / 256);
  if (out_bf16) hipLaunchKernelGGL(reduce_slabs_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, out, n, (int)S);
  else hipLaunchKernelGGL(reduce_slabs_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, out, n, (int)S);
  if (colsum_out)   // bias gradient: sum the per-slice partial column sums of P
    hipLaunchKernelGGL(reduce_slabs_kernel<false>, dim3((unsigned)((R / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)a.colsum, (void*)colsum_out, (long)R, (int)S);
  return clipa_check_launch("gemm_tn_reduce");
}
