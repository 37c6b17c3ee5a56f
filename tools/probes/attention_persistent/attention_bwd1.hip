// Single-sweep attention backward for the image towers at 129 <= L <= 224 tokens, head dim 64, no mask (ViT-S/B/L-16 @ 224:
// 197 tokens = 7 tiles of 32 rows; clipa_torch/open_clip/transformer.py:223-236 through nn.MultiheadAttention's backward).
//
// attention.hip's backward is two sweeps over the (query tile, key tile) pairs of a head - a query-major one for dQ and a
// key-major one for dK / dV - each recomputing the probabilities: 7 matrix products and two exp2 per score where the maths
// needs 5 and one; and a workgroup there is load -> wait -> compute -> store per head with nothing under the loads
// (profiles/r04_pmc_derived.txt: 38 % of wave cycles waiting, MFMA pipe busy 29 %).  This kernel:
//   * ONE sweep.  Wave w of an eight-wave workgroup owns key tile w: its K / V rows sit in registers as MFMA B operands, its
//     dK / dV accumulators stay in registers for the whole head.  It walks the query tiles in the rotated order
//     qt = (w + i) mod NKT, so at step i every wave is on a different query tile.  Per pair: S = Q.K^T and dP = dO.V^T (lane =
//     key, registers = queries), P = exp2(c S - m'), dS = P (dP - D), dV += dO^T.P, dK += Q^T.dS straight from the registers.
//   * dQ needs dS with the OTHER index in the registers (the reduction of an MFMA runs over registers, never over lanes), so
//     the dS tile (bf16, 2 KB) goes through a ring slot in LDS: after the step's barrier the wave that owns QUERY tile
//     (w + i) mod NKT picks it up with transposing reads and adds K_w^T.dS^T to its dQ accumulators, which also stay in
//     registers.  Every step is a perfect matching (each wave writes one tile and reads one), the summation order of every
//     output is fixed: results are bit-reproducible.  5 matrix products, one exp2 per score.
//   * PERSISTENT workgroups (one per CU, 2 waves per SIMD) walking heads, with the NEXT head's operands in flight under the
//     current head's arithmetic: K into the second K image from the start of the head, Q / dO into their images as soon as
//     the last step has released them (under the last dQ products and the output stores), V / O rows and the statistics into
//     registers.  hipcc drains vmcnt in front of every LDS access while an LDS-DMA that MAY alias it is in flight, so the
//     images are separate static __shared__ arrays (distinct objects = provably no alias) and the head loop is unrolled by
//     two with the K image a compile-time choice.
//   * the softmax statistics (c * rowmax, 1 / rowsum) of the forward fold into one exponent offset m' = c * rowmax +
//     log2(rowsum) per query, and the 1 / sqrt(dh) of dS moves to the dQ / dK stores.
// Numerics: the same bf16 operand roundings as the two-sweep kernel (P, dS rounded to bf16 before their products, fp32
// accumulation), different summation order and exponent folding: not bit-identical to it, same tolerance against fp64
// (tests/test_kernels_gpu.py::test_attention*).
#include "common.h"
#include <mutex>
#include "clipa_hip.h"
#include "attention_common.h"

namespace {

constexpr int B1_WAVES = 8;

// dS tile in a ring slot: T[key][query] bf16, 64-byte rows of eight 8-byte chunks (4 queries each), chunk index XORed with
// (key >> 1) & 7: the 16 lanes of a ds_write_b64 group (16 keys, same chunk) and the 32 lanes of a transposing read (4 keys x 8
// chunks) each touch every bank once.
__device__ __forceinline__ int tswz(int key) { return (key >> 1) & 7; }

template <int NKT>
__global__ __launch_bounds__(64 * B1_WAVES, 1) void attn_bwd1_kernel(AttnArgs p) {
  constexpr int LP = NKT * 32, RB = 128, KS = 4, DT = 2, DH = 64, IMG = LP * RB;
  __shared__ __attribute__((aligned(1024))) char sQ[IMG];
  __shared__ __attribute__((aligned(1024))) char sDO[IMG];
  __shared__ __attribute__((aligned(1024))) char sK[IMG];
  __shared__ __attribute__((aligned(1024))) char sRing[B1_WAVES * 4096];      // per wave: two 2 KB dS slots = one 4 KB store window
  __shared__ __attribute__((aligned(16))) float sMp[LP];
  __shared__ __attribute__((aligned(16))) float sD[LP];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool active = wave < NKT;                       // (NKT < 8: the spare waves only help with the DMA)
  const int l31 = lane & 31, hi = lane >> 5, q16 = (lane >> 4) & 1, i16 = lane & 15;
  const long nheads = (long)p.B * p.H;
  const float c = p.scale * 1.4426950408889634f;
  const unsigned nrec = (unsigned)((long)(p.L - 1) * p.ld_qkv * 2 + DH * 2);
  const unsigned nrec_o = (unsigned)((long)(p.L - 1) * p.ld_o * 2 + DH * 2);
  const int tile0 = 32 * min(wave, NKT - 1);            // this wave's key tile = its query tile for dQ (spare waves: any valid tile)
  char* const myring = sRing + wave * 4096;
  const float sbias = (tile0 + l31 < p.L) ? 0.f : -1e30f;   // padded keys (zero K rows) must not reach dQ through dS: P = exp2(-huge) = 0

  bf16x8 fv[KS], fo[KS];       // V rows of the wave's key tile (B operand of dP) / O rows of its query tile (for D)
  float2 st;                   // forward statistics of the wave's query rows

  auto head_off = [&](long hd, long ld) {
    const int b = (int)(hd / p.H), h = (int)(hd - (long)b * p.H);
    return ((size_t)b * p.L * ld + (size_t)h * DH) * 2;
  };
  // operands of head `hd` that travel through registers
  auto prefetch_regs = [&](long hd) {
    load_frags<KS, DH>(make_rsrc(p.v + head_off(hd, p.ld_qkv), nrec), p.ld_qkv, tile0 + l31, hi, fv);
    load_frags<KS, DH>(make_rsrc(p.o_in + head_off(hd, p.ld_o), nrec_o), p.ld_o, tile0 + l31, hi, fo);
    st = make_float2(0.f, 0.f);
    if (tile0 + l31 < p.L) st = *(const float2*)(p.stats + ((size_t)hd * p.L + tile0 + l31) * 2);
  };
  auto dma_q_do = [&](long hd) {
    dma_image<DH>(make_rsrc(p.q + head_off(hd, p.ld_qkv), nrec), sQ, LP, p.ld_qkv, wave, lane, B1_WAVES);
    dma_image<DH>(make_rsrc(p.d_o + head_off(hd, p.ld_o), nrec_o), sDO, LP, p.ld_o, wave, lane, B1_WAVES);
  };
  auto dma_k = [&](long hd) {
    dma_image<DH>(make_rsrc(p.k + head_off(hd, p.ld_qkv), nrec), sK, LP, p.ld_qkv, wave, lane, B1_WAVES);
  };

  long head = blockIdx.x;
  if (head >= nheads) return;
  dma_q_do(head);
  dma_k(head);
  prefetch_regs(head);

  for (; head < nheads; head += gridDim.x) {
    const long next = head + gridDim.x;
    const bool has_next = next < nheads;
    const int b = (int)(head / p.H), h = (int)(head - (long)b * p.H);
    const long row0 = (long)b * p.L;

    // ---- prologue: this head's operands were issued behind the previous head's last step; they must have landed ----
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0), as an instruction hipcc's wait-count pass sees
    __syncthreads();
    {
      float Dq = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 fdo = frag_direct<DH>(sDO, tile0, l31, hi, ks);
        float a[8], o8[8];
        unpack8(__builtin_bit_cast(u32x4, fdo), a);
        unpack8(__builtin_bit_cast(u32x4, fo[ks]), o8);
#pragma unroll
        for (int i = 0; i < 8; ++i) Dq += a[i] * o8[i];
      }
      Dq += __shfl_xor(Dq, 32, 64);
      // P = exp2(c s - c rowmax) / rowsum = exp2(c s - m'), m' = c rowmax + log2(rowsum); padded queries: P = 0
      const float mp = (tile0 + l31 < p.L) ? st.x - __builtin_amdgcn_logf(st.y) : 1e30f;
      if (hi == 0 && active) { sMp[tile0 + l31] = mp; sD[tile0 + l31] = Dq; }
    }
    __syncthreads();

    f32x16 dk[DT], dv[DT], dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; dq[dt][r] = 0.f; }

    // One step = [pair arithmetic + dS tile into the ring] barrier [dQ product from the tile another wave left].  hipcc left to
    // itself issues every LDS read right in front of the MFMA that consumes it (read, wait for the LDS round trip, MFMA: twenty
    // times per step); the fragments of a group of MFMAs are therefore read as a block, fenced from the scheduler, so that the
    // waits count down while the matrix pipe runs.  Not unrolled over the steps (the fragment addresses depend on the step only
    // through qt / src; unrolled, hipcc hoists seven steps' worth of them out of the head loop and spills).
#pragma unroll 1
    for (int i = 0; i < NKT; ++i) {
      if (active && !(p.abl & 1)) {
        int qt = wave + i;
        if (qt >= NKT) qt -= NKT;
        bf16x8 aq[KS], ad[KS], fk[KS];   // (the wave's K rows come back from the image every step: 16 registers less for the head)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          aq[ks] = frag_direct<DH>(sQ, 32 * qt, l31, hi, ks);
          fk[ks] = frag_direct<DH>(sK, tile0, l31, hi, ks);
          ad[ks] = frag_direct<DH>(sDO, 32 * qt, l31, hi, ks);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = sbias; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ks], fk[ks], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ad[ks], fv[ks], dp, 0, 0, 0);
        }
        // the transposed fragments of the dV / dK products land under the score MFMAs and the softmax-gradient arithmetic
        bf16x8 tq[2][DT], td[2][DT];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            td[s2][dt] = frag_trans<DH>(sDO, 32 * qt + 16 * s2, 32 * dt, hi, q16, i16);
            tq[s2][dt] = frag_trans<DH>(sQ, 32 * qt + 16 * s2, 32 * dt, hi, q16, i16);
          }
        __builtin_amdgcn_sched_barrier(0);
        float pr[16], ds[16];
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const float4 m4 = *(const float4*)(sMp + 32 * qt + 8 * rq + 4 * hi);      // query = 32 qt + 8 rq + 4 hi + e
          const float4 d4 = *(const float4*)(sD + 32 * qt + 8 * rq + 4 * hi);
          const float mm[4] = {m4.x, m4.y, m4.z, m4.w};
          const float dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * rq + e;
            const float pe = __builtin_amdgcn_exp2f(fmaf(s[r], c, -mm[e]));
            pr[r] = pe;
            ds[r] = pe * (dp[r] - dd[e]);               // (the 1/sqrt(dh) of dS rides in the dQ / dK stores)
          }
        }
        char* const slot = myring + (i & 1) * 2048;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 pf = pack_frag(pr + 8 * s2);
          const bf16x8 dsf = pack_frag(ds + 8 * s2);
          const u32x4 dw = __builtin_bit_cast(u32x4, dsf);
          // T[key = l31][query = 8 j + 4 hi + e], j = 2 s2 + jj: 8 bytes per (lane, j)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            u32x2 w2;
            w2[0] = dw[2 * jj];
            w2[1] = dw[2 * jj + 1];
            *(u32x2*)(slot + l31 * 64 + (((2 * (2 * s2 + jj) + hi) ^ tswz(l31)) << 3)) = w2;
          }
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(td[s2][dt], pf, dv[dt], 0, 0, 0);
            dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tq[s2][dt], dsf, dk[dt], 0, 0, 0);
          }
        }
      }
      __syncthreads();          // every wave's dS tile of this step is in its slot; (last step: the Q / dO images are free)
      if (i == NKT - 1 && has_next) dma_q_do(next);      // the next head's images go in under the last dQ product and the stores
      if (active && !(p.abl & 1)) {
        // the tile of (query tile = mine, key tile = src) was written by wave src at this step
        int src = wave - i;
        if (src < 0) src += NKT;
        const char* const T = sRing + src * 4096 + (i & 1) * 2048;
        bf16x8 dst[2], kt[2][DT];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int r0 = 16 * s2 + 4 * hi + (i16 >> 2), r1 = r0 + 8, chunk = 4 * q16 + (i16 & 3);
          const bf16x4 ta = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) bf16x4*)(T + r0 * 64 + ((chunk ^ tswz(r0)) << 3)));
          const bf16x4 tb = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) bf16x4*)(T + r1 * 64 + ((chunk ^ tswz(r1)) << 3)));
          dst[s2][0] = ta[0]; dst[s2][1] = ta[1]; dst[s2][2] = ta[2]; dst[s2][3] = ta[3];
          dst[s2][4] = tb[0]; dst[s2][5] = tb[1]; dst[s2][6] = tb[2]; dst[s2][7] = tb[3];
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) kt[s2][dt] = frag_trans<DH>(sK, 32 * src + 16 * s2, 32 * dt, hi, q16, i16);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
            dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt[s2][dt], dst[s2], dq[dt], 0, 0, 0);
      }
    }
    __syncthreads();            // every ring slot has been read: the ring becomes the waves' store windows; the K image is free
    if (has_next) {
      dma_k(next);
      prefetch_regs(next);      // V / O rows and statistics of the next head: land under the stores
    }
    if (active && !(p.abl & 2)) {
      store_tile64(myring, p.dk, p.ld_dqkv, row0 + tile0, p.L - tile0, h * DH, lane, dk[0], dk[1], p.scale);
      store_tile64(myring, p.dv, p.ld_dqkv, row0 + tile0, p.L - tile0, h * DH, lane, dv[0], dv[1], 1.0f);
      store_tile64(myring, p.dq, p.ld_dqkv, row0 + tile0, p.L - tile0, h * DH, lane, dq[0], dq[1], p.scale);
    }
  }
}

int num_cus() {
  static int n[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (!n[dev]) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    n[dev] = v;
  }
  return n[dev];
}

template <int NKT>
int launch_bwd1(const AttnArgs& a, hipStream_t st) {
  const long nheads = (long)a.B * a.H;
  const unsigned grid = (unsigned)(nheads < num_cus() ? nheads : num_cus());
  hipLaunchKernelGGL((attn_bwd1_kernel<NKT>), dim3(grid), dim3(64 * B1_WAVES), 0, st, a);
  return clipa_check_launch("attn_bwd1");
}

}  // namespace

// -> 1 if the single-sweep kernel covers this problem (and was launched: *rc = its return code), 0 if the caller keeps its own
extern "C" int clipa_internal_debug_flags(void);
extern "C" int clipa_attn_bwd1_try(const void* args, int64_t dh, void* stream, int* rc) {
  AttnArgs a = *(const AttnArgs*)args;
  a.abl = (clipa_internal_debug_flags() >> 20) & 3;      // (bits 20 / 21 of the experiment flags)
  const int nkt = (a.L + 31) / 32;
  if (dh != 64 || a.causal || a.seq_len || nkt < 5 || nkt > 7) return 0;
  switch (nkt) {
    case 5: *rc = launch_bwd1<5>(a, (hipStream_t)stream); break;
    case 6: *rc = launch_bwd1<6>(a, (hipStream_t)stream); break;
    default: *rc = launch_bwd1<7>(a, (hipStream_t)stream); break;
  }
  return 1;
}
