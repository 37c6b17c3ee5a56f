// Fused multi-head self-attention forward / backward for short sequences (L <= 288 in one piece, 288 < L <= 1024
// streamed in 256-row chunks; head dim 64 or 80, and - compiled by attention_wide.hip from this
// same source - the wide heads 88 / 104 / 112 of ViT-g/14, ViT-bigG/14, ViT-e/14)
// on gfx950.  Replaces torch scaled_dot_product_attention as called from nn.MultiheadAttention in
// clipa_torch/open_clip/transformer.py:209,223-236 (softmax(q.k^T/sqrt(dh) + mask).v, dropout 0;
// mask = None for the image tower, additive causal triu(1)*-inf for text, transformer.py:618-624).
//
// CLIPA's regime is "short sequence, huge batch": B*H = 65k independent heads of L <= 257 tokens.
// One workgroup (4 waves) owns one (batch, head): K and V (fwd) or Q, K, V, dO (bwd) are DMA'd once
// into LDS straight from the packed [tokens, 3*D] projection output (no head-major copy), and a
// whole softmax row lives in registers, so there is no online rescaling and the [L, L] score matrix
// never exists in HBM.
//
// MFMA mapping (v_mfma_f32_32x32x16_bf16): scores are computed *transposed*, S^T[key][query] =
// K.Q^T, which leaves each lane holding 16 keys of ONE query per 32-key tile - the softmax
// reduction is in-lane plus one cross-half shuffle - and makes P^T directly usable as the B operand
// of O^T[d][query] = V^T.P^T.  V^T / K^T / Q^T / dO^T operands come from row-major LDS images via
// ds_read_b64_tr_b16.  The backward pass is two sweeps (query-major for dQ, key-major for dK/dV)
// that recompute the probabilities from per-row (max, 1/sum) statistics kept in LDS.
#include "common.h"
#include <mutex>
#include "clipa_hip.h"
#include "attention_common.h"
#include "internal_hooks.h"

namespace {

template <int NKT, int DH, bool CAUSAL>
__global__ __launch_bounds__((64 * attn_waves<NKT, DH>()), HD<DH>::WGS) void attn_fwd_kernel(AttnArgs pin) {
  constexpr int LP = NKT * 32, RB = HD<DH>::RB, KS = HD<DH>::KS, DT = HD<DH>::DT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HPW = WGHeads<NKT>::HPW, WPH = attn_waves<NKT, DH>() / HPW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_wg = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slot = wave_wg / WPH, wave = wave_wg % WPH;          // head slot of this wave, wave index within the head
  char* sK = smem + slot * (2 * LP * RB);
  char* sV = sK + LP * RB;
  constexpr bool STAGE = fwd_stages<NKT, DH>();                  // whole-row stores through 4 KB of LDS per wave
  char* stage = smem + HPW * (2 * LP * RB) + wave_wg * Stg<DH>::BYTES;
  const int l31 = lane & 31, hi = lane >> 5, q16 = (lane >> 4) & 1, i16 = lane & 15;
  const long nheads = (long)pin.B * pin.H;
  const long head_raw = (long)blockIdx.x * HPW + slot;
  const bool live = head_raw < nheads;                           // the last workgroup may have empty head slots
  const long head = live ? head_raw : nheads - 1;
  const int b = (int)(head / pin.H), h = (int)(head - (long)b * pin.H);
  ATTN_LOCATE_SEQUENCE
  const size_t hoff = ((size_t)row0 * p.ld_qkv + (size_t)h * DH) * 2;
  const unsigned nrec = (unsigned)((long)(p.L - 1) * p.ld_qkv * 2 + DH * 2);
  const __amdgpu_buffer_rsrc_t rsQ = make_rsrc(p.q + hoff, nrec);
  const __amdgpu_buffer_rsrc_t rsK = make_rsrc(p.k + hoff, nrec);
  const __amdgpu_buffer_rsrc_t rsV = make_rsrc(p.v + hoff, nrec);
  dma_image<DH>(rsK, sK, LP, p.ld_qkv, wave, lane, WPH);
  dma_image<DH>(rsV, sV, LP, p.ld_qkv, wave, lane, WPH);
  // the wave's first query tile rides along with the K / V DMA; each later tile is fetched while the tile
  // before it computes, so no query load sits on the critical path
  // (the one shape whose whole-row softmax already fills the 256-VGPR budget of two workgroups per CU - 9 key
  // tiles at head dim 64 - keeps the plain load)
  constexpr bool PREFETCH = !(NKT == 9 && DH == 64);
  bf16x8 fq[KS], fqn[KS];
  if (PREFETCH) load_frags<KS, DH>(rsQ, p.ld_qkv, 32 * wave + l31, hi, fqn);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int qt = wave; qt < NKT; qt += WPH) {
    const int qg = 32 * qt + l31;
    if (PREFETCH) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) fq[ks] = fqn[ks];
      if (qt + WPH < NKT) load_frags<KS, DH>(rsQ, p.ld_qkv, qg + 32 * WPH, hi, fqn);
    } else {
      load_frags<KS, DH>(rsQ, p.ld_qkv, qg, hi, fq);
    }
    f32x16 s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
      if (CAUSAL && kt > qt) continue;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_direct<DH>(sK, 32 * kt, l31, hi, ks), fq[ks], s[kt], 0, 0, 0);
      if (!CAUSAL && (kt & 1)) __builtin_amdgcn_sched_barrier(0);   // keep the K-fragment reads of later tiles from piling up in registers
    }
    float inv, m2;
    softmax_rows<NKT, CAUSAL>(s, p, qt, qg, hi, inv, m2);

    f32x16 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (CAUSAL && kt > qt) continue;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float pv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pv[e] = s[kt][8 * s2 + e];
        const bf16x8 pf = pack_frag(pv);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_trans<DH>(sV, 32 * kt + 16 * s2, 32 * dt, hi, q16, i16), pf, o[dt], 0, 0, 0);
      }
    }
    if constexpr (STAGE) {
      if (live) store_tile<DH>(stage, p.o, p.ld_o, row0 + 32 * qt, p.L - 32 * qt, h * DH, lane, o, inv);
      if (qg < p.L && live && p.stats && hi == 0) *(float2*)(p.stats + (stat0 + qg) * 2) = make_float2(m2, inv);
    } else if (qg < p.L && live) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        store_frag_T(p.o, p.ld_o, row0 + qg, h * DH + 32 * dt, hi, o[dt], inv, DH - 32 * dt);
      if (p.stats && hi == 0) *(float2*)(p.stats + (stat0 + qg) * 2) = make_float2(m2, inv);
    }
  }
}

// Backward.  Two LDS phases share one 2-image window (so two workgroups fit per CU):
//   phase 1: K, V images; query-major sweep -> dQ   (q / dO / O rows of the wave's tile come from global)
//   phase 2: Q, dO images; key-major sweep   -> dK, dV (k / v rows of the wave's tile come from global)
// Probabilities are recomputed from the forward's (c*rowmax, 1/rowsum) statistics; D_q = <dO_q, O_q>.
// (A software-pipelined variant - next pair's score MFMAs / previous pair's dK,dV MFMAs issued ahead of the softmax-gradient
// arithmetic, bit-identical results - measured 2-14 % SLOWER on every production shape and was dropped:
// profiles/r02_attention_bwd_pipelining_ab.jsonl.  The kernel moves ~13 GB per launch; it is not issue-bound.)
template <int NKT, int DH, bool CAUSAL>
__global__ __launch_bounds__((64 * attn_waves<NKT, DH>()), HD<DH>::WGS) void attn_bwd_kernel(AttnArgs pin) {
  constexpr int LP = NKT * 32, RB = HD<DH>::RB, KS = HD<DH>::KS, DT = HD<DH>::DT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HPW = WGHeads<NKT>::HPW, WPH = attn_waves<NKT, DH>() / HPW;
  // The eight-wave head-dim-80 instantiations fence hipcc's scheduler between the score MFMAs, the softmax-gradient arithmetic
  // and the output MFMAs of a pair: left alone it hoists the next group's LDS reads across them and runs 9 % slower
  // (profiles/r03_attention_bwd_sched_barrier_ab.jsonl; at head dim 64 the same fences move +-2 %, at short head-dim-80
  // sequences they cost up to 46 %: measured per instantiation, not a rule)
  constexpr bool FENCE = DH == 80 && NKT >= 5;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_wg = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slot = wave_wg / WPH, wave = wave_wg % WPH;
  char* img0 = smem + slot * (2 * LP * RB + 3 * LP * 4);
  char* img1 = img0 + LP * RB;
  float* sM = (float*)(img0 + 2 * LP * RB);                      // m' = c rowmax + log2 rowsum per query (round 5: one exponent offset)
  float* sD = sM + 2 * LP;                                       // D = <dO, O> per query  ([LP, 2 LP): free since the statistics folded)
  constexpr bool STAGE = bwd_stages<NKT, DH>();                  // whole-row stores through 4 KB of LDS per wave
  char* stage = smem + HPW * (2 * LP * RB + 3 * LP * 4) + wave_wg * Stg<DH>::BYTES;
  const int l31 = lane & 31, hi = lane >> 5, q16 = (lane >> 4) & 1, i16 = lane & 15;
  const long nheads = (long)pin.B * pin.H;
  const long head_raw = (long)blockIdx.x * HPW + slot;
  const bool live = head_raw < nheads;
  const long head = live ? head_raw : nheads - 1;
  const int b = (int)(head / pin.H), h = (int)(head - (long)b * pin.H);
  ATTN_LOCATE_SEQUENCE
  const size_t hoff = ((size_t)row0 * p.ld_qkv + (size_t)h * DH) * 2;
  const size_t ooff = ((size_t)row0 * p.ld_o + (size_t)h * DH) * 2;
  const unsigned nrec = (unsigned)((long)(p.L - 1) * p.ld_qkv * 2 + DH * 2);
  const unsigned nrec_o = (unsigned)((long)(p.L - 1) * p.ld_o * 2 + DH * 2);
  const __amdgpu_buffer_rsrc_t rsQ = make_rsrc(p.q + hoff, nrec), rsK = make_rsrc(p.k + hoff, nrec),
                               rsV = make_rsrc(p.v + hoff, nrec), rsDO = make_rsrc(p.d_o + ooff, nrec_o),
                               rsO = make_rsrc(p.o_in + ooff, nrec_o);
  const float c = p.scale * 1.4426950408889634f;
  const float* stats = p.stats + stat0 * 2;

  // ---- phase 1: dQ ------------------------------------------------------------------------------
  dma_image<DH>(rsK, img0, LP, p.ld_qkv, wave, lane, WPH);
  dma_image<DH>(rsV, img1, LP, p.ld_qkv, wave, lane, WPH);
  bf16x8 fq[KS], fdo[KS], fo[KS];
  load_frags<KS, DH>(rsQ, p.ld_qkv, 32 * wave + l31, hi, fq);        // first tile's rows ride along with the DMA
  load_frags<KS, DH>(rsDO, p.ld_o, 32 * wave + l31, hi, fdo);
  load_frags<KS, DH>(rsO, p.ld_o, 32 * wave + l31, hi, fo);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int qt = wave; qt < NKT; qt += WPH) {
    const int qg = 32 * qt + l31;
    if (qt != wave) {
      load_frags<KS, DH>(rsQ, p.ld_qkv, qg, hi, fq);
      load_frags<KS, DH>(rsDO, p.ld_o, qg, hi, fdo);
      load_frags<KS, DH>(rsO, p.ld_o, qg, hi, fo);
    }
    // P = exp2(c s - c rowmax) / rowsum = exp2(c s - m'), m' = c rowmax + log2(rowsum): the forward's two statistics fold into
    // one exponent offset per query (one multiply per score less in both sweeps); padded queries: m' = +huge -> P = 0
    float mp = 1e30f;
    if (qg < p.L) {
      const float2 st = *(const float2*)(stats + qg * 2);
      mp = st.x - __builtin_amdgcn_logf(st.y);
    }
    float Dq = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float a[8], o8[8];
      unpack8(__builtin_bit_cast(u32x4, fdo[ks]), a);
      unpack8(__builtin_bit_cast(u32x4, fo[ks]), o8);
#pragma unroll
      for (int i = 0; i < 8; ++i) Dq += a[i] * o8[i];
    }
    Dq += __shfl_xor(Dq, 32, 64);
    if (hi == 0) { sM[qg] = mp; sD[qg] = Dq; }
    const int lim2 = (CAUSAL ? min(p.L, qg + 1) : p.L) - 4 * hi;
    f32x16 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (CAUSAL && kt > qt) continue;
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_direct<DH>(img0, 32 * kt, l31, hi, ks), fq[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_direct<DH>(img1, 32 * kt, l31, hi, ks), fdo[ks], dp, 0, 0, 0);
      }
      if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
      const bool full = CAUSAL ? ((32 * kt + 32 <= p.L) && kt < qt) : (kt < NKT - 1);
      float ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float pe = __builtin_amdgcn_exp2f(fmaf(s[r], c, -mp));
        if (!full) pe = (32 * kt + 8 * (r >> 2) + (r & 3) < lim2) ? pe : 0.f;
        ds[r] = pe * (dp[r] - Dq);                     // (the 1 / sqrt(dh) of dS rides in the dQ / dK stores)
      }
      if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 dsf = pack_frag(ds + 8 * s2);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
          dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_trans<DH>(img0, 32 * kt + 16 * s2, 32 * dt, hi, q16, i16), dsf, dq[dt], 0, 0, 0);
      }
    }
    if constexpr (STAGE) {
      if (live) store_tile<DH>(stage, p.dq, p.ld_dqkv, row0 + 32 * qt, p.L - 32 * qt, h * DH, lane, dq, p.scale);
    } else if (qg < p.L && live) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        store_frag_T(p.dq, p.ld_dqkv, row0 + qg, h * DH + 32 * dt, hi, dq[dt], p.scale, DH - 32 * dt);
    }
  }
  // the wave's first key tile of phase 2 is still in the K / V images: same register layout as the row-wise global load, which
  // costs 8 instructions of 32 quarter-rows each and a DRAM-latency wait (-6 % at 197 tokens, -17 % at 77:
  // profiles/r03_attention_bwd_first_key_tile_from_lds_ab.jsonl)
  bf16x8 fk[KS], fv[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    fk[ks] = frag_direct<DH>(img0, 32 * min(wave, NKT - 1), l31, hi, ks);      // (a wave without a tile reads the last one)
    fv[ks] = frag_direct<DH>(img1, 32 * min(wave, NKT - 1), l31, hi, ks);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();   // everyone is done with the K / V images (and the statistics are in LDS)

  // ---- phase 2: dK, dV --------------------------------------------------------------------------
  dma_image<DH>(rsQ, img0, LP, p.ld_qkv, wave, lane, WPH);
  dma_image<DH>(rsDO, img1, LP, p.ld_o, wave, lane, WPH);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = wave; kt < NKT; kt += WPH) {
    const int kg = 32 * kt + l31;
    const int kgc = l31 - 4 * hi;      // causal test inside the diagonal tile: key <= query
    if (kt != wave) {
      load_frags<KS, DH>(rsK, p.ld_qkv, kg, hi, fk);
      load_frags<KS, DH>(rsV, p.ld_qkv, kg, hi, fv);
    }
    f32x16 dk[DT], dv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
    for (int qt = (CAUSAL ? kt : 0); qt < NKT; ++qt) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_direct<DH>(img0, 32 * qt, l31, hi, ks), fk[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_direct<DH>(img1, 32 * qt, l31, hi, ks), fv[ks], dp, 0, 0, 0);
      }
      if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
      float pr[16], ds[16];
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int q0 = 32 * qt + 8 * rq + 4 * hi;
        const float4 m4 = *(const float4*)(sM + q0);
        const float4 d4 = *(const float4*)(sD + q0);
        const float mm[4] = {m4.x, m4.y, m4.z, m4.w};
        const float dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * rq + e;
          // query = 32qt + 8rq + e + 4hi.  Padded queries carry m' = +huge (P = 0); padded keys need no mask: their
          // K/V rows are zero, their column is never stored and never mixes into other lanes' columns.
          float pe = __builtin_amdgcn_exp2f(fmaf(s[r], c, -mm[e]));
          if (CAUSAL && qt == kt) pe = (kgc <= 8 * rq + e) ? pe : 0.f;
          pr[r] = pe;
          ds[r] = pe * (dp[r] - dd[e]);
        }
      }
      if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = pack_frag(pr + 8 * s2);
        const bf16x8 dsf = pack_frag(ds + 8 * s2);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_trans<DH>(img1, 32 * qt + 16 * s2, 32 * dt, hi, q16, i16), pf, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_trans<DH>(img0, 32 * qt + 16 * s2, 32 * dt, hi, q16, i16), dsf, dk[dt], 0, 0, 0);
        }
      }
    }
    if constexpr (STAGE) {
      if (live) {
        store_tile<DH>(stage, p.dk, p.ld_dqkv, row0 + 32 * kt, p.L - 32 * kt, h * DH, lane, dk, p.scale);
        store_tile<DH>(stage, p.dv, p.ld_dqkv, row0 + 32 * kt, p.L - 32 * kt, h * DH, lane, dv, 1.0f);
      }
    } else if (kg < p.L && live) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        store_frag_T(p.dk, p.ld_dqkv, row0 + kg, h * DH + 32 * dt, hi, dk[dt], p.scale, DH - 32 * dt);
        store_frag_T(p.dv, p.ld_dqkv, row0 + kg, h * DH + 32 * dt, hi, dv[dt], 1.0f, DH - 32 * dt);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// LONG sequences (288 < L <= 1024: the 336-px stage of CLIPA-v2, 577 tokens at 14-px patches; ViT-L-16-320).  A whole
// softmax row no longer fits in registers and the whole K / V (or Q / dO) of a head no longer fits in LDS, so keys
// (forward, dQ sweep) or queries (dK / dV sweep) stream through LDS in chunks of CT 32-row tiles while the four
// waves of the workgroup hold one 32-row tile each ("group"); the forward keeps a running (max, sum) per query row
// and rescales its output accumulator per chunk (online softmax), the backward recomputes the probabilities from the
// forward's final statistics exactly as the short kernels do.  Same fragments, swizzle and MFMA mapping as above.
constexpr int LONG_CT = 8;          // 256 rows per chunk
constexpr int LONG_LMAX = 1024;

template <int DH>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t chunk_rsrc(const char* base, long ld, int row0, int L) {
  const long rows = (long)L - row0;                       // rows of the head at / after row0
  const long bytes = rows > 0 ? (rows - 1) * ld * 2 + DH * 2 : 0;
  return make_rsrc(base + (size_t)row0 * ld * 2, (unsigned)bytes);
}

template <int DH, bool CAUSAL>
__global__ __launch_bounds__(256, HD<DH>::WGS) void attn_fwd_long_kernel(AttnArgs p) {
  constexpr int CT = LONG_CT, RB = HD<DH>::RB, KS = HD<DH>::KS, DT = HD<DH>::DT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* sK = smem;
  char* sV = sK + CT * 32 * RB;
  const int l31 = lane & 31, hi = lane >> 5, q16 = (lane >> 4) & 1, i16 = lane & 15;
  const long head = blockIdx.x;
  const int b = (int)(head / p.H), h = (int)(head - (long)b * p.H);
  const char* qb = p.q + ((size_t)b * p.L * p.ld_qkv + (size_t)h * DH) * 2;
  const char* kb = p.k + ((size_t)b * p.L * p.ld_qkv + (size_t)h * DH) * 2;
  const char* vb = p.v + ((size_t)b * p.L * p.ld_qkv + (size_t)h * DH) * 2;
  const __amdgpu_buffer_rsrc_t rsQ = chunk_rsrc<DH>(qb, p.ld_qkv, 0, p.L);
  const int NT = (p.L + 31) / 32;
  const float c = p.scale * 1.4426950408889634f;

  for (int qt0 = 0; qt0 < NT; qt0 += 4) {
    const int qt = qt0 + wave;
    const int qg = 32 * qt + l31;                        // rows >= L read zeros and are never stored
    bf16x8 fq[KS];
    load_frags<KS, DH>(rsQ, p.ld_qkv, qg, hi, fq);
    const int lim2 = (CAUSAL ? min(p.L, qg + 1) : p.L) - 4 * hi;
    float m_run = -1e30f, l_run = 0.f;
    f32x16 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    for (int kc = 0; kc < NT; kc += CT) {
      if (CAUSAL && kc > qt0 + 3) break;                 // every key of this chunk lies after every query of the group
      const int nct = min(CT, NT - kc);
      __syncthreads();                                   // the previous chunk has been consumed by every wave
      dma_image<DH>(chunk_rsrc<DH>(kb, p.ld_qkv, 32 * kc, p.L), sK, nct * 32, p.ld_qkv, wave, lane, 4);
      dma_image<DH>(chunk_rsrc<DH>(vb, p.ld_qkv, 32 * kc, p.L), sV, nct * 32, p.ld_qkv, wave, lane, 4);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      f32x16 s[CT];
      float mx = -1e30f;
#pragma unroll
      for (int kt = 0; kt < CT; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
        if (kt >= nct || (CAUSAL && kc + kt > qt)) continue;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_direct<DH>(sK, 32 * kt, l31, hi, ks), fq[ks], s[kt], 0, 0, 0);
        const int T = kc + kt;
        const bool full = (32 * T + 32 <= p.L) && (!CAUSAL || T < qt);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = s[kt][r];
          if (!full) v = (32 * T + 8 * (r >> 2) + (r & 3) < lim2) ? v : -1e30f;
          s[kt][r] = v;
          mx = fmaxf(mx, v);
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);      // first chunk: exp2(-huge) = 0 on o = 0, l = 0
      const float m2 = m_new * c;
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < CT; ++kt) {
        if (kt >= nct || (CAUSAL && kc + kt > qt)) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = s[kt][r];
          const float e = (v > -1e29f) ? __builtin_amdgcn_exp2f(fmaf(v, c, -m2)) : 0.f;   // masked entries contribute nothing
          s[kt][r] = e;
          sum += e;
        }
      }
      sum += __shfl_xor(sum, 32, 64);
      l_run = l_run * alpha + sum;
      m_run = m_new;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
#pragma unroll
      for (int kt = 0; kt < CT; ++kt) {
        if (kt >= nct || (CAUSAL && kc + kt > qt)) continue;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          float pv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) pv[e] = s[kt][8 * s2 + e];
          const bf16x8 pf = pack_frag(pv);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_trans<DH>(sV, 32 * kt + 16 * s2, 32 * dt, hi, q16, i16), pf, o[dt], 0, 0, 0);
        }
      }
    }
    if (qg < p.L) {
      const float inv = 1.0f / l_run;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        store_frag_T(p.o, p.ld_o, (long)b * p.L + qg, h * DH + 32 * dt, hi, o[dt], inv, DH - 32 * dt);
      if (p.stats && hi == 0) *(float2*)(p.stats + ((size_t)head * p.L + qg) * 2) = make_float2(m_run * c, inv);
    }
  }
}

template <int DH, bool CAUSAL>
__global__ __launch_bounds__(256, HD<DH>::WGS) void attn_bwd_long_kernel(AttnArgs p) {
  constexpr int CT = LONG_CT, RB = HD<DH>::RB, KS = HD<DH>::KS, DT = HD<DH>::DT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* img0 = smem;
  char* img1 = img0 + CT * 32 * RB;
  float* sM = (float*)(img0 + 2 * CT * 32 * RB);
  float* sL = sM + LONG_LMAX;
  float* sD = sL + LONG_LMAX;
  const int l31 = lane & 31, hi = lane >> 5, q16 = (lane >> 4) & 1, i16 = lane & 15;
  const long head = blockIdx.x;
  const int b = (int)(head / p.H), h = (int)(head - (long)b * p.H);
  const size_t hoff = ((size_t)b * p.L * p.ld_qkv + (size_t)h * DH) * 2;
  const size_t ooff = ((size_t)b * p.L * p.ld_o + (size_t)h * DH) * 2;
  const __amdgpu_buffer_rsrc_t rsQ = chunk_rsrc<DH>(p.q + hoff, p.ld_qkv, 0, p.L), rsK = chunk_rsrc<DH>(p.k + hoff, p.ld_qkv, 0, p.L),
                               rsV = chunk_rsrc<DH>(p.v + hoff, p.ld_qkv, 0, p.L), rsDO = chunk_rsrc<DH>(p.d_o + ooff, p.ld_o, 0, p.L),
                               rsO = chunk_rsrc<DH>(p.o_in + ooff, p.ld_o, 0, p.L);
  const float c = p.scale * 1.4426950408889634f;
  const float* stats = p.stats + (size_t)head * p.L * 2;
  const int NT = (p.L + 31) / 32;

  // ---- sweep 1: dQ (query groups outside, key chunks inside) ----------------------------------------------
  for (int qt0 = 0; qt0 < NT; qt0 += 4) {
    const int qt = qt0 + wave;
    const int qg = 32 * qt + l31;
    bf16x8 fq[KS], fdo[KS], fo[KS];
    load_frags<KS, DH>(rsQ, p.ld_qkv, qg, hi, fq);
    load_frags<KS, DH>(rsDO, p.ld_o, qg, hi, fdo);
    load_frags<KS, DH>(rsO, p.ld_o, qg, hi, fo);
    float2 st = make_float2(0.f, 0.f);                 // padded queries: inv = 0 -> P = 0
    if (qg < p.L) st = *(const float2*)(stats + qg * 2);
    float Dq = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float a[8], o8[8];
      unpack8(__builtin_bit_cast(u32x4, fdo[ks]), a);
      unpack8(__builtin_bit_cast(u32x4, fo[ks]), o8);
#pragma unroll
      for (int i = 0; i < 8; ++i) Dq += a[i] * o8[i];
    }
    Dq += __shfl_xor(Dq, 32, 64);
    if (hi == 0 && qt < NT) { sM[qg] = st.x; sL[qg] = st.y; sD[qg] = Dq; }
    const int lim2 = (CAUSAL ? min(p.L, qg + 1) : p.L) - 4 * hi;
    f32x16 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
    for (int kc = 0; kc < NT; kc += CT) {
      if (CAUSAL && kc > qt0 + 3) break;
      const int nct = min(CT, NT - kc);
      __syncthreads();
      dma_image<DH>(chunk_rsrc<DH>(p.k + hoff, p.ld_qkv, 32 * kc, p.L), img0, nct * 32, p.ld_qkv, wave, lane, 4);
      dma_image<DH>(chunk_rsrc<DH>(p.v + hoff, p.ld_qkv, 32 * kc, p.L), img1, nct * 32, p.ld_qkv, wave, lane, 4);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      for (int kt = 0; kt < nct; ++kt) {
        const int T = kc + kt;
        if (CAUSAL && T > qt) break;
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_direct<DH>(img0, 32 * kt, l31, hi, ks), fq[ks], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_direct<DH>(img1, 32 * kt, l31, hi, ks), fdo[ks], dp, 0, 0, 0);
        }
        const bool full = (32 * T + 32 <= p.L) && (!CAUSAL || T < qt);
        float ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float pe = __builtin_amdgcn_exp2f(fmaf(s[r], c, -st.x)) * st.y;
          if (!full) pe = (32 * T + 8 * (r >> 2) + (r & 3) < lim2) ? pe : 0.f;
          ds[r] = pe * (dp[r] - Dq) * p.scale;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 dsf = pack_frag(ds + 8 * s2);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
            dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_trans<DH>(img0, 32 * kt + 16 * s2, 32 * dt, hi, q16, i16), dsf, dq[dt], 0, 0, 0);
        }
      }
    }
    if (qg < p.L) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        store_frag_T(p.dq, p.ld_dqkv, (long)b * p.L + qg, h * DH + 32 * dt, hi, dq[dt], 1.0f, DH - 32 * dt);
    }
  }

  // ---- sweep 2: dK, dV (key groups outside, query chunks inside) -----------------------------------------
  for (int kt0 = 0; kt0 < NT; kt0 += 4) {
    const int kt = kt0 + wave;
    const int kg = 32 * kt + l31;
    const int kgc = l31 - 4 * hi;      // causal test inside the diagonal tile: key <= query
    bf16x8 fk[KS], fv[KS];
    load_frags<KS, DH>(rsK, p.ld_qkv, kg, hi, fk);
    load_frags<KS, DH>(rsV, p.ld_qkv, kg, hi, fv);
    f32x16 dk[DT], dv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
    for (int qc = 0; qc < NT; qc += CT) {
      const int nct = min(CT, NT - qc);
      if (CAUSAL && qc + nct <= kt0) continue;          // every query of the chunk lies before every key of the group
      __syncthreads();                                  // (first pass: also orders sweep 1's statistics before their readers)
      dma_image<DH>(chunk_rsrc<DH>(p.q + hoff, p.ld_qkv, 32 * qc, p.L), img0, nct * 32, p.ld_qkv, wave, lane, 4);
      dma_image<DH>(chunk_rsrc<DH>(p.d_o + ooff, p.ld_o, 32 * qc, p.L), img1, nct * 32, p.ld_o, wave, lane, 4);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      for (int qi = 0; qi < nct; ++qi) {
        const int T = qc + qi;                          // global query tile
        if (CAUSAL && T < kt) continue;
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_direct<DH>(img0, 32 * qi, l31, hi, ks), fk[ks], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_direct<DH>(img1, 32 * qi, l31, hi, ks), fv[ks], dp, 0, 0, 0);
        }
        float pr[16], ds[16];
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int q0 = 32 * T + 8 * rq + 4 * hi;
          const float4 m4 = *(const float4*)(sM + q0);
          const float4 l4 = *(const float4*)(sL + q0);
          const float4 d4 = *(const float4*)(sD + q0);
          const float mm[4] = {m4.x, m4.y, m4.z, m4.w};
          const float ll[4] = {l4.x, l4.y, l4.z, l4.w};
          const float dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * rq + e;
            float pe = __builtin_amdgcn_exp2f(fmaf(s[r], c, -mm[e])) * ll[e];
            if (CAUSAL && T == kt) pe = (kgc <= 8 * rq + e) ? pe : 0.f;
            pr[r] = pe;
            ds[r] = pe * (dp[r] - dd[e]) * p.scale;
          }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const bf16x8 pf = pack_frag(pr + 8 * s2);
          const bf16x8 dsf = pack_frag(ds + 8 * s2);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_trans<DH>(img1, 32 * qi + 16 * s2, 32 * dt, hi, q16, i16), pf, dv[dt], 0, 0, 0);
            dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_trans<DH>(img0, 32 * qi + 16 * s2, 32 * dt, hi, q16, i16), dsf, dk[dt], 0, 0, 0);
          }
        }
      }
    }
    if (kg < p.L) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        store_frag_T(p.dk, p.ld_dqkv, (long)b * p.L + kg, h * DH + 32 * dt, hi, dk[dt], 1.0f, DH - 32 * dt);
        store_frag_T(p.dv, p.ld_dqkv, (long)b * p.L + kg, h * DH + 32 * dt, hi, dv[dt], 1.0f, DH - 32 * dt);
      }
    }
  }
}

constexpr int ATTN_MAX_DEVICES = 64;

template <int NKT, int DH, bool CAUSAL>
int launch_fwd_c(const AttnArgs& a, hipStream_t st) {
  constexpr int HPW = WGHeads<NKT>::HPW;
  const int lds = HPW * 2 * NKT * 32 * HD<DH>::RB + (fwd_stages<NKT, DH>() ? attn_waves<NKT, DH>() * Stg<DH>::BYTES : 0);
  // the LDS opt-in is a per-device function attribute: once per (kernel instantiation, device), thread-safe (the
  // forward runs on the Python main thread, the backward on autograd's worker thread)
  static std::once_flag once[ATTN_MAX_DEVICES];
  static int rc_dev[ATTN_MAX_DEVICES];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ATTN_MAX_DEVICES) { clipa_set_error("attn_fwd: bad device"); return CLIPA_ERR_LAUNCH; }
  std::call_once(once[dev], [&]() {
    const hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_kernel<NKT, DH, CAUSAL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    rc_dev[dev] = 0;
    if (e != hipSuccess) { clipa_set_error("attn_fwd attr: %s", hipGetErrorString(e)); rc_dev[dev] = CLIPA_ERR_LAUNCH; }
  });
  if (rc_dev[dev]) return rc_dev[dev];
  hipLaunchKernelGGL((attn_fwd_kernel<NKT, DH, CAUSAL>), dim3((unsigned)(((long)a.B * a.H + HPW - 1) / HPW)), dim3(64 * attn_waves<NKT, DH>()), lds, st, a);
  return clipa_check_launch("attn_fwd");
}
template <int NKT, int DH>
int launch_fwd(const AttnArgs& a, hipStream_t st) {
  return a.causal ? launch_fwd_c<NKT, DH, true>(a, st) : launch_fwd_c<NKT, DH, false>(a, st);
}
template <int NKT, int DH, bool CAUSAL>
int launch_bwd_c(const AttnArgs& a, hipStream_t st) {
  constexpr int HPW = WGHeads<NKT>::HPW;
  const int lds = HPW * (2 * NKT * 32 * HD<DH>::RB + 3 * NKT * 32 * 4) + (bwd_stages<NKT, DH>() ? attn_waves<NKT, DH>() * Stg<DH>::BYTES : 0);
  // the LDS opt-in is a per-device function attribute: once per (kernel instantiation, device), thread-safe (the
  // forward runs on the Python main thread, the backward on autograd's worker thread)
  static std::once_flag once[ATTN_MAX_DEVICES];
  static int rc_dev[ATTN_MAX_DEVICES];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ATTN_MAX_DEVICES) { clipa_set_error("attn_bwd: bad device"); return CLIPA_ERR_LAUNCH; }
  std::call_once(once[dev], [&]() {
    const hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_kernel<NKT, DH, CAUSAL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    rc_dev[dev] = 0;
    if (e != hipSuccess) { clipa_set_error("attn_bwd attr: %s", hipGetErrorString(e)); rc_dev[dev] = CLIPA_ERR_LAUNCH; }
  });
  if (rc_dev[dev]) return rc_dev[dev];
  hipLaunchKernelGGL((attn_bwd_kernel<NKT, DH, CAUSAL>), dim3((unsigned)(((long)a.B * a.H + HPW - 1) / HPW)), dim3(64 * attn_waves<NKT, DH>()), lds, st, a);
  return clipa_check_launch("attn_bwd");
}
template <int NKT, int DH>
int launch_bwd(const AttnArgs& a, hipStream_t st) {
  return a.causal ? launch_bwd_c<NKT, DH, true>(a, st) : launch_bwd_c<NKT, DH, false>(a, st);
}

template <int DH, bool CAUSAL, bool BWD>
int launch_long(const AttnArgs& a, hipStream_t st) {
  const int lds = 2 * LONG_CT * 32 * HD<DH>::RB + (BWD ? 3 * LONG_LMAX * 4 : 0);
  const void* fn = BWD ? (const void*)attn_bwd_long_kernel<DH, CAUSAL> : (const void*)attn_fwd_long_kernel<DH, CAUSAL>;
  static std::once_flag once[ATTN_MAX_DEVICES];
  static int rc_dev[ATTN_MAX_DEVICES];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ATTN_MAX_DEVICES) { clipa_set_error("attention: bad device"); return CLIPA_ERR_LAUNCH; }
  std::call_once(once[dev], [&]() {
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    rc_dev[dev] = 0;
    if (e != hipSuccess) { clipa_set_error("attention (long) attr: %s", hipGetErrorString(e)); rc_dev[dev] = CLIPA_ERR_LAUNCH; }
  });
  if (rc_dev[dev]) return rc_dev[dev];
  const dim3 grid((unsigned)((long)a.B * a.H));
  if (BWD) hipLaunchKernelGGL((attn_bwd_long_kernel<DH, CAUSAL>), grid, dim3(256), lds, st, a);
  else hipLaunchKernelGGL((attn_fwd_long_kernel<DH, CAUSAL>), grid, dim3(256), lds, st, a);
  return clipa_check_launch(BWD ? "attn_bwd_long" : "attn_fwd_long");
}
#ifndef CLIPA_ATTN_WIDE
template <bool BWD>
int dispatch_long(const AttnArgs& a, int64_t dh, hipStream_t st) {
  if (dh == 80) return a.causal ? launch_long<80, true, BWD>(a, st) : launch_long<80, false, BWD>(a, st);
  return a.causal ? launch_long<64, true, BWD>(a, st) : launch_long<64, false, BWD>(a, st);
}
#endif

bool wide_head(int64_t dh) { return dh == 88 || dh == 104 || dh == 112; }

int check_args(int64_t B, int64_t H, int64_t L, int64_t dh, int64_t ld_qkv, int64_t ld_o, int causal) {
  if (dh != 64 && dh != 80 && !wide_head(dh)) { clipa_set_error("attention: head dim %ld unsupported (64, 80, 88, 104, 112)", (long)dh); return CLIPA_ERR_ARG; }
  if (wide_head(dh) && causal) { clipa_set_error("attention: head dim %ld is compiled for image towers only (no causal mask)", (long)dh); return CLIPA_ERR_ARG; }
  if (L <= 0 || L > LONG_LMAX) { clipa_set_error("attention: L=%ld outside (0, %d]", (long)L, LONG_LMAX); return CLIPA_ERR_ARG; }
  if (ld_qkv % 8 != 0 || ld_o % 8 != 0) { clipa_set_error("attention: row strides must be multiples of 8 elements"); return CLIPA_ERR_ARG; }
  if (B * H <= 0 || B * H > 0x7fffffffL) { clipa_set_error("attention: bad B*H"); return CLIPA_ERR_ARG; }
  return 0;
}

}  // namespace

#define ATTN_DISPATCH_DH(fn, DH, a, st)                                     \
  switch (((a).L + 31) / 32) {                                               \
    case 1: return fn<1, DH>(a, st); case 2: return fn<2, DH>(a, st);        \
    case 3: return fn<3, DH>(a, st); case 4: return fn<4, DH>(a, st);        \
    case 5: return fn<5, DH>(a, st); case 6: return fn<6, DH>(a, st);        \
    case 7: return fn<7, DH>(a, st); case 8: return fn<8, DH>(a, st);        \
    default: return fn<9, DH>(a, st);                                        \
  }
#ifdef CLIPA_ATTN_WIDE
// attention_wide.hip: the non-causal kernels of the wide heads (ViT-g/14 88, ViT-bigG/14 104, ViT-e/14 112; image towers only),
// a translation unit of their own so that the build stays parallel.  `args` is this file's AttnArgs.
template <int NKT, int DH> int wide_fwd(const AttnArgs& a, hipStream_t st) { return launch_fwd_c<NKT, DH, false>(a, st); }
template <int NKT, int DH> int wide_bwd(const AttnArgs& a, hipStream_t st) { return launch_bwd_c<NKT, DH, false>(a, st); }
template <int DH>
int wide_dh(const AttnArgs& a, int bwd, hipStream_t st) {
  if (a.L > 288) return bwd ? launch_long<DH, false, true>(a, st) : launch_long<DH, false, false>(a, st);
  if (bwd) { ATTN_DISPATCH_DH(wide_bwd, DH, a, st) }
  ATTN_DISPATCH_DH(wide_fwd, DH, a, st)
}
extern "C" int clipa_attn_wide_launch(const void* args, int64_t dh, int bwd, void* stream) {
  const AttnArgs& a = *(const AttnArgs*)args;
  if (dh == 88) return wide_dh<88>(a, bwd, (hipStream_t)stream);
  if (dh == 104) return wide_dh<104>(a, bwd, (hipStream_t)stream);
  return wide_dh<112>(a, bwd, (hipStream_t)stream);
}
#else
extern "C" int clipa_attn_wide_launch(const void* args, int64_t dh, int bwd, void* stream);   // attention_wide.hip
#ifdef CLIPA_ATTN_PERSISTENT_EXPERIMENT
// tools/probes/attention_persistent/: the single-sweep backward / persistent forward of round 5, measured slower than the
// kernels of this file (profiles/r05_attention_persistent_single_sweep.md); a probe build links them in and selects them with
// clipa_internal_debug_set flags 262144 (backward) / 524288 (forward)
extern "C" int clipa_attn_bwd1_try(const void* args, int64_t dh, void* stream, int* rc);
extern "C" int clipa_attn_fwd1_try(const void* args, int64_t dh, void* stream, int* rc);
#endif

#define ATTN_DISPATCH(fn, dh, a, st)                 \
  if ((dh) == 80) { ATTN_DISPATCH_DH(fn, 80, a, st) } \
  ATTN_DISPATCH_DH(fn, 64, a, st)

extern "C" int clipa_attention_fwd(const void* q, const void* k, const void* v, void* out, float* stats,
                                   int64_t B, int64_t H, int64_t L, int64_t dh, int64_t ld_qkv, int64_t ld_o,
                                   float scale, int causal, void* stream) {
  if (B * H == 0) return CLIPA_OK;
  if (int rc = check_args(B, H, L, dh, ld_qkv, ld_o, causal)) return rc;
  AttnArgs a = {};
  a.q = (const char*)q; a.k = (const char*)k; a.v = (const char*)v; a.ld_qkv = ld_qkv;
  a.o = (char*)out; a.ld_o = ld_o; a.B = (int)B; a.H = (int)H; a.L = (int)L; a.scale = scale; a.causal = causal;
  a.stats = stats;
  if (wide_head(dh)) return clipa_attn_wide_launch(&a, dh, 0, stream);
  if (L > 288) return dispatch_long<false>(a, dh, (hipStream_t)stream);
#ifdef CLIPA_ATTN_PERSISTENT_EXPERIMENT
  int rc1 = 0;
  if ((clipa_internal_debug_flags() & 524288) && clipa_attn_fwd1_try(&a, dh, stream, &rc1)) return rc1;
#endif
  ATTN_DISPATCH(launch_fwd, dh, a, (hipStream_t)stream)
}

extern "C" int clipa_attention_bwd(const void* q, const void* k, const void* v, const void* out,
                                   const void* d_out, const float* stats, void* dq, void* dk, void* dv, int64_t B, int64_t H,
                                   int64_t L, int64_t dh, int64_t ld_qkv, int64_t ld_o, int64_t ld_dqkv,
                                   float scale, int causal, void* stream) {
  if (B * H == 0) return CLIPA_OK;
  if (int rc = check_args(B, H, L, dh, ld_qkv, ld_o, causal)) return rc;
  if (ld_dqkv % 8 != 0) { clipa_set_error("attention_bwd: ld_dqkv must be a multiple of 8"); return CLIPA_ERR_ARG; }
  if (!stats) { clipa_set_error("attention_bwd: needs the forward's softmax statistics"); return CLIPA_ERR_ARG; }
  AttnArgs a = {};
  a.q = (const char*)q; a.k = (const char*)k; a.v = (const char*)v; a.ld_qkv = ld_qkv;
  a.o_in = (const char*)out; a.d_o = (const char*)d_out; a.ld_o = ld_o;
  a.dq = (char*)dq; a.dk = (char*)dk; a.dv = (char*)dv; a.ld_dqkv = ld_dqkv;
  a.B = (int)B; a.H = (int)H; a.L = (int)L; a.scale = scale; a.causal = causal;
  a.stats = const_cast<float*>(stats);
  if (wide_head(dh)) return clipa_attn_wide_launch(&a, dh, 1, stream);
  if (L > 288) return dispatch_long<true>(a, dh, (hipStream_t)stream);
#ifdef CLIPA_ATTN_PERSISTENT_EXPERIMENT
  int rc1 = 0;
  if ((clipa_internal_debug_flags() & 262144) && clipa_attn_bwd1_try(&a, dh, stream, &rc1)) return rc1;
#endif
  ATTN_DISPATCH(launch_bwd, dh, a, (hipStream_t)stream)
}
// ---- packed variable-length sequences -------------------------------------------------------------------------------------
// The text tower's captions end at their EOT token; under the causal mask no later position can reach the pooled output
// (model.py:251-254: x[arange, text.argmax(-1)]) or receive gradient, so the engine may run the tower on the tokens up to EOT
// only, packed back to back in one [T, 3D] matrix.  One launch handles the `nseq` sequences listed in seq_ids (or all of
// 0 .. nseq-1) whose lengths are <= 32 * tiles; seq_start / seq_len are indexed by sequence id.  Statistics: [T, H] pairs laid
// out per sequence as [H][len].  Head dims 64 / 80, tiles <= 9.
namespace {
int varlen_args(const int* seq_start, const int* seq_len, int64_t nseq, int64_t tiles, int64_t dh) {
  if (!seq_start || !seq_len) { clipa_set_error("attention (varlen): seq_start / seq_len missing"); return CLIPA_ERR_ARG; }
  if (tiles < 1 || tiles > 9) { clipa_set_error("attention (varlen): tiles=%ld outside [1, 9] (sequences of up to 288 tokens)", (long)tiles); return CLIPA_ERR_ARG; }
  if (dh != 64 && dh != 80) { clipa_set_error("attention (varlen): head dim %ld unsupported (64 and 80)", (long)dh); return CLIPA_ERR_ARG; }
  (void)nseq;
  return 0;
}
}  // namespace

extern "C" int clipa_attention_fwd_varlen(const void* q, const void* k, const void* v, void* out, float* stats,
                                          const int32_t* seq_start, const int32_t* seq_len, const int32_t* seq_ids, int64_t nseq,
                                          int64_t tiles, int64_t H, int64_t dh, int64_t ld_qkv, int64_t ld_o, float scale,
                                          int causal, void* stream) {
  if (nseq * H == 0) return CLIPA_OK;
  if (int rc = check_args(nseq, H, 32 * tiles, dh, ld_qkv, ld_o, causal)) return rc;
  if (int rc = varlen_args(seq_start, seq_len, nseq, tiles, dh)) return rc;
  AttnArgs a = {};
  a.q = (const char*)q; a.k = (const char*)k; a.v = (const char*)v; a.ld_qkv = ld_qkv;
  a.o = (char*)out; a.ld_o = ld_o; a.B = (int)nseq; a.H = (int)H; a.L = (int)(32 * tiles); a.scale = scale; a.causal = causal;
  a.stats = stats;
  a.seq_start = seq_start; a.seq_len = seq_len; a.seq_ids = seq_ids;
  ATTN_DISPATCH(launch_fwd, dh, a, (hipStream_t)stream)
}

extern "C" int clipa_attention_bwd_varlen(const void* q, const void* k, const void* v, const void* out, const void* d_out,
                                          const float* stats, void* dq, void* dk, void* dv, const int32_t* seq_start,
                                          const int32_t* seq_len, const int32_t* seq_ids, int64_t nseq, int64_t tiles, int64_t H,
                                          int64_t dh, int64_t ld_qkv, int64_t ld_o, int64_t ld_dqkv, float scale, int causal,
                                          void* stream) {
  if (nseq * H == 0) return CLIPA_OK;
  if (int rc = check_args(nseq, H, 32 * tiles, dh, ld_qkv, ld_o, causal)) return rc;
  if (int rc = varlen_args(seq_start, seq_len, nseq, tiles, dh)) return rc;
  if (ld_dqkv % 8 != 0) { clipa_set_error("attention_bwd (varlen): ld_dqkv must be a multiple of 8"); return CLIPA_ERR_ARG; }
  if (!stats) { clipa_set_error("attention_bwd (varlen): needs the forward's softmax statistics"); return CLIPA_ERR_ARG; }
  AttnArgs a = {};
  a.q = (const char*)q; a.k = (const char*)k; a.v = (const char*)v; a.ld_qkv = ld_qkv;
  a.o_in = (const char*)out; a.d_o = (const char*)d_out; a.ld_o = ld_o;
  a.dq = (char*)dq; a.dk = (char*)dk; a.dv = (char*)dv; a.ld_dqkv = ld_dqkv;
  a.B = (int)nseq; a.H = (int)H; a.L = (int)(32 * tiles); a.scale = scale; a.causal = causal;
  a.stats = const_cast<float*>(stats);
  a.seq_start = seq_start; a.seq_len = seq_len; a.seq_ids = seq_ids;
  ATTN_DISPATCH(launch_bwd, dh, a, (hipStream_t)stream)
}
#endif   // CLIPA_ATTN_WIDE
