// Persistent attention forward for the image towers at 129 <= L <= 224 tokens, head dim 64, no mask (ViT-S/B/L-16 @ 224;
// clipa_torch/open_clip/transformer.py:223-236): the arithmetic of attention.hip's attn_fwd_kernel - S^T = K.Q^T with a whole
// softmax row in registers, O^T = V^T.P^T, whole-row stores through a per-wave LDS window - in ONE eight-wave workgroup per CU
// that walks heads with the next head's K / V images landing in a second pair of LDS images under the current head's
// arithmetic (attn_fwd_kernel: load -> wait -> compute -> store per workgroup, two workgroups per CU as the only overlap).
// The four images are separate static __shared__ arrays and the head loop is unrolled by two, so that hipcc can prove that an
// LDS-DMA into one pair never aliases the reads of the other (it drains vmcnt in front of every LDS access otherwise).
// Bit-identical to attn_fwd_kernel (same operand order in every MFMA chain and in the softmax).
#include "common.h"
#include <mutex>
#include "clipa_hip.h"
#include "attention_common.h"

namespace {

constexpr int F1_WAVES = 8;

template <int NKT>
__global__ __launch_bounds__(64 * F1_WAVES, 1) void attn_fwd1_kernel(AttnArgs p) {
  constexpr int LP = NKT * 32, RB = 128, KS = 4, DT = 2, DH = 64, IMG = LP * RB;
  __shared__ __attribute__((aligned(1024))) char sK0[IMG];
  __shared__ __attribute__((aligned(1024))) char sV0[IMG];
  __shared__ __attribute__((aligned(1024))) char sK1[IMG];
  __shared__ __attribute__((aligned(1024))) char sV1[IMG];
  __shared__ __attribute__((aligned(1024))) char sStage[F1_WAVES * 4096];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool active = wave < NKT;
  const int l31 = lane & 31, hi = lane >> 5, q16 = (lane >> 4) & 1, i16 = lane & 15;
  const long nheads = (long)p.B * p.H;
  const unsigned nrec = (unsigned)((long)(p.L - 1) * p.ld_qkv * 2 + DH * 2);
  const int qt = min(wave, NKT - 1), qg = 32 * qt + l31;
  char* const stage = sStage + wave * 4096;

  bf16x8 fq[KS];               // the wave's query rows, next head's (issued one head ahead)

  auto hoff = [&](long hd) {
    const int b = (int)(hd / p.H), h = (int)(hd - (long)b * p.H);
    return ((size_t)b * p.L * p.ld_qkv + (size_t)h * DH) * 2;
  };
  auto dma_kv = [&](long hd, char* sK, char* sV) {
    dma_image<DH>(make_rsrc(p.k + hoff(hd), nrec), sK, LP, p.ld_qkv, wave, lane, F1_WAVES);
    dma_image<DH>(make_rsrc(p.v + hoff(hd), nrec), sV, LP, p.ld_qkv, wave, lane, F1_WAVES);
  };
  auto prefetch_q = [&](long hd) { load_frags<KS, DH>(make_rsrc(p.q + hoff(hd), nrec), p.ld_qkv, qg, hi, fq); };

  auto head_body = [&](long head, const char* sK, const char* sV, char* sKn, char* sVn) {
    const long next = head + gridDim.x;
    const int b = (int)(head / p.H), h = (int)(head - (long)b * p.H);
    const long row0 = (long)b * p.L;
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0) as an instruction hipcc's wait-count pass sees (a raw asm wait
    __syncthreads();                                     // leaves it believing the query prefetch is still pending); images landed
    if (next < nheads) dma_kv(next, sKn, sVn);
    if (!active) return;       // (NKT < 8: the spare waves only help with the DMA)
    f32x16 s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_direct<DH>(sK, 32 * kt, l31, hi, ks), fq[ks], s[kt], 0, 0, 0);
      if (kt & 1) __builtin_amdgcn_sched_barrier(0);   // keep the K-fragment reads of later tiles from piling up in registers
    }
    float inv, m2;
    softmax_rows<NKT, false>(s, p, qt, qg, hi, inv, m2);
    f32x16 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
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
    if (next < nheads) prefetch_q(next);               // (lands under the stores and the next head's barrier)
    store_tile64(stage, p.o, p.ld_o, row0 + 32 * qt, p.L - 32 * qt, h * DH, lane, o[0], o[1], inv);
    if (qg < p.L && p.stats && hi == 0) *(float2*)(p.stats + ((size_t)head * p.L + qg) * 2) = make_float2(m2, inv);
  };

  long head = blockIdx.x;
  if (head >= nheads) return;
  dma_kv(head, sK0, sV0);
  prefetch_q(head);
  for (;;) {
    head_body(head, sK0, sV0, sK1, sV1);
    head += gridDim.x;
    if (head >= nheads) break;
    head_body(head, sK1, sV1, sK0, sV0);
    head += gridDim.x;
    if (head >= nheads) break;
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
int launch_fwd1(const AttnArgs& a, hipStream_t st) {
  const long nheads = (long)a.B * a.H;
  const unsigned grid = (unsigned)(nheads < num_cus() ? nheads : num_cus());
  hipLaunchKernelGGL((attn_fwd1_kernel<NKT>), dim3(grid), dim3(64 * F1_WAVES), 0, st, a);
  return clipa_check_launch("attn_fwd1");
}

}  // namespace

// -> 1 if the persistent kernel covers this problem (and was launched: *rc = its return code), 0 if the caller keeps its own
extern "C" int clipa_attn_fwd1_try(const void* args, int64_t dh, void* stream, int* rc) {
  const AttnArgs& a = *(const AttnArgs*)args;
  const int nkt = (a.L + 31) / 32;
  if (dh != 64 || a.causal || a.seq_len || nkt < 5 || nkt > 7) return 0;
  switch (nkt) {
    case 5: *rc = launch_fwd1<5>(a, (hipStream_t)stream); break;
    case 6: *rc = launch_fwd1<6>(a, (hipStream_t)stream); break;
    default: *rc = launch_fwd1<7>(a, (hipStream_t)stream); break;
  }
  return 1;
}
