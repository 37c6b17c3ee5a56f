// Shared pieces of the fused attention kernels (attention.hip, attention_wide.hip, attention_bwd1.hip): argument block, LDS image
// geometry per head dim, the chunk swizzle that serves direct and transposed fragment reads, LDS-DMA of a head's rows, MFMA
// fragment loads, whole-row output stores through a per-wave LDS window, the in-register row softmax.
#pragma once
#include "common.h"

namespace {

struct AttnArgs {
  const char* q; const char* k; const char* v; long ld_qkv;   // bf16 rows, stride in elements
  char* o; const char* o_in; const char* d_o; long ld_o;
  char* dq; char* dk; char* dv; long ld_dqkv;
  float* stats;   // [B*H*L][2] = (c*rowmax, 1/rowsum), written by fwd, read by bwd
  int B, H, L;
  float scale;
  int causal;
  // packed variable-length sequences (short kernels only; clipa_attention_*_varlen): sequence i of this launch is
  // seq_ids[i] (or i), its rows are [seq_start[s], seq_start[s] + seq_len[s]) of the token matrix, B = sequences of the launch
  const int* seq_start; const int* seq_len; const int* seq_ids;
  int abl;        // timing ablations of the persistent kernels (internal_hooks.h; wrong results): 1 = no pair arithmetic, 2 = no output stores
};

// Geometry per head dim.  dh = 64 (ViT-S/B/L, every text tower): 128-byte LDS rows, 4 k-steps, 2 output tiles.
// dh = 80 (ViT-H/14, transformer.py:126 with head_width 80): 256-byte LDS rows of which 160 B carry data
// (power-of-two row stride keeps the XOR swizzle), 5 k-steps, 3 output tiles whose last 16 columns are
// never stored; one workgroup per CU instead of two.
template <int DH>
struct HD {
  static constexpr int RB = DH == 64 ? 128 : 256;   // LDS row bytes
  static constexpr int NCH = RB / 16;               // 16-byte chunk positions per row
  static constexpr int KS = (DH + 15) / 16;         // 16-wide reduction steps over the head dim (a ragged last step sees zeros)
  static constexpr int DT = (DH + 31) / 32;         // 32-wide output tiles over the head dim
  static constexpr int WGS = DH == 64 ? 2 : 1;      // workgroups per CU the LDS image allows
};
// Short sequences (CLIPA's 84-112 px images = 26-50 tokens, 8-32 token captions) fit one or two 32-row tiles: a
// workgroup then takes 4 or 2 heads (one or two waves each) instead of idling three or two of its waves.
template <int NKT>
struct WGHeads {
  static constexpr int HPW = NKT == 1 ? 4 : (NKT == 2 ? 2 : 1);   // heads per workgroup
  static constexpr int WPH = 4 / HPW;                             // waves per head
};
// Waves per workgroup.  Head dim 80 with five or more key tiles (ViT-H/14: 257 tokens = 9 tiles) has LDS for ONE workgroup per
// CU; four waves there leave every SIMD a single wave and nothing to overlap its MFMA -> softmax -> MFMA chain with, so those
// instantiations run eight waves (two per SIMD, one 32-row tile each for all but one wave).
template <int NKT, int DH>
// (Round 6, "one tile per wave": 7 or 8 waves for the 7 tiles of 197 tokens need 154-165 registers each, which leaves ONE
// workgroup per CU - forward 1.63 -> 2.03 ms, backward 4.76 -> 5.61 ms; 9 waves for the 9 tiles of 257 tokens at head dim 80 cap a
// wave at 168 registers: 53-57 spills.  profiles/r06_attention_waves_per_workgroup_ab.jsonl)
__host__ __device__ constexpr int attn_waves() { return (DH == 80 && NKT >= 5) ? 8 : 4; }   // (wider heads spill at 256 registers per wave)

// chunk permutation of an LDS row: conflict-free for both the direct ds_read_b128 operand reads and the
// transposed ds_read_b64_tr_b16 reads (tools/lds_bank_sim.py) - 8 chunks per 128-B row, 16 per 256-B row
template <int DH>
__device__ __forceinline__ int swz_u(int row) {
  if (DH == 64) return (((row >> 1) & 1) << 2) | ((row >> 2) & 1) | (((row >> 3) & 1) << 1);
  return ((row & 3) << 2) | ((row >> 2) & 3);
}

// DMA rows [0, LP) of one head into a swizzled LDS image; rows >= L and chunks >= dh read zeros
template <int DH>
__device__ __forceinline__ void dma_image(const __amdgpu_buffer_rsrc_t rs, char* img, int LP, long ld,
                                          int wave, int lane, int nwaves = 4) {
  constexpr int NCH = HD<DH>::NCH, RPP = 1024 / HD<DH>::RB;
  for (int pc = wave; pc < LP / RPP; pc += nwaves) {
    const int row = pc * RPP + lane / NCH;
    const int chunk = (lane & (NCH - 1)) ^ swz_u<DH>(row);
    const unsigned oob = (chunk * 8 >= DH) ? 0x80000000u : 0u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(img + pc * 1024), 16,
                                             (unsigned)(row * ld * 2 + chunk * 16) | oob, 0, 0, 0);
  }
}

// direct operand fragment: lane -> image row (rowbase + l31), k-chunk 2*ks+hi
template <int DH>
__device__ __forceinline__ bf16x8 frag_direct(const char* img, int rowbase, int l31, int hi, int ks) {
  return *(const bf16x8*)(img + (rowbase + l31) * HD<DH>::RB + (((2 * ks + hi) ^ swz_u<DH>(l31)) << 4));
}

// transposed operand fragment for a 16-wide reduction step: rows rowbase16 + {4hi+0..3, 8+4hi+0..3},
// column (colbase + 16*q16 + i16) -> 8 k-slots matching the register order of a 32x32 C fragment
template <int DH>
__device__ __forceinline__ bf16x8 frag_trans(const char* img, int rowbase16, int colbase, int hi, int q16, int i16) {
  constexpr int RB = HD<DH>::RB;
  const int col = colbase + 16 * q16 + 4 * (i16 & 3);
  const int r0 = rowbase16 + 4 * hi + (i16 >> 2);
  const int r1 = r0 + 8;
  const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) bf16x4*)(img + r0 * RB + (((col >> 3) ^ swz_u<DH>(r0)) << 4) + (col & 7) * 2));
  const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) bf16x4*)(img + r1 * RB + (((col >> 3) ^ swz_u<DH>(r1)) << 4) + (col & 7) * 2));
  bf16x8 f;
  f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3];
  f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
  return f;
}

__device__ __forceinline__ bf16x8 pack_frag(const float* v) {
  u32x4 w;
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = pack2bf(v[2 * i], v[2 * i + 1]);
  return __builtin_bit_cast(bf16x8, w);
}

// store a 32x32 fragment held as X^T[d][row] (lane: row = l31, d = 8*(r>>2)+4*hi+(r&3)) into a
// row-major bf16 matrix: 4 consecutive d per store
__device__ __forceinline__ void store_frag_T(char* base, long ld, long row, int col0, int hi, const f32x16& a, float mul,
                                             int ncols = 32) {
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    if (8 * qd >= ncols) continue;     // partial last tile of a head dim that is not a multiple of 32
    u32x2 w;
    w[0] = pack2bf(a[4 * qd + 0] * mul, a[4 * qd + 1] * mul);
    w[1] = pack2bf(a[4 * qd + 2] * mul, a[4 * qd + 3] * mul);
    *(u32x2*)(base + (row * ld + col0 + 8 * qd + 4 * hi) * 2) = w;
  }
}

// The same tile (head dim 64: both 32-column fragments of a 32-row tile = 32 rows of 128 B) stored as WHOLE 128-byte rows:
// store_frag_T writes 8 bytes per lane to 32 different rows per instruction (16 instructions, every one touching 32 cache
// lines); here the tile passes through 4 KB of LDS private to the wave (16-byte chunks XOR-swizzled by the row, conflict-free
// both ways) and leaves as 4 instructions of 8 full rows each.  The global stores of the backward cost 21 % of its time
// in the scattered form (profiles/r03_attention_bwd_ablations.jsonl).
__device__ __forceinline__ void store_tile64(char* stage, char* base, long ld, long row0, int rows_valid, int col0, int lane,
                                             const f32x16& a0, const f32x16& a1, float mul) {
  const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    const f32x16& a = dt ? a1 : a0;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      u32x2 w;
      w[0] = pack2bf(a[4 * qd + 0] * mul, a[4 * qd + 1] * mul);
      w[1] = pack2bf(a[4 * qd + 2] * mul, a[4 * qd + 3] * mul);
      *(u32x2*)(stage + l31 * 128 + (((4 * dt + qd) ^ (l31 & 7)) << 4) + 8 * hi) = w;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // one wave: LDS operations retire in order, only the compiler needs telling
  const int chunk = lane & 7;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 8 * j + (lane >> 3);
    const u32x4 v = *(const u32x4*)(stage + row * 128 + ((chunk ^ (row & 7)) << 4));
    if (row < rows_valid) *(u32x4*)(base + ((row0 + row) * ld + col0) * 2 + chunk * 16) = v;
  }
  asm volatile("" ::: "memory");
}

// Head dim 80 (ViT-H/14): rows of 160 bytes = 10 chunks, three 32-column fragments of which the last 16 columns do not exist.
// Its 256-byte-row images leave 10-13 KB of LDS per workgroup, so the tile goes through the window in two halves of 16 rows
// (2560 B per wave); 160 = 10 x 16 makes the read side linear in the lane index.
__device__ __forceinline__ void store_tile80(char* stage, char* base, long ld, long row0, int rows_valid, int col0, int lane,
                                             const f32x16& a0, const f32x16& a1, const f32x16& a2, float mul) {
  const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if ((l31 >> 4) == h) {
#pragma unroll
      for (int dt = 0; dt < 3; ++dt) {
        const f32x16& a = dt == 0 ? a0 : (dt == 1 ? a1 : a2);
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          if (4 * dt + qd >= 10) continue;
          u32x2 w;
          w[0] = pack2bf(a[4 * qd + 0] * mul, a[4 * qd + 1] * mul);
          w[1] = pack2bf(a[4 * qd + 2] * mul, a[4 * qd + 3] * mul);
          *(u32x2*)(stage + (l31 & 15) * 160 + (4 * dt + qd) * 16 + 8 * hi) = w;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int task = 64 * j + lane;               // 16 rows x 10 chunks
      if (task < 160) {
        const u32x4 v = *(const u32x4*)(stage + task * 16);
        const int row = task / 10, chunk = task - 10 * row;
        if (16 * h + row < rows_valid) *(u32x4*)(base + ((row0 + 16 * h + row) * ld + col0) * 2 + chunk * 16) = v;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}
template <int DH>
struct Stg { static constexpr int BYTES = DH == 64 ? 4096 : 2560; };      // LDS window per wave
template <int DH>
__device__ __forceinline__ void store_tile(char* stage, char* base, long ld, long row0, int rows_valid, int col0, int lane,
                                           const f32x16* a, float mul) {
  if constexpr (DH == 64) store_tile64(stage, base, ld, row0, rows_valid, col0, lane, a[0], a[1], mul);
  else store_tile80(stage, base, ld, row0, rows_valid, col0, lane, a[0], a[1], a[2], mul);
}

// Row softmax over the transposed score fragments of one 32-query tile.  In: raw q.k scores.  Out: s =
// exp2(c*(s - rowmax)) (un-normalised, c = scale*log2 e folded into one FMA), inv = 1/rowsum, m2 = c*rowmax.
// key(kt, r) = 32kt + 8(r>>2) + (r&3) + 4hi is valid iff < lim (padding / causal bound); tiles valid for
// every lane skip the compare/select: without a causal mask only the last key tile can be partial (NKT =
// ceil(L / 32) exactly), so the image tower pays 16 compares per row instead of 16 * NKT.
template <int NKT, bool CAUSAL>
__device__ __forceinline__ void softmax_rows(f32x16 (&s)[NKT], const AttnArgs& p, int qt, int qg, int hi,
                                             float& inv, float& m2) {
  const float c = p.scale * 1.4426950408889634f;
  const int lim2 = (CAUSAL ? min(p.L, qg + 1) : p.L) - 4 * hi;
  float mx = -1e30f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    const bool full = CAUSAL ? ((32 * kt + 32 <= p.L) && kt < qt) : (kt < NKT - 1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = s[kt][r];
      if (!full) v = (32 * kt + 8 * (r >> 2) + (r & 3) < lim2) ? v : -1e30f;
      s[kt][r] = v;
      mx = fmaxf(mx, v);
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  m2 = mx * c;
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __builtin_amdgcn_exp2f(fmaf(s[kt][r], c, -m2));
      s[kt][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 32, 64);
  inv = 1.0f / sum;
}

// load the k-step fragments of one 32-row tile straight from global memory (rows >= L read zeros)
// (chunks at or beyond the head dim - the ragged last k-step of head dims 88 / 104 - would be the NEXT head's columns: their
// offset is pushed out of the descriptor's range, which returns zeros, as dma_image does for the LDS images)
template <int KS, int DH>
__device__ __forceinline__ void load_frags(const __amdgpu_buffer_rsrc_t rs, long ld, int row, int hi, bf16x8 (&f)[KS]) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const unsigned oob = ((2 * ks + hi) * 8 >= DH) ? 0x80000000u : 0u;
    f[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(row * ld * 2 + (2 * ks + hi) * 16) | oob, 0, 0));
  }
}

// whole-row stores (store_tile) need 16 / 10 KB of LDS more per workgroup: taken where the workgroups per CU stay what they are
template <int NKT, int DH>
__host__ __device__ constexpr bool fwd_stages() {
  return (DH == 64 || DH == 80) && (WGHeads<NKT>::HPW * 2 * NKT * 32 * HD<DH>::RB + attn_waves<NKT, DH>() * Stg<DH>::BYTES) * HD<DH>::WGS <= 160 * 1024;
}
template <int NKT, int DH>
__host__ __device__ constexpr bool bwd_stages() {
  return (DH == 64 || DH == 80) && (WGHeads<NKT>::HPW * (2 * NKT * 32 * HD<DH>::RB + 3 * NKT * 32 * 4) + attn_waves<NKT, DH>() * Stg<DH>::BYTES) * HD<DH>::WGS <= 160 * 1024;
}

// Where the (batch, head) pair of this workgroup slot lives: fixed-length batches have sequence b at rows b * L; packed
// variable-length launches (the text tower on the tokens up to each caption's EOT) look the sequence up.  `p` is the kernel's
// view of the arguments with L = THIS sequence's length; statistics sit at [row0 * H + h * L + query].
#define ATTN_LOCATE_SEQUENCE                                                     \
  AttnArgs p = pin;                                                              \
  long row0 = (long)b * pin.L;                                                   \
  size_t stat0 = (size_t)head * pin.L;                                           \
  if (pin.seq_len) {                                                             \
    const int sid = pin.seq_ids ? pin.seq_ids[b] : b;                            \
    p.L = pin.seq_len[sid];                                                      \
    row0 = pin.seq_start[sid];                                                   \
    stat0 = (size_t)row0 * pin.H + (size_t)h * p.L;                              \
  }

}  // namespace
