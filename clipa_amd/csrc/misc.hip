// HBM-bound front-end / back-end kernels of the CLIP training step on gfx950.
// Each cites the reference lines whose arithmetic it reproduces (paths relative to
// /root/reference/clipa_torch).
#include "common.h"
#include "stream.h"
#include "clipa_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------
// K1+K2 front half: uint8/float image -> normalised bf16 patch matrix [B*g*g, Kp].
// training/train.py:191-197 (x/255, (x-mean)/std, cast) fused with the im2col of conv1
// (open_clip/transformer.py:371,491-493).  Patch elements are emitted in (ph, pw, c) order - the
// memory order of an NHWC image, so channels_last inputs are read in contiguous 3*P runs - and the
// host permutes conv1.weight [D,3,P,P] -> [D,P,P,3] to match.  Columns >= 3*P*P are zero padding.
// Pixels beyond g*P (S mod P != 0) are never read (VALID conv, transformer.py:359).
template <int DT>   // 0 = u8, 1 = bf16, 2 = f32
__global__ void patchify_kernel(const void* __restrict__ img, unsigned short* __restrict__ out,
                                int B, int S, int P, int g, int Kp, int nhwc, int normalize,
                                float m0, float m1, float m2, float is0, float is1, float is2) {
  const long total = (long)B * g * g * (Kp / 8);
  const int K = 3 * P * P;
  const bool fast8 = nhwc && (3 * P) % 8 == 0 && (3 * S) % 8 == 0 && ((size_t)img & 7) == 0;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int kc = (int)(idx % (Kp / 8));
    const long patch = idx / (Kp / 8);
    const int px = (int)(patch % g);
    const int py = (int)((patch / g) % g);
    const long b = patch / ((long)g * g);
    float f[8];
    if (DT == 0 && fast8) {
      // uint8 NHWC, 3 * P and 3 * S multiples of 8 (16- and 32-pixel patches at the usual image sizes): the 8 elements of this
      // chunk are 8 CONSECUTIVE BYTES of the image row (a patch row is 3 * P contiguous bytes in (pw, c) order) - one aligned
      // 8-byte load per thread, consecutive threads read consecutive bytes of the same patch row
      const int k0 = kc * 8;
      if (k0 < K) {
        const int ph = k0 / (3 * P), off = k0 - ph * 3 * P;
        const size_t src = (((size_t)b * S + (py * P + ph)) * S + px * P) * 3 + off;
        const uint2 raw = *(const uint2*)((const unsigned char*)img + src);
        int c = off % 3;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float v = (float)(((i < 4 ? raw.x : raw.y) >> (8 * (i & 3))) & 0xffu);
          if (normalize) {
            const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
            const float istd = c == 0 ? is0 : (c == 1 ? is1 : is2);
            v = (v * (1.0f / 255.0f) - mean) * istd;
          }
          f[i] = v;
          c = c == 2 ? 0 : c + 1;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = 0.f;
      }
      *(u32x4*)(out + (size_t)patch * Kp + kc * 8) = pack8(f);
      continue;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = kc * 8 + i;
      float v = 0.f;
      if (k < K) {
        const int c = k % 3, pw = (k / 3) % P, ph = k / (3 * P);
        const int y = py * P + ph, x = px * P + pw;
        const size_t src = nhwc ? (((size_t)b * S + y) * S + x) * 3 + c : (((size_t)b * 3 + c) * S + y) * S + x;
        if (DT == 0) v = (float)((const unsigned char*)img)[src];
        else if (DT == 1) v = bf2f(((const unsigned short*)img)[src]);
        else v = ((const float*)img)[src];
        if (normalize) {
          const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
          const float istd = c == 0 ? is0 : (c == 1 ? is1 : is2);
          v = (v * (1.0f / 255.0f) - mean) * istd;
        }
      }
      f[i] = v;
    }
    *(u32x4*)(out + (size_t)patch * Kp + kc * 8) = pack8(f);
  }
}

// ------------------------------------------------------------------------------------------------
// K3: tokens[b,0,:] = cls + pos[0]; tokens[b,1+i,:] = patch[b*g2+i,:] + pos[1+i]
// (transformer.py:496-499: cat(class_embedding) then + positional_embedding)
__global__ void assemble_tokens_kernel(const unsigned short* __restrict__ patch, const float* __restrict__ cls,
                                       const float* __restrict__ pos, unsigned short* __restrict__ tok,
                                       long B, int L, int D) {
  const int dc = D / 8;
  const long total = B * L * dc;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % dc);
    const long row = idx / dc;
    const int l = (int)(row % L);
    const long b = row / L;
    float v[8], pe[8];
    *(float4*)pe = *(const float4*)(pos + (size_t)l * D + c * 8);
    *(float4*)(pe + 4) = *(const float4*)(pos + (size_t)l * D + c * 8 + 4);
    if (l == 0) {
      *(float4*)v = *(const float4*)(cls + c * 8);
      *(float4*)(v + 4) = *(const float4*)(cls + c * 8 + 4);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = bf2f(f2bf(v[i]));   // class_embedding.to(x.dtype)
    } else {
      unpack8(*(const u32x4*)(patch + ((size_t)b * (L - 1) + (l - 1)) * D + c * 8), v);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += bf2f(f2bf(pe[i]));     // positional_embedding.to(x.dtype)
    *(u32x4*)(tok + (size_t)row * D + c * 8) = pack8(v);
  }
}

// dpos[l,:] = sum_b dtok[b,l,:]; dcls = dpos[0] (before pos add, the cls row only sees cls);
// dpatch[b*g2+i,:] = dtok[b,1+i,:]
// Batch sums are two-level and ORDERED (per-chunk partials in a workspace, then a fixed-order reduce): no float atomics,
// so dcls / dpos are bit-reproducible from run to run (recompute == stored tests compare gradients bit for bit).
__global__ void assemble_tokens_bwd_kernel(const unsigned short* __restrict__ dtok, unsigned short* __restrict__ dpatch,
                                           float* __restrict__ part, long B, int L, int D, int bchunk) {
  const int dc = D / 8;
  const int c = (int)((blockIdx.x * (long)blockDim.x + threadIdx.x) % dc);
  const int l = (int)((blockIdx.x * (long)blockDim.x + threadIdx.x) / dc);
  if (l >= L) return;
  const long b0 = (long)blockIdx.y * bchunk;
  const long b1 = b0 + bchunk < B ? b0 + bchunk : B;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long b = b0; b < b1; ++b) {
    const u32x4 raw = *(const u32x4*)(dtok + ((size_t)b * L + l) * D + c * 8);
    float v[8];
    unpack8(raw, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += v[i];
    if (l > 0 && dpatch) *(u32x4*)(dpatch + ((size_t)b * (L - 1) + (l - 1)) * D + c * 8) = raw;
  }
  float* o = part + ((size_t)blockIdx.y * L + l) * D + c * 8;
  *(float4*)o = make_float4(acc[0], acc[1], acc[2], acc[3]);
  *(float4*)(o + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
}

// dpos[l][d] = sum over chunks (in chunk order) of part[chunk][l][d]; dcls = row 0 of the same sums
__global__ void assemble_tokens_bwd_reduce_kernel(const float* __restrict__ part, float* __restrict__ dcls,
                                                  float* __restrict__ dpos, int L, int D, int nchunk) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = dpos ? (long)L * D : (long)D;       // without dpos only the cls row is needed
  if (i >= n) return;
  float a = 0.f;
  for (int ch = 0; ch < nchunk; ++ch) a += part[(size_t)ch * L * D + i];
  if (dpos) dpos[i] = a;
  if (dcls && i < D) dcls[i] = a;
}

// ------------------------------------------------------------------------------------------------
// K14: x[b,t,:] = token_embedding[ids[b,t]].to(bf16) + positional_embedding[t].to(bf16)
// (open_clip/model.py:245-247)
template <bool TBF16>
__global__ void embed_tokens_kernel(const long* __restrict__ ids, const void* __restrict__ table,
                                    const float* __restrict__ pos, unsigned short* __restrict__ out,
                                    long B, int T, int D, int vocab, int* __restrict__ oob) {
  const int dc = D / 8;
  const long total = B * T * dc;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % dc);
    const long row = idx / dc;
    const int t = (int)(row % T);
    long id = ids[row];
    if (id < 0 || id >= vocab) {          // nn.Embedding raises here; report through the caller's counter
      if (c == 0 && oob) atomicAdd(oob, 1);
      id = 0;
    }
    float v[8], pe[8];
    if (TBF16) {
      unpack8(*(const u32x4*)((const unsigned short*)table + (size_t)id * D + c * 8), v);
    } else {
      *(float4*)v = *(const float4*)((const float*)table + (size_t)id * D + c * 8);
      *(float4*)(v + 4) = *(const float4*)((const float*)table + (size_t)id * D + c * 8 + 4);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = bf2f(f2bf(v[i]));
    }
    *(float4*)pe = *(const float4*)(pos + (size_t)t * D + c * 8);
    *(float4*)(pe + 4) = *(const float4*)(pos + (size_t)t * D + c * 8 + 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += bf2f(f2bf(pe[i]));
    *(u32x4*)(out + (size_t)row * D + c * 8) = pack8(v);
  }
}

// dtable[ids[b,t],:] += dx[b,t,:] (all-zero rows - every position after EOT under the
// causal mask - are skipped); dpos via assemble_tokens_bwd-style reduction is done by a second call.
// The scatter-add runs on 64-bit FIXED-POINT integers (value * 2^44): integer atomics commute, so the table gradient is
// bit-identical from run to run whatever order the rows arrive in (float atomics are not: the same step gave different
// last bits on different runs).  bf16 inputs convert exactly (anything below 2^-45 rounds to 0), the sum is exact, and
// one rounding to f32 happens in the convert kernel.  Range: |element| <= 2^17, |sum| < 2^19 - far above any gradient
// of a training run that has not diverged; rows that receive a non-finite element are flagged and come out as NaN (what
// the float sum would give); a finite element beyond 2^17 SATURATES at +-2^17 instead of poisoning its row.  The token-id
// range check comes first, so an out-of-range id is counted even when its gradient row is all zero.
constexpr double EMB_FIX = 17592186044416.0;         // 2^44
__global__ void embed_tokens_bwd_kernel(const long* __restrict__ ids, const unsigned short* __restrict__ dx,
                                        unsigned long long* __restrict__ acc, unsigned char* __restrict__ bad_row, long B, int T,
                                        int D, int vocab, int* __restrict__ oob) {
  const int dc = D / 8;
  const long total = B * T * dc;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % dc);
    const long row = idx / dc;
    const long id = ids[row];
    if (id < 0 || id >= vocab) {          // never scatter outside the table: count (once per row) and skip
      if (c == 0 && oob) atomicAdd(oob, 1);
      continue;
    }
    const u32x4 raw = *(const u32x4*)(dx + (size_t)row * D + c * 8);
    if (((raw[0] | raw[1] | raw[2] | raw[3]) & 0x7fff7fffu) == 0) continue;
    float v[8];
    unpack8(raw, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (v[i] == 0.f) continue;
      if (!(fabsf(v[i]) <= 3.0e38f)) { bad_row[id] = 1; continue; }        // NaN / Inf: the row comes out as NaN
      const float vs = fminf(fmaxf(v[i], -131072.0f), 131072.0f);          // finite but beyond the fixed-point range: saturate
      atomicAdd(acc + (size_t)id * D + c * 8 + i, (unsigned long long)__double2ll_rn((double)vs * EMB_FIX));
    }
  }
}
__global__ void embed_fix_to_f32_kernel(const long long* __restrict__ acc, const unsigned char* __restrict__ bad_row,
                                        float* __restrict__ out, long n, int D) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = bad_row[i / D] ? __int_as_float(0x7fc00000) : (float)((double)acc[i] * (1.0 / EMB_FIX));
}

// first-occurrence argmax over token ids (text.argmax(dim=-1), model.py:254)
__global__ void argmax_tokens_kernel(const long* __restrict__ ids, int* __restrict__ out, long B, int T) {
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  long best = ids[b * T];
  int bi = 0;
  for (int t = 1; t < T; ++t) {
    const long v = ids[b * T + t];
    if (v > best) { best = v; bi = t; }
  }
  out[b] = bi;
}

// ------------------------------------------------------------------------------------------------
// K12/K15 pooling (transformer.py:472-478, model.py:254-260): [B,L,D] bf16 -> [B,D] f32
__global__ void pool_fwd_kernel(const unsigned short* __restrict__ x, const int* __restrict__ idx,
                                float* __restrict__ out, long B, int L, int D, int mode) {
  const int dc = D / 8;
  const long total = B * dc;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % dc);
    const long b = i / dc;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int l0, l1;
    if (mode == CLIPA_POOL_FIRST) { l0 = 0; l1 = 1; }
    else if (mode == CLIPA_POOL_LAST) { l0 = L - 1; l1 = L; }
    else if (mode == CLIPA_POOL_INDEX) { l0 = idx[b]; l1 = l0 + 1; }
    else if (mode == CLIPA_POOL_MEAN_ALL) { l0 = 0; l1 = L; }
    else { l0 = 1; l1 = L; }
    for (int l = l0; l < l1; ++l) {
      float v[8];
      unpack8(*(const u32x4*)(x + ((size_t)b * L + l) * D + c * 8), v);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += v[k];
    }
    const float sc = 1.0f / (float)(l1 - l0);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] *= sc;
    *(float4*)(out + (size_t)b * D + c * 8) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *(float4*)(out + (size_t)b * D + c * 8 + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
}

__global__ void pool_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ idx,
                                unsigned short* __restrict__ dx, long B, int L, int D, int mode) {
  const int dc = D / 8;
  const long total = B * L * dc;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % dc);
    const long row = i / dc;
    const int l = (int)(row % L);
    const long b = row / L;
    int l0, l1;
    if (mode == CLIPA_POOL_FIRST) { l0 = 0; l1 = 1; }
    else if (mode == CLIPA_POOL_LAST) { l0 = L - 1; l1 = L; }
    else if (mode == CLIPA_POOL_INDEX) { l0 = idx[b]; l1 = l0 + 1; }
    else if (mode == CLIPA_POOL_MEAN_ALL) { l0 = 0; l1 = L; }
    else { l0 = 1; l1 = L; }
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (l >= l0 && l < l1) {
      const float sc = 1.0f / (float)(l1 - l0);
      const float4 a = *(const float4*)(dout + (size_t)b * D + c * 8);
      const float4 d = *(const float4*)(dout + (size_t)b * D + c * 8 + 4);
      v[0] = a.x * sc; v[1] = a.y * sc; v[2] = a.z * sc; v[3] = a.w * sc;
      v[4] = d.x * sc; v[5] = d.y * sc; v[6] = d.z * sc; v[7] = d.w * sc;
    }
    *(u32x4*)(dx + (size_t)row * D + c * 8) = pack8(v);
  }
}

// ------------------------------------------------------------------------------------------------
// K16: F.normalize(x, dim=-1) (model.py:240,263): y = x / max(||x||, eps); one wave per row.
__global__ void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned short* __restrict__ ybf,
                                  float* __restrict__ inv_norm, long rows, int E, float eps) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= rows) return;
  float ss = 0.f;
  for (int i = lane; i < E; i += 64) { const float v = x[(size_t)r * E + i]; ss += v * v; }
  ss = wave_sum(ss);
  const float inv = 1.0f / fmaxf(sqrtf(ss), eps);
  for (int i = lane; i < E; i += 64) {
    const float v = x[(size_t)r * E + i] * inv;
    y[(size_t)r * E + i] = v;
    if (ybf) ybf[(size_t)r * E + i] = f2bf(v);
  }
  if (lane == 0 && inv_norm) inv_norm[r] = inv;
}
// dx = inv * (dy - y * <y, dy>)   (exact when ||x|| > eps)
__global__ void l2norm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ inv_norm,
                                  const float* __restrict__ dy, float* __restrict__ dx, long rows, int E) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= rows) return;
  float dot = 0.f;
  for (int i = lane; i < E; i += 64) dot += y[(size_t)r * E + i] * dy[(size_t)r * E + i];
  dot = wave_sum(dot);
  const float inv = inv_norm[r];
  for (int i = lane; i < E; i += 64) dx[(size_t)r * E + i] = inv * (dy[(size_t)r * E + i] - y[(size_t)r * E + i] * dot);
}

// ------------------------------------------------------------------------------------------------
// bias gradient: out[n] = sum_m dY[m,n]  (bf16 in, f32 out). Stage 1: partial[blk][N]
__global__ __launch_bounds__(256) void colsum_partial_kernel(const unsigned short* __restrict__ dy, float* __restrict__ part,
                                                             long M, int N, long ld, int rows_per_blk) {
  __shared__ float red[4][512];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + lane) * 8;
  const long r0 = (long)blockIdx.y * rows_per_blk;
  const long r1 = r0 + rows_per_blk < M ? r0 + rows_per_blk : M;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (col < N) {
    for (long r = r0 + wv; r < r1; r += 4) {
      float v[8];
      unpack8(*(const u32x4*)(dy + (size_t)r * ld + col), v);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += v[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[wv][lane * 8 + i] = acc[i];
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 256) {
    const int c = blockIdx.x * 512 + i;
    if (c < N) part[(size_t)blockIdx.y * N + c] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
  }
}
// 64 columns per block, 4 waves striding the partial rows
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int nblk, int N) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float a0 = 0.f, a1 = 0.f;
  if (c < N) {
    int s = grp;
    for (; s + 4 < nblk; s += 8) {
      a0 += part[(size_t)s * N + c];
      a1 += part[(size_t)(s + 4) * N + c];
    }
    for (; s < nblk; s += 4) a0 += part[(size_t)s * N + c];
  }
  red[grp][lane] = a0 + a1;
  __syncthreads();
  if (grp == 0 && c < N) out[c] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}

// ------------------------------------------------------------------------------------------------
// casts and weight transposes
template <bool IN_F32>
__global__ void cast_to_bf16_kernel(const void* __restrict__ in, unsigned short* __restrict__ out, long n) {
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += (long)gridDim.x * blockDim.x * 8) {
    float v[8];
    if (i + 8 <= n) {
      if (IN_F32) { *(float4*)v = *(const float4*)((const float*)in + i); *(float4*)(v + 4) = *(const float4*)((const float*)in + i + 4); }
      else unpack8(*(const u32x4*)((const unsigned short*)in + i), v);
      *(u32x4*)(out + i) = pack8(v);
    } else {
      for (long j = i; j < n; ++j) out[j] = IN_F32 ? f2bf(((const float*)in)[j]) : ((const unsigned short*)in)[j];
    }
  }
}
__global__ void cast_bf16_to_f32_kernel(const unsigned short* __restrict__ in, float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = bf2f(in[i]);
}
// out[c][r] = bf16(in[r][c]); 64x64 tiles through LDS
template <bool IN_F32>
__global__ __launch_bounds__(256) void transpose_to_bf16_kernel(const void* __restrict__ in, unsigned short* __restrict__ out,
                                                                int R, int C, long ldi, long ldo) {
  __shared__ unsigned short tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    unsigned short v = 0;
    if (r0 + r < R && c0 + c < C)
      v = IN_F32 ? f2bf(((const float*)in)[(size_t)(r0 + r) * ldi + c0 + c]) : ((const unsigned short*)in)[(size_t)(r0 + r) * ldi + c0 + c];
    tile[r][c] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (r0 + r < R && c0 + c < C) out[(size_t)(c0 + c) * ldo + r0 + r] = tile[r][c];
  }
}

// out[0] = scale * sum_i in[i]  (single workgroup; n is small)
__global__ __launch_bounds__(256) void sum_scale_kernel(const float* __restrict__ in, float* __restrict__ out, long n, float scale, int accumulate) {
  __shared__ float red[4];
  float a = 0.f;
  for (long i = threadIdx.x; i < n; i += 256) a += in[i];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = (red[0] + red[1] + red[2] + red[3]) * scale;
    out[0] = accumulate ? out[0] + v : v;
  }
}

// one-shot by default: a thread per work item (a persistent grid striding a large array streams a third slower than one block
// per 4 KB - tools/probes/stream_ab.hip, copy 4.6 vs 6.2 TB/s); the kernels keep their grid-stride loops for the > 2^31 case
inline unsigned grid_for(long work, int block = 256, long cap = 0x7fffffffL) {
  long g = (work + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

extern "C" int clipa_patchify(const void* img, void* out, int64_t B, int64_t S, int64_t P, int64_t Kp,
                              int in_dtype, int nhwc, int normalize, const float* mean3,
                              const float* std3, void* stream) {
  const int64_t g = S / P;
  if (g <= 0 || Kp % 8 != 0 || Kp < 3 * P * P) { clipa_set_error("patchify: bad geometry S=%ld P=%ld Kp=%ld", (long)S, (long)P, (long)Kp); return CLIPA_ERR_ARG; }
  if (B <= 0) return CLIPA_OK;
  float m[3] = {0, 0, 0}, is[3] = {1, 1, 1};
  if (normalize) {
    if (!mean3 || !std3) { clipa_set_error("patchify: normalize needs mean/std"); return CLIPA_ERR_ARG; }
    for (int i = 0; i < 3; ++i) { m[i] = mean3[i]; is[i] = 1.0f / std3[i]; }
  }
  const long total = B * g * g * (Kp / 8);
  const unsigned grid = grid_for(total);
  hipStream_t st = (hipStream_t)stream;
  if (in_dtype == CLIPA_DT_U8) hipLaunchKernelGGL(patchify_kernel<0>, dim3(grid), dim3(256), 0, st, img, (unsigned short*)out, (int)B, (int)S, (int)P, (int)g, (int)Kp, nhwc, normalize, m[0], m[1], m[2], is[0], is[1], is[2]);
  else if (in_dtype == CLIPA_DT_BF16) hipLaunchKernelGGL(patchify_kernel<1>, dim3(grid), dim3(256), 0, st, img, (unsigned short*)out, (int)B, (int)S, (int)P, (int)g, (int)Kp, nhwc, normalize, m[0], m[1], m[2], is[0], is[1], is[2]);
  else if (in_dtype == CLIPA_DT_F32) hipLaunchKernelGGL(patchify_kernel<2>, dim3(grid), dim3(256), 0, st, img, (unsigned short*)out, (int)B, (int)S, (int)P, (int)g, (int)Kp, nhwc, normalize, m[0], m[1], m[2], is[0], is[1], is[2]);
  else { clipa_set_error("patchify: unsupported input dtype %d", in_dtype); return CLIPA_ERR_ARG; }
  return clipa_check_launch("patchify");
}

extern "C" int clipa_assemble_tokens(const void* patch, const float* cls, const float* pos, void* tokens,
                                     int64_t B, int64_t L, int64_t D, void* stream) {
  if (D % 8 != 0) { clipa_set_error("assemble_tokens: D%%8 != 0"); return CLIPA_ERR_ARG; }
  if (B <= 0) return CLIPA_OK;
  hipLaunchKernelGGL(assemble_tokens_kernel, dim3(grid_for(B * L * (D / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)patch, cls, pos, (unsigned short*)tokens, (long)B, (int)L, (int)D);
  return clipa_check_launch("assemble_tokens");
}

namespace {
void assemble_bwd_chunks(long B, long L, long D, int* nchunk, int* bchunk) {
  const long threads = L * (D / 8);
  long nc = (4096L * 256 + threads - 1) / threads;
  if (nc > B) nc = B;
  if (nc < 1) nc = 1;
  const long bc = (B + nc - 1) / nc;
  *bchunk = (int)bc;
  *nchunk = (int)((B + bc - 1) / bc);
}
}  // namespace

extern "C" int64_t clipa_assemble_tokens_bwd_workspace(int64_t B, int64_t L, int64_t D) {
  if (B <= 0) return 0;
  int nchunk, bchunk;
  assemble_bwd_chunks(B, L, D, &nchunk, &bchunk);
  return (int64_t)nchunk * L * D * (int64_t)sizeof(float);
}

extern "C" int clipa_assemble_tokens_bwd(const void* dtokens, void* dpatch, float* dcls, float* dpos,
                                         int64_t B, int64_t L, int64_t D, void* workspace, int64_t workspace_bytes,
                                         void* stream) {
  if (D % 8 != 0) { clipa_set_error("assemble_tokens_bwd: D%%8 != 0"); return CLIPA_ERR_ARG; }
  if (B <= 0) return CLIPA_OK;
  if (!workspace || workspace_bytes < clipa_assemble_tokens_bwd_workspace(B, L, D)) { clipa_set_error("assemble_tokens_bwd: workspace too small"); return CLIPA_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  int nchunk, bchunk;
  assemble_bwd_chunks(B, L, D, &nchunk, &bchunk);
  const long threads = L * (D / 8);
  hipLaunchKernelGGL(assemble_tokens_bwd_kernel, dim3((unsigned)((threads + 255) / 256), (unsigned)nchunk), dim3(256), 0, st,
                     (const unsigned short*)dtokens, (unsigned short*)dpatch, (float*)workspace, (long)B, (int)L, (int)D, bchunk);
  if (int rc = clipa_check_launch("assemble_tokens_bwd")) return rc;
  if (!dcls && !dpos) return CLIPA_OK;
  const long n = dpos ? L * D : D;
  hipLaunchKernelGGL(assemble_tokens_bwd_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)workspace,
                     dcls, dpos, (int)L, (int)D, nchunk);
  return clipa_check_launch("assemble_tokens_bwd_reduce");
}

extern "C" int clipa_embed_tokens(const int64_t* ids, const void* table, int table_bf16, const float* pos,
                                  void* out, int64_t B, int64_t T, int64_t D, int64_t vocab, int32_t* oob_count,
                                  void* stream) {
  if (D % 8 != 0) { clipa_set_error("embed_tokens: D%%8 != 0"); return CLIPA_ERR_ARG; }
  if (B <= 0) return CLIPA_OK;
  const unsigned grid = grid_for(B * T * (D / 8));
  if (table_bf16) hipLaunchKernelGGL(embed_tokens_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const long*)ids, table, pos, (unsigned short*)out, (long)B, (int)T, (int)D, (int)vocab, oob_count);
  else hipLaunchKernelGGL(embed_tokens_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const long*)ids, table, pos, (unsigned short*)out, (long)B, (int)T, (int)D, (int)vocab, oob_count);
  return clipa_check_launch("embed_tokens");
}

extern "C" int64_t clipa_embed_tokens_bwd_workspace(int64_t B, int64_t T, int64_t D, int64_t vocab, int need_table,
                                                   int need_pos) {
  int64_t w = need_table ? vocab * D * (int64_t)sizeof(long long) + (vocab + 7) / 8 * 8 : 0;   // fixed-point accumulators + row flags
  const int64_t pw = need_pos ? clipa_assemble_tokens_bwd_workspace(B, T, D) : 0;
  return w > pw ? w : pw;                                                   // the two phases run one after the other
}

extern "C" int clipa_embed_tokens_bwd(const int64_t* ids, const void* dx, float* dtable, float* dpos,
                                      int64_t B, int64_t T, int64_t D, int64_t vocab, int32_t* oob_count, void* workspace,
                                      int64_t workspace_bytes, void* stream) {
  if (D % 8 != 0) { clipa_set_error("embed_tokens_bwd: D%%8 != 0"); return CLIPA_ERR_ARG; }
  if (B <= 0) return CLIPA_OK;
  if (!workspace || workspace_bytes < clipa_embed_tokens_bwd_workspace(B, T, D, vocab, dtable != nullptr, dpos != nullptr)) {
    clipa_set_error("embed_tokens_bwd: workspace too small");
    return CLIPA_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  if (dtable) {
    unsigned char* bad_row = (unsigned char*)workspace + vocab * D * sizeof(long long);
    (void)hipMemsetAsync(workspace, 0, vocab * D * sizeof(long long) + (vocab + 7) / 8 * 8, st);
    hipLaunchKernelGGL(embed_tokens_bwd_kernel, dim3(grid_for(B * T * (D / 8))), dim3(256), 0, st, (const long*)ids, (const unsigned short*)dx, (unsigned long long*)workspace, bad_row, (long)B, (int)T, (int)D, (int)vocab, oob_count);
    if (int rc = clipa_check_launch("embed_tokens_bwd")) return rc;
    hipLaunchKernelGGL(embed_fix_to_f32_kernel, dim3(grid_for(vocab * D)), dim3(256), 0, st, (const long long*)workspace, bad_row, dtable, (long)(vocab * D), (int)D);
    if (int rc = clipa_check_launch("embed_tokens_bwd_convert")) return rc;
  }
  if (dpos) return clipa_assemble_tokens_bwd(dx, nullptr, nullptr, dpos, B, T, D, workspace, workspace_bytes, stream);
  return CLIPA_OK;
}

extern "C" int clipa_argmax_tokens(const int64_t* ids, int32_t* out, int64_t B, int64_t T, void* stream) {
  if (B <= 0) return CLIPA_OK;
  hipLaunchKernelGGL(argmax_tokens_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const long*)ids, out, (long)B, (int)T);
  return clipa_check_launch("argmax_tokens");
}

extern "C" int clipa_pool_fwd(const void* x, const int32_t* idx, float* out, int64_t B, int64_t L, int64_t D,
                              int mode, void* stream) {
  if (D % 8 != 0 || (mode == CLIPA_POOL_INDEX && !idx)) { clipa_set_error("pool_fwd: bad args"); return CLIPA_ERR_ARG; }
  if (B <= 0) return CLIPA_OK;
  hipLaunchKernelGGL(pool_fwd_kernel, dim3(grid_for(B * (D / 8))), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, idx, out, (long)B, (int)L, (int)D, mode);
  return clipa_check_launch("pool_fwd");
}
extern "C" int clipa_pool_bwd(const float* dout, const int32_t* idx, void* dx, int64_t B, int64_t L, int64_t D,
                              int mode, void* stream) {
  if (D % 8 != 0 || (mode == CLIPA_POOL_INDEX && !idx)) { clipa_set_error("pool_bwd: bad args"); return CLIPA_ERR_ARG; }
  if (B <= 0) return CLIPA_OK;
  hipLaunchKernelGGL(pool_bwd_kernel, dim3(grid_for(B * L * (D / 8))), dim3(256), 0, (hipStream_t)stream, dout, idx, (unsigned short*)dx, (long)B, (int)L, (int)D, mode);
  return clipa_check_launch("pool_bwd");
}

// PatchDropout (open_clip/transformer.py:53-83, applied at :501-502): the kept tokens are a ROW SELECTION of the token matrix.
// gather: out[r, :] = x[rows[r], :]; scatter (its backward): dx = 0, dx[rows[r], :] = dy[r, :] (the rows are distinct).
namespace {
__global__ void gather_rows_kernel(const unsigned short* __restrict__ x, const int64_t* __restrict__ rows,
                                   unsigned short* __restrict__ out, long n_out, long n_src, int D) {
  const int dc = D / 8;
  const long total = n_out * dc;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long r = idx / dc;
    const int c = (int)(idx - r * dc);
    const long src = rows[r];
    u32x4 v = {0u, 0u, 0u, 0u};
    if (src >= 0 && src < n_src) v = *(const u32x4*)(x + (size_t)src * D + c * 8);
    *(u32x4*)(out + (size_t)r * D + c * 8) = v;
  }
}
__global__ void scatter_rows_kernel(const unsigned short* __restrict__ dy, const int64_t* __restrict__ rows,
                                    unsigned short* __restrict__ dx, long n_src, long n_dst, int D) {
  const int dc = D / 8;
  const long total = n_src * dc;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long r = idx / dc;
    const int c = (int)(idx - r * dc);
    const long dst = rows[r];
    if (dst >= 0 && dst < n_dst) *(u32x4*)(dx + (size_t)dst * D + c * 8) = *(const u32x4*)(dy + (size_t)r * D + c * 8);
  }
}
}  // namespace

extern "C" int clipa_gather_rows(const void* x, const int64_t* rows, void* out, int64_t n_out, int64_t n_src, int64_t D,
                                 void* stream) {
  if (D % 8 != 0 || !rows) { clipa_set_error("gather_rows: D%%8 != 0 or no row list"); return CLIPA_ERR_ARG; }
  if (n_out <= 0) return CLIPA_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(n_out * (D / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)x, rows, (unsigned short*)out, (long)n_out, (long)n_src, (int)D);
  return clipa_check_launch("gather_rows");
}
extern "C" int clipa_scatter_rows(const void* dy, const int64_t* rows, void* dx, int64_t n_src, int64_t n_dst, int64_t D,
                                  void* stream) {
  if (D % 8 != 0 || !rows) { clipa_set_error("scatter_rows: D%%8 != 0 or no row list"); return CLIPA_ERR_ARG; }
  if (n_dst <= 0) return CLIPA_OK;
  const hipError_t e = hipMemsetAsync(dx, 0, (size_t)n_dst * D * 2, (hipStream_t)stream);
  if (e != hipSuccess) { clipa_set_error("scatter_rows: memset: %s", hipGetErrorString(e)); return CLIPA_ERR_LAUNCH; }
  if (n_src <= 0) return CLIPA_OK;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for(n_src * (D / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)dy, rows, (unsigned short*)dx, (long)n_src, (long)n_dst, (int)D);
  return clipa_check_launch("scatter_rows");
}

extern "C" int clipa_l2norm_fwd(const float* x, float* y, void* y_bf16, float* inv_norm, int64_t rows, int64_t E,
                                float eps, void* stream) {
  if (rows <= 0) return CLIPA_OK;
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, y, (unsigned short*)y_bf16, inv_norm, (long)rows, (int)E, eps);
  return clipa_check_launch("l2norm_fwd");
}
extern "C" int clipa_l2norm_bwd(const float* y, const float* inv_norm, const float* dy, float* dx, int64_t rows,
                                int64_t E, void* stream) {
  if (rows <= 0) return CLIPA_OK;
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, y, inv_norm, dy, dx, (long)rows, (int)E);
  return clipa_check_launch("l2norm_bwd");
}

static int colsum_blocks(int64_t M) {
  long nb = (M + 511) / 512;
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  return (int)nb;
}
extern "C" int64_t clipa_colsum_workspace(int64_t M, int64_t N) { return (int64_t)colsum_blocks(M) * N * sizeof(float); }
extern "C" int clipa_colsum(const void* dy, float* out, int64_t M, int64_t N, int64_t ld, void* workspace,
                            int64_t workspace_bytes, void* stream) {
  if (N % 8 != 0 || ld % 8 != 0) { clipa_set_error("colsum: N, ld must be multiples of 8"); return CLIPA_ERR_ARG; }
  if (!workspace || workspace_bytes < clipa_colsum_workspace(M, N)) { clipa_set_error("colsum: workspace too small"); return CLIPA_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  if (M <= 0) { (void)hipMemsetAsync(out, 0, N * sizeof(float), st); return CLIPA_OK; }
  const int nb = colsum_blocks(M);
  const int rpb = (int)((M + nb - 1) / nb);
  const int nb2 = (int)((M + rpb - 1) / rpb);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)((N + 511) / 512), (unsigned)nb2), dim3(256), 0, st, (const unsigned short*)dy, (float*)workspace, (long)M, (int)N, (long)ld, rpb);
  if (int rc = clipa_check_launch("colsum_partial")) return rc;
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, st, (const float*)workspace, out, nb2, (int)N);
  return clipa_check_launch("colsum_final");
}

extern "C" int clipa_cast_to_bf16(const void* in, int in_f32, void* out, int64_t n, void* stream) {
  if (n <= 0) return CLIPA_OK;
  const unsigned grid = grid_for((n + 7) / 8);
  if (in_f32) hipLaunchKernelGGL(cast_to_bf16_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, (unsigned short*)out, (long)n);
  else hipLaunchKernelGGL(cast_to_bf16_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, (unsigned short*)out, (long)n);
  return clipa_check_launch("cast_to_bf16");
}
extern "C" int clipa_cast_bf16_to_f32(const void* in, float* out, int64_t n, void* stream) {
  if (n <= 0) return CLIPA_OK;
  hipLaunchKernelGGL(cast_bf16_to_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)in, out, (long)n);
  return clipa_check_launch("cast_bf16_to_f32");
}
extern "C" int clipa_transpose_to_bf16(const void* in, int in_f32, void* out, int64_t R, int64_t C, int64_t ldi,
                                       int64_t ldo, void* stream) {
  if (R <= 0 || C <= 0) return CLIPA_OK;
  const dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R + 63) / 64));
  if (in_f32) hipLaunchKernelGGL(transpose_to_bf16_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, in, (unsigned short*)out, (int)R, (int)C, (long)ldi, (long)ldo);
  else hipLaunchKernelGGL(transpose_to_bf16_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, in, (unsigned short*)out, (int)R, (int)C, (long)ldi, (long)ldo);
  return clipa_check_launch("transpose_to_bf16");
}

// out = act(in), bf16 -> bf16 (re-materialises the MLP activation from the stored pre-activation in the
// backward of a block whose GEMM outputs were kept; transformer.py:218 / model.py:128-129)
namespace {
template <int ACT>
__global__ void activation_kernel(const unsigned short* __restrict__ in, unsigned short* __restrict__ out, long n8) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float f[8];
    unpack8(ld_stream<u32x4>(in + i * 8), f);
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      const f32x2 r = act_fwd2<ACT>(f32x2{f[k], f[k + 1]});
      f[k] = r.x;
      f[k + 1] = r.y;
    }
    st_stream<u32x4>(out + i * 8, pack8(f));
  }
}
}  // namespace
extern "C" int clipa_activation_fwd(const void* in, void* out, int64_t n, int act, void* stream) {
  if (n % 8 != 0) { clipa_set_error("activation_fwd: n must be a multiple of 8"); return CLIPA_ERR_ARG; }
  if (n <= 0) return CLIPA_OK;
  const unsigned grid = grid_for(n / 8);
  hipStream_t st = (hipStream_t)stream;
  if (act == ACT_GELU_ERF) hipLaunchKernelGGL(activation_kernel<ACT_GELU_ERF>, dim3(grid), dim3(256), 0, st, (const unsigned short*)in, (unsigned short*)out, (long)(n / 8));
  else if (act == ACT_GELU_TANH) hipLaunchKernelGGL(activation_kernel<ACT_GELU_TANH>, dim3(grid), dim3(256), 0, st, (const unsigned short*)in, (unsigned short*)out, (long)(n / 8));
  else if (act == ACT_QUICK_GELU) hipLaunchKernelGGL(activation_kernel<ACT_QUICK_GELU>, dim3(grid), dim3(256), 0, st, (const unsigned short*)in, (unsigned short*)out, (long)(n / 8));
  else { clipa_set_error("activation_fwd: unknown activation %d", act); return CLIPA_ERR_ARG; }
  return clipa_check_launch("activation_fwd");
}

extern "C" int clipa_sum_scale(const float* in, float* out, int64_t n, float scale, int accumulate, void* stream) {
  hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, in, out, (long)n, scale, accumulate);
  return clipa_check_launch("sum_scale");
}
