// fp8 MFMA GEMM for gfx950 (MI355X):  C[M,N] = epi(alpha * sa[m] * sb[n] * A8[M,K] . B8[N,K]^T + bias[N])
//
// BASELINE.json configs[3] ("fp8 MFMA weights/activations", ViT-H/14): the reference has no fp8 mode (its precision
// menu is clipa_torch/training/params.py:195-200), so this is the engine's own recipe for the same call sites as
// gemm_nt.hip - the four linear layers of a residual block, forward and input gradient (clipa_torch/open_clip/
// transformer.py:209,217-219,234): operands are OCP e4m3 / e5m2 bytes with one f32 de-quantisation scale per ROW
// (per token for activations / gradients, per output channel for weights: clipa_quantize_rows), products on
// v_mfma_f32_16x16x128_f8f6f4 (twice the bf16 MFMA rate), fp32 accumulation, the bf16 kernel's fused epilogues.
//
// Geometry = gemm_nt2_kernel<bf16, M16> with the K tile counted in BYTES: an operand image is still 256 rows x 128 B
// (now 128 fp8 k-values), so the LDS-DMA pattern, the (row >> 1) & 7 chunk swizzle, the 2-slot ring and the
// LDS-window epilogue are unchanged, and one MFMA consumes a whole image row: 32 MFMAs per K tile per wave instead
// of 64 at equal bytes moved - the same timeline as the bf16 kernel with twice the FLOPs in it.
// A lane's 32 operand bytes are chunks g and 4 + g of its row (g = lane >> 4), i.e. the two conflict-free 16-B reads
// of the bf16 kernel; which k-values a lane holds is immaterial as long as both operands use the same rule.
#include "gemm_common.h"

namespace clipa_gemm {
namespace {

constexpr int F8_CBUF_OFF = 2 * STAGE_BYTES;            // 131072: epilogue window behind the ring
constexpr int F8_LDS_BYTES = F8_CBUF_OFF + 64 * 512;    // 163840

typedef int i32x8 __attribute__((ext_vector_type(8)));

struct F8Args {
  const char* A; const char* B; char* C; char* C2; const float* bias; const char* aux;
  const float* sa; const float* sb;     // de-quantisation scales: per row of A [M], per row of B [N]; NULL = 1
  int M, N, K;
  long lda, ldb, ldc, ldaux;            // element strides (A, B: bytes)
  float alpha;
  int epi, act, abl;
};

__device__ __forceinline__ i32x8 frag8(const char* rowp, int g4, int sw16) {
  const u32x4 lo = *(const u32x4*)(rowp + ((g4 ^ sw16) << 4));
  const u32x4 hi = *(const u32x4*)(rowp + (((4 + g4) ^ sw16) << 4));
  i32x8 f;
  f[0] = (int)lo[0]; f[1] = (int)lo[1]; f[2] = (int)lo[2]; f[3] = (int)lo[3];
  f[4] = (int)hi[0]; f[5] = (int)hi[1]; f[6] = (int)hi[2]; f[7] = (int)hi[3];
  return f;
}

// FMT_A / FMT_B: 0 = e4m3, 1 = e5m2 of the A (activation / gradient) and B (weight) operand
template <int FMT_A, int FMT_B>
__global__ __launch_bounds__(NTHREADS) void gemm_nt_f8_kernel(F8Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l15 = lane & 15, g4 = lane >> 4, sw16 = (l15 >> 1) & 7;

  const int tilesN = (p.N + BN - 1) / BN;
  const int tilesM = (p.M + BM - 1) / BM;
  const unsigned ntiles = (unsigned)(tilesM * tilesN);
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
    voffA[j] = (unsigned)(row * p.lda + chunk * 16);
    voffB[j] = (unsigned)(row * p.ldb + chunk * 16);
    kel[j] = chunk * 16;
  }
  const int nkt = (p.K + 127) / 128;

  auto tile_origin = [&](unsigned t, int& m0, int& n0) {
    const int GM = 4;
    const int per = GM * tilesN;
    const int g = (int)t / per, r = (int)t - g * per;
    const int gm = min(GM, tilesM - g * GM);
    const int tn = r / gm, mm = r - tn * gm;
    m0 = (g * GM + mm) * BM;
    n0 = tn * BN;
  };
  auto stage = [&](int buf, int m0, int n0, int k0) {
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A + (size_t)m0 * p.lda, (unsigned)(min(BM, p.M - m0) * p.lda));
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(p.B + (size_t)n0 * p.ldb, (unsigned)(min(BN, p.N - n0) * p.ldb));
    char* sA = smem + buf * STAGE_BYTES;
    char* sB = sA + IMG_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pc = j * 8 + wave;
      const unsigned oob = (k0 + kel[j] >= p.K) ? 0x80000000u : 0u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(sA + pc * 1024), 16, voffA[j] | oob, k0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(sB + pc * 1024), 16, voffB[j] | oob, k0, 0, 0);
    }
  };

  if (idx >= len) return;
  unsigned it = idx;
  int m0, n0;
  tile_origin(base + it, m0, n0);
  stage(0, m0, n0, 0);
  unsigned gk = 0;
  for (;;) {
    const bool has_next = it + gx < len;
    int m1 = 0, n1 = 0;
    if (has_next) tile_origin(base + it + gx, m1, n1);

    f32x4v acc16[4][8];      // [n block of 16][m block of 16]
#pragma unroll
    for (int bj = 0; bj < 4; ++bj)
#pragma unroll
      for (int ai = 0; ai < 8; ++ai) acc16[bj][ai] = f32x4v{0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < nkt; ++kt, ++gk) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (kt + 1 < nkt) stage((gk + 1) & 1, m0, n0, (kt + 1) * 128);
      else {
        if (has_next) stage((gk + 1) & 1, m1, n1, 0);
        // the tile's 256 row scales ride the last K step as ONE 1 KiB LDS-DMA into the (idle) epilogue window: a per-lane
        // gather of 8 scattered floats cost 64 extra wave-instructions of the CU's memory pipe per tile
        if (p.sa && wave == 0) {
          const __amdgpu_buffer_rsrc_t rsS = make_rsrc(p.sa + m0, (unsigned)(max(0, min(BM, p.M - m0)) * 4));
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, LDS_PTR(smem + F8_CBUF_OFF), 16, (unsigned)(lane * 16), 0, 0, 0);
        }
      }
      const char* sA = smem + (gk & 1) * STAGE_BYTES;
      const char* sB = sA + IMG_BYTES;
      const char* pa = sA + (wm * 128 + l15) * 128;
      const char* pb = sB + (wn * 64 + l15) * 128;
      // 8 sub-steps per K tile: A block u against the four B blocks (4 MFMAs of 16x16x128); A fragments double-buffered
      i32x8 gb[4], ga[2];
#pragma unroll
      for (int bj = 0; bj < 4; ++bj) gb[bj] = frag8(pb + bj * 2048, g4, sw16);
      ga[0] = frag8(pa, g4, sw16);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (u < 7) ga[(u + 1) & 1] = frag8(pa + (u + 1) * 2048, g4, sw16);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int bj = 0; bj < 4; ++bj)
          acc16[bj][u] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(gb[bj], ga[u & 1], acc16[bj][u], FMT_B, FMT_A, 0, 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
    }

    // ---- epilogue of tile (m0, n0); the ring keeps filling for the next tile meanwhile ----
    if (p.abl & 2) {   // ablation: keep the accumulators live, write nothing
      float t = 0.f;
#pragma unroll
      for (int bj = 0; bj < 4; ++bj)
#pragma unroll
        for (int ai = 0; ai < 8; ++ai) t += acc16[bj][ai][0] + acc16[bj][ai][1] + acc16[bj][ai][2] + acc16[bj][ai][3];
      if (t == 1.2345e-30f) ((float*)p.C)[tid] = t;
    } else {
      char* cb = smem + F8_CBUF_OFF;
      const int epi = p.epi, act = p.act;
      float4 bias4[4], sb4[4];    // the 4 consecutive features of n block bj this lane holds
#pragma unroll
      for (int bj = 0; bj < 4; ++bj) {
        const int n = n0 + wn * 64 + bj * 16 + 4 * g4;
        bias4[bj] = (p.bias && n < p.N) ? *(const float4*)(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        sb4[bj] = (p.sb && n < p.N) ? *(const float4*)(p.sb + n) : make_float4(1.f, 1.f, 1.f, 1.f);
      }
      float sam[8];               // alpha * row scale of the 8 m blocks' row this lane holds
      if (p.sa) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();          // the scale vector has landed in the window (pass 0 overwrites it after its own barrier)
        const float* sc = (const float*)(smem + F8_CBUF_OFF);
#pragma unroll
        for (int ai = 0; ai < 8; ++ai) sam[ai] = p.alpha * sc[wm * 128 + ai * 16 + l15];
      } else {
#pragma unroll
        for (int ai = 0; ai < 8; ++ai) sam[ai] = p.alpha;
      }
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
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        if ((pass & 1) == 0) fetch_aux(pass);
        LDS_BARRIER();   // readers of the previous pass are done with the window
        if (wm == (pass >> 1)) {
#pragma unroll
          for (int a2 = 0; a2 < 4; ++a2) {
            const int ai = 4 * (pass & 1) + a2;
            const int row = a2 * 16 + l15;
            const float s = sam[ai];
#pragma unroll
            for (int bj = 0; bj < 4; ++bj) {
              const int nl = wn * 64 + bj * 16 + 4 * g4;
              const float4 b4 = bias4[bj], q4 = sb4[bj];
              u32x2 w;
              w[0] = pack2bf(acc16[bj][ai][0] * (s * q4.x) + b4.x, acc16[bj][ai][1] * (s * q4.y) + b4.y);
              w[1] = pack2bf(acc16[bj][ai][2] * (s * q4.z) + b4.z, acc16[bj][ai][3] * (s * q4.w) + b4.w);
              *(u32x2*)(cb + row * 512 + ((((nl >> 3) ^ row) & 31) << 4) + (nl & 7) * 2) = w;
            }
          }
        }
        LDS_BARRIER();
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
          if (m < p.M && n < p.N) {
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

std::once_flag g_f8_once[MAX_DEVICES];
int g_f8_rc[MAX_DEVICES];

int ensure_f8_attrs(int dev) {
  std::call_once(g_f8_once[dev], [dev]() {
    int rc = 0;
    const void* ks[4] = {(const void*)gemm_nt_f8_kernel<0, 0>, (const void*)gemm_nt_f8_kernel<1, 0>,
                         (const void*)gemm_nt_f8_kernel<0, 1>, (const void*)gemm_nt_f8_kernel<1, 1>};
    for (int i = 0; i < 4; ++i) {
      const hipError_t e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, F8_LDS_BYTES);
      if (e != hipSuccess) { clipa_set_error("hipFuncSetAttribute(gemm_nt_f8): %s", hipGetErrorString(e)); rc = CLIPA_ERR_LAUNCH; }
    }
    g_f8_rc[dev] = rc;
  });
  return g_f8_rc[dev];
}

}  // namespace
}  // namespace clipa_gemm

using namespace clipa_gemm;

extern "C" int clipa_gemm_nt_f8(const void* A8, const void* B8, const float* scale_a, const float* scale_b, void* C,
                                void* C2, const float* bias, const void* aux, int64_t M, int64_t N, int64_t K,
                                int64_t lda, int64_t ldb, int64_t ldc, int64_t ldaux, float alpha, int epi, int act,
                                int fmt_a, int fmt_b, void* stream) {
  if (M <= 0 || N <= 0) return CLIPA_OK;
  if (K <= 0 || K % 16 != 0) { clipa_set_error("gemm_nt_f8: K=%ld must be a positive multiple of 16", (long)K); return CLIPA_ERR_ARG; }
  if (lda % 16 != 0 || ldb % 16 != 0) { clipa_set_error("gemm_nt_f8: lda, ldb must be multiples of 16 bytes"); return CLIPA_ERR_ARG; }
  if (N % 8 != 0 || ldc % 8 != 0) { clipa_set_error("gemm_nt_f8: N, ldc must be multiples of 8"); return CLIPA_ERR_ARG; }
  if (epi < CLIPA_EPI_NONE || epi > CLIPA_EPI_DACT) { clipa_set_error("gemm_nt_f8: unknown epilogue %d", epi); return CLIPA_ERR_ARG; }
  if ((epi == CLIPA_EPI_ADD || epi == CLIPA_EPI_DACT) && (!aux || ldaux % 8 != 0)) { clipa_set_error("gemm_nt_f8: epilogue %d needs aux with ldaux%%8==0", epi); return CLIPA_ERR_ARG; }
  if (C2 && epi != CLIPA_EPI_ACT) { clipa_set_error("gemm_nt_f8: C2 (pre-activation copy) goes with CLIPA_EPI_ACT only"); return CLIPA_ERR_ARG; }
  if ((fmt_a != 0 && fmt_a != 1) || (fmt_b != 0 && fmt_b != 1)) { clipa_set_error("gemm_nt_f8: formats are 0 (e4m3) or 1 (e5m2)"); return CLIPA_ERR_ARG; }
  if (256 * lda >= (1L << 30) || 256 * ldb >= (1L << 30) || 256 * ldc * 2 >= (1L << 30) || 256 * ldaux * 2 >= (1L << 30)) { clipa_set_error("gemm_nt_f8: leading dimension too large"); return CLIPA_ERR_ARG; }
  int dev = 0;
  if (int rc = current_device(&dev)) return rc;
  if (int rc = ensure_f8_attrs(dev)) return rc;
  const int num_cu = gemm_num_cu(dev);
  F8Args a;
  a.A = (const char*)A8; a.B = (const char*)B8; a.C = (char*)C; a.C2 = (char*)C2; a.bias = bias; a.aux = (const char*)aux;
  a.sa = scale_a; a.sb = scale_b;
  a.M = (int)M; a.N = (int)N; a.K = (int)K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldaux = ldaux;
  a.alpha = alpha; a.epi = epi; a.act = act; a.abl = g_abl.load(std::memory_order_relaxed);
  hipStream_t st = (hipStream_t)stream;
  const long tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  const dim3 grid((unsigned)(tiles < num_cu ? tiles : num_cu)), block(NTHREADS);
  if (fmt_a == 0 && fmt_b == 0) hipLaunchKernelGGL((gemm_nt_f8_kernel<0, 0>), grid, block, F8_LDS_BYTES, st, a);
  else if (fmt_a == 1 && fmt_b == 0) hipLaunchKernelGGL((gemm_nt_f8_kernel<1, 0>), grid, block, F8_LDS_BYTES, st, a);
  else if (fmt_a == 0 && fmt_b == 1) hipLaunchKernelGGL((gemm_nt_f8_kernel<0, 1>), grid, block, F8_LDS_BYTES, st, a);
  else hipLaunchKernelGGL((gemm_nt_f8_kernel<1, 1>), grid, block, F8_LDS_BYTES, st, a);
  return clipa_check_launch("gemm_nt_f8");
}
