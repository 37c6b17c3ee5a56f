// A/B harness (test infrastructure, not part of libclipa_hip.so): gemm_f8a (four waves, hand-scheduled main loop) against
// gemm_nt_f8_kernel (the round-2 fp8 kernel) through clipa_gemm_nt_f8; operands are quantised by clipa_quantize_rows.
//   * outputs compared BIT FOR BIT for every epilogue, e4m3 and e5m2 A operands;
//   * timing: interleaved rounds in one process, median TF/s per variant (+ main loop alone).
// Build:  hipcc --offload-arch=gfx950 -O2 -I include tools/probes/gemm_f8a_ab.hip -o tools/probes/gemm_f8a_ab -Lclipa_amd/lib -lclipa_hip -Wl,-rpath,'$ORIGIN/../../clipa_amd/lib'
// Run:    tools/probes/gemm_f8a_ab [quick]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "clipa_hip.h"
#include "../../clipa_amd/csrc/internal_hooks.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__global__ void fill_bf16(unsigned short* p, size_t n, unsigned seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    const float u = ((h & 0xffffff) * (1.0f / 8388608.0f) - 1.0f) * scale;       // uniform [-scale, scale)
    p[i] = (unsigned short)(__float_as_uint(u) >> 16);
  }
}
__global__ void fill_f32(float* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (h & 0xffff) * (1.0f / 32768.0f) - 1.0f;
  }
}
__global__ void diff_count(const unsigned short* a, const unsigned short* b, size_t n, unsigned long long* cnt, unsigned long long* first) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (a[i] != b[i]) { atomicAdd(cnt, 1ull); atomicMin(first, (unsigned long long)i); }
}

struct Epi { const char* name; int epi; bool pre; bool aux; };
static const Epi EPIS[] = {{"bias", CLIPA_EPI_NONE, false, false}, {"gelu", CLIPA_EPI_ACT, false, false}, {"gelu+pre", CLIPA_EPI_ACT, true, false},
                           {"residual", CLIPA_EPI_ADD, false, true}, {"gelu_bwd", CLIPA_EPI_DACT, false, true}};

int main(int argc, char** argv) {
  setenv("CLIPA_DEBUG_HOOKS", "1", 1);   // csrc/internal_hooks.h: the experiment hooks are off in production processes
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  struct Shape { long M, N, K; };
  std::vector<Shape> shapes = {{131584, 5120, 1280}, {131584, 1280, 5120}, {131584, 3840, 1280}, {131584, 1280, 1280}, {200704, 4096, 1024}, {78848, 768, 3072}, {4096, 512, 512}, {512, 256, 768}};
  if (quick) shapes = {{131584, 5120, 1280}, {131584, 1280, 5120}, {512, 256, 768}};
  const int VARS[] = {1, 0};   // clipa_internal_debug_set variant: 1 = gemm_nt_f8_kernel, 0 = default (gemm_f8a where eligible)
  const int NV = 2;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  unsigned long long* d_cnt;
  CK(hipMalloc(&d_cnt, 16));
  for (const Shape& s : shapes) {
    const long M = s.M, N = s.N, K = s.K;
    unsigned short *A, *B, *AUX, *C[2], *C2[2];
    float* bias;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&AUX, (size_t)M * N * 2));
    CK(hipMalloc(&bias, N * 4));
    for (int i = 0; i < 2; ++i) { CK(hipMalloc(&C[i], (size_t)M * N * 2)); CK(hipMalloc(&C2[i], (size_t)M * N * 2)); }
    fill_bf16<<<2048, 256, 0, st>>>(A, (size_t)M * K, 11u, 1.0f);
    fill_bf16<<<2048, 256, 0, st>>>(B, (size_t)N * K, 22u, 0.05f);
    fill_bf16<<<2048, 256, 0, st>>>(AUX, (size_t)M * N, 33u, 1.5f);
    fill_f32<<<64, 256, 0, st>>>(bias, N, 44u);
    unsigned char *A8[2], *B8;
    float *sa[2], *sb;
    for (int f = 0; f < 2; ++f) { CK(hipMalloc(&A8[f], (size_t)M * K)); CK(hipMalloc(&sa[f], M * 4)); }
    CK(hipMalloc(&B8, (size_t)N * K)); CK(hipMalloc(&sb, N * 4));
    for (int f = 0; f < 2; ++f)
      if (clipa_quantize_rows(A, A8[f], sa[f], M, K, K, K, f, st)) { printf("quantize_rows: %s\n", clipa_last_error()); exit(3); }
    if (clipa_quantize_rows(B, B8, sb, N, K, K, K, 0, st)) { printf("quantize_rows: %s\n", clipa_last_error()); exit(3); }
    CK(hipStreamSynchronize(st));
    for (int fmt = 0; fmt < 2; ++fmt)
    for (const Epi& e : EPIS) {
      if (fmt == 1 && e.epi != CLIPA_EPI_NONE && e.epi != CLIPA_EPI_DACT) continue;     // e5m2 = gradient operand: the input-gradient GEMMs
      auto run = [&](int variant, int slot) {
        clipa_internal_debug_set(variant, 0);
        const int rc = clipa_gemm_nt_f8(A8[fmt], B8, sa[fmt], sb, C[slot], e.pre ? C2[slot] : nullptr, bias, e.aux ? AUX : nullptr, M, N, K, K, K, N, N,
                                        0.75f, e.epi, 0, fmt, 0, st);
        if (rc) { printf("clipa_gemm_nt_f8 rc=%d: %s\n", rc, clipa_last_error()); exit(3); }
      };
      // ---- bit-exactness vs gemm_nt2 ----
      CK(hipMemsetAsync(C[0], 0xff, (size_t)M * N * 2, st));
      run(1, 0);
      CK(hipStreamSynchronize(st));
      for (int vi = 1; vi < NV; ++vi) {
        const int v = VARS[vi];
        CK(hipMemsetAsync(C[1], 0x7f, (size_t)M * N * 2, st));
        if (e.pre) CK(hipMemsetAsync(C2[1], 0x7f, (size_t)M * N * 2, st));
        run(v, 1);
        unsigned long long h[2] = {0ull, ~0ull};
        CK(hipMemcpyAsync(d_cnt, h, 16, hipMemcpyHostToDevice, st));
        diff_count<<<2048, 256, 0, st>>>(C[0], C[1], (size_t)M * N, d_cnt, d_cnt + 1);
        if (e.pre) diff_count<<<2048, 256, 0, st>>>(C2[0], C2[1], (size_t)M * N, d_cnt, d_cnt + 1);
        CK(hipMemcpyAsync(h, d_cnt, 16, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        if (h[0] && M * N <= 1048576) {     // small case: where do they differ?
          std::vector<unsigned short> h0((size_t)M * N), h1((size_t)M * N);
          CK(hipMemcpy(h0.data(), C[0], (size_t)M * N * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), C[1], (size_t)M * N * 2, hipMemcpyDeviceToHost));
          int shown = 0;
          std::vector<int> rows(M, 0), cols(N, 0);
          for (long i = 0; i < M * N; ++i) if (h0[i] != h1[i]) { rows[i / N]++; cols[i % N]++; if (shown++ < 6) printf("  diff at row %ld col %ld: %04x vs %04x\n", i / N, i % N, h0[i], h1[i]); }
          if (e.epi == CLIPA_EPI_ACT && !e.pre) {
            printf("  row 124 old:"); for (int c = 0; c < 32; ++c) printf(" %04x", h0[124 * N + c]); printf("\n  row 124 new:"); for (int c = 0; c < 32; ++c) printf(" %04x", h1[124 * N + c]);
            printf("\n  row 123 new:"); for (int c = 0; c < 32; ++c) printf(" %04x", h1[123 * N + c]);
            printf("\n  row 127 new:"); for (int c = 0; c < 32; ++c) printf(" %04x", h1[127 * N + c]);
            long hits = 0; for (long i = 0; i < M * N; ++i) if (h0[i] == h1[124 * N + 7]) { if (hits++ < 4) printf("\n  old has %04x at row %ld col %ld", h0[i], i / N, i % N); }
            printf("\n");
          }
          printf("  rows:"); for (long r = 0; r < M; ++r) if (rows[r]) printf(" %ld(%d)", r, rows[r]); printf("\n  cols:");
          for (long c = 0; c < N; ++c) if (cols[c]) printf(" %ld", c); printf("\n");
        }
        printf("{\"check\": \"bits\", \"M\": %ld, \"N\": %ld, \"K\": %ld, \"epi\": \"%s\", \"fmt_a\": %d, \"kernel\": %d, \"mismatches\": %llu, \"first\": %lld}\n", M, N, K, e.name, fmt,
               clipa_internal_last_gemm(), h[0], h[0] ? (long long)h[1] : -1ll);
        fflush(stdout);
      }
      // ---- timing: interleaved rounds ----
      if (M < 50000) continue;
      const int rounds = quick ? 3 : 5, reps = 3;
      std::vector<std::vector<float>> ms(16);
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int r = 0; r < rounds; ++r)
        for (int vi = 0; vi < NV; ++vi) {
          const int v = VARS[vi];
          run(v, 1);                                  // warm
          CK(hipEventRecord(e0, st));
          for (int k = 0; k < reps; ++k) run(v, 1);
          CK(hipEventRecord(e1, st));
          CK(hipEventSynchronize(e1));
          float t;
          CK(hipEventElapsedTime(&t, e0, e1));
          ms[v].push_back(t / reps);
        }
      printf("{\"check\": \"time\", \"M\": %ld, \"N\": %ld, \"K\": %ld, \"epi\": \"%s\", \"fmt_a\": %d", M, N, K, e.name, fmt);
      for (int vi = 0; vi < NV; ++vi) {
        const int v = VARS[vi];
        std::sort(ms[v].begin(), ms[v].end());
        const float med = ms[v][ms[v].size() / 2];
        printf(", \"v%d_ms\": %.4f, \"v%d_tflops\": %.1f", v, med, v, 2.0 * M * N * K / (med * 1e-3) / 1e12);
      }
      printf("}\n");
      fflush(stdout);
      if (e.epi == CLIPA_EPI_NONE) {     // main loop only (ablation flag 2): what the epilogues cost on top
        printf("{\"check\": \"mainloop\", \"M\": %ld, \"N\": %ld, \"K\": %ld", M, N, K);
        for (int vi = 0; vi < NV; ++vi) {
          const int v = VARS[vi];
          std::vector<float> t;
          for (int r = 0; r < 3; ++r) {
            clipa_internal_debug_set(v, 2);
            clipa_gemm_nt_f8(A8[fmt], B8, sa[fmt], sb, C[1], nullptr, bias, nullptr, M, N, K, K, K, N, N, 1.0f, 0, 0, fmt, 0, st);
            CK(hipEventRecord(e0, st));
            for (int k = 0; k < reps; ++k) clipa_gemm_nt_f8(A8[fmt], B8, sa[fmt], sb, C[1], nullptr, bias, nullptr, M, N, K, K, K, N, N, 1.0f, 0, 0, fmt, 0, st);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float x;
            CK(hipEventElapsedTime(&x, e0, e1));
            t.push_back(x / reps);
          }
          std::sort(t.begin(), t.end());
          printf(", \"v%d_tflops\": %.1f", v, 2.0 * M * N * K / (t[1] * 1e-3) / 1e12);
        }
        printf("}\n");
        fflush(stdout);
      }
    }
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(AUX)); CK(hipFree(bias)); CK(hipFree(B8)); CK(hipFree(sb));
    for (int f = 0; f < 2; ++f) { CK(hipFree(A8[f])); CK(hipFree(sa[f])); }
    for (int i = 0; i < 2; ++i) { CK(hipFree(C[i])); CK(hipFree(C2[i])); }
  }
  clipa_internal_debug_set(0, 0);
  return 0;
}
