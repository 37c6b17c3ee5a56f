// A/B harness (test infrastructure, not part of libclipa_hip.so): gemm_nta (four waves, hand-scheduled main loop, schedules
// 0..2 of gemm_nta_asm.inc) against gemm_nt2 (the round-1/2 production kernel) through the C ABI of the library both live in.
//   * outputs compared BIT FOR BIT for every epilogue (bias, GELU, GELU + pre-activation copy, residual add, GELU backward);
//   * timing: interleaved rounds in one process, median TF/s per variant.
// Build:  hipcc --offload-arch=gfx950 -O2 -I include tools/probes/gemm_nta_ab.hip -o tools/probes/gemm_nta_ab -Lclipa_amd/lib -lclipa_hip -Wl,-rpath,'$ORIGIN/../../clipa_amd/lib'
// Run:    tools/probes/gemm_nta_ab [quick]
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
  std::vector<Shape> shapes = {{200704, 4096, 1024}, {200704, 1024, 4096}, {200704, 3072, 1024}, {200704, 1024, 1024}, {78848, 768, 3072}, {4096, 512, 256}, {512, 256, 384}};
  if (quick) shapes = {{200704, 4096, 1024}, {200704, 1024, 4096}, {512, 256, 384}};
  const int VARS[] = {1, 2, 5, 6};   // clipa_internal_debug_set variant: 1 = gemm_nt2, 2 + s = gemm_nta schedule s (0, 3, 4 are compiled in)
  const int NV = 4;
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
    CK(hipStreamSynchronize(st));
    for (const Epi& e : EPIS) {
      auto run = [&](int variant, int slot) {
        clipa_internal_debug_set(variant, 0);
        const int rc = clipa_gemm_nt(A, B, C[slot], e.pre ? C2[slot] : nullptr, bias, e.aux ? AUX : nullptr, M, N, K, K, K, N, N, 1.0f, e.epi, 0, 0, st);
        if (rc) { printf("clipa_gemm_nt rc=%d: %s\n", rc, clipa_last_error()); exit(3); }
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
        printf("{\"check\": \"bits\", \"M\": %ld, \"N\": %ld, \"K\": %ld, \"epi\": \"%s\", \"variant\": %d, \"mismatches\": %llu, \"first\": %lld}\n", M, N, K, e.name, v,
               h[0], h[0] ? (long long)h[1] : -1ll);
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
      printf("{\"check\": \"time\", \"M\": %ld, \"N\": %ld, \"K\": %ld, \"epi\": \"%s\"", M, N, K, e.name);
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
            clipa_gemm_nt(A, B, C[1], nullptr, bias, nullptr, M, N, K, K, K, N, N, 1.0f, 0, 0, 0, st);
            CK(hipEventRecord(e0, st));
            for (int k = 0; k < reps; ++k) clipa_gemm_nt(A, B, C[1], nullptr, bias, nullptr, M, N, K, K, K, N, N, 1.0f, 0, 0, 0, st);
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
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(AUX)); CK(hipFree(bias));
    for (int i = 0; i < 2; ++i) { CK(hipFree(C[i])); CK(hipFree(C2[i])); }
  }
  clipa_internal_debug_set(0, 0);
  return 0;
}
