// Experiment (not part of the library): tile-group size (A panels per XCD group) of the persistent NT kernels at the real
// launch shapes, M = 806 912.  clipa_internal_debug_set flags bits 20..25 override nt_group_size.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "clipa_hip.h"
#include "../../clipa_amd/csrc/internal_hooks.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
__global__ void fill_bf16(unsigned short* p, size_t n, unsigned seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    p[i] = (unsigned short)(__float_as_uint(((h & 0xffffff) * (1.0f / 8388608.0f) - 1.0f) * scale) >> 16);
  }
}
int main() {
  setenv("CLIPA_DEBUG_HOOKS", "1", 1);   // csrc/internal_hooks.h: the experiment hooks are off in production processes
  hipStream_t st; CK(hipStreamCreate(&st));
  struct Shape { long M, N, K; };
  const Shape shapes[] = {{806912, 4096, 1024}, {806912, 1024, 4096}, {806912, 3072, 1024}, {806912, 1024, 1024}};
  const int gms[] = {0, 2, 4, 6, 8, 12, 16, 24, 32};
  for (const Shape& s : shapes) {
    unsigned short *A, *B, *C; float* bias;
    CK(hipMalloc(&A, (size_t)s.M * s.K * 2)); CK(hipMalloc(&B, (size_t)s.N * s.K * 2)); CK(hipMalloc(&C, (size_t)s.M * s.N * 2)); CK(hipMalloc(&bias, s.N * 4));
    fill_bf16<<<2048, 256, 0, st>>>(A, (size_t)s.M * s.K, 1u, 1.0f); fill_bf16<<<2048, 256, 0, st>>>(B, (size_t)s.N * s.K, 2u, 0.05f);
    CK(hipMemsetAsync(bias, 0, s.N * 4, st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("{\"M\": %ld, \"N\": %ld, \"K\": %ld", s.M, s.N, s.K);
    for (int gm : gms) {
      std::vector<float> t;
      for (int r = 0; r < 3; ++r) {
        clipa_internal_debug_set(0, gm << 20);
        clipa_gemm_nt(A, B, C, nullptr, bias, nullptr, s.M, s.N, s.K, s.K, s.K, s.N, s.N, 1.0f, 0, 0, 0, st);
        CK(hipEventRecord(e0, st));
        for (int k = 0; k < 2; ++k) clipa_gemm_nt(A, B, C, nullptr, bias, nullptr, s.M, s.N, s.K, s.K, s.K, s.N, s.N, 1.0f, 0, 0, 0, st);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float x; CK(hipEventElapsedTime(&x, e0, e1)); t.push_back(x / 2);
      }
      std::sort(t.begin(), t.end());
      printf(", \"gm%d\": %.1f", gm, 2.0 * s.M * s.N * s.K / (t[1] * 1e-3) / 1e12);
    }
    printf("}\n"); fflush(stdout);
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(bias));
  }
  clipa_internal_debug_set(0, 0);
  return 0;
}
