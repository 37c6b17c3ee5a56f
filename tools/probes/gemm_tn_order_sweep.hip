// Experiment harness (not part of the library): clipa_gemm_tn at production shapes under the work-order flags of
// clipa_internal_debug_set (0 = the library's choice, 4096 = slice-per-XCD order forced, 8192 = tile-per-XCD order forced).
// Build:  hipcc --offload-arch=gfx950 -O2 -I include tools/probes/gemm_tn_order_sweep.hip -o tools/probes/gemm_tn_order_sweep -Lclipa_amd/lib -lclipa_hip -Wl,-rpath,'$ORIGIN/../../clipa_amd/lib'
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
  struct Shape { long M, R, C; };
  const Shape shapes[] = {{526336, 1280, 5120}, {526336, 5120, 1280}, {526336, 3840, 1280}, {526336, 1280, 1280},
                          {806912, 1024, 4096}, {806912, 4096, 1024}, {806912, 3072, 1024}, {806912, 1024, 1024}};
  const int flags[] = {0, 4096, 8192, 0, 4096, 8192};
  for (const Shape& s : shapes) {
    unsigned short *P, *Q; float *out, *cs; void* ws;
    CK(hipMalloc(&P, (size_t)s.M * s.R * 2)); CK(hipMalloc(&Q, (size_t)s.M * s.C * 2)); CK(hipMalloc(&out, (size_t)s.R * s.C * 4)); CK(hipMalloc(&cs, s.R * 4));
    fill_bf16<<<2048, 256, 0, st>>>(P, (size_t)s.M * s.R, 1u, 1.0f); fill_bf16<<<2048, 256, 0, st>>>(Q, (size_t)s.M * s.C, 2u, 1.0f);
    int64_t wsb = 0;
    for (int f : {0, 4096, 8192}) { clipa_internal_debug_set(0, f); int64_t ns; wsb = std::max(wsb, clipa_gemm_tn_workspace(s.M, s.R, s.C, &ns)); }
    CK(hipMalloc(&ws, wsb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("{\"M\": %ld, \"R\": %ld, \"C\": %ld", s.M, s.R, s.C);
    for (int f : flags) {
      clipa_internal_debug_set(0, f);
      int64_t ns = 0; clipa_gemm_tn_workspace(s.M, s.R, s.C, &ns);
      std::vector<float> t;
      for (int r = 0; r < 3; ++r) {
        if (clipa_gemm_tn(P, Q, out, cs, s.M, s.R, s.C, s.R, s.C, 0, ws, wsb, st)) { printf("gemm_tn failed: %s\n", clipa_last_error()); return 3; }
        CK(hipEventRecord(e0, st));
        for (int k = 0; k < 2; ++k) clipa_gemm_tn(P, Q, out, cs, s.M, s.R, s.C, s.R, s.C, 0, ws, wsb, st);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float x; CK(hipEventElapsedTime(&x, e0, e1)); t.push_back(x / 2);
      }
      std::sort(t.begin(), t.end());
      printf(", \"f%d(S=%ld)\": %.1f", f, (long)ns, 2.0 * s.M * s.R * s.C / (t[1] * 1e-3) / 1e12);
    }
    printf("}\n"); fflush(stdout);
    CK(hipFree(P)); CK(hipFree(Q)); CK(hipFree(out)); CK(hipFree(cs)); CK(hipFree(ws));
  }
  clipa_internal_debug_set(0, 0);
  return 0;
}
