// Experiment harness (not part of the library): clipa_gemm_nt at the real launch shapes (M = 806 912) under a list of
// clipa_internal_debug_set experiment flags (command line; default: the XCD re-alignment periods of gemm_nta), interleaved rounds,
// output checksums compared with the first flag's.  Flags the library still has: 64 = epilogue stores dropped by the bounds
// check, 128 = every tile stores to tile 0 (L2 hits).  Round 3 also ran it with kernel-side switches that were removed after
// the measurement: workgroups of an XCD started out of phase (profiles/r03_gemm_nta_intra_xcd_stagger.jsonl) and 8 rows x
// 128 B per store instruction (profiles/r03_gemm_nta_store_row_width_timing.jsonl).
// Build:  hipcc --offload-arch=gfx950 -O2 -I include tools/probes/gemm_flag_sweep.hip -o tools/probes/gemm_flag_sweep -Lclipa_amd/lib -lclipa_hip -Wl,-rpath,'$ORIGIN/../../clipa_amd/lib'
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
__global__ void checksum_u32(const unsigned* p, size_t n, unsigned long long* out) {
  unsigned long long h = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) h += (unsigned long long)p[i] * (2 * (i & 1023) + 1);
  atomicAdd(out, h);
}
int main(int argc, char** argv) {
  setenv("CLIPA_DEBUG_HOOKS", "1", 1);   // csrc/internal_hooks.h: the experiment hooks are off in production processes
  hipStream_t st; CK(hipStreamCreate(&st));
  struct Case { long M, N, K; int epi; const char* name; };
  const Case cases[] = {{806912, 4096, 1024, 0, "bias"}, {806912, 4096, 1024, 1, "gelu"}, {806912, 4096, 1024, 3, "dact"},
                        {806912, 3072, 1024, 0, "bias"}, {806912, 1024, 4096, 2, "add"}, {806912, 1024, 4096, 0, "bias"},
                        {806912, 1024, 3072, 0, "bias"}, {806912, 1024, 1024, 2, "add"}, {315392, 3072, 768, 1, "gelu"},
                        {315392, 2304, 768, 0, "bias"}, {315392, 768, 3072, 2, "add"}, {315392, 768, 3072, 0, "bias"},
                        {315392, 768, 2304, 0, "bias"}, {315392, 3072, 768, 3, "dact"}, {315392, 768, 768, 2, "add"}};
  // flags from the command line (decimal or 0x..); default: XCD re-alignment period (bits 26..29: 15 = off, p = every p tiles),
  // each also with the main loop alone (| 2)
  std::vector<int> flags;
  for (int i = 1; i < argc; ++i) flags.push_back((int)strtol(argv[i], nullptr, 0));
  if (flags.empty()) for (int p : {15, 1, 2, 4, 8}) { flags.push_back(p << 26); }
  if (argc <= 1) for (int p : {15, 2, 4}) flags.push_back((p << 26) | 2);
  unsigned long long* d_sum; CK(hipMalloc(&d_sum, 8));
  const int from = getenv("FLAGSWEEP_FROM") ? atoi(getenv("FLAGSWEEP_FROM")) : 0;
  int case_no = 0;
  for (const Case& s : cases) {
    if (case_no++ < from) continue;
    unsigned short *A, *B, *C, *aux = nullptr; float* bias;
    CK(hipMalloc(&A, (size_t)s.M * s.K * 2)); CK(hipMalloc(&B, (size_t)s.N * s.K * 2)); CK(hipMalloc(&C, (size_t)s.M * s.N * 2)); CK(hipMalloc(&bias, s.N * 4));
    if (s.epi >= 2) { CK(hipMalloc(&aux, (size_t)s.M * s.N * 2)); fill_bf16<<<2048, 256, 0, st>>>(aux, (size_t)s.M * s.N, 3u, 1.0f); }
    fill_bf16<<<2048, 256, 0, st>>>(A, (size_t)s.M * s.K, 1u, 1.0f); fill_bf16<<<2048, 256, 0, st>>>(B, (size_t)s.N * s.K, 2u, 0.05f);
    CK(hipMemsetAsync(bias, 0, s.N * 4, st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("{\"M\": %ld, \"N\": %ld, \"K\": %ld, \"epi\": \"%s\"", s.M, s.N, s.K, s.name);
    std::vector<std::vector<float>> t(flags.size());
    std::vector<unsigned long long> sums(flags.size(), 0);
    for (int r = 0; r < 4; ++r)                       // interleaved rounds: every flag sees the same clock / thermal state
      for (size_t f = 0; f < flags.size(); ++f) {
        clipa_internal_debug_set(0, flags[f]);
        if (r == 0) CK(hipMemsetAsync(C, 0, (size_t)s.M * s.N * 2, st));
        if (clipa_gemm_nt(A, B, C, nullptr, bias, aux, s.M, s.N, s.K, s.K, s.K, s.N, s.N, 1.0f, s.epi, 0, 0, st)) { printf("gemm failed: %s\n", clipa_last_error()); return 3; }
        if (r == 0) {
          CK(hipMemsetAsync(d_sum, 0, 8, st));
          checksum_u32<<<1024, 256, 0, st>>>((const unsigned*)C, (size_t)s.M * s.N / 2, d_sum);
          CK(hipMemcpyAsync(&sums[f], d_sum, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
          continue;
        }
        CK(hipEventRecord(e0, st));
        for (int k = 0; k < 3; ++k) clipa_gemm_nt(A, B, C, nullptr, bias, aux, s.M, s.N, s.K, s.K, s.K, s.N, s.N, 1.0f, s.epi, 0, 0, st);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float x; CK(hipEventElapsedTime(&x, e0, e1)); t[f].push_back(x / 3);
      }
    for (size_t f = 0; f < flags.size(); ++f) {
      std::sort(t[f].begin(), t[f].end());
      printf(", \"f%d\": %.1f", flags[f], 2.0 * s.M * s.N * s.K / (t[f][1] * 1e-3) / 1e12);
    }
    printf(", \"same_output\": [");
    for (size_t f = 0; f < flags.size(); ++f) printf("%s%d", f ? ", " : "", (int)(sums[f] == sums[0]));
    printf("], \"last_gemm\": %d}\n", clipa_internal_last_gemm()); fflush(stdout);
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(bias)); if (aux) CK(hipFree(aux));
  }
  clipa_internal_debug_set(0, 0);
  return 0;
}
