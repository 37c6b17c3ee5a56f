// A/B harness (test infrastructure, not part of libclipa_hip.so): gemm_tna (four waves, hand-scheduled main loop) against
// gemm_tn2 / gemm_tn3 through the C ABI.  Values: sampled entries against an fp64 dot product computed on the device by a
// trivial kernel, plus the largest difference between the two kernels' full outputs (slice boundaries differ, so the fp32
// summation order and the last bits do).  Timing: interleaved rounds, median.
// Build:  hipcc --offload-arch=gfx950 -O2 -I include tools/probes/gemm_tna_ab.hip -o tools/probes/gemm_tna_ab -Lclipa_amd/lib -lclipa_hip -Wl,-rpath,'$ORIGIN/../../clipa_amd/lib'
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
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
    const float u = ((h & 0xffffff) * (1.0f / 8388608.0f) - 1.0f) * scale;
    p[i] = (unsigned short)(__float_as_uint(u) >> 16);
  }
}
__device__ float bf(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
// one block per sampled (r, c): fp64 sum over m; r < 0: column sum of P column c
__global__ void ref_dots(const unsigned short* P, const unsigned short* Q, long M, long R, long C, const int* rs, const int* cs, double* out) {
  __shared__ double red[256];
  const int r = rs[blockIdx.x], c = cs[blockIdx.x];
  double s = 0.0;
  for (long m = threadIdx.x; m < M; m += 256) s += r >= 0 ? (double)bf(P[m * R + r]) * (double)bf(Q[m * C + c]) : (double)bf(P[m * R + c]);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w]; __syncthreads(); }
  if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}
__global__ void max_diff(const float* a, const float* b, size_t n, float* out) {   // out[0] = max |a-b|, out[1] = max |a|
  float d = 0.f, m = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { d = fmaxf(d, fabsf(a[i] - b[i])); m = fmaxf(m, fabsf(a[i])); }
  atomicMax((int*)out, __float_as_int(d));
  atomicMax((int*)out + 1, __float_as_int(m));
}

int main(int argc, char** argv) {
  setenv("CLIPA_DEBUG_HOOKS", "1", 1);   // csrc/internal_hooks.h: the experiment hooks are off in production processes
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  struct Shape { long M, R, C; };
  std::vector<Shape> shapes = {{200704, 4096, 1024}, {200704, 1024, 4096}, {200704, 3072, 1024}, {200704, 1024, 1024}, {78848, 3072, 768}, {78848, 768, 3072}, {78848, 768, 768}, {2048, 256, 512}};
  if (quick) shapes = {{200704, 4096, 1024}, {200704, 1024, 4096}, {2048, 256, 512}};
  const int NS = 256;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  int *d_rs, *d_cs; double* d_ref; float* d_md;
  CK(hipMalloc(&d_rs, NS * 4)); CK(hipMalloc(&d_cs, NS * 4)); CK(hipMalloc(&d_ref, NS * 8)); CK(hipMalloc(&d_md, 8));
  const int flags[3] = {16384, 0, 32768};      // gemm_tn2/3, gemm_tna schedule 0, schedule 1
  for (const Shape& s : shapes) {
    const long M = s.M, R = s.R, C = s.C;
    unsigned short *P, *Q; float *out[2], *csum[2]; void* ws;
    int64_t nsl = 0;
    const int64_t wsb = clipa_gemm_tn_workspace(M, R, C, &nsl);
    CK(hipMalloc(&P, (size_t)M * R * 2)); CK(hipMalloc(&Q, (size_t)M * C * 2)); CK(hipMalloc(&ws, wsb));
    for (int i = 0; i < 2; ++i) { CK(hipMalloc(&out[i], (size_t)R * C * 4)); CK(hipMalloc(&csum[i], R * 4)); }
    fill_bf16<<<2048, 256, 0, st>>>(P, (size_t)M * R, 5u, 0.05f);
    fill_bf16<<<2048, 256, 0, st>>>(Q, (size_t)M * C, 6u, 1.0f);
    std::vector<int> rs(NS), cs(NS);
    for (int i = 0; i < NS; ++i) { rs[i] = i < NS - 32 ? (int)((i * 7919L + 13) % R) : -1; cs[i] = i < NS - 32 ? (int)((i * 104729L + 7) % C) : (int)((i * 613L) % R); }
    rs[0] = 0; cs[0] = 0; rs[1] = (int)R - 1; cs[1] = (int)C - 1; rs[2] = 255; cs[2] = 256; rs[3] = 256; cs[3] = 255;
    CK(hipMemcpyAsync(d_rs, rs.data(), NS * 4, hipMemcpyHostToDevice, st));
    CK(hipMemcpyAsync(d_cs, cs.data(), NS * 4, hipMemcpyHostToDevice, st));
    ref_dots<<<NS, 256, 0, st>>>(P, Q, M, R, C, d_rs, d_cs, d_ref);
    std::vector<double> ref(NS);
    CK(hipMemcpyAsync(ref.data(), d_ref, NS * 8, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    auto run = [&](int v, int slot) {
      clipa_internal_debug_set(0, flags[v]);
      const int rc = clipa_gemm_tn(P, Q, out[slot], csum[slot], M, R, C, R, C, 0, ws, wsb, st);
      if (rc) { printf("clipa_gemm_tn rc=%d: %s\n", rc, clipa_last_error()); exit(3); }
    };
    for (int v = 0; v < 3; ++v) {
      const int slot = v ? 1 : 0;
      CK(hipMemsetAsync(out[slot], 0xff, (size_t)R * C * 4, st));
      CK(hipMemsetAsync(csum[slot], 0xff, R * 4, st));
      run(v, slot);
      std::vector<float> o((size_t)R * C), c(R);
      CK(hipMemcpyAsync(o.data(), out[slot], (size_t)R * C * 4, hipMemcpyDeviceToHost, st));
      CK(hipMemcpyAsync(c.data(), csum[slot], R * 4, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
      double worst = 0.0, worst_cs = 0.0; int nan = 0;
      for (int i = 0; i < NS; ++i) {
        const double got = rs[i] >= 0 ? (double)o[(size_t)rs[i] * C + cs[i]] : (double)c[cs[i]];
        if (!(got == got)) { ++nan; continue; }
        const double err = fabs(got - ref[i]) / (fabs(ref[i]) + 1e-2 * sqrt((double)M) * 0.03);
        if (rs[i] >= 0) worst = std::max(worst, err); else worst_cs = std::max(worst_cs, err);
      }
      float md[2] = {0.f, 0.f};
      if (v) {
        CK(hipMemsetAsync(d_md, 0, 8, st));
        max_diff<<<1024, 256, 0, st>>>(out[0], out[1], (size_t)R * C, d_md);
        CK(hipMemcpyAsync(md, d_md, 8, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
      }
      printf("{\"check\": \"values\", \"kernel\": %d, \"M\": %ld, \"R\": %ld, \"C\": %ld, \"variant\": %d, \"nan\": %d, \"worst_rel_err_vs_fp64\": %.3g, \"worst_colsum_rel_err\": %.3g, \"max_abs_diff_vs_old\": %.3g, \"max_abs\": %.3g}\n",
             clipa_internal_last_gemm(), M, R, C, v, nan, worst, worst_cs, md[0], md[1]);
      fflush(stdout);
    }
    if (M >= 50000) {
      const int rounds = quick ? 3 : 5, reps = 3;
      std::vector<std::vector<float>> ms(3);
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int r = 0; r < rounds; ++r)
        for (int v = 0; v < 3; ++v) {
          run(v, 1);
          CK(hipEventRecord(e0, st));
          for (int k = 0; k < reps; ++k) run(v, 1);
          CK(hipEventRecord(e1, st));
          CK(hipEventSynchronize(e1));
          float t;
          CK(hipEventElapsedTime(&t, e0, e1));
          ms[v].push_back(t / reps);
        }
      printf("{\"check\": \"time\", \"M\": %ld, \"R\": %ld, \"C\": %ld, \"slices_planned\": %ld", M, R, C, (long)nsl);
      for (int v = 0; v < 3; ++v) {
        std::sort(ms[v].begin(), ms[v].end());
        const float med = ms[v][ms[v].size() / 2];
        printf(", \"v%d_ms\": %.4f, \"v%d_tflops\": %.1f", v, med, v, 2.0 * M * R * C / (med * 1e-3) / 1e12);
      }
      printf("}\n");
      fflush(stdout);
      // memory ablations of gemm_tna (wrong results): operands never fetched / every step re-reads the same 64 rows
      const int ab[3] = {0, 65536, 131072};
      printf("{\"check\": \"ablation\", \"M\": %ld, \"R\": %ld, \"C\": %ld", M, R, C);
      for (int a = 0; a < 3; ++a) {
        std::vector<float> tt;
        for (int r = 0; r < 3; ++r) {
          clipa_internal_debug_set(0, ab[a]);
          clipa_gemm_tn(P, Q, out[1], csum[1], M, R, C, R, C, 0, ws, wsb, st);
          CK(hipEventRecord(e0, st));
          for (int k = 0; k < reps; ++k) clipa_gemm_tn(P, Q, out[1], csum[1], M, R, C, R, C, 0, ws, wsb, st);
          CK(hipEventRecord(e1, st));
          CK(hipEventSynchronize(e1));
          float x;
          CK(hipEventElapsedTime(&x, e0, e1));
          tt.push_back(x / reps);
        }
        std::sort(tt.begin(), tt.end());
        printf(", \"%s_tflops\": %.1f", a == 0 ? "normal" : a == 1 ? "no_fetch" : "l2_resident", 2.0 * M * R * C / (tt[1] * 1e-3) / 1e12);
      }
      printf(", \"kernel\": %d}\n", clipa_internal_last_gemm());
      fflush(stdout);
    }
    CK(hipFree(P)); CK(hipFree(Q)); CK(hipFree(ws));
    for (int i = 0; i < 2; ++i) { CK(hipFree(out[i])); CK(hipFree(csum[i])); }
  }
  clipa_internal_debug_set(0, 0);
  return 0;
}
