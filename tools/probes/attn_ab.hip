// A/B of clipa_attention_fwd / clipa_attention_bwd between builds of libclipa_hip.so loaded side by side (first = baseline):
// interleaved rounds, median, outputs compared bit for bit on the device.  (tools/attn_lib_ab.py without the torch import.)
// Build:  hipcc --offload-arch=gfx950 -O2 tools/probes/attn_ab.hip -o tools/probes/attn_ab -ldl
// Run:    tools/probes/attn_ab clipa_amd/lib/libclipa_hip_head.so clipa_amd/lib/libclipa_hip.so
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
typedef int (*fwd_t)(const void*, const void*, const void*, void*, float*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, float, int, void*);
typedef int (*bwd_t)(const void*, const void*, const void*, const void*, const void*, const float*, void*, void*, void*, int64_t, int64_t, int64_t,
                     int64_t, int64_t, int64_t, int64_t, float, int, void*);
typedef const char* (*err_t)(void);
typedef int (*dbg_t)(int, int);
__global__ void fill_bf16(unsigned short* p, size_t n, unsigned seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    // roughly normal: sum of three uniforms
    float u = ((h & 0x3ff) + ((h >> 10) & 0x3ff) + ((h >> 20) & 0x3ff)) * (1.0f / 1024.0f) - 1.5f;
    p[i] = (unsigned short)(__float_as_uint(u * 2.0f * scale) >> 16);
  }
}
__global__ void diff_words(const unsigned* a, const unsigned* b, size_t n, unsigned long long* cnt) {
  unsigned long long c = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
  if (c) atomicAdd(cnt, c);
}
__global__ void max_diff(const unsigned short* a, const unsigned short* b, size_t n, unsigned* mx, unsigned* mxval) {
  float m = 0.f, v = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float x = __uint_as_float((unsigned)a[i] << 16), y = __uint_as_float((unsigned)b[i] << 16);
    m = fmaxf(m, fabsf(x - y)); v = fmaxf(v, fabsf(x));
  }
  atomicMax(mx, __float_as_uint(m)); atomicMax(mxval, __float_as_uint(v));     // non-negative floats order like their bit patterns
}
__global__ void nan_words(const unsigned short* a, size_t n, unsigned long long* cnt) {
  unsigned long long c = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += (a[i] & 0x7f80) == 0x7f80;
  if (c) atomicAdd(cnt, c);
}
int main(int argc, char** argv) {
  const int NL = argc - 1;
  if (NL < 1) { printf("usage: attn_ab libA.so [libB.so ...]\n"); return 1; }
  std::vector<fwd_t> F(NL); std::vector<bwd_t> Bw(NL); std::vector<err_t> E(NL); std::vector<dbg_t> Dbg(NL);
  for (int i = 0; i < NL; ++i) {
    void* h = dlopen(argv[i + 1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { printf("dlopen %s: %s\n", argv[i + 1], dlerror()); return 1; }
    F[i] = (fwd_t)dlsym(h, "clipa_attention_fwd"); Bw[i] = (bwd_t)dlsym(h, "clipa_attention_bwd"); E[i] = (err_t)dlsym(h, "clipa_last_error"); Dbg[i] = (dbg_t)dlsym(h, "clipa_internal_debug_set");
  }
  hipStream_t st; CK(hipStreamCreate(&st));
  unsigned long long* cnt; CK(hipMalloc(&cnt, 8));
  struct Shape { long B, H, L, dh; int causal; };
  const Shape shapes[] = {{4096, 16, 197, 64, 0}, {4096, 12, 197, 64, 0}, {4096, 12, 77, 64, 1}, {2048, 16, 257, 80, 0},
                          {2048, 16, 145, 64, 0}, {4096, 16, 50, 64, 0}, {4096, 8, 32, 64, 1}, {1024, 16, 256, 64, 1}, {512, 16, 200, 64, 0},
                          {1024, 16, 50, 80, 0}, {512, 16, 77, 80, 1}, {256, 16, 100, 80, 0}, {256, 16, 26, 80, 0}, {128, 16, 288, 80, 0}};
  const int nshapes = getenv("ATTN_AB_NSHAPES") ? atoi(getenv("ATTN_AB_NSHAPES")) : 1000;
  int shape_no = 0;
  for (const Shape& s : shapes) {
    if (shape_no++ >= nshapes) break;
    const long D = s.H * s.dh, T = s.B * s.L;
    unsigned short *qkv, *dO; CK(hipMalloc(&qkv, (size_t)T * 3 * D * 2)); CK(hipMalloc(&dO, (size_t)T * D * 2));
    fill_bf16<<<4096, 256, 0, st>>>(qkv, (size_t)T * 3 * D, 1u, 1.0f); fill_bf16<<<4096, 256, 0, st>>>(dO, (size_t)T * D, 2u, 1.0f);
    std::vector<unsigned short*> out(NL), dqkv(NL); std::vector<float*> stats(NL);
    for (int i = 0; i < NL; ++i) {
      CK(hipMalloc(&out[i], (size_t)T * D * 2)); CK(hipMalloc(&dqkv[i], (size_t)T * 3 * D * 2)); CK(hipMalloc(&stats[i], (size_t)s.B * s.H * s.L * 8));
      CK(hipMemsetAsync(out[i], 0, (size_t)T * D * 2, st)); CK(hipMemsetAsync(dqkv[i], 0, (size_t)T * 3 * D * 2, st)); CK(hipMemsetAsync(stats[i], 0, (size_t)s.B * s.H * s.L * 8, st));
    }
    const float scale = 1.0f / sqrtf((float)s.dh);
    auto fwd = [&](int i) {
      if (F[i](qkv, qkv + D, qkv + 2 * D, out[i], stats[i], s.B, s.H, s.L, s.dh, 3 * D, D, scale, s.causal, st)) { printf("fwd failed: %s\n", E[i]()); exit(3); }
    };
    auto bwd = [&](int i) {
      if (Bw[i](qkv, qkv + D, qkv + 2 * D, out[i], dO, stats[i], dqkv[i], dqkv[i] + D, dqkv[i] + 2 * D, s.B, s.H, s.L, s.dh, 3 * D, D, 3 * D, scale, s.causal, st)) { printf("bwd failed: %s\n", E[i]()); exit(3); }
    };
    for (int i = 0; i < NL; ++i) { fwd(i); bwd(i); }
    unsigned long long d_out = 0, d_st = 0, d_g = 0, nan_g = 0, tmp;
    for (int i = 1; i < NL; ++i) {
      CK(hipMemsetAsync(cnt, 0, 8, st)); diff_words<<<2048, 256, 0, st>>>((unsigned*)out[0], (unsigned*)out[i], (size_t)T * D / 2, cnt); CK(hipMemcpyAsync(&tmp, cnt, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); d_out += tmp;
      CK(hipMemsetAsync(cnt, 0, 8, st)); diff_words<<<2048, 256, 0, st>>>((unsigned*)stats[0], (unsigned*)stats[i], (size_t)s.B * s.H * s.L * 2, cnt); CK(hipMemcpyAsync(&tmp, cnt, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); d_st += tmp;
      CK(hipMemsetAsync(cnt, 0, 8, st)); diff_words<<<2048, 256, 0, st>>>((unsigned*)dqkv[0], (unsigned*)dqkv[i], (size_t)T * 3 * D / 2, cnt); CK(hipMemcpyAsync(&tmp, cnt, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); d_g += tmp;
      CK(hipMemsetAsync(cnt, 0, 8, st)); nan_words<<<2048, 256, 0, st>>>(dqkv[i], (size_t)T * 3 * D, cnt); CK(hipMemcpyAsync(&tmp, cnt, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); nan_g += tmp;
    }
    unsigned* mx; CK(hipMalloc(&mx, 8)); CK(hipMemsetAsync(mx, 0, 8, st));
    float mxh[2] = {0.f, 0.f};
    if (NL > 1) { max_diff<<<2048, 256, 0, st>>>(dqkv[0], dqkv[NL - 1], (size_t)T * 3 * D, mx, mx + 1); CK(hipMemcpyAsync(mxh, mx, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); }
    CK(hipFree(mx));
    printf("{\"B\": %ld, \"H\": %ld, \"L\": %ld, \"dh\": %ld, \"causal\": %d, \"diff_words_out\": %llu, \"diff_words_stats\": %llu, \"diff_words_dqkv\": %llu, \"nonfinite_dqkv\": %llu, \"max_abs_diff_dqkv\": %.3g, \"max_abs_dqkv\": %.3g",
           s.B, s.H, s.L, s.dh, s.causal, d_out, d_st, d_g, nan_g, mxh[0], mxh[1]);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int which = 0; which < 2; ++which) {
      std::vector<std::vector<float>> ts(NL);
      for (int r = 0; r < 5; ++r)
        for (int i = 0; i < NL; ++i) {
          CK(hipEventRecord(e0, st));
          for (int k = 0; k < 3; ++k) { if (which) bwd(i); else fwd(i); }
          CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
          float x; CK(hipEventElapsedTime(&x, e0, e1)); ts[i].push_back(x / 3);
        }
      printf(", \"%s_ms\": [", which ? "bwd" : "fwd");
      for (int i = 0; i < NL; ++i) { std::sort(ts[i].begin(), ts[i].end()); printf("%s%.4f", i ? ", " : "", ts[i][2]); }
      printf("]");
    }
    // ATTN_AB_ABL="f1,f2,..." (with CLIPA_DEBUG_HOOKS=1): the LAST library's backward timed under each experiment-flag word
    // (clipa_internal_debug_set(0, f): timing ablations, wrong results) - medians of 5 x 3 launches
    if (const char* abl = getenv("ATTN_AB_ABL")) {
      printf(", \"bwd_ablation_ms\": {");
      const char* q = abl; bool first = true;
      while (*q) {
        const int f = (int)strtol(q, (char**)&q, 10); if (*q == ',') ++q;
        if (Dbg[NL - 1](0, f)) { printf("\"%d\": \"refused: %s\"", f, E[NL - 1]()); break; }
        std::vector<float> ts;
        for (int r = 0; r < 5; ++r) {
          CK(hipEventRecord(e0, st));
          for (int k = 0; k < 3; ++k) bwd(NL - 1);
          CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
          float x; CK(hipEventElapsedTime(&x, e0, e1)); ts.push_back(x / 3);
        }
        std::sort(ts.begin(), ts.end());
        printf("%s\"%d\": %.4f", first ? "" : ", ", f, ts[2]); first = false;
      }
      Dbg[NL - 1](0, 0);
      printf("}");
    }
    printf("}\n"); fflush(stdout);
    CK(hipFree(qkv)); CK(hipFree(dO));
    for (int i = 0; i < NL; ++i) { CK(hipFree(out[i])); CK(hipFree(dqkv[i])); CK(hipFree(stats[i])); }
  }
  return 0;
}
