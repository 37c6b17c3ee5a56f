// Stand-alone probe (not part of libclipa_hip.so): how many bytes per clock can one CU pull through `buffer_load_dwordx4 ... lds`
// (the GEMMs' operand path) and push through 16-byte global stores (their epilogue path)?
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/ldsdma_probe tools/probes/ldsdma_probe.hip && tools/probes/ldsdma_probe
// Each workgroup (one per CU) repeats a "K step": every wave issues P LDS-DMA pieces of 1 KiB (64 lanes x 16 B) into a 2-slot
// LDS ring, waits for the previous step's pieces, barrier - the GEMM main loop without its MFMAs.  Source rows: 128-byte
// segments at a 2 KiB row stride (the A / B tiles of a K = 1024 bf16 GEMM); "l2" re-reads a 1 MiB window per workgroup pair
// (L2 / Infinity-Cache resident), "hbm" streams a region that is touched once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// one non-template kernel per piece count (P = LDS-DMA pieces per wave per step)
#define P 4
#define DMA_NAME dma_kernel_4
__global__ __launch_bounds__(512) void DMA_NAME(const char* src, long region_bytes, long wg_stride, int steps, int* sink, int gemm_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* base = src + (long)blockIdx.x * wg_stride;
  // piece pc of a step covers 8 rows x 128 B, rows at a 2 KiB stride (K = 1024 bf16)
  unsigned voff[P];
  for (int j = 0; j < P; ++j) {
    const int pc = j * 8 + wave;
    const int row = pc * 8 + (lane >> 3);
    voff[j] = (unsigned)(row * 2048 + (lane & 7) * 16);
  }
  if (gemm_n > 0) {
    // GEMM-shaped addressing: persistent workgroup walks output tiles t = (ti, tj) of a [4096*256] x [gemm_n*256] x 1024 bf16 GEMM,
    // row-panel-major; half of a step's pieces come from A panel ti (streamed once), half from B panel tj (gemm_n * 512 KiB, hot)
    for (int j = 0; j < P; ++j) {
      const int pc = (j % (P / 2)) * 8 + wave;
      voff[j] = (unsigned)((pc * 8 + (lane >> 3)) * 2048 + (lane & 7) * 16);
    }
    for (int s = 0; s < steps; ++s) {
      const long t = blockIdx.x + (long)(s >> 3) * gridDim.x;
      const char* pa = src + ((t / gemm_n) % 4096) * (512L << 10);
      const char* pb = src + (4L << 30) + (t % gemm_n) * (512L << 10);
      const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(pa), 0, 0x7fffffff, 0x00020000);
      const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(pb), 0, 0x7fffffff, 0x00020000);
      char* dst = smem + (s & 1) * (P * 8 * 1024);
#pragma unroll
      for (int j = 0; j < P; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(j < P / 2 ? ra : rb, LDS_PTR(dst + (j * 8 + wave) * 1024), 16, voff[j], (s & 7) * 128, 0, 0);
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(P) : "memory");
      __builtin_amdgcn_s_barrier();
    }
  } else {
  long off = 0;
  for (int s = 0; s < steps; ++s) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base + off), 0, 0x7fffffff, 0x00020000);
    char* dst = smem + (s & 1) * (P * 8 * 1024);
#pragma unroll
    for (int j = 0; j < P; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(dst + (j * 8 + wave) * 1024), 16, voff[j], (s & 7) * 128, 0, 0);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(P) : "memory");      // the previous step's pieces have landed
    __builtin_amdgcn_s_barrier();
    if ((s & 7) == 7) off += (long)P * 64 * 2048;                    // next block of rows after 8 k-slabs
    if (off + (long)P * 64 * 2048 > region_bytes) off = 0;
  }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (smem[threadIdx.x] == 77 && sink) sink[0] = 1;
}


#undef P
#undef DMA_NAME
#define P 8
#define DMA_NAME dma_kernel_8
__global__ __launch_bounds__(512) void DMA_NAME(const char* src, long region_bytes, long wg_stride, int steps, int* sink, int gemm_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* base = src + (long)blockIdx.x * wg_stride;
  // piece pc of a step covers 8 rows x 128 B, rows at a 2 KiB stride (K = 1024 bf16)
  unsigned voff[P];
  for (int j = 0; j < P; ++j) {
    const int pc = j * 8 + wave;
    const int row = pc * 8 + (lane >> 3);
    voff[j] = (unsigned)(row * 2048 + (lane & 7) * 16);
  }
  if (gemm_n > 0) {
    // GEMM-shaped addressing: persistent workgroup walks output tiles t = (ti, tj) of a [4096*256] x [gemm_n*256] x 1024 bf16 GEMM,
    // row-panel-major; half of a step's pieces come from A panel ti (streamed once), half from B panel tj (gemm_n * 512 KiB, hot)
    for (int j = 0; j < P; ++j) {
      const int pc = (j % (P / 2)) * 8 + wave;
      voff[j] = (unsigned)((pc * 8 + (lane >> 3)) * 2048 + (lane & 7) * 16);
    }
    for (int s = 0; s < steps; ++s) {
      const long t = blockIdx.x + (long)(s >> 3) * gridDim.x;
      const char* pa = src + ((t / gemm_n) % 4096) * (512L << 10);
      const char* pb = src + (4L << 30) + (t % gemm_n) * (512L << 10);
      const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(pa), 0, 0x7fffffff, 0x00020000);
      const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(pb), 0, 0x7fffffff, 0x00020000);
      char* dst = smem + (s & 1) * (P * 8 * 1024);
#pragma unroll
      for (int j = 0; j < P; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(j < P / 2 ? ra : rb, LDS_PTR(dst + (j * 8 + wave) * 1024), 16, voff[j], (s & 7) * 128, 0, 0);
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(P) : "memory");
      __builtin_amdgcn_s_barrier();
    }
  } else {
  long off = 0;
  for (int s = 0; s < steps; ++s) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base + off), 0, 0x7fffffff, 0x00020000);
    char* dst = smem + (s & 1) * (P * 8 * 1024);
#pragma unroll
    for (int j = 0; j < P; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(dst + (j * 8 + wave) * 1024), 16, voff[j], (s & 7) * 128, 0, 0);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(P) : "memory");      // the previous step's pieces have landed
    __builtin_amdgcn_s_barrier();
    if ((s & 7) == 7) off += (long)P * 64 * 2048;                    // next block of rows after 8 k-slabs
    if (off + (long)P * 64 * 2048 > region_bytes) off = 0;
  }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (smem[threadIdx.x] == 77 && sink) sink[0] = 1;
}


#undef P
#undef DMA_NAME
#define P 16
#define DMA_NAME dma_kernel_16
__global__ __launch_bounds__(512) void DMA_NAME(const char* src, long region_bytes, long wg_stride, int steps, int* sink, int gemm_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* base = src + (long)blockIdx.x * wg_stride;
  // piece pc of a step covers 8 rows x 128 B, rows at a 2 KiB stride (K = 1024 bf16)
  unsigned voff[P];
  for (int j = 0; j < P; ++j) {
    const int pc = j * 8 + wave;
    const int row = pc * 8 + (lane >> 3);
    voff[j] = (unsigned)(row * 2048 + (lane & 7) * 16);
  }
  if (gemm_n > 0) {
    // GEMM-shaped addressing: persistent workgroup walks output tiles t = (ti, tj) of a [4096*256] x [gemm_n*256] x 1024 bf16 GEMM,
    // row-panel-major; half of a step's pieces come from A panel ti (streamed once), half from B panel tj (gemm_n * 512 KiB, hot)
    for (int j = 0; j < P; ++j) {
      const int pc = (j % (P / 2)) * 8 + wave;
      voff[j] = (unsigned)((pc * 8 + (lane >> 3)) * 2048 + (lane & 7) * 16);
    }
    for (int s = 0; s < steps; ++s) {
      const long t = blockIdx.x + (long)(s >> 3) * gridDim.x;
      const char* pa = src + ((t / gemm_n) % 4096) * (512L << 10);
      const char* pb = src + (4L << 30) + (t % gemm_n) * (512L << 10);
      const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(pa), 0, 0x7fffffff, 0x00020000);
      const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(pb), 0, 0x7fffffff, 0x00020000);
      char* dst = smem + (s & 1) * (P * 8 * 1024);
#pragma unroll
      for (int j = 0; j < P; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(j < P / 2 ? ra : rb, LDS_PTR(dst + (j * 8 + wave) * 1024), 16, voff[j], (s & 7) * 128, 0, 0);
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(P) : "memory");
      __builtin_amdgcn_s_barrier();
    }
  } else {
  long off = 0;
  for (int s = 0; s < steps; ++s) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base + off), 0, 0x7fffffff, 0x00020000);
    char* dst = smem + (s & 1) * (P * 8 * 1024);
#pragma unroll
    for (int j = 0; j < P; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(dst + (j * 8 + wave) * 1024), 16, voff[j], (s & 7) * 128, 0, 0);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(P) : "memory");      // the previous step's pieces have landed
    __builtin_amdgcn_s_barrier();
    if ((s & 7) == 7) off += (long)P * 64 * 2048;                    // next block of rows after 8 k-slabs
    if (off + (long)P * 64 * 2048 > region_bytes) off = 0;
  }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (smem[threadIdx.x] == 77 && sink) sink[0] = 1;
}


#undef P
#undef DMA_NAME
__global__ __launch_bounds__(512) void store_kernel(char* dst, long wg_stride, int steps) {
  // every thread stores 16 B; a wave instruction = 2 rows of 512 B (the GEMM epilogue's coalesced rows)
  char* base = dst + (long)blockIdx.x * wg_stride;
  const uint4 v = make_uint4(threadIdx.x, 1, 2, 3);
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = j * 512 + threadIdx.x;
      *(uint4*)(base + ((long)s * 64 + (c >> 5)) * 8192 + (c & 31) * 16) = v;      // 64 rows x 512 B per step, row stride 8 KiB (N = 4096 bf16)
    }
  }
}

int main() {
  int dev = 0; CHECK(hipSetDevice(dev));
  hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, dev));
  const int ncu = pr.multiProcessorCount;
  const long big = 8L << 30;
  char* buf; CHECK(hipMalloc(&buf, big)); CHECK(hipMemset(buf, 1, big));
  int* sink; CHECK(hipMalloc(&sink, 4));
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
#define RUN(name, P, region, stride, nsteps) RUNG(name, P, region, stride, nsteps, 0)
#define RUNG(name, P, region, stride, nsteps, gemm_n_)                                                                                   \
  do {                                                                                                                           \
    CHECK(hipFuncSetAttribute((const void*)dma_kernel_##P, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P * 8 * 1024));          \
    for (int rep = 0; rep < 2; ++rep) {                                                                                          \
      CHECK(hipEventRecord(a));                                                                                                  \
      hipLaunchKernelGGL(dma_kernel_##P, dim3(ncu), dim3(512), 2 * P * 8 * 1024, 0, (const char*)buf, (long)(region), (long)(stride), nsteps, sink, gemm_n_); \
      CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));                                                                   \
    }                                                                                                                            \
    float ms_; CHECK(hipEventElapsedTime(&ms_, a, b));                                                                           \
    const double bytes_ = (double)ncu * nsteps * P * 8 * 1024;                                                                   \
    printf("{\"probe\": \"lds_dma\", \"source\": \"%s\", \"pieces_per_wave_per_step\": %d, \"KiB_per_step_per_CU\": %d, \"steps\": %d, \"ms\": %.3f, " \
           "\"TB_s\": %.2f, \"GB_s_per_CU\": %.1f, \"B_per_clk_per_CU_at_2.0GHz\": %.1f, \"us_per_step\": %.3f}\n",                   \
           name, P, P * 8, nsteps, ms_, bytes_ / ms_ / 1e9, bytes_ / ms_ / 1e6 / ncu, bytes_ / ms_ / 1e6 / ncu / 2.0, ms_ * 1e3 / nsteps); \
  } while (0)
  const int steps = 4000;
  // L2 / Infinity-Cache resident: every workgroup cycles over its own 1 MiB
  RUN("own_1MiB_per_WG_256MiB_total_infinity_cache", 8, 1L << 20, 1L << 20, steps);
  RUN("own_1MiB_per_WG_256MiB_total_infinity_cache", 4, 1L << 20, 1L << 20, steps);
  RUN("shared_1MiB_all_WGs_l2_resident", 8, 1L << 20, 0, steps);
  RUN("shared_1MiB_all_WGs_l2_resident", 4, 1L << 20, 0, steps);
  // shared hot window: all workgroups read the same 2 MiB (the weight tile of a GEMM)
  RUN("shared_2MiB_all_WGs", 8, 2L << 20, 0, steps);
  // streaming from HBM: 32 MiB per workgroup, touched once
  RUN("hbm_stream_32MiB_per_WG", 8, 32L << 20, 32L << 20, 512);
  // GEMM-shaped: A row panels streamed (each read by gemm_n workgroups), B panels hot
  RUNG("gemm_shaped_N4096_K1024_bf16", 8, 0, 0, steps, 16);
  RUNG("gemm_shaped_N1024_K1024_bf16", 8, 0, 0, steps, 4);
  // epilogue-style stores
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(store_kernel, dim3(ncu), dim3(512), 0, 0, buf, 24L << 20, 48);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
  }
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  const double sb = (double)ncu * 48 * 32768;
  printf("{\"probe\": \"global_store_16B\", \"KiB_per_step_per_CU\": 32, \"steps\": 48, \"ms\": %.3f, \"TB_s\": %.2f, \"B_per_clk_per_CU_at_2.0GHz\": %.1f}\n",
         ms, sb / ms / 1e9, sb / ms / 1e6 / ncu / 2.0);
  return 0;
}
