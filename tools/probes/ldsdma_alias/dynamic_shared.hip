#include "common.h"
extern "C" __global__ __launch_bounds__(256) void k(const char* g, float* out, int n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* A = smem; char* B = smem + 32768;
  const int lane = threadIdx.x;
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(g, 1u << 30);
  // fill B normally
  *(float4*)(B + lane * 16) = make_float4(1, 2, 3, 4);
  __syncthreads();
  float acc = 0.f;
  for (int it = 0; it < n; ++it) {
    // DMA into A (not read in this iteration)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(A + (it & 1) * 16384), 16, (unsigned)(lane * 16 + it * 4096), 0, 0, 0);
    // compute from B
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float4 v = *(const float4*)(B + ((lane * 16 + j * 4096) & 32767));
      acc += v.x * v.y + v.z * v.w;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  acc += *(const float*)(A + lane * 4);
  out[lane] = acc;
}
