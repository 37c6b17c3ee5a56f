#include "common.h"
extern "C" __global__ __launch_bounds__(256) void k(const char* g, float* out, int n) {
  __shared__ __attribute__((aligned(16))) char A[32768];
  __shared__ __attribute__((aligned(16))) char B[32768];
  __shared__ __attribute__((aligned(16))) char C[32768];
  const int lane = threadIdx.x;
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(g, 1u << 30);
  *(float4*)(B + lane * 16) = make_float4(1, 2, 3, 4);
  *(float4*)(C + lane * 16) = make_float4(1, 2, 3, 4);
  __syncthreads();
  float acc = 0.f;
  for (int it = 0; it < n; ++it) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(A + (it & 1) * 16384), 16, (unsigned)(lane * 16 + it * 4096), 0, 0, 0);
    char* R = (it & 2) ? B : C;     // runtime select between two arrays that are NOT the DMA target
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float4 v = *(const float4*)(R + ((lane * 16 + j * 4096 + it * 64) & 32767));
      acc += v.x * v.y + v.z * v.w;
    }
    *(float*)(R + ((lane * 4 + it * 1024) & 32767)) = acc;      // and an LDS write
    __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  acc += *(const float*)(A + lane * 4);
  out[lane] = acc;
}
