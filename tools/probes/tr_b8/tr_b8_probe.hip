// Probe of gfx950's ds_read_b64_tr_b8: which LDS byte lands in which (lane, byte) of the result.
//   hipcc --offload-arch=gfx950 -O2 -o tr_b8_probe tr_b8_probe.hip && ./tr_b8_probe
// Every lane supplies its own 8-byte-aligned address (lane * STRIDE); the LDS holds a 16-bit id per byte in two planes
// (low / high byte of the byte address, two runs), so each output byte names the address it was read from.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void probe(const int* addr, unsigned long long* out, int plane) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = plane ? (unsigned char)(i >> 8) : (unsigned char)(i & 255);
  __syncthreads();
  const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds + (unsigned)addr[threadIdx.x];
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b8 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  out[threadIdx.x] = v;
}

int main() {
  int* daddr; unsigned long long* dout;
  hipMalloc(&daddr, 64 * 4); hipMalloc(&dout, 64 * 8);
  const int strides[3] = {8, 264, 520};
  for (int s = 0; s < 3; ++s) {
    std::vector<int> addr(64);
    for (int l = 0; l < 64; ++l) addr[l] = (l * strides[s]) & ~7;
    if (s == 2) for (int l = 0; l < 64; ++l) addr[l] = (((l * 37) % 64) * 136) & ~7;      // scrambled
    hipMemcpy(daddr, addr.data(), 256, hipMemcpyHostToDevice);
    unsigned long long lo[64], hi[64];
    probe<<<1, 64>>>(daddr, dout, 0); hipMemcpy(lo, dout, 512, hipMemcpyDeviceToHost);
    probe<<<1, 64>>>(daddr, dout, 1); hipMemcpy(hi, dout, 512, hipMemcpyDeviceToHost);
    printf("case %d (lane address = %s)\n", s, s == 2 ? "((37 l) %% 64) * 136" : (s ? "264 l" : "8 l"));
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int b = 0; b < 8; ++b) {
        const int a = (int)((lo[l] >> (8 * b)) & 255) | ((int)((hi[l] >> (8 * b)) & 255) << 8);
        int src = -1, off = -1;
        for (int k = 0; k < 64; ++k) if (a >= addr[k] && a < addr[k] + 8) { src = k; off = a - addr[k]; }
        printf(" L%02d+%d", src, off);
      }
      printf("\n");
    }
  }
  return 0;
}
