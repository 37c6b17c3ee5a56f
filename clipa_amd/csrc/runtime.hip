// Error plumbing of the C ABI + the fused AdamW kernel.
#include "common.h"
#include "clipa_hip.h"
#include <cstdarg>
#include <cstdio>

static thread_local char g_err[512] = "";

void clipa_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int clipa_check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    clipa_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return CLIPA_ERR_LAUNCH;
  }
  return CLIPA_OK;
}

extern "C" const char* clipa_last_error(void) { return g_err; }
extern "C" int clipa_version(void) { return 1; }

namespace {
// torch.optim.AdamW semantics (decoupled weight decay, bias correction), one pass over p/g/m/v.
template <bool PF32, bool GF32>
__global__ void adamw_kernel(void* __restrict__ param, const void* __restrict__ grad, float* __restrict__ m,
                             float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float wd,
                             float bc1, float bc2_sqrt, float gscale) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float p = PF32 ? ((float*)param)[i] : bf2f(((unsigned short*)param)[i]);
    const float g = (GF32 ? ((const float*)grad)[i] : bf2f(((const unsigned short*)grad)[i])) * gscale;
    p *= 1.0f - lr * wd;
    const float mi = b1 * m[i] + (1.0f - b1) * g;
    const float vi = b2 * v[i] + (1.0f - b2) * g * g;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p -= (lr / bc1) * (mi / denom);
    if (PF32) ((float*)param)[i] = p;
    else ((unsigned short*)param)[i] = f2bf(p);
  }
}
// Multi-tensor form: up to MT_MAX tensors per launch, their pointers and sizes travel in the kernel argument
// (no device-side table to fill); block b works on a 4096-element chunk of the tensor whose block range holds b.
constexpr int MT_MAX = 48;
constexpr int MT_CHUNK = 4096;
struct MTArgs {
  void* p[MT_MAX];
  const void* g[MT_MAX];
  float* m[MT_MAX];
  float* v[MT_MAX];
  long n[MT_MAX];
  int blk0[MT_MAX + 1];
  int count;
  float lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale;
  const float* gscale_dev;     // optional: gradients are additionally scaled by gscale_dev[0] (the clip coefficient)
  int clamp_slot;              // >= 0: the updated values of that tensor are clamped to [clamp_lo, clamp_hi]
  float clamp_lo, clamp_hi;
};
template <bool PF32, bool GF32>
__global__ void adamw_multi_kernel(MTArgs a) {
  int t = 0;
  while (t + 1 < a.count && (int)blockIdx.x >= a.blk0[t + 1]) ++t;
  const long base = (long)((int)blockIdx.x - a.blk0[t]) * MT_CHUNK;
  const long n = a.n[t];
  void* param = a.p[t];
  const void* grad = a.g[t];
  float* m = a.m[t];
  float* v = a.v[t];
  const float gscale = a.gscale_dev ? a.gscale * a.gscale_dev[0] : a.gscale;
  const bool clamp = t == a.clamp_slot;
  for (long i = base + threadIdx.x; i < n && i < base + MT_CHUNK; i += blockDim.x) {
    float p = PF32 ? ((float*)param)[i] : bf2f(((unsigned short*)param)[i]);
    const float g = (GF32 ? ((const float*)grad)[i] : bf2f(((const unsigned short*)grad)[i])) * gscale;
    p *= 1.0f - a.lr * a.wd;
    const float mi = a.b1 * m[i] + (1.0f - a.b1) * g;
    const float vi = a.b2 * v[i] + (1.0f - a.b2) * g * g;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / a.bc2_sqrt + a.eps;
    p -= (a.lr / a.bc1) * (mi / denom);
    if (clamp) p = fminf(fmaxf(p, a.clamp_lo), a.clamp_hi);
    if (PF32) ((float*)param)[i] = p;
    else ((unsigned short*)param)[i] = f2bf(p);
  }
}

// sum of squares of many gradient tensors (clip_grad_norm_'s total norm, train.py:270-277): one partial per block, summed
// in a FIXED order by sqnorm_reduce_kernel - no float atomics, so the clip coefficient (and with it every parameter update)
// is bit-reproducible run to run, like the other reductions of the step
template <bool GF32>
__global__ void sqnorm_multi_kernel(MTArgs a, float* __restrict__ partials) {
  __shared__ float red[4];
  int t = 0;
  while (t + 1 < a.count && (int)blockIdx.x >= a.blk0[t + 1]) ++t;
  const long base = (long)((int)blockIdx.x - a.blk0[t]) * MT_CHUNK;
  const long n = a.n[t];
  const void* grad = a.g[t];
  float s = 0.f;
  for (long i = base + threadIdx.x; i < n && i < base + MT_CHUNK; i += blockDim.x) {
    const float g = GF32 ? ((const float*)grad)[i] : bf2f(((const unsigned short*)grad)[i]);
    s += g * g;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// acc[0] += sum of partials[0..n): thread t adds partials t, t+256, ... in order, then a fixed tree over the 256 threads
__global__ void sqnorm_reduce_kernel(const float* __restrict__ partials, long n, float* __restrict__ acc) {
  __shared__ float red[256];
  float s = 0.f;
  for (long i = threadIdx.x; i < n; i += 256) s += partials[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) acc[0] += red[0];
}
__global__ void clip_coef_kernel(const float* __restrict__ acc, float max_norm, float* __restrict__ norm_out, float* __restrict__ coef_out) {
  const float n = sqrtf(acc[0]);
  if (norm_out) norm_out[0] = n;
  const float c = max_norm / (n + 1e-6f);      // torch.nn.utils.clip_grad_norm_: clip_coef clamped to <= 1
  coef_out[0] = c < 1.0f ? c : 1.0f;
}
}  // namespace

extern "C" int clipa_grad_sqnorm_multi(const void* const* grads, const int64_t* numel, int count, int grad_f32, float* acc,
                                       float* partials, int64_t partials_cap, void* stream) {
  if (count <= 0) return CLIPA_OK;
  if (!grads || !numel || !acc || !partials) { clipa_set_error("grad_sqnorm_multi: null argument"); return CLIPA_ERR_ARG; }
  long need = 0;
  for (int i = 0; i < count; ++i) need += numel[i] > 0 ? (numel[i] + MT_CHUNK - 1) / MT_CHUNK : 0;
  if (need > partials_cap) { clipa_set_error("grad_sqnorm_multi: partials holds %ld floats, %ld needed (one per %d elements of every tensor)", (long)partials_cap, need, MT_CHUNK); return CLIPA_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  long done = 0;
  for (int next = 0; next < count;) {
    MTArgs a;
    a.count = 0;
    a.blk0[0] = 0;
    int i = next;
    for (; i < count && a.count < MT_MAX; ++i) {
      if (numel[i] <= 0) continue;
      const int c = a.count++;
      a.g[c] = grads[i]; a.n[c] = (long)numel[i];
      a.blk0[c + 1] = a.blk0[c] + (int)((numel[i] + MT_CHUNK - 1) / MT_CHUNK);
    }
    next = i;
    if (a.count == 0) continue;
    const unsigned grid = (unsigned)a.blk0[a.count];
    if (grad_f32) hipLaunchKernelGGL(sqnorm_multi_kernel<true>, dim3(grid), dim3(256), 0, st, a, partials + done);
    else hipLaunchKernelGGL(sqnorm_multi_kernel<false>, dim3(grid), dim3(256), 0, st, a, partials + done);
    if (int rc = clipa_check_launch("grad_sqnorm_multi")) return rc;
    done += grid;
  }
  if (done > 0) {
    hipLaunchKernelGGL(sqnorm_reduce_kernel, dim3(1), dim3(256), 0, st, partials, done, acc);
    if (int rc = clipa_check_launch("grad_sqnorm_reduce")) return rc;
  }
  return CLIPA_OK;
}

extern "C" int clipa_clip_coef(const float* acc, float max_norm, float* norm_out, float* coef_out, void* stream) {
  if (!acc || !coef_out) { clipa_set_error("clip_coef: null argument"); return CLIPA_ERR_ARG; }
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, acc, max_norm, norm_out, coef_out);
  return clipa_check_launch("clip_coef");
}

extern "C" int clipa_adamw_multi(void* const* params, const void* const* grads, float* const* exp_avg,
                                 float* const* exp_avg_sq, const int64_t* numel, int count, int param_f32,
                                 int grad_f32, float lr, float beta1, float beta2, float eps, float weight_decay,
                                 int64_t step, float grad_scale, const float* grad_scale_dev, int clamp_index,
                                 float clamp_lo, float clamp_hi, void* stream) {
  if (count <= 0) return CLIPA_OK;
  if (step < 1) { clipa_set_error("adamw_multi: step must be >= 1"); return CLIPA_ERR_ARG; }
  if (!params || !grads || !exp_avg || !exp_avg_sq || !numel) { clipa_set_error("adamw_multi: null table"); return CLIPA_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  for (int next = 0; next < count;) {
    MTArgs a;
    a.count = 0;
    a.blk0[0] = 0;
    a.clamp_slot = -1;
    a.clamp_lo = clamp_lo; a.clamp_hi = clamp_hi; a.gscale_dev = grad_scale_dev;
    int i = next;                      // consumed index: empty tensors are skipped without using up a slot
    for (; i < count && a.count < MT_MAX; ++i) {
      if (numel[i] <= 0) continue;
      const int c = a.count++;
      if (i == clamp_index) a.clamp_slot = c;
      a.p[c] = params[i]; a.g[c] = grads[i]; a.m[c] = exp_avg[i]; a.v[c] = exp_avg_sq[i]; a.n[c] = (long)numel[i];
      a.blk0[c + 1] = a.blk0[c] + (int)((numel[i] + MT_CHUNK - 1) / MT_CHUNK);
    }
    next = i;
    if (a.count == 0) continue;
    a.lr = lr; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.wd = weight_decay; a.gscale = grad_scale;
    a.bc1 = 1.0f - powf(beta1, (float)step);
    a.bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
    const unsigned grid = (unsigned)a.blk0[a.count];
#define LAUNCH(P, G) hipLaunchKernelGGL((adamw_multi_kernel<P, G>), dim3(grid), dim3(256), 0, st, a)
    if (param_f32 && grad_f32) LAUNCH(true, true);
    else if (param_f32) LAUNCH(true, false);
    else if (grad_f32) LAUNCH(false, true);
    else LAUNCH(false, false);
#undef LAUNCH
    if (int rc = clipa_check_launch("adamw_multi")) return rc;
  }
  return CLIPA_OK;
}

extern "C" int clipa_adamw(void* param, const void* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                           int param_f32, int grad_f32, float lr, float beta1, float beta2, float eps,
                           float weight_decay, int64_t step, float grad_scale, void* stream) {
  if (n <= 0) return CLIPA_OK;
  if (step < 1) { clipa_set_error("adamw: step must be >= 1"); return CLIPA_ERR_ARG; }
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
  long g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH(P, G) hipLaunchKernelGGL((adamw_kernel<P, G>), dim3((unsigned)g), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq, (long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale)
  if (param_f32 && grad_f32) LAUNCH(true, true);
  else if (param_f32) LAUNCH(true, false);
  else if (grad_f32) LAUNCH(false, true);
  else LAUNCH(false, false);
#undef LAUNCH
  return clipa_check_launch("adamw");
}

// out[i] = scale * sum_{w < W} in[w * n + i]: the local half of a one-hop gradient reduce-scatter (every rank sends shard j
// of its gradients straight to rank j - an all-to-all over the xGMI mesh - and sums the W pieces it received; replaces the
// ring reduction inside DDP's bucket all-reduce, clipa_torch/training/main.py:292-299).  Pieces are summed in rank order in
// fp32: the same result on every run.  HBM-bound, 16-byte accesses.
namespace {
template <bool IN_F32, bool OUT_F32>
__global__ void reduce_shards_kernel(const void* __restrict__ in, void* __restrict__ out, long n, int W, float scale) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i >= n) return;
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int w = 0; w < W; ++w) {
    float v[8];
    if (IN_F32) {
      const float4 x = *(const float4*)((const float*)in + (size_t)w * n + i), y = *(const float4*)((const float*)in + (size_t)w * n + i + 4);
      v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
    } else {
      unpack8(*(const u32x4*)((const unsigned short*)in + (size_t)w * n + i), v);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] += v[k];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] *= scale;
  if (OUT_F32) {
    *(float4*)((float*)out + i) = make_float4(a[0], a[1], a[2], a[3]);
    *(float4*)((float*)out + i + 4) = make_float4(a[4], a[5], a[6], a[7]);
  } else {
    *(u32x4*)((unsigned short*)out + i) = pack8(a);
  }
}
}  // namespace

extern "C" int clipa_reduce_shards(const void* in, void* out, int64_t n, int W, int in_f32, int out_f32, float scale,
                                   void* stream) {
  if (n <= 0) return CLIPA_OK;
  if (n % 8 != 0 || W <= 0) { clipa_set_error("reduce_shards: n must be a multiple of 8 and W > 0"); return CLIPA_ERR_ARG; }
  const dim3 grid((unsigned)((n / 8 + 255) / 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (in_f32 && out_f32) hipLaunchKernelGGL((reduce_shards_kernel<true, true>), grid, block, 0, st, in, out, (long)n, W, scale);
  else if (in_f32) hipLaunchKernelGGL((reduce_shards_kernel<true, false>), grid, block, 0, st, in, out, (long)n, W, scale);
  else if (out_f32) hipLaunchKernelGGL((reduce_shards_kernel<false, true>), grid, block, 0, st, in, out, (long)n, W, scale);
  else hipLaunchKernelGGL((reduce_shards_kernel<false, false>), grid, block, 0, st, in, out, (long)n, W, scale);
  return clipa_check_launch("reduce_shards");
}
