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
}  // namespace

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
