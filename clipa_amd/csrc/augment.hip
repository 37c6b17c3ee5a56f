// Device-side image augmentation for gfx950: RandomResizedCrop (bicubic, antialiased) + Grayscale on uint8 NHWC batches.
//
// Replaces, after the host->device copy, the per-sample CPU work of the reference's train transform
// (clipa_torch/open_clip/transform.py:152-168: torchvision RandomResizedCrop(..., BICUBIC) and gray_scale on PIL images,
// then PILToTensor for the --to-float-on-device wire format, :171-174; consumed at training/train.py:187-197).  On a PIL
// image RandomResizedCrop is crop(box).resize(size, BICUBIC) = Pillow's ImagingResample and Grayscale(3) is Pillow's rgb2l;
// the kernels below reproduce those integer pipelines BIT FOR BIT (oracle/resize_oracle.py is the restatement, pinned to
// Pillow itself): coefficient windows normalised in double and rounded to 22-bit fixed point, a horizontal pass into a
// uint8 intermediate, a vertical pass, out = clip8((2^21 + sum p*k) >> 22).  Crop boxes are sampled on the host
// (clipa_amd/data.py, torchvision's get_params) - a few bytes per sample.
// HBM-bound byte work: one thread per output pixel, neighbouring threads read neighbouring source pixels.
#include "common.h"
#include "clipa_hip.h"

namespace {

constexpr int RRC_KMAX = 48;          // coefficients per output coordinate: ksize = 2 * ceil(2 * scale) + 1 -> scale <= 11.5
constexpr int RRC_PREC = 22;          // Resample.c PRECISION_BITS for 8-bit channels

// Pillow's filter and coefficient code in the same double operations, in the same order: no FMA contraction here
#pragma clang fp contract(off)
__device__ double bicubic_w(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

// thread = (sample b, axis, output coordinate xx); axis 0 = horizontal (source extent = crop width), 1 = vertical
__global__ void rrc_coeffs_kernel(const int* __restrict__ boxes, int* __restrict__ bounds, int* __restrict__ coef, int B, int S,
                                  int Hs, int Ws, int* __restrict__ err) {
#pragma clang fp contract(off)
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)B * 2 * S) return;
  const int xx = (int)(t % S), axis = (int)((t / S) & 1), b = (int)(t / (2 * S));
  const int top = boxes[b * 4 + 0], left = boxes[b * 4 + 1], bh = boxes[b * 4 + 2], bw = boxes[b * 4 + 3];
  const bool inside = top >= 0 && left >= 0 && bh > 0 && bw > 0 && top + bh <= Hs && left + bw <= Ws;
  const int in_size = axis == 0 ? bw : bh;
  int* bo = bounds + t * 2;
  int* ko = coef + t * RRC_KMAX;
  const double scale = (double)(float)in_size / S;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 2.0 * filterscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  // a sample is rejected as a whole: bad box, or either axis needing more than RRC_KMAX coefficients per output pixel
  const double scale_o = (double)(float)(axis == 0 ? bh : bw) / S;
  const int ksize_o = (int)ceil(2.0 * (scale_o < 1.0 ? 1.0 : scale_o)) * 2 + 1;
  if (ksize > RRC_KMAX || ksize_o > RRC_KMAX || !inside) {      // reported to the host; the sample's output is zeros
    if (err && xx == 0 && axis == 0) atomicAdd(err, 1);
    bo[0] = 0; bo[1] = 0;
    return;
  }
  const double center = 0.0 + (xx + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  double k[RRC_KMAX];
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) {
    const double w = bicubic_w((x + xmin - center + 0.5) * ss);
    k[x] = w;
    ww += w;
  }
  for (int x = 0; x < xmax; ++x) {
    double w = k[x];
    if (ww != 0.0) w /= ww;
    ko[x] = w < 0 ? (int)(-0.5 + w * (1 << RRC_PREC)) : (int)(0.5 + w * (1 << RRC_PREC));
  }
  bo[0] = xmin;
  bo[1] = xmax;
}

__device__ __forceinline__ unsigned char clip8(int v) {
  v >>= RRC_PREC;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: tmp[b][r][xx][c], r < crop height; thread = (b, r, xx)
__global__ void rrc_horizontal_kernel(const unsigned char* __restrict__ src, const int* __restrict__ boxes,
                                      const int* __restrict__ bounds, const int* __restrict__ coef,
                                      unsigned char* __restrict__ tmp, int B, int Hs, int Ws, int S) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)B * Hs * S) return;
  const int xx = (int)(t % S), r = (int)((t / S) % Hs), b = (int)(t / ((long)S * Hs));
  const int top = boxes[b * 4 + 0], left = boxes[b * 4 + 1], h = boxes[b * 4 + 2];
  if (r >= h) return;
  const long ct = ((long)b * 2 + 0) * S + xx;
  const int xmin = bounds[ct * 2], n = bounds[ct * 2 + 1];
  const int* k = coef + ct * RRC_KMAX;
  const unsigned char* p = src + (((long)b * Hs + top + r) * Ws + left + xmin) * 3;
  int s0 = 1 << (RRC_PREC - 1), s1 = s0, s2 = s0;
  for (int x = 0; x < n; ++x) {
    const int kx = k[x];
    s0 += p[3 * x + 0] * kx;
    s1 += p[3 * x + 1] * kx;
    s2 += p[3 * x + 2] * kx;
  }
  unsigned char* o = tmp + (((long)b * Hs + r) * S + xx) * 3;
  o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

// vertical pass (+ optional rgb2l grayscale of flagged samples): out[b][yy][xx][c]; thread = (b, yy, xx)
__global__ void rrc_vertical_kernel(const unsigned char* __restrict__ tmp, const int* __restrict__ bounds,
                                    const int* __restrict__ coef, const unsigned char* __restrict__ gray,
                                    unsigned char* __restrict__ out, int B, int Hs, int S) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)B * S * S) return;
  const int xx = (int)(t % S), yy = (int)((t / S) % S), b = (int)(t / ((long)S * S));
  const long ct = ((long)b * 2 + 1) * S + yy;
  const int ymin = bounds[ct * 2], n = bounds[ct * 2 + 1];
  const int* k = coef + ct * RRC_KMAX;
  const unsigned char* p = tmp + (((long)b * Hs + ymin) * S + xx) * 3;
  int s0 = 1 << (RRC_PREC - 1), s1 = s0, s2 = s0;
  for (int y = 0; y < n; ++y) {
    const int ky = k[y];
    const unsigned char* q = p + (long)y * S * 3;
    s0 += q[0] * ky;
    s1 += q[1] * ky;
    s2 += q[2] * ky;
  }
  unsigned char c0 = clip8(s0), c1 = clip8(s1), c2 = clip8(s2);
  if (n == 0) { c0 = c1 = c2 = 0; }           // rejected sample (coefficient window too wide): zeros, error counted
  if (gray && gray[b]) {                      // Pillow Convert.c rgb2l, replicated (Grayscale(num_output_channels=3))
    const unsigned char l = (unsigned char)((c0 * 19595 + c1 * 38470 + c2 * 7471 + 0x8000) >> 16);
    c0 = c1 = c2 = l;
  }
  unsigned char* o = out + t * 3;
  o[0] = c0; o[1] = c1; o[2] = c2;
}

}  // namespace

extern "C" int64_t clipa_resized_crop_workspace(int64_t B, int64_t Hs, int64_t S) {
  const int64_t tmp = (B * Hs * S * 3 + 255) / 256 * 256;
  return tmp + B * 2 * S * (2 + RRC_KMAX) * (int64_t)sizeof(int);
}

extern "C" int clipa_resized_crop_u8(const void* src, const int32_t* boxes, const uint8_t* gray_flags, void* out, int64_t B,
                                     int64_t Hs, int64_t Ws, int64_t S, void* workspace, int64_t workspace_bytes,
                                     int32_t* err_count, void* stream) {
  if (B <= 0) return CLIPA_OK;
  if (Hs <= 0 || Ws <= 0 || S <= 0 || S > 4096 || Hs > 16384 || Ws > 16384) { clipa_set_error("resized_crop: bad sizes"); return CLIPA_ERR_ARG; }
  if (!workspace || workspace_bytes < clipa_resized_crop_workspace(B, Hs, S)) { clipa_set_error("resized_crop: workspace too small"); return CLIPA_ERR_ARG; }
  if (B * Hs * S / 256 + 1 > 0x7fffffffL || B * S * S / 256 + 1 > 0x7fffffffL) { clipa_set_error("resized_crop: batch too large"); return CLIPA_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  unsigned char* tmp = (unsigned char*)workspace;
  int* bounds = (int*)(tmp + (B * Hs * S * 3 + 255) / 256 * 256);
  int* coef = bounds + B * 2 * S * 2;
  hipLaunchKernelGGL(rrc_coeffs_kernel, dim3((unsigned)((B * 2 * S + 255) / 256)), dim3(256), 0, st, boxes, bounds, coef, (int)B, (int)S, (int)Hs, (int)Ws, err_count);
  if (int rc = clipa_check_launch("rrc_coeffs")) return rc;
  hipLaunchKernelGGL(rrc_horizontal_kernel, dim3((unsigned)((B * Hs * S + 255) / 256)), dim3(256), 0, st, (const unsigned char*)src, boxes,
                     bounds, coef, tmp, (int)B, (int)Hs, (int)Ws, (int)S);
  if (int rc = clipa_check_launch("rrc_horizontal")) return rc;
  hipLaunchKernelGGL(rrc_vertical_kernel, dim3((unsigned)((B * S * S + 255) / 256)), dim3(256), 0, st, tmp, bounds, coef, gray_flags,
                     (unsigned char*)out, (int)B, (int)Hs, (int)S);
  return clipa_check_launch("rrc_vertical");
}
