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

// ---- ColorJitter + Grayscale on the resized uint8 batch, in place ----------------------------------------------------------
// torchvision ColorJitter on PIL images (open_clip/transform.py:61-72): brightness / contrast / saturation are Pillow's
// Image.blend(degenerate, image, factor) - libImaging Blend.c: (UINT8)(in1 + alpha * (in2 - in1)) in C float, clipped when
// alpha leaves [0, 1] - with degenerate = black / solid int(mean(L) + 0.5) / L; hue is convert('HSV'), h += uint8(f * 255),
// convert('RGB') (Convert.c rgb2hsv_row / hsv2rgb: colorsys in mixed float / double).  The four run in a per-sample random
// order, so the batch takes four passes: pass k applies every sample's k-th operation (preceded by the per-image sum of L
// that a contrast step needs - integer atomics, exact).  Same float / double operations in the same order as the C code
// (FMA contraction is off in this file): bit-exact against Pillow (oracle/color_oracle.py).
__device__ __forceinline__ int rgb2l_i(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

__device__ __forceinline__ unsigned char blend_u8(int d, int p, float a) {
  const float t = (float)d + a * (float)(p - d);
  if (a >= 0.f && a <= 1.f) return (unsigned char)(int)t;
  return t <= 0.f ? (unsigned char)0 : (t >= 255.f ? (unsigned char)255 : (unsigned char)(int)t);
}

__device__ __forceinline__ int clip8i(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

__device__ void hue_u8(int& r, int& g, int& b, int shift) {
  // rgb2hsv_row
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  int uh = 0, us = 0;
  const int uv = maxc;
  if (minc != maxc) {
    const float cr = (float)(maxc - minc);
    const float sf = cr / (float)maxc;
    const float rc = ((float)(maxc - r)) / cr, gc = ((float)(maxc - g)) / cr, bc = ((float)(maxc - b)) / cr;
    float h;
    if (r == maxc) h = bc - gc;
    else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
    else h = (float)(4.0 + (double)gc - (double)rc);
    h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
    uh = clip8i((int)((double)h * 255.0));
    us = clip8i((int)((double)sf * 255.0));
  }
  uh = (uh + shift) & 0xff;
  // hsv2rgb
  if (us == 0) { r = g = b = uv; return; }
  const double hf = (double)(float)uh * 6.0 / 255.0;
  const int i = (int)floor(hf);
  const float f = (float)(hf - (double)(float)i);
  const float fs = (float)((double)(float)us / 255.0);
  const double vf = (double)(float)uv;
  const int p = clip8i((int)round(vf * (1.0 - (double)fs)));
  const int q = clip8i((int)round(vf * (1.0 - (double)fs * (double)f)));
  const int t = clip8i((int)round(vf * (1.0 - (double)fs * (1.0 - (double)f))));
  switch (i % 6) {
    case 0: r = uv; g = t; b = p; break;
    case 1: r = q; g = uv; b = p; break;
    case 2: r = p; g = uv; b = t; break;
    case 3: r = p; g = q; b = uv; break;
    case 4: r = t; g = p; b = uv; break;
    default: r = uv; g = p; b = q; break;
  }
}

// lsum[b] += sum over the image of L (only where sample b's operation of this pass is a contrast step)
__global__ void jitter_lsum_kernel(const unsigned char* __restrict__ img, const unsigned char* __restrict__ apply,
                                   const int* __restrict__ order, unsigned long long* __restrict__ lsum, int B, int S, int step) {
  const int b = blockIdx.y;
  if ((apply && !apply[b]) || order[b * 4 + step] != 1) return;
  const long n = (long)S * S;
  const unsigned char* im = img + (long)b * n * 3;
  unsigned long long acc = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    acc += (unsigned long long)rgb2l_i(im[3 * i], im[3 * i + 1], im[3 * i + 2]);
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(lsum + b, acc);
}

__global__ void jitter_step_kernel(unsigned char* __restrict__ img, const unsigned char* __restrict__ apply,
                                   const int* __restrict__ order, const float* __restrict__ factors,
                                   const unsigned long long* __restrict__ lsum, int B, int S, int step) {
  const long n = (long)S * S;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)B * n) return;
  const int b = (int)(t / n);
  if (apply && !apply[b]) return;
  const int op = order[b * 4 + step];
  const float f = factors[b * 4 + op];
  unsigned char* px = img + t * 3;
  int r = px[0], g = px[1], bl = px[2];
  if (op == 0) {                 // brightness: blend with black
    r = blend_u8(0, r, f); g = blend_u8(0, g, f); bl = blend_u8(0, bl, f);
  } else if (op == 1) {          // contrast: blend with the solid mean-of-L image
    const int mean = (int)((double)lsum[b] / (double)n + 0.5);
    r = blend_u8(mean, r, f); g = blend_u8(mean, g, f); bl = blend_u8(mean, bl, f);
  } else if (op == 2) {          // saturation: blend with L
    const int l = rgb2l_i(r, g, bl);
    r = blend_u8(l, r, f); g = blend_u8(l, g, f); bl = blend_u8(l, bl, f);
  } else {                       // hue
    hue_u8(r, g, bl, (int)((double)f * 255.0) & 0xff);      // torchvision: np.uint8(hue_factor * 255), a double product
  }
  px[0] = (unsigned char)r; px[1] = (unsigned char)g; px[2] = (unsigned char)bl;
}

__global__ void grayscale_kernel(unsigned char* __restrict__ img, const unsigned char* __restrict__ gray, int B, int S) {
  const long n = (long)S * S;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)B * n || !gray[t / n]) return;
  unsigned char* px = img + t * 3;
  const unsigned char l = (unsigned char)rgb2l_i(px[0], px[1], px[2]);
  px[0] = px[1] = px[2] = l;
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

extern "C" int clipa_color_jitter_u8(void* img, const uint8_t* apply, const int32_t* order, const float* factors,
                                     const uint8_t* gray_flags, int64_t B, int64_t S, void* workspace, int64_t workspace_bytes,
                                     void* stream) {
  if (B <= 0) return CLIPA_OK;
  if (S <= 0 || S > 4096 || B * S * S / 256 + 1 > 0x7fffffffL || B > 65535) { clipa_set_error("color_jitter: bad sizes"); return CLIPA_ERR_ARG; }
  if (order && (!factors || !workspace || workspace_bytes < B * 8)) { clipa_set_error("color_jitter: needs factors and a workspace of 8 bytes per sample"); return CLIPA_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  const unsigned nblk = (unsigned)((B * S * S + 255) / 256);
  if (order) {
    for (int step = 0; step < 4; ++step) {
      (void)hipMemsetAsync(workspace, 0, B * 8, st);
      hipLaunchKernelGGL(jitter_lsum_kernel, dim3(32, (unsigned)B), dim3(256), 0, st, (const unsigned char*)img, apply, order,
                         (unsigned long long*)workspace, (int)B, (int)S, step);
      hipLaunchKernelGGL(jitter_step_kernel, dim3(nblk), dim3(256), 0, st, (unsigned char*)img, apply, order, factors,
                         (const unsigned long long*)workspace, (int)B, (int)S, step);
      if (int rc = clipa_check_launch("color_jitter")) return rc;
    }
  }
  if (gray_flags) {
    hipLaunchKernelGGL(grayscale_kernel, dim3(nblk), dim3(256), 0, st, (unsigned char*)img, gray_flags, (int)B, (int)S);
    if (int rc = clipa_check_launch("grayscale")) return rc;
  }
  return CLIPA_OK;
}
