// Input-side image operators (LoadMultiViewImageFromFiles_SemanticKitti, datasets/pipelines/
// loading_semkitti.py:76-302): Pillow's antialiased resize (`img.resize(resize_dims)`: bicubic, 8 bits per channel) and
// crop + flip + mmcv `imnormalize` + HWC -> CHW, on the GPU.  The resize is BYTE-EXACT with Pillow: the reference's
// pixels are whatever libImaging's fixed-point convolution produces (Resample.c: coefficients normalised in double,
// converted to 22-bit fixed point with round-half-away, accumulator seeded with 1 << 21, shifted and clamped to 0..255;
// horizontal pass to an 8-bit intermediate, then vertical), so the kernels consume the host-computed integer
// coefficient tables and do the same integer arithmetic.  Pure byte work, HBM-bound; one thread per output byte triple.
#include "common.h"

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;

__device__ __forceinline__ unsigned char clip8(int v) {
  v >>= kPrecisionBits;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// dst[y][x][c] = clip8(half + sum_k src[y][xmin + k][c] * kk[x][k]),   src [H][Ws][C], dst [H][Wd][C]
__global__ void __launch_bounds__(256)
resample_h_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, const int* __restrict__ kk,
                  const int* __restrict__ bounds, int ksize, int H, int Ws, int Wd, int C) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)H * Wd) return;
  const int x = (int)(i % Wd), y = (int)(i / Wd);
  const int xmin = bounds[2 * x], n = bounds[2 * x + 1];
  const int* k = kk + (long)x * ksize;
  const unsigned char* row = src + ((long)y * Ws + xmin) * C;
  for (int c = 0; c < C; ++c) {
    int ss = 1 << (kPrecisionBits - 1);
    for (int j = 0; j < n; ++j) ss += (int)row[j * C + c] * k[j];
    dst[i * C + c] = clip8(ss);
  }
}

// dst[y][x][c] = clip8(half + sum_k src[ymin + k][x][c] * kk[y][k]),   src [Hs][W][C], dst [Hd][W][C]
__global__ void __launch_bounds__(256)
resample_v_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, const int* __restrict__ kk,
                  const int* __restrict__ bounds, int ksize, int Hs, int Hd, int W, int C) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)Hd * W) return;
  const int x = (int)(i % W), y = (int)(i / W);
  const int ymin = bounds[2 * y], n = bounds[2 * y + 1];
  const int* k = kk + (long)y * ksize;
  const unsigned char* col = src + ((long)ymin * W + x) * C;
  for (int c = 0; c < C; ++c) {
    int ss = 1 << (kPrecisionBits - 1);
    for (int j = 0; j < n; ++j) ss += (int)col[(long)j * W * C + c] * k[j];
    dst[i * C + c] = clip8(ss);
  }
}

struct NormParams { float mean[3], stdinv[3]; int x0, y0, flip, swap_rb; };

// out[c][y][x] = (src[y0 + y][x0 + (flip ? w - 1 - x : x)][swap ? 2 - c : c] - mean[c]) * stdinv[c]
__global__ void __launch_bounds__(256)
crop_normalize_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, NormParams p, int Hs, int Ws, int h, int w) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)h * w) return;
  const int x = (int)(i % w), y = (int)(i / w);
  const int sx = p.x0 + (p.flip ? w - 1 - x : x), sy = p.y0 + y;
  const bool in = sx >= 0 && sx < Ws && sy >= 0 && sy < Hs;            // PIL's crop pads with zeros outside the image
  const unsigned char* px = src + ((long)sy * Ws + sx) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = in ? (float)px[p.swap_rb ? 2 - c : c] : 0.0f;
    dst[(long)c * h * w + i] = (v - p.mean[c]) * p.stdinv[c];
  }
}

}  // namespace

extern "C" {

int ssbev_resize_pil_u8(const uint8_t* src, int Hs, int Ws, int C, const int32_t* kk_h, const int32_t* bounds_h, int ksize_h,
                        const int32_t* kk_v, const int32_t* bounds_v, int ksize_v, uint8_t* tmp, uint8_t* dst, int Hd, int Wd,
                        ssbev_stream_t stream) {
  if (!src || !dst || Hs <= 0 || Ws <= 0 || Hd <= 0 || Wd <= 0 || C <= 0 || C > 4) return SSBEV_EINVAL;
  const bool need_h = Wd != Ws, need_v = Hd != Hs;
  if ((need_h && (!kk_h || !bounds_h || ksize_h <= 0)) || (need_v && (!kk_v || !bounds_v || ksize_v <= 0))) return SSBEV_EINVAL;
  if (need_h && need_v && !tmp) return SSBEV_EINVAL;
  hipStream_t st = as_stream(stream);
  const uint8_t* cur = src;
  if (need_h) {
    uint8_t* out = need_v ? tmp : dst;
    hipLaunchKernelGGL(resample_h_kernel, dim3(cdiv((size_t)Hs * Wd, 256)), dim3(256), 0, st, cur, out, kk_h, bounds_h, ksize_h,
                       Hs, Ws, Wd, C);
    cur = out;
  }
  if (need_v)
    hipLaunchKernelGGL(resample_v_kernel, dim3(cdiv((size_t)Hd * Wd, 256)), dim3(256), 0, st, cur, dst, kk_v, bounds_v, ksize_v,
                       Hs, Hd, Wd, C);
  if (!need_h && !need_v && hipMemcpyAsync(dst, src, (size_t)Hs * Ws * C, hipMemcpyDeviceToDevice, st) != hipSuccess)
    return SSBEV_ELAUNCH;
  return ssbev_launch_status();
}

int ssbev_crop_normalize_u8(const uint8_t* src, int Hs, int Ws, float* dst, int x0, int y0, int w, int h, int flip,
                            const float* mean, const float* stdinv, int swap_rb, ssbev_stream_t stream) {
  if (!src || !dst || !mean || !stdinv || Hs <= 0 || Ws <= 0 || w <= 0 || h <= 0) return SSBEV_EINVAL;
  NormParams p;
  for (int c = 0; c < 3; ++c) { p.mean[c] = mean[c]; p.stdinv[c] = stdinv[c]; }       // host pointers (3 floats each)
  p.x0 = x0; p.y0 = y0; p.flip = flip ? 1 : 0; p.swap_rb = swap_rb ? 1 : 0;
  hipLaunchKernelGGL(crop_normalize_kernel, dim3(cdiv((size_t)h * w, 256)), dim3(256), 0, as_stream(stream), src, dst, p, Hs, Ws,
                     h, w);
  return ssbev_launch_status();
}

}  // extern "C"
