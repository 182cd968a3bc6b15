// Stereo plane-sweep cost volume (SURVEY a2+a3), gfx950: group-wise correlation fused with the
// disparity -> metric-depth bilinear resample of the reference's `warp` (VT:104-156).
//
//   cost[g', d](w) = 1/cpg * sum_{c in g'} L[w, c] * R[w - d, c]          (0 for w < d)
//   vol[k, g](w)   = sum_{ty in {y0,y0+1}} sum_{tx in {x0,x0+1}} wy*wx * cost[ty, tx](w)
// with (x0, wx) from ix(k) = unnormalise(2*(calib/(4*down)/(k+1))/(D-1) - 1) and (y0, wy) from
// iy(g) = unnormalise(2*g/(G-1) - 1): exactly grid_sample's arithmetic, both conventions.
//
// MI355X mapping: one workgroup per (batch, image row, depth-plane chunk).  The right-view row
// (W x C fp32, 40 KB at 160x64) is staged ONCE into LDS with a +4-float row pad (conflict-free
// 8-byte reads); the left-view operands live in registers for the whole chunk; every plane is
// written as W*G contiguous floats (channels-last volume) => the kernel is a pure streaming
// write of the 189 MB volume (D=192), its only HBM-sized traffic.  The 2-channel group
// reduction happens inside one lane (no cross-lane traffic needed for cpg=2..8).
#include "common.h"

#include <algorithm>

namespace {

constexpr int KCHUNK = 16;

struct XTap { int x0; float w0, w1; };

// grid_sample coordinate un-normalisation (ATen grid_sampler_unnormalize)
__device__ __forceinline__ float unnormalise(float coord, int size, int align_corners) {
  return align_corners ? __fmul_rn(__fdiv_rn(__fadd_rn(coord, 1.0f), 2.0f), (float)(size - 1))
                       : __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(coord, 1.0f), (float)size), 1.0f), 2.0f);
}

__device__ __forceinline__ XTap depth_tap(float calib, int k, int D, float down, int align_corners) {
  const float xx = __fdiv_rn(__fdiv_rn(calib, __fmul_rn(down, 4.0f)), (float)(k + 1));
  const float gx = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, xx), (float)max(D - 1, 1)), 1.0f);
  const float ix = unnormalise(gx, D, align_corners);
  const float fl = floorf(ix);
  XTap t;
  // clamp before the int cast: only taps inside [-1, D] can ever contribute
  t.x0 = (int)fminf(fmaxf(fl, -2.0f), (float)D + 1.0f);
  t.w1 = __fsub_rn(ix, fl);
  t.w0 = __fsub_rn(1.0f, t.w1);
  if (!(ix == ix)) { t.x0 = -2; t.w0 = t.w1 = 0.0f; }
  return t;
}

__device__ __forceinline__ void group_tap(int g, int G, int align_corners, int* y0, float* w0, float* w1) {
  const float gy = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, (float)g), (float)max(G - 1, 1)), 1.0f);
  const float iy = unnormalise(gy, G, align_corners);
  const float fl = floorf(iy);
  *y0 = (int)fl;
  *w1 = __fsub_rn(iy, fl);
  *w0 = __fsub_rn(1.0f, *w1);
}

// stage one [W, C] row into LDS with padded stride
__device__ __forceinline__ void stage_row(const float* __restrict__ src, float* lds, int W, int C, int stride) {
  const int vec_per_row = C >> 2;
  for (int i = threadIdx.x; i < W * vec_per_row; i += blockDim.x) {
    const int w = i / vec_per_row, v = i - w * vec_per_row;
    const float4 t = reinterpret_cast<const float4*>(src)[i];
    *reinterpret_cast<float4*>(lds + w * stride + 4 * v) = t;
  }
}

template <int CPG>
__global__ void __launch_bounds__(256)
gwc_warp_fwd_kernel(const float* __restrict__ left, const float* __restrict__ right,
                    const float* __restrict__ calib, float* __restrict__ vol, int B, int C, int G, int D, int H,
                    int W, float down, int align_corners) {
  extern __shared__ __align__(16) float lds[];
  const int stride = C + 4;
  float* Rrow = lds;                                   // [W][stride]
  XTap* taps = reinterpret_cast<XTap*>(lds + W * stride);  // [KCHUNK]
  const int bh = blockIdx.x;
  const int b = bh / H, h = bh - b * H;
  const int k_begin = blockIdx.y * KCHUNK;
  const int k_end = min(D, k_begin + KCHUNK);
  const float* Lrow_g = left + ((size_t)b * H + h) * W * C;
  stage_row(right + ((size_t)b * H + h) * W * C, Rrow, W, C, stride);
  if (threadIdx.x < KCHUNK && k_begin + threadIdx.x < D)
    taps[threadIdx.x] = depth_tap(calib[b], k_begin + threadIdx.x, D, down, align_corners);
  __syncthreads();

  const float inv_cpg = 1.0f / (float)CPG;
  for (int item = threadIdx.x; item < W * G; item += blockDim.x) {
    const int w = item / G, g = item - w * G;
    int y0;
    float wy0, wy1;
    group_tap(g, G, align_corners, &y0, &wy0, &wy1);
    const bool ok0 = y0 >= 0 && y0 < G, ok1 = y0 + 1 >= 0 && y0 + 1 < G;
    float l0[CPG], l1[CPG];
#pragma unroll
    for (int c = 0; c < CPG; ++c) {
      l0[c] = ok0 ? Lrow_g[w * C + y0 * CPG + c] : 0.0f;
      l1[c] = ok1 ? Lrow_g[w * C + (y0 + 1) * CPG + c] : 0.0f;
    }
    const int cy0 = ok0 ? y0 * CPG : 0, cy1 = ok1 ? (y0 + 1) * CPG : 0;
    float* dst = vol + ((((size_t)b * D + k_begin) * H + h) * W + w) * G + g;
    const size_t plane = (size_t)H * W * G;
    for (int k = k_begin; k < k_end; ++k) {
      const XTap t = taps[k - k_begin];
      float acc = 0.0f;
#pragma unroll
      for (int tx = 0; tx < 2; ++tx) {
        const int d = t.x0 + tx;
        const float wx = tx ? t.w1 : t.w0;
        if (d >= 0 && d < D && w >= d) {   // w >= d: the reference volume is zero left of the disparity
          const float* r = Rrow + (w - d) * stride;
          float c0 = 0.0f, c1 = 0.0f;
#pragma unroll
          for (int c = 0; c < CPG; ++c) {
            c0 += l0[c] * r[cy0 + c];
            c1 += l1[c] * r[cy1 + c];
          }
          acc += (wy0 * wx) * (c0 * inv_cpg) + (wy1 * wx) * (c1 * inv_cpg);
        }
      }
      *dst = acc;
      dst += plane;
    }
  }
}

// Backward.  For a source group g' the output groups that read it are those g with y0(g) == g'
// (weight wy0(g)) or y0(g)+1 == g' (weight wy1(g)); y0 is monotone in g so there are at most a few.
//   S(k, w)      = sum_{(g, wy)} wy * gvol[k, w, g]
//   gL[w , c]    = 1/cpg * sum_k sum_tx wx * S(k, w)       * R[w - d, c]      d = x0(k)+tx, w >= d
//   gR[w', c]    = 1/cpg * sum_k sum_tx wx * S(k, w' + d)  * L[w' + d, c]     w' + d < W
// One workgroup per (batch, row, 32-pixel tile); each thread owns (pixel, source group) and walks
// all D planes, so every gradient element is produced by exactly one thread (no atomics).
constexpr int MAX_SRC = 4;
constexpr int WTILE = 32;

template <int CPG, bool FOR_LEFT>
__global__ void __launch_bounds__(1024)
gwc_warp_bwd_kernel(const float* __restrict__ gvol, const float* __restrict__ other,
                    const float* __restrict__ calib, float* __restrict__ gout, int B, int C, int G, int D,
                    int H, int W, float down, int align_corners) {
  extern __shared__ __align__(16) float lds[];
  const int stride = C + 4;
  float* Orow = lds;                                         // the OTHER view's row [W][stride]
  XTap* taps = reinterpret_cast<XTap*>(lds + W * stride);    // [D]
  int* src_g = reinterpret_cast<int*>(taps + D);             // [G][MAX_SRC]
  float* src_w = reinterpret_cast<float*>(src_g + G * MAX_SRC);
  int* src_n = reinterpret_cast<int*>(src_w + G * MAX_SRC);  // [G]
  const int bh = blockIdx.x;
  const int b = bh / H, h = bh - b * H;
  const int w_begin = blockIdx.y * WTILE;
  stage_row(other + ((size_t)b * H + h) * W * C, Orow, W, C, stride);
  for (int k = threadIdx.x; k < D; k += blockDim.x) taps[k] = depth_tap(calib[b], k, D, down, align_corners);
  for (int gs = threadIdx.x; gs < G; gs += blockDim.x) {
    int n = 0;
    for (int g = 0; g < G; ++g) {
      int y0;
      float w0, w1;
      group_tap(g, G, align_corners, &y0, &w0, &w1);
      if (y0 == gs && n < MAX_SRC) { src_g[gs * MAX_SRC + n] = g; src_w[gs * MAX_SRC + n] = w0; ++n; }
      if (y0 + 1 == gs && n < MAX_SRC) { src_g[gs * MAX_SRC + n] = g; src_w[gs * MAX_SRC + n] = w1; ++n; }
    }
    src_n[gs] = n;
  }
  __syncthreads();
  const float inv_cpg = 1.0f / (float)CPG;
  const size_t plane = (size_t)H * W * G;
  const float* grow = gvol + (((size_t)b * D) * H + h) * W * G;   // plane 0 of this row
  for (int item = threadIdx.x; item < WTILE * G; item += blockDim.x) {
    const int w = w_begin + item / G, gs = item % G;
    if (w >= W) continue;
    const int n = src_n[gs];
    int sg[MAX_SRC];
    float sw[MAX_SRC];
#pragma unroll
    for (int i = 0; i < MAX_SRC; ++i) {
      sg[i] = i < n ? src_g[gs * MAX_SRC + i] : 0;
      sw[i] = i < n ? src_w[gs * MAX_SRC + i] : 0.0f;
    }
    float acc[CPG];
#pragma unroll
    for (int c = 0; c < CPG; ++c) acc[c] = 0.0f;
#pragma unroll 4
    for (int k = 0; k < D; ++k) {
      const XTap t = taps[k];
      const float* gk = grow + (size_t)k * plane;
      float s_here = 0.0f;
      if (FOR_LEFT) {
#pragma unroll
        for (int i = 0; i < MAX_SRC; ++i)
          if (i < n) s_here += sw[i] * gk[(size_t)w * G + sg[i]];
      }
#pragma unroll
      for (int tx = 0; tx < 2; ++tx) {
        const int d = t.x0 + tx;
        const float wx = tx ? t.w1 : t.w0;
        if (d < 0 || d >= D) continue;
        if (FOR_LEFT) {
          if (w < d) continue;
          const float* r = Orow + (w - d) * stride + gs * CPG;
          const float coef = wx * s_here * inv_cpg;
#pragma unroll
          for (int c = 0; c < CPG; ++c) acc[c] += coef * r[c];
        } else {
          const int ws = w + d;
          if (ws >= W) continue;
          float s = 0.0f;
#pragma unroll
          for (int i = 0; i < MAX_SRC; ++i)
            if (i < n) s += sw[i] * gk[(size_t)ws * G + sg[i]];
          const float* l = Orow + ws * stride + gs * CPG;
          const float coef = wx * s * inv_cpg;
#pragma unroll
          for (int c = 0; c < CPG; ++c) acc[c] += coef * l[c];
        }
      }
    }
    float* dst = gout + (((size_t)b * H + h) * W + w) * C + gs * CPG;
#pragma unroll
    for (int c = 0; c < CPG; ++c) dst[c] = acc[c];
  }
}

bool gwc_dims_ok(const ssbev_gwc_dims* d) {
  if (!d || d->B <= 0 || d->C <= 0 || d->G <= 0 || d->D <= 0 || d->H <= 0 || d->W <= 0) return false;
  if (d->C % d->G != 0 || d->C % 4 != 0 || d->down != 1.0f) return false;
  const int cpg = d->C / d->G;
  return cpg == 1 || cpg == 2 || cpg == 4 || cpg == 8;
}

size_t fwd_lds_bytes(const ssbev_gwc_dims* d) { return (size_t)d->W * (d->C + 4) * 4 + KCHUNK * sizeof(XTap); }
size_t bwd_lds_bytes(const ssbev_gwc_dims* d) {
  return (size_t)d->W * (d->C + 4) * 4 + d->D * sizeof(XTap) + (size_t)d->G * MAX_SRC * 8 + d->G * 4;
}

template <int CPG>
int launch_fwd(const float* l, const float* r, const float* calib, float* vol, const ssbev_gwc_dims* d,
               hipStream_t st) {
  const size_t lds = fwd_lds_bytes(d);
  if (lds > 160 * 1024) return SSBEV_EINVAL;
  auto kern = gwc_warp_fwd_kernel<CPG>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
          hipSuccess)
    return SSBEV_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3(d->B * d->H, cdiv(d->D, KCHUNK)), dim3(256), lds, st, l, r, calib, vol, d->B, d->C,
                     d->G, d->D, d->H, d->W, d->down, d->align_corners);
  return ssbev_launch_status();
}

template <int CPG>
int launch_bwd(const float* gvol, const float* l, const float* r, const float* calib, float* gl, float* gr,
               const ssbev_gwc_dims* d, hipStream_t st) {
  const size_t lds = bwd_lds_bytes(d);
  if (lds > 160 * 1024) return SSBEV_EINVAL;
  auto kl = gwc_warp_bwd_kernel<CPG, true>;
  auto kr = gwc_warp_bwd_kernel<CPG, false>;
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kl), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(kr), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
      return SSBEV_ELAUNCH;
  }
  // one thread per (pixel, source group) of the tile: the 192-plane walk is the only serial loop left
  dim3 grid(d->B * d->H, cdiv(d->W, WTILE)), block(std::min(1024, std::max(256, WTILE * d->G)));
  hipLaunchKernelGGL(kl, grid, block, lds, st, gvol, r, calib, gl, d->B, d->C, d->G, d->D, d->H, d->W, d->down,
                     d->align_corners);
  hipLaunchKernelGGL(kr, grid, block, lds, st, gvol, l, calib, gr, d->B, d->C, d->G, d->D, d->H, d->W, d->down,
                     d->align_corners);
  return ssbev_launch_status();
}

}  // namespace

extern "C" {

int ssbev_gwc_warp_fwd(const float* left, const float* right, const float* calib, float* vol,
                       const ssbev_gwc_dims* d, ssbev_stream_t stream) {
  if (!gwc_dims_ok(d) || !left || !right || !calib || !vol) return SSBEV_EINVAL;
  hipStream_t st = as_stream(stream);
  switch (d->C / d->G) {
    case 1: return launch_fwd<1>(left, right, calib, vol, d, st);
    case 2: return launch_fwd<2>(left, right, calib, vol, d, st);
    case 4: return launch_fwd<4>(left, right, calib, vol, d, st);
    default: return launch_fwd<8>(left, right, calib, vol, d, st);
  }
}

int ssbev_gwc_warp_bwd(const float* grad_vol, const float* left, const float* right,
                       const float* calib, float* grad_left, float* grad_right,
                       const ssbev_gwc_dims* d, ssbev_stream_t stream) {
  if (!gwc_dims_ok(d) || !grad_vol || !left || !right || !calib || !grad_left || !grad_right) return SSBEV_EINVAL;
  hipStream_t st = as_stream(stream);
  switch (d->C / d->G) {
    case 1: return launch_bwd<1>(grad_vol, left, right, calib, grad_left, grad_right, d, st);
    case 2: return launch_bwd<2>(grad_vol, left, right, calib, grad_left, grad_right, d, st);
    case 4: return launch_bwd<4>(grad_vol, left, right, calib, grad_left, grad_right, d, st);
    default: return launch_bwd<8>(grad_vol, left, right, calib, grad_left, grad_right, d, st);
  }
}

}  // extern "C"
