// Softmax over a strided axis: x viewed as [outer][C][inner] (inner contiguous), softmax along C.
// Used where the reference calls F.softmax(dim=1) / softmax(dim=2) on depth-major volumes (the 48x160x192 matching
// distribution `pred3`, ViewTransformerLSSVoxel.py:255-259, and the BRI confidence, attention.py:66-68): ATen runs
// those through its generic "spatial" soft-max (142 us for a 5.9 MB tensor).  Consecutive threads read consecutive
// addresses at every step along C; long axes are sliced across the threads of a block.
#include "common.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

// Block = COLS adjacent columns x SL slices of the softmax axis (COLS * SL = 256): a thread walks C / SL elements per
// pass, the slices are folded through LDS.  With one thread per column the 48x160 maps gave 7680 threads (120 waves on a
// 1024-SIMD chip) walking 192 strided elements each: 100 us for a 5.9 MB tensor; sliced: ~8x the parallelism per column.
template <int SL>
__global__ void __launch_bounds__(256)
softmax_axis_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long outer, int C, long inner) {
  constexpr int COLS = 256 / SL;
  __shared__ float red[SL][COLS];
  const int col = threadIdx.x % COLS, sl = threadIdx.x / COLS;
  const long i = (long)blockIdx.x * COLS + col;
  const bool ok = i < outer * inner;
  const long o = ok ? i / inner : 0, r = ok ? i - o * inner : 0;
  const float* px = x + o * C * inner + r;
  float* py = y + o * C * inner + r;
  const int c0 = (int)((long)C * sl / SL), c1 = (int)((long)C * (sl + 1) / SL);
  float m = -INFINITY;
  if (ok)
    for (int c = c0; c < c1; ++c) m = fmaxf(m, px[c * inner]);
  red[sl][col] = m;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < SL; ++k) m = fmaxf(m, red[k][col]);
  __syncthreads();
  float s = 0.0f;
  if (ok)
    for (int c = c0; c < c1; ++c) s += __expf(px[c * inner] - m);
  red[sl][col] = s;
  __syncthreads();
  float tot = 0.0f;
#pragma unroll
  for (int k = 0; k < SL; ++k) tot += red[k][col];          // fixed order: every slice of a column gets the same sum
  const float inv = 1.0f / tot;
  if (ok)
    for (int c = c0; c < c1; ++c) py[c * inner] = __expf(px[c * inner] - m) * inv;
}

template <int SL>
__global__ void __launch_bounds__(256)
softmax_axis_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gy, float* __restrict__ gx, long outer,
                        int C, long inner) {
  constexpr int COLS = 256 / SL;
  __shared__ float red[SL][COLS];
  const int col = threadIdx.x % COLS, sl = threadIdx.x / COLS;
  const long i = (long)blockIdx.x * COLS + col;
  const bool ok = i < outer * inner;
  const long o = ok ? i / inner : 0, r = ok ? i - o * inner : 0;
  const long base = o * C * inner + r;
  const int c0 = (int)((long)C * sl / SL), c1 = (int)((long)C * (sl + 1) / SL);
  float dot = 0.0f;
  if (ok)
    for (int c = c0; c < c1; ++c) dot += y[base + c * inner] * gy[base + c * inner];
  red[sl][col] = dot;
  __syncthreads();
  float tot = 0.0f;
#pragma unroll
  for (int k = 0; k < SL; ++k) tot += red[k][col];
  if (ok)
    for (int c = c0; c < c1; ++c) gx[base + c * inner] = y[base + c * inner] * (gy[base + c * inner] - tot);
}

// Softmax along the INNERMOST axis of long rows (BRI attention, attention.py:66-68 on the [T, T] energy with T = 7680):
// one workgroup per row, the row lives in registers between the passes (NV float4 per thread), so a row is read once and
// written once -- 2 tensor passes forward, 3 backward -- where the tensor-op formulation of the backward
// (att * (g - sum(g * att))) made 4 launches and 9 passes over the 236 MB matrices.  y may alias x (gx may alias gy): a
// workgroup only touches its own row and reads it completely before the first store.
template <int NV>
__global__ void __launch_bounds__(256)
softmax_row_fwd_kernel(const float* x, float* y, int C4) {              // y may alias x: no __restrict__
  __shared__ float red[4];
  const v4f* px = reinterpret_cast<const v4f*>(x) + (long)blockIdx.x * C4;
  v4f* py = reinterpret_cast<v4f*>(y) + (long)blockIdx.x * C4;
  v4f v[NV];
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = threadIdx.x + k * 256;
    if (i < C4) {
      v[k] = px[i];
      m = fmaxf(fmaxf(m, fmaxf(v[k][0], v[k][1])), fmaxf(v[k][2], v[k][3]));
    }
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = threadIdx.x + k * 256;
    if (i < C4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[k][j] = __expf(v[k][j] - m); s += v[k][j]; }
    }
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float inv = 1.0f / (((red[0] + red[1]) + red[2]) + red[3]);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = threadIdx.x + k * 256;
    if (i < C4) py[i] = v[k] * inv;
  }
}

template <int NV>
__global__ void __launch_bounds__(256)
softmax_row_bwd_kernel(const float* __restrict__ y, const float* gy, float* gx, int C4) {   // gx may alias gy
  __shared__ float red[4];
  const v4f* py = reinterpret_cast<const v4f*>(y) + (long)blockIdx.x * C4;
  const v4f* pg = reinterpret_cast<const v4f*>(gy) + (long)blockIdx.x * C4;
  v4f* po = reinterpret_cast<v4f*>(gx) + (long)blockIdx.x * C4;
  v4f a[NV], g[NV];
  float dot = 0.0f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = threadIdx.x + k * 256;
    if (i < C4) {
      a[k] = py[i];
      g[k] = pg[i];
      dot += (a[k][0] * g[k][0] + a[k][1] * g[k][1]) + (a[k][2] * g[k][2] + a[k][3] * g[k][3]);
    }
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) dot += __shfl_xor(dot, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
  __syncthreads();
  const float tot = ((red[0] + red[1]) + red[2]) + red[3];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = threadIdx.x + k * 256;
    if (i < C4) po[i] = a[k] * (g[k] - tot);
  }
}

}  // namespace

extern "C" {

int ssbev_softmax_axis_fwd(const float* x, float* y, int64_t outer, int C, int64_t inner, ssbev_stream_t stream) {
  if (!x || !y || outer <= 0 || C <= 0 || inner <= 0) return SSBEV_EINVAL;
  if (C >= 64 && inner >= 32)
    hipLaunchKernelGGL(softmax_axis_fwd_kernel<8>, dim3(cdiv((size_t)(outer * inner), 32)), dim3(256), 0, as_stream(stream),
                       x, y, (long)outer, C, (long)inner);
  else
    hipLaunchKernelGGL(softmax_axis_fwd_kernel<1>, dim3(cdiv((size_t)(outer * inner), 256)), dim3(256), 0, as_stream(stream),
                       x, y, (long)outer, C, (long)inner);
  return ssbev_launch_status();
}

int ssbev_softmax_axis_bwd(const float* y, const float* gy, float* gx, int64_t outer, int C, int64_t inner,
                           ssbev_stream_t stream) {
  if (!y || !gy || !gx || outer <= 0 || C <= 0 || inner <= 0) return SSBEV_EINVAL;
  if (C >= 64 && inner >= 32)
    hipLaunchKernelGGL(softmax_axis_bwd_kernel<8>, dim3(cdiv((size_t)(outer * inner), 32)), dim3(256), 0, as_stream(stream),
                       y, gy, gx, (long)outer, C, (long)inner);
  else
    hipLaunchKernelGGL(softmax_axis_bwd_kernel<1>, dim3(cdiv((size_t)(outer * inner), 256)), dim3(256), 0, as_stream(stream),
                       y, gy, gx, (long)outer, C, (long)inner);
  return ssbev_launch_status();
}

// Rows of C contiguous floats (C % 4 == 0, C <= 8192); y may alias x, gx may alias gy.
int ssbev_softmax_rows_fwd(const float* x, float* y, int64_t rows, int C, ssbev_stream_t stream) {
  if (!x || !y || rows <= 0 || C <= 0 || (C & 3) || C > 8192 || rows > 0x7fffffffLL) return SSBEV_EINVAL;
  const int C4 = C / 4;
  if (C4 <= 1024)
    hipLaunchKernelGGL(softmax_row_fwd_kernel<4>, dim3((unsigned)rows), dim3(256), 0, as_stream(stream), x, y, C4);
  else
    hipLaunchKernelGGL(softmax_row_fwd_kernel<8>, dim3((unsigned)rows), dim3(256), 0, as_stream(stream), x, y, C4);
  return ssbev_launch_status();
}

int ssbev_softmax_rows_bwd(const float* y, const float* gy, float* gx, int64_t rows, int C, ssbev_stream_t stream) {
  if (!y || !gy || !gx || rows <= 0 || C <= 0 || (C & 3) || C > 8192 || rows > 0x7fffffffLL) return SSBEV_EINVAL;
  const int C4 = C / 4;
  if (C4 <= 1024)
    hipLaunchKernelGGL(softmax_row_bwd_kernel<4>, dim3((unsigned)rows), dim3(256), 0, as_stream(stream), y, gy, gx, C4);
  else
    hipLaunchKernelGGL(softmax_row_bwd_kernel<8>, dim3((unsigned)rows), dim3(256), 0, as_stream(stream), y, gy, gx, C4);
  return ssbev_launch_status();
}

}  // extern "C"
