// Softmax over a strided axis: x viewed as [outer][C][inner] (inner contiguous), softmax along C.
// Used where the reference calls F.softmax(dim=1) / softmax(dim=2) on depth-major volumes (the 48x160x192 matching
// distribution `pred3`, ViewTransformerLSSVoxel.py:255-259, and the BRI confidence, attention.py:66-68): ATen runs
// those through its generic "spatial" soft-max (142 us for a 5.9 MB tensor).  Consecutive threads read consecutive
// addresses at every step along C; long axes are sliced across the threads of a block.
#include "common.h"

namespace {

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

}  // extern "C"
