// Fused AdamW + global-norm gradient clipping over ONE flat fp32 parameter buffer, gfx950.
// Mirrors the reference's optimiser recipe (projects/configs/.../stereoscene.py:203-209: AdamW lr 1e-4,
// weight_decay 0.01, grad_clip max_norm 5 / norm_type 2) as two streaming kernels over the flat buffers
// that the data-parallel exchange already uses (stereoscene_amd/dp.py): HBM-bound, 7 floats of traffic
// per parameter instead of ~20 elementwise launches per tensor x 341 tensors.
#include "common.h"

namespace {

__global__ void __launch_bounds__(256)
sumsq_partial_kernel(const float* __restrict__ g, long n, double* __restrict__ partial) {
  __shared__ double red[256];
  double acc = 0.0;
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  if (blockIdx.x == 0)
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) acc += (double)g[i] * g[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void sumsq_final_kernel(const double* __restrict__ partial, int nblocks, float* __restrict__ norm_out) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) acc += partial[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) norm_out[0] = (float)sqrt(red[0]);
}

// torch.optim.AdamW semantics (decoupled weight decay), gradient pre-scaled by the clip coefficient
// min(1, max_norm / (norm + 1e-6)) read from device memory (no host round trip).
__global__ void __launch_bounds__(256)
adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
             float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2_sqrt, float max_norm,
             const float* __restrict__ norm) {
  float coef = 1.0f;
  if (max_norm > 0.0f && norm) coef = fminf(1.0f, max_norm / (norm[0] + 1e-6f));
  const float step = lr / bc1;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gi = g[i] * coef;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    pi -= step * mi / (sqrtf(vi) / bc2_sqrt + eps);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

constexpr int NORM_BLOCKS = 1024;

}  // namespace

extern "C" {

size_t ssbev_grad_norm_workspace(void) { return NORM_BLOCKS * sizeof(double); }

int ssbev_grad_norm(const float* g, int64_t n, float* norm_out, void* ws, size_t ws_bytes, ssbev_stream_t stream) {
  if (!g || n <= 0 || !norm_out || !ws) return SSBEV_EINVAL;
  if (ws_bytes < ssbev_grad_norm_workspace()) return SSBEV_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(NORM_BLOCKS), dim3(256), 0, st, g, (long)n, static_cast<double*>(ws));
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, st, static_cast<const double*>(ws), NORM_BLOCKS, norm_out);
  return ssbev_launch_status();
}

int ssbev_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, const ssbev_adamw_cfg* c,
                     const float* grad_norm, ssbev_stream_t stream) {
  if (!p || !g || !m || !v || n <= 0 || !c || c->step < 1) return SSBEV_EINVAL;
  const float bc1 = 1.0f - powf(c->beta1, (float)c->step);
  const float bc2 = 1.0f - powf(c->beta2, (float)c->step);
  hipLaunchKernelGGL(adamw_kernel, dim3(2048), dim3(256), 0, as_stream(stream), p, g, m, v, (long)n, c->lr, c->beta1,
                     c->beta2, c->eps, c->weight_decay, bc1, sqrtf(bc2), c->max_grad_norm, grad_norm);
  return ssbev_launch_status();
}

}  // extern "C"
