// LiDAR -> sparse depth map / image-view segmentation (CreateDepthFromLiDAR, the producer of the
// `gt_depths` slot of img_inputs that the depth loss a16 consumes; datasets/pipelines/occ_to_depth.py:216-303).
// Upstream projects every point on the CPU, sorts the visible ones by depth (descending) and lets an index_put with
// "last write wins" keep the nearest point per pixel.  Here: one thread per point projects it with the same operation
// order (built with -ffp-contract=off), and the nearest point per pixel is a 64-bit atomicMin on
// (depth bits << 32 | point index) -- for positive floats the IEEE bit pattern is monotone, so the minimum is the
// nearest point (lowest index among exact ties; upstream's unstable argsort leaves ties unspecified).  Integer /
// bit-pattern work: HBM-bound and tiny (~125 k points, 0.5 M pixels).
#include "common.h"

namespace {

struct LidarCam {
  float inv_rot[9];     // rots^-1, row major
  float trans[3];
  float K[16];          // intrins, 4 x 4 row major
  float post_rot[4];    // post_rots[:2, :2]
  float post_trans[2];
};

__global__ void __launch_bounds__(256)
lidar_project_kernel(const float* __restrict__ pts, LidarCam c, float* __restrict__ uvd, unsigned char* __restrict__ valid,
                     unsigned long long* __restrict__ zbuf, int N, int H, int W) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  // lidar -> camera: R^-1 (p - t)                       (project_points, occ_to_depth.py:218-222)
  const float x = pts[i * 3 + 0] - c.trans[0], y = pts[i * 3 + 1] - c.trans[1], z = pts[i * 3 + 2] - c.trans[2];
  const float cx = (c.inv_rot[0] * x + c.inv_rot[1] * y) + c.inv_rot[2] * z;
  const float cy = (c.inv_rot[3] * x + c.inv_rot[4] * y) + c.inv_rot[5] * z;
  const float cz = (c.inv_rot[6] * x + c.inv_rot[7] * y) + c.inv_rot[8] * z;
  // camera -> raw pixel: K [p; 1]                        (:225-228)
  const float px = ((c.K[0] * cx + c.K[1] * cy) + c.K[2] * cz) + c.K[3];
  const float py = ((c.K[4] * cx + c.K[5] * cy) + c.K[6] * cz) + c.K[7];
  const float d = ((c.K[8] * cx + c.K[9] * cy) + c.K[10] * cz) + c.K[11];
  const float u0 = px / d, v0 = py / d;
  // raw pixel -> augmented pixel                         (:231-233)
  const float u = (c.post_rot[0] * u0 + c.post_rot[1] * v0) + c.post_trans[0];
  const float v = (c.post_rot[2] * u0 + c.post_rot[3] * v0) + c.post_trans[1];
  uvd[i * 3 + 0] = u; uvd[i * 3 + 1] = v; uvd[i * 3 + 2] = d;
  const bool ok = u >= 0.0f && v >= 0.0f && u <= (float)(W - 1) && v <= (float)(H - 1) && d > 0.0f;    // :252-256
  valid[i] = ok ? 1 : 0;
  if (ok) {
    const int col = (int)rintf(u), row = (int)rintf(v);          // torch.round: half to even
    const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)i;
    atomicMin(&zbuf[(size_t)row * W + col], key);
  }
}

__global__ void __launch_bounds__(256)
lidar_resolve_kernel(const unsigned long long* __restrict__ zbuf, const float* __restrict__ labels, float* __restrict__ depth,
                     float* __restrict__ seg, int HW) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= HW) return;
  const unsigned long long key = zbuf[i];
  const bool hit = key != ~0ull;
  depth[i] = hit ? __uint_as_float((unsigned)(key >> 32)) : 0.0f;
  if (seg) seg[i] = (hit && labels) ? labels[(unsigned)(key & 0xffffffffu)] : 0.0f;
}

// ---------------------------------------------------------------- depth BCE loss (VT:349-416)
// get_downsampled_gt_depth + get_depth_loss as three launches instead of ~35 ATen ops: per feature pixel the nearest LiDAR depth of
// its ds x ds block (zeros = no return), its bin ((g - (d0 - dd / 2)) / dd, fp32 subtract and IEEE divide as in the tensor
// expression), the one-hot row it stands for, and the binary cross entropy of the predicted depth distribution against it,
// summed over the rows that HAVE a return and divided by their number.  Elementwise terms are ATen's (log / log1p clamped at
// -100, backward (p - y) / max((1 - p) p, 1e-12)); the sum is taken in double.
struct DepthBce { int BN, D, fH, fW, ds; float c0, dd; };   // c0 = d0 - dd / 2

__device__ __forceinline__ int depth_bin(const float* __restrict__ gt, const DepthBce& g, int bn, int fh, int fw) {
  const int H = g.fH * g.ds, W = g.fW * g.ds;
  float m = 1e5f;
  for (int i = 0; i < g.ds; ++i)
    for (int j = 0; j < g.ds; ++j) {
      const float v = gt[((size_t)bn * H + fh * g.ds + i) * W + fw * g.ds + j];
      m = fminf(m, v == 0.0f ? 1e5f : v);
    }
  const float b = __fdiv_rn(__fsub_rn(m, g.c0), g.dd);
  return (b < (float)(g.D + 1) && b >= 0.0f) ? (int)b : 0;   // class 0 = "no label" (dropped by the [:, 1:] slice)
}

constexpr int BCE_DCH = 8;       // depth chunks per pixel block (parallelism: 7680 pixels alone are 30 workgroups)

__global__ void __launch_bounds__(256)
depth_bce_fwd_kernel(const float* __restrict__ gt, const float* __restrict__ pred, DepthBce g, int* __restrict__ label,
                     double* __restrict__ partial) {
  __shared__ double red[256];
  const int HW = g.fH * g.fW, npix = g.BN * HW;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  double acc = 0.0, cnt = 0.0;
  if (pix < npix) {
    const int bn = pix / HW, r = pix - bn * HW;
    const int L = depth_bin(gt, g, bn, r / g.fW, r % g.fW);
    if (blockIdx.y == 0) { label[pix] = L; cnt = L >= 1 ? 1.0 : 0.0; }
    if (L >= 1) {
      const int per = (g.D + BCE_DCH - 1) / BCE_DCH, d0 = blockIdx.y * per, d1 = min(g.D, d0 + per);
      for (int d = d0; d < d1; ++d) {
        const float p = pred[((size_t)bn * g.D + d) * HW + r];
        const float y = d == L - 1 ? 1.0f : 0.0f;
        const float lp = fmaxf(logf(p), -100.0f), lq = fmaxf(log1pf(-p), -100.0f);
        acc += (double)((y - 1.0f) * lq - y * lp);
      }
    }
  }
  // fixed-order block sums of (bce, count)
  for (int pass = 0; pass < 2; ++pass) {
    red[threadIdx.x] = pass ? cnt : acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + pass] = red[0];
    __syncthreads();
  }
}

// loss = weight * float(sum bce) / max(float(rows with a return), 1); out[0] = loss, out[1] = the divisor
__global__ void depth_bce_final_kernel(const double* __restrict__ partial, int n, float weight, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s = 0.0, c = 0.0;
  for (int i = 0; i < n; ++i) { s += partial[2 * i]; c += partial[2 * i + 1]; }
  const float den = fmaxf((float)c, 1.0f);
  out[0] = weight * ((float)s / den);
  out[1] = den;
}

__global__ void __launch_bounds__(256)
depth_bce_bwd_kernel(const float* __restrict__ pred, const int* __restrict__ label, const float* __restrict__ gloss,
                     const float* __restrict__ out, float weight, DepthBce g, float* __restrict__ gpred) {
  const int HW = g.fH * g.fW;
  const size_t total = (size_t)g.BN * g.D * HW;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int r = (int)(i % HW), d = (int)((i / HW) % g.D), bn = (int)(i / ((size_t)HW * g.D));
  const int L = label[bn * HW + r];
  float v = 0.0f;
  if (L >= 1) {
    const float p = pred[i], y = d == L - 1 ? 1.0f : 0.0f;
    const float coef = (gloss[0] * weight) / out[1];
    v = coef * (p - y) / fmaxf((1.0f - p) * p, 1e-12f);
  }
  gpred[i] = v;
}

}  // namespace

extern "C" {

size_t ssbev_depth_bce_workspace(int BN, int fH, int fW) {
  if (BN <= 0 || fH <= 0 || fW <= 0) return 0;
  const size_t npix = (size_t)BN * fH * fW, nb = (npix + 255) / 256;
  return npix * sizeof(int) + nb * BCE_DCH * 2 * sizeof(double) + 256;
}

/* see include/ssbev.h */
int ssbev_depth_bce_fwd(const float* gt_depths, const float* depth_pred, float* out2, int BN, int D, int fH, int fW, int ds,
                        float c0, float dd, float weight, void* ws, size_t ws_bytes, ssbev_stream_t stream) {
  if (!gt_depths || !depth_pred || !out2 || !ws || BN <= 0 || D <= 0 || fH <= 0 || fW <= 0 || ds <= 0) return SSBEV_EINVAL;
  if (ws_bytes < ssbev_depth_bce_workspace(BN, fH, fW)) return SSBEV_EWORKSPACE;
  const size_t npix = (size_t)BN * fH * fW, nb = (npix + 255) / 256;
  int* label = static_cast<int*>(ws);
  double* partial = reinterpret_cast<double*>(static_cast<char*>(ws) + ((npix * sizeof(int) + 255) & ~(size_t)255));
  DepthBce g{BN, D, fH, fW, ds, c0, dd};
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(depth_bce_fwd_kernel, dim3((unsigned)nb, BCE_DCH), dim3(256), 0, st, gt_depths, depth_pred, g, label, partial);
  hipLaunchKernelGGL(depth_bce_final_kernel, dim3(1), dim3(64), 0, st, partial, (int)(nb * BCE_DCH), weight, out2);
  return ssbev_launch_status();
}

int ssbev_depth_bce_bwd(const float* depth_pred, const float* grad_loss, const float* out2, float* grad_pred, int BN, int D, int fH,
                        int fW, int ds, float c0, float dd, float weight, const void* ws, ssbev_stream_t stream) {
  if (!depth_pred || !grad_loss || !out2 || !grad_pred || !ws || BN <= 0 || D <= 0 || fH <= 0 || fW <= 0) return SSBEV_EINVAL;
  DepthBce g{BN, D, fH, fW, ds, c0, dd};
  const size_t total = (size_t)BN * D * fH * fW;
  hipLaunchKernelGGL(depth_bce_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), depth_pred,
                     static_cast<const int*>(ws), grad_loss, out2, weight, g, grad_pred);
  return ssbev_launch_status();
}

size_t ssbev_lidar_depth_workspace(int H, int W) { return (H > 0 && W > 0) ? (size_t)H * W * sizeof(unsigned long long) : 0; }

int ssbev_lidar_depth_map(const float* points, int n_points, const float* cam, const float* labels, float* uvd,
                          unsigned char* valid, float* depth, float* seg, int H, int W, void* ws, size_t ws_bytes,
                          ssbev_stream_t stream) {
  if (n_points < 0 || H <= 0 || W <= 0 || !cam || !depth || !ws || (n_points && (!points || !uvd || !valid))) return SSBEV_EINVAL;
  if (ws_bytes < ssbev_lidar_depth_workspace(H, W)) return SSBEV_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  LidarCam c;
  hipError_t e = hipMemcpyAsync(&c, cam, sizeof(LidarCam), hipMemcpyDefault, st);     // 34 floats, host or device
  if (e != hipSuccess) return SSBEV_ELAUNCH;
  if (hipStreamSynchronize(st) != hipSuccess) return SSBEV_ELAUNCH;
  unsigned long long* zbuf = static_cast<unsigned long long*>(ws);
  if (hipMemsetAsync(zbuf, 0xff, (size_t)H * W * sizeof(unsigned long long), st) != hipSuccess) return SSBEV_ELAUNCH;
  if (n_points)
    hipLaunchKernelGGL(lidar_project_kernel, dim3(cdiv((size_t)n_points, 256)), dim3(256), 0, st, points, c, uvd, valid, zbuf,
                       n_points, H, W);
  hipLaunchKernelGGL(lidar_resolve_kernel, dim3(cdiv((size_t)H * W, 256)), dim3(256), 0, st, zbuf, labels, depth, seg, H * W);
  return ssbev_launch_status();
}

}  // extern "C"
