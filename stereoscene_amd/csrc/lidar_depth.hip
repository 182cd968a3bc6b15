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

}  // namespace

extern "C" {

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
