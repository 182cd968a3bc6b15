// Deformable convolution v1 (mmcv DCN = DeformConv2dPack, used by DepthNet at bevdepth.py:490-498), split as mmcv
// does into a sampling stage and a grouped GEMM -- but with the GEMM on the MFMA convolution kernels of this library:
//   im2col:  cols[g][pixel][tap * Cg + c] = bilinear(x[.., g*Cg + c], pixel + tap + offset[pixel][tap])
//   col2im:  gradient of cols back onto x (four weighted corner adds per sample) and onto the offsets.
// Channels-last everywhere: x [B,H,W,C], offsets [B,H,W,2*K] (channel 2k = dy, 2k+1 = dx of tap k, mmcv's order),
// cols [G][B*H*W][K*Cg] -- i.e. group g's slab IS a channels-last [B, K*Cg, H, W] tensor, so the grouped
// contraction is one dense 1x1 convolution per group.  One workgroup per output pixel; a thread owns float4s of
// channels (consecutive threads -> consecutive 16 bytes of one pixel: coalesced gathers).
// The x gradient is a GATHER (round 6): the (output pixel, tap, corner) samples are sorted by the input pixel they touch with the
// library's own stable counting sort (ssbev_pool_prepare, the CSR build of the voxel scatter), and one workgroup per input pixel
// adds its list in ascending sample order -- run-to-run identical, no atomics, no memset.  (mmcv's col2im, and rounds 1-5 here,
// scatter with fp32 atomics: unordered sums, reproducible to rounding only.)  The offset gradient is a deterministic workgroup
// reduction.
#include "common.h"

namespace {

struct DcnGeom { int B, C, H, W, G, k, pad, dil; };

struct Sample {          // bilinear footprint of one (pixel, tap)
  int h0, w0;
  float lh, lw;
  bool inside, ok00, ok01, ok10, ok11;
};

__device__ __forceinline__ Sample make_sample(const float* __restrict__ off, long pix, int oh, int ow, int tap,
                                              const DcnGeom& g) {
  const int K = g.k * g.k;
  const float dy = off[pix * 2 * K + 2 * tap], dx = off[pix * 2 * K + 2 * tap + 1];
  const float hh = (float)(oh - g.pad + (tap / g.k) * g.dil) + dy;
  const float ww = (float)(ow - g.pad + (tap % g.k) * g.dil) + dx;
  Sample s;
  s.inside = hh > -1.0f && ww > -1.0f && hh < (float)g.H && ww < (float)g.W;
  const float fh = floorf(hh), fw = floorf(ww);
  s.h0 = (int)fh; s.w0 = (int)fw;
  s.lh = hh - fh; s.lw = ww - fw;
  const bool hlo = s.h0 >= 0, hhi = s.h0 + 1 <= g.H - 1, wlo = s.w0 >= 0, whi = s.w0 + 1 <= g.W - 1;
  s.ok00 = s.inside && hlo && wlo && s.h0 < g.H && s.w0 < g.W;
  s.ok01 = s.inside && hlo && whi && s.h0 < g.H;
  s.ok10 = s.inside && hhi && wlo && s.w0 < g.W;
  s.ok11 = s.inside && hhi && whi;
  return s;
}

__device__ __forceinline__ float4 ld4(const float* p, bool ok) {
  return ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ void __launch_bounds__(256)
dcn_im2col_kernel(const float* __restrict__ x, const float* __restrict__ off, float* __restrict__ cols, DcnGeom g) {
  const long pix = blockIdx.x;                       // b * H * W + oh * W + ow
  const int ow = (int)(pix % g.W), oh = (int)((pix / g.W) % g.H);
  const long b = pix / ((long)g.W * g.H);
  const int K = g.k * g.k, q = g.C >> 2, Cg = g.C / g.G;
  const long BHW = (long)g.B * g.H * g.W;
  const float* xb = x + b * (long)g.H * g.W * g.C;
  for (int i = threadIdx.x; i < K * q; i += 256) {
    const int tap = i / q, c = (i - tap * q) * 4;
    const Sample s = make_sample(off, pix, oh, ow, tap, g);
    const float* p00 = xb + ((long)s.h0 * g.W + s.w0) * g.C + c;
    const float4 v00 = ld4(p00, s.ok00), v01 = ld4(p00 + g.C, s.ok01);
    const float4 v10 = ld4(p00 + (long)g.W * g.C, s.ok10), v11 = ld4(p00 + (long)g.W * g.C + g.C, s.ok11);
    const float w00 = (1.f - s.lh) * (1.f - s.lw), w01 = (1.f - s.lh) * s.lw, w10 = s.lh * (1.f - s.lw), w11 = s.lh * s.lw;
    float4 r;
    r.x = w00 * v00.x + w01 * v01.x + w10 * v10.x + w11 * v11.x;
    r.y = w00 * v00.y + w01 * v01.y + w10 * v10.y + w11 * v11.y;
    r.z = w00 * v00.z + w01 * v01.z + w10 * v10.z + w11 * v11.z;
    r.w = w00 * v00.w + w01 * v01.w + w10 * v10.w + w11 * v11.w;
    const int grp = c / Cg, cg = c - grp * Cg;
    *reinterpret_cast<float4*>(cols + (((long)grp * BHW + pix) * K + tap) * Cg + cg) = r;
  }
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// Offset gradient: one workgroup per output pixel, deterministic reduction over the channels.
__global__ void __launch_bounds__(256)
dcn_coord_grad_kernel(const float* __restrict__ x, const float* __restrict__ off, const float* __restrict__ gcols,
                      float* __restrict__ goff, DcnGeom g) {
  __shared__ float red[4][32];
  const long pix = blockIdx.x;
  const int ow = (int)(pix % g.W), oh = (int)((pix / g.W) % g.H);
  const long b = pix / ((long)g.W * g.H);
  const int K = g.k * g.k, q = g.C >> 2, Cg = g.C / g.G;
  const long BHW = (long)g.B * g.H * g.W;
  const float* xb = x + b * (long)g.H * g.W * g.C;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int tap = 0; tap < K; ++tap) {
    const Sample s = make_sample(off, pix, oh, ow, tap, g);
    float dh = 0.f, dw = 0.f;
    for (int cq = threadIdx.x; cq < q; cq += 256) {
      const int c = cq * 4, grp = c / Cg, cg = c - grp * Cg;
      const float4 gc = *reinterpret_cast<const float4*>(gcols + (((long)grp * BHW + pix) * K + tap) * Cg + cg);
      const long o00 = ((long)s.h0 * g.W + s.w0) * g.C + c;
      const float4 v00 = ld4(xb + o00, s.ok00), v01 = ld4(xb + o00 + g.C, s.ok01);
      const float4 v10 = ld4(xb + o00 + (long)g.W * g.C, s.ok10), v11 = ld4(xb + o00 + (long)g.W * g.C + g.C, s.ok11);
      const float d00 = dot4(gc, v00), d01 = dot4(gc, v01), d10 = dot4(gc, v10), d11 = dot4(gc, v11);
      // d val / d h = -(1-lw) v00 - lw v01 + (1-lw) v10 + lw v11 ;  d val / d w = -(1-lh) v00 + (1-lh) v01 - lh v10 + lh v11
      dh += (1.f - s.lw) * (d10 - d00) + s.lw * (d11 - d01);
      dw += (1.f - s.lh) * (d01 - d00) + s.lh * (d11 - d10);
    }
    // lanes by xor-shuffle, waves through LDS in fixed order
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      dh += __shfl_xor(dh, m, 64);
      dw += __shfl_xor(dw, m, 64);
    }
    if (lane == 0) { red[wave][2 * tap] = dh; red[wave][2 * tap + 1] = dw; }
  }
  __syncthreads();
  if (threadIdx.x < 2 * K) {
    const float t = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    goff[pix * 2 * K + threadIdx.x] = t;
  }
}

// ---- input gradient, gather form -------------------------------------------------------------------------------------------
// sample id = ((pix * K + tap) * 4 + corner); key = the input pixel (b * H + h) * W + w the corner lands on, or -1 (outside the
// map / zero-padded corner); wts[id] = the bilinear weight of that corner.
__global__ void __launch_bounds__(256)
dcn_corner_keys_kernel(const float* __restrict__ off, int32_t* __restrict__ keys, float* __restrict__ wts, DcnGeom g, long n_samples) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;          // (pix, tap)
  if (i >= n_samples) return;
  const int K = g.k * g.k;
  const long pix = i / K;
  const int tap = (int)(i - pix * K);
  const int ow = (int)(pix % g.W), oh = (int)((pix / g.W) % g.H);
  const long b = pix / ((long)g.W * g.H);
  const Sample s = make_sample(off, pix, oh, ow, tap, g);
  const bool okc[4] = {s.ok00, s.ok01, s.ok10, s.ok11};
  const float wgt[4] = {(1.f - s.lh) * (1.f - s.lw), (1.f - s.lh) * s.lw, s.lh * (1.f - s.lw), s.lh * s.lw};
  int4 k4;
  int* kk = reinterpret_cast<int*>(&k4);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int hc = s.h0 + (c >> 1), wc = s.w0 + (c & 1);
    kk[c] = okc[c] ? (int)((b * g.H + hc) * g.W + wc) : -1;
  }
  *reinterpret_cast<int4*>(keys + i * 4) = k4;
  *reinterpret_cast<float4*>(wts + i * 4) = make_float4(wgt[0], wgt[1], wgt[2], wgt[3]);
}

// One workgroup per INPUT pixel; thread t owns the float4 of channels 4t.. (C / 4 <= 256 lanes per pass).  The list of a pixel
// (~4 k^2 samples when the offsets are small) is walked in ascending sample id = the canonical summation order.
__global__ void __launch_bounds__(256)
dcn_input_grad_gather_kernel(const float* __restrict__ gcols, const int32_t* __restrict__ starts, const int32_t* __restrict__ order,
                             const float* __restrict__ wts, float* __restrict__ gx, DcnGeom g) {
  const long ipix = blockIdx.x;
  const int K = g.k * g.k, q = g.C >> 2, Cg = g.C / g.G;
  const long BHW = (long)g.B * g.H * g.W;
  const int e0 = starts[ipix], e1 = starts[ipix + 1];
  for (int cq = threadIdx.x; cq < q; cq += 256) {
    const int c = cq * 4, grp = c / Cg, cg = c - grp * Cg;
    const float* gbase = gcols + (long)grp * BHW * K * Cg + cg;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int e = e0;
    for (; e + 4 <= e1; e += 4) {                 // four list entries in flight; the adds stay in list order
      int id[4];
      float w[4];
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) id[u] = order[e + u];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        w[u] = wts[id[u]];
        v[u] = *reinterpret_cast<const float4*>(gbase + (long)(id[u] >> 2) * Cg);      // (pix * K + tap) * Cg
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc.x += w[u] * v[u].x; acc.y += w[u] * v[u].y; acc.z += w[u] * v[u].z; acc.w += w[u] * v[u].w;
      }
    }
    for (; e < e1; ++e) {
      const int id = order[e];
      const float w = wts[id];
      const float4 v = *reinterpret_cast<const float4*>(gbase + (long)(id >> 2) * Cg);
      acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
    }
    *reinterpret_cast<float4*>(gx + ipix * g.C + c) = acc;
  }
}

bool dcn_ok(const ssbev_dcn_dims* d) {
  return d && d->B > 0 && d->C > 0 && d->H > 0 && d->W > 0 && d->G > 0 && d->C % d->G == 0 && (d->C / d->G) % 4 == 0 &&
         (d->C / d->G) % 8 == 0 && d->k >= 1 && d->k * d->k <= 16 && d->dil >= 1 && d->pad >= 0;
}

DcnGeom to_geom(const ssbev_dcn_dims* d) { return DcnGeom{d->B, d->C, d->H, d->W, d->G, d->k, d->pad, d->dil}; }

}  // namespace

extern "C" {

int ssbev_dcn_im2col(const float* x, const float* offset, float* cols, const ssbev_dcn_dims* d, ssbev_stream_t stream) {
  if (!dcn_ok(d) || !x || !offset || !cols) return SSBEV_EINVAL;
  hipLaunchKernelGGL(dcn_im2col_kernel, dim3((unsigned)((long)d->B * d->H * d->W)), dim3(256), 0, as_stream(stream), x,
                     offset, cols, to_geom(d));
  return ssbev_launch_status();
}

static ssbev_pool_dims dcn_sort_dims(const ssbev_dcn_dims* d) {
  ssbev_pool_dims p = {};
  p.B = d->B; p.nx = d->H; p.ny = d->W; p.nz = 1; p.C = 4; p.P = d->H * d->W * d->k * d->k * 4;
  p.dx[0] = p.dx[1] = p.dx[2] = 1.0f;
  return p;
}

// workspace of ssbev_dcn_col2im: keys[n] + weights[n] + order[n] + starts[B*H*W + 1] + the counting sort's own, n = 4 k^2 B H W
size_t ssbev_dcn_col2im_workspace(const ssbev_dcn_dims* d) {
  if (!dcn_ok(d)) return 0;
  const size_t n = (size_t)d->B * d->H * d->W * d->k * d->k * 4;
  const size_t nv = (size_t)d->B * d->H * d->W;
  if (n >= (1ull << 31)) return 0;
  const ssbev_pool_dims p = dcn_sort_dims(d);
  return 3 * n * 4 + ((nv + 1 + 3) & ~(size_t)3) * 4 + ssbev_pool_prepare_workspace((int)n, &p) + 256;
}

int ssbev_dcn_col2im(const float* x, const float* offset, const float* gcols, float* gx, float* goffset,
                     const ssbev_dcn_dims* d, void* ws, size_t ws_bytes, ssbev_stream_t stream) {
  if (!dcn_ok(d) || !x || !offset || !gcols || !gx || !goffset || !ws) return SSBEV_EINVAL;
  const size_t need = ssbev_dcn_col2im_workspace(d);
  if (need == 0) return SSBEV_EINVAL;
  if (ws_bytes < need) return SSBEV_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  const DcnGeom g = to_geom(d);
  const long npix = (long)d->B * d->H * d->W, ns = npix * d->k * d->k, n = ns * 4;
  hipLaunchKernelGGL(dcn_coord_grad_kernel, dim3((unsigned)npix), dim3(256), 0, st, x, offset, gcols, goffset, g);
  char* w = static_cast<char*>(ws);
  int32_t* keys = reinterpret_cast<int32_t*>(w);            w += n * 4;
  float* wts = reinterpret_cast<float*>(w);                  w += n * 4;
  int32_t* order = reinterpret_cast<int32_t*>(w);           w += n * 4;
  int32_t* starts = reinterpret_cast<int32_t*>(w);          w += ((npix + 1 + 3) & ~3L) * 4;
  w = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(w) + 255) & ~(uintptr_t)255);
  hipLaunchKernelGGL(dcn_corner_keys_kernel, dim3(cdiv((size_t)ns, 256)), dim3(256), 0, st, offset, keys, wts, g, ns);
  const ssbev_pool_dims p = dcn_sort_dims(d);
  const int rc = ssbev_pool_prepare(keys, (int)n, starts, order, &p, w, ssbev_pool_prepare_workspace((int)n, &p), stream);
  if (rc != SSBEV_OK) return rc;
  hipLaunchKernelGGL(dcn_input_grad_gather_kernel, dim3((unsigned)npix), dim3(256), 0, st, gcols, starts, order, wts, gx, g);
  return ssbev_launch_status();
}

}  // extern "C"
