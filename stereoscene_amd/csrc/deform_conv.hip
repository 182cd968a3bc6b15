// Deformable convolution v1 (mmcv DCN = DeformConv2dPack, used by DepthNet at bevdepth.py:490-498), split as mmcv
// does into a sampling stage and a grouped GEMM -- but with the GEMM on the MFMA convolution kernels of this library:
//   im2col:  cols[g][pixel][tap * Cg + c] = bilinear(x[.., g*Cg + c], pixel + tap + offset[pixel][tap])
//   col2im:  gradient of cols back onto x (four weighted corner adds per sample) and onto the offsets.
// Channels-last everywhere: x [B,H,W,C], offsets [B,H,W,2*K] (channel 2k = dy, 2k+1 = dx of tap k, mmcv's order),
// cols [G][B*H*W][K*Cg] -- i.e. group g's slab IS a channels-last [B, K*Cg, H, W] tensor, so the grouped
// contraction is one dense 1x1 convolution per group.  One workgroup per output pixel; a thread owns float4s of
// channels (consecutive threads -> consecutive 16 bytes of one pixel: coalesced gathers).
// The x gradient uses hardware fp32 atomics like mmcv's col2im (unordered sums); the offset gradient is a
// deterministic workgroup reduction.
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

__device__ __forceinline__ void atomic_add4(float* p, float w, const float4& gval) {
  unsafeAtomicAdd(p + 0, w * gval.x);
  unsafeAtomicAdd(p + 1, w * gval.y);
  unsafeAtomicAdd(p + 2, w * gval.z);
  unsafeAtomicAdd(p + 3, w * gval.w);
}

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

// Input gradient.  Every sample adds to four corner pixels (unordered fp32 atomics in mmcv).  Sent straight to L2 that is
// 4 x 9 atomics per output pixel and channel (177 M per DepthNet step: 2.2 ms).  A workgroup therefore owns a TH x TW tile
// of output pixels and a 32-channel slice and accumulates into an LDS window that covers the tile plus a reach of R
// pixels (samples that land outside the window -- offsets larger than R -- go to global memory directly); the window
// is flushed with one global atomic per touched element (~9x fewer L2 atomics).
// LDS float atomics are NOT used for the window: ds_add_f32 retires ~0.4 lanes per clock on gfx950 (PMC: 155 LDS-busy
// cycles per wave instruction, 0.85 ms for this kernel).  Instead every (tap, corner) phase is made collision-free by a
// claim round: the threads of a pixel store their pixel id into an owner map (plain ds_write), and after a barrier
// only the pixels that still own their target add with plain 16-byte read-modify-writes; losers (two pixels of the tile
// whose offsets send the same corner to the same window position -- rare, offsets vary smoothly) retry.  36 phases of a
// few LDS instructions and barriers each.
constexpr int kDcnTH = 4, kDcnTW = 16, kDcnR = 3, kDcnCC = 32;
constexpr int kDcnWH = kDcnTH + 2 * kDcnR + 1, kDcnWW = kDcnTW + 2 * kDcnR + 1, kDcnPitch = kDcnCC + 4;

__global__ void __launch_bounds__(256)
dcn_input_grad_kernel(const float* __restrict__ off, const float* __restrict__ gcols, float* __restrict__ gx, DcnGeom g) {
  __shared__ __align__(16) float win[kDcnWH * kDcnWW * kDcnPitch];
  __shared__ int owner[kDcnWH * kDcnWW];
  const int tiles_w = (g.W + kDcnTW - 1) / kDcnTW;
  const int th = blockIdx.x / tiles_w, tw = blockIdx.x % tiles_w;
  const int c0 = blockIdx.y * kDcnCC;
  const long b = blockIdx.z;
  const int h_lo = th * kDcnTH - kDcnR, w_lo = tw * kDcnTW - kDcnR;
  const int K = g.k * g.k, Cg = g.C / g.G;
  const long BHW = (long)g.B * g.H * g.W;
  for (int i = threadIdx.x; i < kDcnWH * kDcnWW * kDcnPitch; i += 256) win[i] = 0.0f;
  __syncthreads();
  float* gxb = gx + b * (long)g.H * g.W * g.C;
  const int px = threadIdx.x >> 2, qsub = threadIdx.x & 3;            // 64 pixels x 4 threads, 8 channels each
  const int oh = th * kDcnTH + px / kDcnTW, ow = tw * kDcnTW + px % kDcnTW;
  const int c = c0 + qsub * 8;
  const bool live = oh < g.H && ow < g.W && c < g.C;
  const long pix = live ? (b * g.H + oh) * g.W + ow : 0;
  const int grp = live ? c / Cg : 0, cg = c - grp * Cg;
  for (int tap = 0; tap < K; ++tap) {
    Sample s;
    s.inside = false;
    float gv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (live) {
      s = make_sample(off, pix, oh, ow, tap, g);
      if (s.inside) {
        const float* gp = gcols + (((long)grp * BHW + pix) * K + tap) * Cg + cg;
        const float4 ga = *reinterpret_cast<const float4*>(gp), gb = *reinterpret_cast<const float4*>(gp + 4);
        gv[0] = ga.x; gv[1] = ga.y; gv[2] = ga.z; gv[3] = ga.w; gv[4] = gb.x; gv[5] = gb.y; gv[6] = gb.z; gv[7] = gb.w;
      }
    }
    const bool use = live && s.inside;
    const float wgt[4] = {(1.f - s.lh) * (1.f - s.lw), (1.f - s.lh) * s.lw, s.lh * (1.f - s.lw), s.lh * s.lw};
    const bool okc[4] = {s.ok00, s.ok01, s.ok10, s.ok11};
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      int pos = -1;
      if (use && okc[k4]) {
        const int hc = s.h0 + (k4 >> 1), wc = s.w0 + (k4 & 1);
        const int wh = hc - h_lo, wwc = wc - w_lo;
        if (wh >= 0 && wh < kDcnWH && wwc >= 0 && wwc < kDcnWW) {
          pos = wh * kDcnWW + wwc;
        } else {                                           // beyond the reach of the window: straight to L2
          float* dst = gxb + ((long)hc * g.W + wc) * g.C + c;
#pragma unroll
          for (int e = 0; e < 8; ++e) unsafeAtomicAdd(dst + e, wgt[k4] * gv[e]);
        }
      }
      bool pending = pos >= 0;
      while (__syncthreads_or(pending ? 1 : 0)) {
        if (pending) owner[pos] = px;
        __syncthreads();
        if (pending && owner[pos] == px) {
          float4* d = reinterpret_cast<float4*>(win + pos * kDcnPitch + qsub * 8);
          float4 a0 = d[0], a1 = d[1];
          const float w = wgt[k4];
          a0.x += w * gv[0]; a0.y += w * gv[1]; a0.z += w * gv[2]; a0.w += w * gv[3];
          a1.x += w * gv[4]; a1.y += w * gv[5]; a1.z += w * gv[6]; a1.w += w * gv[7];
          d[0] = a0; d[1] = a1;
          pending = false;
        }
        __syncthreads();
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kDcnWH * kDcnWW * kDcnCC; i += 256) {
    const int e = i % kDcnCC, wp = i / kDcnCC;
    const int hc = h_lo + wp / kDcnWW, wc = w_lo + wp % kDcnWW;
    const float v = win[wp * kDcnPitch + e];
    if (v != 0.0f && hc >= 0 && hc < g.H && wc >= 0 && wc < g.W && c0 + e < g.C)
      unsafeAtomicAdd(gxb + ((long)hc * g.W + wc) * g.C + c0 + e, v);
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

int ssbev_dcn_col2im(const float* x, const float* offset, const float* gcols, float* gx, float* goffset,
                     const ssbev_dcn_dims* d, ssbev_stream_t stream) {
  if (!dcn_ok(d) || !x || !offset || !gcols || !gx || !goffset) return SSBEV_EINVAL;
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(gx, 0, (size_t)d->B * d->H * d->W * d->C * sizeof(float), st) != hipSuccess) return SSBEV_ELAUNCH;
  hipLaunchKernelGGL(dcn_coord_grad_kernel, dim3((unsigned)((long)d->B * d->H * d->W)), dim3(256), 0, st, x, offset,
                     gcols, goffset, to_geom(d));
  const int tiles = ((d->H + kDcnTH - 1) / kDcnTH) * ((d->W + kDcnTW - 1) / kDcnTW);
  hipLaunchKernelGGL(dcn_input_grad_kernel, dim3(tiles, (d->C + kDcnCC - 1) / kDcnCC, d->B), dim3(256), 0, st, offset,
                     gcols, gx, to_geom(d));
  return ssbev_launch_status();
}

}  // extern "C"
