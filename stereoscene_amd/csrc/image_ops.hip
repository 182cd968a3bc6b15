// Image-branch operators (CustomEfficientNet-B7 + SECONDFPN, the step before a1/a5 of the hot path):
// depthwise k x k convolution with TensorFlow-style "same" padding (mmcv Conv2dAdaptivePadding: pad so that
// Ho = ceil(Hi / stride), the odd row/column of padding goes to the bottom/right), Swish, and the squeeze-excitation
// pair (global average pool, per-(sample, channel) rescale).  All of them are HBM-bound streaming passes over
// channels-last [B, H, W, C] fp32 tensors: one thread owns 4 consecutive channels (16-byte accesses, a wave covers
// 1 KB of one pixel line), neighbouring output pixels re-read their k x k input neighbourhood from L1/L2.
// The dense 1x1 / stem / neck convolutions of the branch run on the MFMA kernels of conv_mfma.hip.
#include "common.h"

#include <algorithm>

namespace {

struct DwGeom { int B, C, Hi, Wi, Ho, Wo, k, s, pt, pl; };

// y[b, ho, wo, c] = sum_{i,j} x[b, ho*s - pt + i, wo*s - pl + j, c] * w[i*k + j][c]
template <int K>
__global__ void __launch_bounds__(256)
dw_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, DwGeom g, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int C4 = g.C >> 2;
  const int c = (int)(i % C4) << 2;
  long p = i / C4;
  const int wo = (int)(p % g.Wo); p /= g.Wo;
  const int ho = (int)(p % g.Ho);
  const int b = (int)(p / g.Ho);
  const int h0 = ho * g.s - g.pt, w0 = wo * g.s - g.pl;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int a = 0; a < K; ++a) {
    const int h = h0 + a;
    if (h < 0 || h >= g.Hi) continue;
#pragma unroll
    for (int e = 0; e < K; ++e) {
      const int ww = w0 + e;
      if (ww < 0 || ww >= g.Wi) continue;
      const float4 xv = *reinterpret_cast<const float4*>(x + (((long)b * g.Hi + h) * g.Wi + ww) * g.C + c);
      const float4 wv = *reinterpret_cast<const float4*>(w + (long)(a * K + e) * g.C + c);
      acc.x += xv.x * wv.x; acc.y += xv.y * wv.y; acc.z += xv.z * wv.z; acc.w += xv.w * wv.w;
    }
  }
  *reinterpret_cast<float4*>(y + i * 4) = acc;
}

// gx[b, h, w, c] = sum_{i,j : (h + pt - i) % s == 0, ...} gy[b, (h + pt - i)/s, (w + pl - j)/s, c] * w[i*k + j][c]
template <int K>
__global__ void __launch_bounds__(256)
dw_bwd_data_kernel(const float* __restrict__ gy, const float* __restrict__ w, float* __restrict__ gx, DwGeom g, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int C4 = g.C >> 2;
  const int c = (int)(i % C4) << 2;
  long p = i / C4;
  const int wi = (int)(p % g.Wi); p /= g.Wi;
  const int hi = (int)(p % g.Hi);
  const int b = (int)(p / g.Hi);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int a = 0; a < K; ++a) {
    const int hn = hi + g.pt - a;
    if (hn < 0 || hn % g.s != 0) continue;
    const int ho = hn / g.s;
    if (ho >= g.Ho) continue;
#pragma unroll
    for (int e = 0; e < K; ++e) {
      const int wn = wi + g.pl - e;
      if (wn < 0 || wn % g.s != 0) continue;
      const int wo = wn / g.s;
      if (wo >= g.Wo) continue;
      const float4 gv = *reinterpret_cast<const float4*>(gy + (((long)b * g.Ho + ho) * g.Wo + wo) * g.C + c);
      const float4 wv = *reinterpret_cast<const float4*>(w + (long)(a * K + e) * g.C + c);
      acc.x += gv.x * wv.x; acc.y += gv.y * wv.y; acc.z += gv.z * wv.z; acc.w += gv.w * wv.w;
    }
  }
  *reinterpret_cast<float4*>(gx + i * 4) = acc;
}

// Partial weight gradients: block (chunk of output pixels) x (64 channel quads); the 4 pixel lanes of a block are
// folded through LDS in a fixed order and each chunk writes its own [k*k][C] slab (deterministic; folded by
// dw_reduce_kernel).
// Thread layout: Q channel quads x L pixel lanes with Q * L <= 256 (Q = all quads when C <= 1024, else 64 per block), so
// that narrow layers (C = 32: 8 quads x 32 lanes) keep every thread busy.
template <int K>
__global__ void __launch_bounds__(256)
dw_bwd_weight_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ part, DwGeom g,
                     long npix, int pix_per_chunk, int Q, int L) {
  __shared__ float4 red[256];
  const int ql = threadIdx.x % Q, pl = threadIdx.x / Q;
  const int quad = blockIdx.y * Q + ql;
  const bool qok = quad < (g.C >> 2) && pl < L;
  const int c = quad << 2;
  float4 acc[K * K];
#pragma unroll
  for (int t = 0; t < K * K; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  const long p0 = (long)blockIdx.x * pix_per_chunk, p1 = min(npix, p0 + pix_per_chunk);
  if (qok) {
    for (long p = p0 + pl; p < p1; p += L) {
      long r = p;
      const int wo = (int)(r % g.Wo); r /= g.Wo;
      const int ho = (int)(r % g.Ho);
      const int b = (int)(r / g.Ho);
      const float4 gv = *reinterpret_cast<const float4*>(gy + p * g.C + c);
      const int h0 = ho * g.s - g.pt, w0 = wo * g.s - g.pl;
#pragma unroll
      for (int a = 0; a < K; ++a) {
        const int h = h0 + a;
        if (h < 0 || h >= g.Hi) continue;
#pragma unroll
        for (int e = 0; e < K; ++e) {
          const int ww = w0 + e;
          if (ww < 0 || ww >= g.Wi) continue;
          const float4 xv = *reinterpret_cast<const float4*>(x + (((long)b * g.Hi + h) * g.Wi + ww) * g.C + c);
          float4& t = acc[a * K + e];
          t.x += gv.x * xv.x; t.y += gv.y * xv.y; t.z += gv.z * xv.z; t.w += gv.w * xv.w;
        }
      }
    }
  }
  float* dst = part + (long)blockIdx.x * (K * K) * g.C;
#pragma unroll
  for (int t = 0; t < K * K; ++t) {
    red[threadIdx.x] = acc[t];
    __syncthreads();
    if (pl == 0 && qok) {
      float4 s = red[ql];
      for (int l = 1; l < L; ++l) {
        const float4 v = red[l * Q + ql];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      *reinterpret_cast<float4*>(dst + (long)t * g.C + c) = s;
    }
    __syncthreads();
  }
}

// one wave per output element, lanes striding over the chunk slabs (fixed butterfly order: deterministic)
__global__ void __launch_bounds__(256)
dw_reduce_kernel(const float* __restrict__ part, float* __restrict__ gw, int n, int nchunks) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  float s = 0.0f;
  for (int ch = lane; ch < nchunks; ch += 64) s += part[(long)ch * n + i];
  s = wave_sum(s);
  if (lane == 0) gw[i] = s;
}

__global__ void __launch_bounds__(256)
swish_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n4) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    float4 o;
    o.x = v.x / (1.0f + __expf(-v.x)); o.y = v.y / (1.0f + __expf(-v.y));
    o.z = v.z / (1.0f + __expf(-v.z)); o.w = v.w / (1.0f + __expf(-v.w));
    reinterpret_cast<float4*>(y)[i] = o;
  }
}

__device__ __forceinline__ float swish_grad(float x, float g) {
  const float s = 1.0f / (1.0f + __expf(-x));
  return g * (s + x * s * (1.0f - s));
}

__global__ void __launch_bounds__(256)
swish_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gx, long n4) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(x)[i], g = reinterpret_cast<const float4*>(gy)[i];
    float4 o;
    o.x = swish_grad(v.x, g.x); o.y = swish_grad(v.y, g.y); o.z = swish_grad(v.z, g.z); o.w = swish_grad(v.w, g.w);
    reinterpret_cast<float4*>(gx)[i] = o;
  }
}

// out[b][c] = scale * sum_s a[b][s][c] * (bmul ? bmul[b][s][c] : 1): global average pool (scale = 1/S) and the
// gate gradient of the SE rescale (sum_s gy * x).  grid (C/4 quads / 64, B, pixel chunks), 4 pixel lanes per block;
// chunk partials [chunk][B][C] are folded in chunk order by chan_fold_kernel (deterministic).
__global__ void __launch_bounds__(256)
chan_sum_kernel(const float* __restrict__ a, const float* __restrict__ bmul, float* __restrict__ out, long S, int C,
                long per_chunk) {
  __shared__ float4 red[4][64];
  const int ql = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int quad = blockIdx.x * 64 + ql;
  const bool ok = quad < (C >> 2);
  const int c = quad << 2;
  const long base = (long)blockIdx.y * S * C;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const long s0 = (long)blockIdx.z * per_chunk, s1 = min(S, s0 + per_chunk);
  if (ok) {
    for (long s = s0 + pl; s < s1; s += 4) {
      float4 v = *reinterpret_cast<const float4*>(a + base + s * C + c);
      if (bmul) {
        const float4 m = *reinterpret_cast<const float4*>(bmul + base + s * C + c);
        v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
      }
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  red[pl][ql] = acc;
  __syncthreads();
  if (pl == 0 && ok) {
    float4 s = red[0][ql];
#pragma unroll
    for (int l = 1; l < 4; ++l) { const float4 v = red[l][ql]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    *reinterpret_cast<float4*>(out + ((long)blockIdx.z * gridDim.y + blockIdx.y) * C + c) = s;
  }
}

__global__ void __launch_bounds__(256)
chan_fold_kernel(const float* __restrict__ part, float* __restrict__ out, int n, int nchunks, float scale) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  float s = 0.0f;
  for (int ch = lane; ch < nchunks; ch += 64) s += part[(long)ch * n + i];
  s = wave_sum(s);
  if (lane == 0) out[i] = s * scale;
}

// y[b][s][c] = x[b][s][c] * gate[b][c] (+ optionally add[b][s][c]); also the broadcast of the pool gradient (x = null:
// y = gate * scale)
__global__ void __launch_bounds__(256)
chan_scale_kernel(const float* __restrict__ x, const float* __restrict__ gate, float* __restrict__ y, long S, int C,
                  long total4, float scale) {
  const long stride = (long)gridDim.x * 256;
  const int C4 = C >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
    const int c = (int)(i % C4) << 2;
    const long b = i / ((long)C4 * S);
    const float4 gt = *reinterpret_cast<const float4*>(gate + b * C + c);
    float4 o;
    if (x) {
      const float4 v = reinterpret_cast<const float4*>(x)[i];
      o.x = v.x * gt.x; o.y = v.y * gt.y; o.z = v.z * gt.z; o.w = v.w * gt.w;
    } else {
      o.x = gt.x * scale; o.y = gt.y * scale; o.z = gt.z * scale; o.w = gt.w * scale;
    }
    reinterpret_cast<float4*>(y)[i] = o;
  }
}

bool dw_ok(const ssbev_dw_dims* d) {
  return d && d->B > 0 && d->C > 0 && d->C % 4 == 0 && d->Hi > 0 && d->Wi > 0 && d->Ho > 0 && d->Wo > 0 &&
         (d->k == 3 || d->k == 5) && (d->stride == 1 || d->stride == 2) && d->pad_t >= 0 && d->pad_l >= 0;
}

DwGeom dw_geom(const ssbev_dw_dims* d) { return DwGeom{d->B, d->C, d->Hi, d->Wi, d->Ho, d->Wo, d->k, d->stride, d->pad_t, d->pad_l}; }

void dw_layout(const ssbev_dw_dims* d, int* Q, int* L, int* qblocks) {
  const int quads = d->C >> 2;
  *Q = quads <= 256 ? quads : 64;
  *L = 256 / *Q;
  *qblocks = (quads + *Q - 1) / *Q;
}

void dw_chunks(const ssbev_dw_dims* d, long* npix, int* ppc, int* nchunks) {
  *npix = (long)d->B * d->Ho * d->Wo;
  int Q, L, qblocks;
  dw_layout(d, &Q, &L, &qblocks);
  long want = std::max(1L, 2048L / qblocks);                 // ~2k workgroups in total
  long per = std::max(64L, (*npix + want - 1) / want);
  *ppc = (int)per;
  *nchunks = (int)((*npix + per - 1) / per);
}

void chan_chunks(int B, long S, int C, long* per, int* nchunks) {
  const long blocks = (long)B * (((C >> 2) + 63) / 64);
  long want = std::max(1L, 1024L / blocks);
  *per = std::max(64L, (S + want - 1) / want);
  *nchunks = (int)((S + *per - 1) / *per);
}

unsigned stream_blocks(long n4) { return (unsigned)std::min<long>((n4 + 255) / 256, 256L * 32); }

}  // namespace

extern "C" {

int ssbev_dwconv2d_fwd(const float* x, const float* w, float* y, const ssbev_dw_dims* d, ssbev_stream_t stream) {
  if (!dw_ok(d) || !x || !w || !y) return SSBEV_EINVAL;
  const long total = (long)d->B * d->Ho * d->Wo * (d->C >> 2);
  const DwGeom g = dw_geom(d);
  if (d->k == 3) hipLaunchKernelGGL(dw_fwd_kernel<3>, dim3(cdiv((size_t)total, 256)), dim3(256), 0, as_stream(stream), x, w, y, g, total);
  else hipLaunchKernelGGL(dw_fwd_kernel<5>, dim3(cdiv((size_t)total, 256)), dim3(256), 0, as_stream(stream), x, w, y, g, total);
  return ssbev_launch_status();
}

int ssbev_dwconv2d_bwd_data(const float* gy, const float* w, float* gx, const ssbev_dw_dims* d, ssbev_stream_t stream) {
  if (!dw_ok(d) || !gy || !w || !gx) return SSBEV_EINVAL;
  const long total = (long)d->B * d->Hi * d->Wi * (d->C >> 2);
  const DwGeom g = dw_geom(d);
  if (d->k == 3) hipLaunchKernelGGL(dw_bwd_data_kernel<3>, dim3(cdiv((size_t)total, 256)), dim3(256), 0, as_stream(stream), gy, w, gx, g, total);
  else hipLaunchKernelGGL(dw_bwd_data_kernel<5>, dim3(cdiv((size_t)total, 256)), dim3(256), 0, as_stream(stream), gy, w, gx, g, total);
  return ssbev_launch_status();
}

size_t ssbev_dwconv2d_bwd_weight_workspace(const ssbev_dw_dims* d) {
  if (!dw_ok(d)) return 0;
  long npix; int ppc, nchunks;
  dw_chunks(d, &npix, &ppc, &nchunks);
  return (size_t)nchunks * d->k * d->k * d->C;          // floats
}

int ssbev_dwconv2d_bwd_weight(const float* x, const float* gy, float* gw, const ssbev_dw_dims* d, float* ws,
                              size_t ws_elems, ssbev_stream_t stream) {
  if (!dw_ok(d) || !x || !gy || !gw || !ws) return SSBEV_EINVAL;
  if (ws_elems < ssbev_dwconv2d_bwd_weight_workspace(d)) return SSBEV_EWORKSPACE;
  long npix; int ppc, nchunks;
  dw_chunks(d, &npix, &ppc, &nchunks);
  const DwGeom g = dw_geom(d);
  int Q, L, qblocks;
  dw_layout(d, &Q, &L, &qblocks);
  const dim3 grid(nchunks, qblocks), block(256);
  hipStream_t st = as_stream(stream);
  if (d->k == 3) hipLaunchKernelGGL(dw_bwd_weight_kernel<3>, grid, block, 0, st, x, gy, ws, g, npix, ppc, Q, L);
  else hipLaunchKernelGGL(dw_bwd_weight_kernel<5>, grid, block, 0, st, x, gy, ws, g, npix, ppc, Q, L);
  const int n = d->k * d->k * d->C;
  hipLaunchKernelGGL(dw_reduce_kernel, dim3(cdiv(n, 4)), dim3(256), 0, st, ws, gw, n, nchunks);
  return ssbev_launch_status();
}

int ssbev_swish_fwd(const float* x, float* y, int64_t n, ssbev_stream_t stream) {
  if (!x || !y || n <= 0 || n % 4 != 0) return SSBEV_EINVAL;
  hipLaunchKernelGGL(swish_fwd_kernel, dim3(stream_blocks(n / 4)), dim3(256), 0, as_stream(stream), x, y, (long)(n / 4));
  return ssbev_launch_status();
}

int ssbev_swish_bwd(const float* x, const float* gy, float* gx, int64_t n, ssbev_stream_t stream) {
  if (!x || !gy || !gx || n <= 0 || n % 4 != 0) return SSBEV_EINVAL;
  hipLaunchKernelGGL(swish_bwd_kernel, dim3(stream_blocks(n / 4)), dim3(256), 0, as_stream(stream), x, gy, gx, (long)(n / 4));
  return ssbev_launch_status();
}

size_t ssbev_chan_sum_workspace(int B, int64_t S, int C) {
  if (B <= 0 || S <= 0 || C <= 0 || C % 4 != 0) return 0;
  long per; int nchunks;
  chan_chunks(B, (long)S, C, &per, &nchunks);
  return (size_t)nchunks * B * C;                       // floats
}

int ssbev_chan_sum(const float* a, const float* bmul, float* out, int B, int64_t S, int C, float scale, float* ws,
                   size_t ws_elems, ssbev_stream_t stream) {
  if (!a || !out || !ws || B <= 0 || S <= 0 || C <= 0 || C % 4 != 0) return SSBEV_EINVAL;
  if (ws_elems < ssbev_chan_sum_workspace(B, S, C)) return SSBEV_EWORKSPACE;
  long per; int nchunks;
  chan_chunks(B, (long)S, C, &per, &nchunks);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(chan_sum_kernel, dim3(cdiv(C >> 2, 64), B, nchunks), dim3(256), 0, st, a, bmul, ws, (long)S, C, per);
  hipLaunchKernelGGL(chan_fold_kernel, dim3(cdiv((size_t)B * C, 4)), dim3(256), 0, st, ws, out, B * C, nchunks, scale);
  return ssbev_launch_status();
}

int ssbev_chan_scale(const float* x, const float* gate, float* y, int B, int64_t S, int C, float scale, ssbev_stream_t stream) {
  if (!gate || !y || B <= 0 || S <= 0 || C <= 0 || C % 4 != 0) return SSBEV_EINVAL;
  const long total4 = (long)B * S * (C >> 2);
  hipLaunchKernelGGL(chan_scale_kernel, dim3(stream_blocks(total4)), dim3(256), 0, as_stream(stream), x, gate, y, (long)S, C,
                     total4, scale);
  return ssbev_launch_status();
}

}  // extern "C"
