// Thin 3x3x3 layers of the cost-volume stack on the matrix pipe, HBM-streaming (round 2).
//
// The stereo / MIE heads hold three stride-1 3x3x3 "same" convolutions with a 1..4-channel side next to a 32-channel
// side on the 192x48x160 volume (ViewTransformerLSSVoxel.py:185-187, 239-241: the two 32 -> 1 classifiers and the
// 2 -> 32 entry of volume_interaction).  The wide side is a 188.7 MB tensor, the thin side 5.9-11.8 MB, and the FLOPs are
// nothing (10 GF): these launches should cost one pass over the wide tensor.  The r1 kernels (conv_thin_kernel's VALU
// dot products over an LDS row ring, the padded-to-4 gather kernel) took 0.3-0.6 ms each, 5-10x that.
//
//  thin -> 32  (forward of 2 -> 32; data gradient of 32 -> 1)             conv_thinin_kernel<CIN>
//      y[v][n] = sum_{tap, c} W[n][tap][c] x[v + off(tap)][c]: a GEMM with M = 32 output channels, N = voxels and
//      K = 27 CIN, the B operand gathered 4 bytes per lane straight from the (L2-resident) thin tensor -- lane (li, lk)
//      of v_mfma_f32_32x32x2_f32 step s wants K index 2 s + lk of voxel li, and consecutive lanes are consecutive voxels,
//      so every gather is one contiguous 128-256 byte wave load.  No LDS, no barriers, 16 accumulators: the kernel is
//      the 188.7 MB output stream.
//
//  32 -> thin  (forward of 32 -> 1; data gradient of 2 -> 32)              conv_thinout_u_kernel<NOUT> + conv_thinout_sum_kernel
//      out[v][n] = sum_tap W[n][tap] . x[v + off(tap)].  Per INPUT voxel v' the 27 NOUT dot products
//      T[tap][n][v'] = W[n][tap] . x[v'] are a GEMM with M = 27 NOUT rows, K = 32 channels, N = voxels whose B operand is
//      the wide tensor read ONCE in its natural layout (one float4 per lane, as the gather kernels do); the output is
//      the shifted sum out[v] = sum_tap T[tap][v + off(tap)].  The kw part of the shift is done on the accumulators
//      (rows (kd, kh, n, kw = 0..2) of one group sit in four consecutive accumulator registers of a lane; lane = voxel
//      along w, two lane shifts give U[kd,kh,n][v] = T0[v-1] + T1[v] + T2[v+1]), the 9 NOUT planes U go to a workspace
//      (53 MB for NOUT = 1) and a second streaming kernel adds the nine (kd, kh)-shifted planes, bias and ReLU.
//      Traffic: 189 + 2 x 53 + 6 MB instead of the 27 LDS re-reads per voxel of conv_thin_kernel.
//
//  weight gradient of both kinds                                            wgrad_thinside_kernel<NT, SGN> + wgrad_thinside_reduce_kernel
//      gw[n][c][tap] = sum_v gy[v][n] x[v + off(tap)][c] with one 32-channel ("wide") and one NT-channel ("thin") tensor:
//      a GEMM whose K axis is the voxels of the WIDE tensor in memory order (A / B operand = 4 bytes per lane, a wave load
//      is two whole voxel lines), M = (tap, thin channel) rows gathered from the thin tensor at v -+ off(tap), N = the 32
//      wide channels.  Each wave owns a few rows of the volume and one 32 x 32 partial tile per M-tile; a second kernel
//      sums the partials in fixed order (deterministic) into the torch layout.
#include <algorithm>

#include "common.h"
#include "conv_thin_mfma.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

struct ThinGeom {
  int B, D, H, W;        // volume (input == output extent)
  int ntw, ntiles;       // 32-voxel tiles per row, in total
  int relu, has_bias;
};

// Wp[s * 64 + lane] = Weff[n = lane & 31][kk = 2 s + (lane >> 5)], kk = tap * CIN + c (zero beyond 27 CIN)
//   mode 0 (forward of a CIN -> 32 layer):        Weff[n][tap][c] = w[(n * CIN + c) * 27 + tap]
//   mode 1 (data gradient of a 32 -> CIN layer):  Weff[n][tap][c] = w[(c * 32 + n) * 27 + 26 - tap]
__global__ void __launch_bounds__(256)
pack_thinin_kernel(const float* __restrict__ w, float* __restrict__ wp, int cin, int mode) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int KS = (27 * cin + 1) / 2;
  if (i >= KS * 64) return;
  const int lane = i & 63, s = i >> 6, n = lane & 31, kk = 2 * s + (lane >> 5);
  float v = 0.0f;
  if (kk < 27 * cin) {
    const int tap = kk / cin, c = kk % cin;
    v = mode == 0 ? w[((size_t)n * cin + c) * 27 + tap] : w[((size_t)c * 32 + n) * 27 + (26 - tap)];
  }
  wp[i] = v;
}

template <int CIN>
__global__ void __launch_bounds__(256)
conv_thinin_kernel(const float* __restrict__ X, const float* __restrict__ Wp, const float* __restrict__ bias,
                   float* __restrict__ Y, ThinGeom g) {
  constexpr int KS = (27 * CIN + 1) / 2;
  const int lane = threadIdx.x & 63, li = lane & 31, lk = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float a[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) a[s] = Wp[s * 64 + lane];
  float bv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bv[r] = g.has_bias ? bias[(r & 3) + 8 * (r >> 2) + 4 * lk] : 0.0f;
  const int sH = g.W * CIN, sD = g.H * sH;
  const int nwaves = gridDim.x * 4;
  const float lo = g.relu ? 0.0f : -INFINITY;
  for (int tile = blockIdx.x * 4 + wave; tile < g.ntiles; tile += nwaves) {
    const int tw = tile % g.ntw, row = tile / g.ntw;
    const int h = row % g.H, bd = row / g.H, d = bd % g.D;
    const int w = tw * 32 + li;
    const float* px = X + (long)row * sH + (long)w * CIN;
    const bool okc = w < g.W, okl = w >= 1 && w - 1 < g.W, okr = w + 1 < g.W;
    float bval[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      // K index 2 s + lk: both candidates are compile-time (tap, c) pairs, the lane picks by lk
      const int k0 = 2 * s, k1 = 2 * s + 1;
      const int t0 = k0 / CIN, c0 = k0 % CIN, t1 = (k1 < 27 * CIN ? k1 : k0) / CIN, c1 = (k1 < 27 * CIN ? k1 : k0) % CIN;
      const int kd0 = t0 / 9, kh0 = (t0 / 3) % 3, kw0 = t0 % 3, kd1 = t1 / 9, kh1 = (t1 / 3) % 3, kw1 = t1 % 3;
      const bool r0 = (unsigned)(d + kd0 - 1) < (unsigned)g.D && (unsigned)(h + kh0 - 1) < (unsigned)g.H;
      const bool r1 = (unsigned)(d + kd1 - 1) < (unsigned)g.D && (unsigned)(h + kh1 - 1) < (unsigned)g.H && k1 < 27 * CIN;
      const int o0 = (kd0 - 1) * sD + (kh0 - 1) * sH + (kw0 - 1) * CIN + c0;
      const int o1 = (kd1 - 1) * sD + (kh1 - 1) * sH + (kw1 - 1) * CIN + c1;
      const bool c0ok = kw0 == 0 ? okl : (kw0 == 1 ? okc : okr), c1ok = kw1 == 0 ? okl : (kw1 == 1 ? okc : okr);
      const bool ok = lk ? (r1 && c1ok) : (r0 && c0ok);
      const float* q = ok ? px + (lk ? o1 : o0) : X;       // both arms in the global address space (no flat loads)
      const float v = *q;
      bval[s] = ok ? v : 0.0f;
    }
    __builtin_amdgcn_sched_barrier(0);          // all gathers in flight before the first MFMA waits on one
    v16f acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bv[r];
#pragma unroll
    for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bval[s], acc, 0, 0, 0);
    if (okc) {
      float* py = Y + ((long)row * g.W + w) * 32 + 4 * lk;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v4f o = {fmaxf(acc[4 * q], lo), fmaxf(acc[4 * q + 1], lo), fmaxf(acc[4 * q + 2], lo), fmaxf(acc[4 * q + 3], lo)};
        *reinterpret_cast<v4f*>(py + 8 * q) = o;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ 32 -> thin
// Wp[(t * 16 + s) * 64 + lane]: A operand of M-tile t, k-step s: row li -> group G = 8 t + 2 (li >> 3) + ((li >> 2) & 1)
// = (kd * 3 + kh) * NOUT + n, kw = li & 3 (row 3 of a group is padding); channel 8 (s >> 2) + 4 lk + (s & 3).
//   mode 0 (forward of a 32 -> NOUT layer):        Weff = w[(n * 32 + c) * 27 + tap]
//   mode 1 (data gradient of a NOUT -> 32 layer):  Weff = w[(c * NOUT + n) * 27 + 26 - tap]
__global__ void __launch_bounds__(256)
pack_thinout_kernel(const float* __restrict__ w, float* __restrict__ wp, int nout, int mt, int mode) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= mt * 16 * 64) return;
  const int lane = i & 63, s = (i >> 6) & 15, t = i >> 10, li = lane & 31, lk = lane >> 5;
  const int G = 8 * t + 2 * (li >> 3) + ((li >> 2) & 1), kw = li & 3, c = 8 * (s >> 2) + 4 * lk + (s & 3);
  float v = 0.0f;
  if (G < 9 * nout && kw < 3) {
    const int n = G % nout, kdh = G / nout, tap = kdh * 3 + kw;
    v = mode == 0 ? w[((size_t)n * 32 + c) * 27 + tap] : w[((size_t)c * nout + n) * 27 + (26 - tap)];
  }
  wp[i] = v;
}

// Tile = 32 input voxels w0 - 1 .. w0 + 30 of one row (w0 = 30 tw); lanes 1..30 own the outputs w0 .. w0 + 29.
template <int NOUT>
__global__ void __launch_bounds__(256)
conv_thinout_u_kernel(const float* __restrict__ X, const float* __restrict__ Wp, float* __restrict__ U, ThinGeom g) {
  constexpr int NG = 9 * NOUT, MT = (NG + 7) / 8;
  const int lane = threadIdx.x & 63, li = lane & 31, lk = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float a[MT][16];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int s = 0; s < 16; ++s) a[t][s] = Wp[(t * 16 + s) * 64 + lane];
  const int nwaves = gridDim.x * 4;
  const long nrows = (long)g.B * g.D * g.H;
  for (int tile = blockIdx.x * 4 + wave; tile < g.ntiles; tile += nwaves) {
    const int tw = tile % g.ntw, row = tile / g.ntw;
    const int wi = tw * 30 - 1 + li;
    const bool ok = (unsigned)wi < (unsigned)g.W;
    const v4f* px = reinterpret_cast<const v4f*>(X + ((long)row * g.W + (ok ? wi : 0)) * 32 + 4 * lk);
    v4f xq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) xq[q] = px[2 * q];
    if (!ok) {
#pragma unroll
      for (int q = 0; q < 4; ++q) xq[q] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
    }
    const bool outok = li >= 1 && li <= 30 && ok;
    float* pu = U + (long)row * g.W + wi;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      v16f acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
      for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][s], xq[s >> 2][s & 3], acc, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // group 8 t + 2 q + lk: T0 from the left neighbour, T2 from the right one
        const float u = __shfl_up(acc[4 * q], 1, 32) + acc[4 * q + 1] + __shfl_down(acc[4 * q + 2], 1, 32);
        const int G = 8 * t + 2 * q + lk;
        if (outok && G < NG) pu[(long)G * nrows * g.W] = u;
      }
    }
  }
}

// out[row][v][n] = bias[n] + sum_{kd,kh} U[(kd * 3 + kh) * NOUT + n][row + (kd - 1) H + (kh - 1)][v], zero outside the volume
template <int NOUT>
__global__ void __launch_bounds__(256)
conv_thinout_sum_kernel(const float* __restrict__ U, const float* __restrict__ bias, float* __restrict__ Y, ThinGeom g) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long nrows = (long)g.B * g.D * g.H, plane = nrows * g.W;
  if (i >= plane) return;
  const int v = (int)(i % g.W);
  const long row = i / g.W;
  const int h = (int)(row % g.H), d = (int)((row / g.H) % g.D);
  float acc[NOUT];
#pragma unroll
  for (int n = 0; n < NOUT; ++n) acc[n] = g.has_bias ? bias[n] : 0.0f;
#pragma unroll
  for (int kd = 0; kd < 3; ++kd)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const bool ok = (unsigned)(d + kd - 1) < (unsigned)g.D && (unsigned)(h + kh - 1) < (unsigned)g.H;
      const long src = ok ? (row + (kd - 1) * g.H + (kh - 1)) * g.W + v : 0;
#pragma unroll
      for (int n = 0; n < NOUT; ++n) {
        const float u = U[(long)((kd * 3 + kh) * NOUT + n) * plane + src];
        acc[n] += ok ? u : 0.0f;
      }
    }
#pragma unroll
  for (int n = 0; n < NOUT; ++n) Y[i * NOUT + n] = g.relu ? fmaxf(acc[n], 0.0f) : acc[n];
}

// ------------------------------------------------------------------------------------------------ weight gradient
struct ThinWgGeom {
  int B, D, H, W;
  int rows_per_wave, nchunks;     // chunk = rows_per_wave consecutive rows (b, d, h) of the volume, one per wave
};

// Wide[v][32], Thin[v][NT].  SGN = -1: wide = x, thin = gy (32 -> NT layer), thin voxel = v - off(tap);
// SGN = +1: wide = gy, thin = x (NT -> 32 layer), thin voxel = v + off(tap).
// ws[(chunk * MT + t) * 1024 + row * 32 + col]: row m = 32 t + row = tap * NT + j, col = wide channel.
template <int NT, int SGN>
__global__ void __launch_bounds__(256)
wgrad_thinside_kernel(const float* __restrict__ Wide, const float* __restrict__ Thin, float* __restrict__ ws, ThinWgGeom g) {
  constexpr int MT = (27 * NT + 31) / 32, UN = MT == 1 ? 16 : 8;
  const int lane = threadIdx.x & 63, li = lane & 31, lk = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int chunk = blockIdx.x * 4 + wave;
  if (chunk >= g.nchunks) return;
  int dd[MT], dh[MT], dwv[MT], jj[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int m = 32 * t + li, tap = m / NT;
    const bool mv = m < 27 * NT;
    dd[t] = mv ? SGN * (tap / 9 - 1) : 1 << 20;            // an invalid row index for the padding rows
    dh[t] = SGN * ((tap / 3) % 3 - 1);
    dwv[t] = SGN * (tap % 3 - 1);
    jj[t] = m % NT;
  }
  v16f acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  const int nrows = g.B * g.D * g.H;
  const int r0 = chunk * g.rows_per_wave, r1 = min(nrows, r0 + g.rows_per_wave);
  const int nsteps = (g.W + 1) / 2;
  for (int row = r0; row < r1; ++row) {
    const int h = row % g.H, d = (row / g.H) % g.D;
    const float* pw = Wide + (long)row * g.W * 32 + li;
    const float* pt[MT];
    bool rv[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      rv[t] = (unsigned)(d + dd[t]) < (unsigned)g.D && (unsigned)(h + dh[t]) < (unsigned)g.H;
      pt[t] = Thin + ((long)(row + dd[t] * g.H + dh[t]) * g.W + dwv[t]) * NT + jj[t];
    }
    for (int s0 = 0; s0 < nsteps; s0 += UN) {
      float bw[UN], at[MT][UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int w = 2 * (s0 + u) + lk;
        const bool wok = w < g.W;
        const float v = *(wok ? pw + w * 32 : Wide);
        bw[u] = wok ? v : 0.0f;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const bool ok = rv[t] && wok && (unsigned)(w + dwv[t]) < (unsigned)g.W;
          const float a = *(ok ? pt[t] + w * NT : Thin);
          at[t][u] = ok ? a : 0.0f;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < UN; ++u)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(at[t][u], bw[u], acc[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      ws[((long)chunk * MT + t) * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * lk) * 32 + li] = acc[t][r];
}

// gw (torch layout [Cout][Cin][27]) <- sum over the chunks.  One 1024-thread workgroup per row m = tap * NT + j of the
// partial tiles: thread (col, slice) adds the chunks c = slice (mod 32) in ascending order (128-byte coalesced reads), the
// 32 slice sums are folded in fixed order -- run-to-run deterministic.  thin_is_out: 32 -> NT layer (n = j, c = col),
// else NT -> 32 layer (n = col, c = j).
__global__ void __launch_bounds__(1024)
wgrad_thinside_reduce_kernel(const float* __restrict__ ws, float* __restrict__ gw, int nchunks, int nt, int mt, int thin_is_out) {
  __shared__ float part[32][33];
  const int m = blockIdx.x, col = threadIdx.x & 31, slice = threadIdx.x >> 5, tap = m / nt, j = m % nt;
  const float* p = ws + (long)(m >> 5) * 1024 + (m & 31) * 32 + col;
  const long cs = (long)mt * 1024;
  float s0 = 0.0f, s1 = 0.0f;
  int c = slice;
  for (; c + 32 < nchunks; c += 64) {
    s0 += p[c * cs];
    s1 += p[(c + 32) * cs];
  }
  if (c < nchunks) s0 += p[c * cs];
  part[slice][col] = s0 + s1;
  __syncthreads();
  if (slice == 0) {
    float tot = 0.0f;
#pragma unroll
    for (int k = 0; k < 32; ++k) tot += part[k][col];
    if (thin_is_out) gw[((long)j * 32 + col) * 27 + tap] = tot;
    else gw[((long)col * nt + j) * 27 + tap] = tot;
  }
}

ThinGeom make_geom(const ssbev_conv_dims* d, int mode) {
  ThinGeom g;
  g.B = d->B; g.D = d->Do; g.H = d->Ho; g.W = d->Wo;
  g.ntw = (g.W + 31) / 32;
  g.ntiles = g.B * g.D * g.H * g.ntw;
  g.relu = mode == 0 ? d->relu : 0;
  g.has_bias = 0;
  return g;
}

bool same_3x3x3(const ssbev_conv_dims* d) {
  if (d->transposed || d->kd != 3 || d->kh != 3 || d->kw != 3 || d->sd != 1 || d->sh != 1 || d->sw != 1) return false;
  if (d->pd != 1 || d->ph != 1 || d->pw != 1 || d->dd != 1 || d->dh != 1 || d->dw != 1) return false;
  if (d->Di != d->Do || d->Hi != d->Ho || d->Wi != d->Wo || d->accumulate) return false;
  if (d->tile_hint == 6 || d->tile_hint == 8 || d->tile_hint == 9) return false;    // hints that select the older kernel families
  return (long)d->B * d->Do * d->Ho * d->Wo * 32 < (1L << 31);       // 32-bit element offsets inside the kernels
}

}  // namespace

namespace ssbev_thin {

bool thinin_applicable(const ssbev_conv_dims* d, int mode) {
  if (!same_3x3x3(d)) return false;
  const int K = mode == 0 ? d->Cin : d->Cout, N = mode == 0 ? d->Cout : d->Cin;
  return N == 32 && (K == 1 || K == 2 || K == 4);
}

int thinin_pack(const float* w, float* wp, const ssbev_conv_dims* d, int mode, hipStream_t st) {
  const int K = mode == 0 ? d->Cin : d->Cout;
  const int KS = (27 * K + 1) / 2;
  hipLaunchKernelGGL(pack_thinin_kernel, dim3(cdiv((size_t)KS * 64, 256)), dim3(256), 0, st, w, wp, K, mode);
  return ssbev_launch_status();
}

int thinin_launch(const float* x, const float* wp, const float* bias, float* y, const ssbev_conv_dims* d, int mode,
                  hipStream_t st) {
  ThinGeom g = make_geom(d, mode);
  g.has_bias = (mode == 0 && bias) ? 1 : 0;
  const int K = mode == 0 ? d->Cin : d->Cout;
  const unsigned wgs = (unsigned)std::min<long>((g.ntiles + 3) / 4, 2048);
  if (K == 1)
    hipLaunchKernelGGL(conv_thinin_kernel<1>, dim3(wgs), dim3(256), 0, st, x, wp, bias, y, g);
  else if (K == 2)
    hipLaunchKernelGGL(conv_thinin_kernel<2>, dim3(wgs), dim3(256), 0, st, x, wp, bias, y, g);
  else
    hipLaunchKernelGGL(conv_thinin_kernel<4>, dim3(wgs), dim3(256), 0, st, x, wp, bias, y, g);
  return ssbev_launch_status();
}

bool thinout_applicable(const ssbev_conv_dims* d, int mode) {
  if (!same_3x3x3(d)) return false;
  const int K = mode == 0 ? d->Cin : d->Cout, N = mode == 0 ? d->Cout : d->Cin;
  return K == 32 && (N == 1 || N == 2 || N == 4);
}

size_t thinout_workspace(const ssbev_conv_dims* d, int mode) {
  const int N = mode == 0 ? d->Cout : d->Cin;
  return (size_t)9 * N * d->B * d->Do * d->Ho * d->Wo * sizeof(float);
}

size_t thinout_packed_elems(const ssbev_conv_dims* d, int mode) {
  const int N = mode == 0 ? d->Cout : d->Cin;
  return (size_t)((9 * N + 7) / 8) * 16 * 64;
}

int thinout_pack(const float* w, float* wp, const ssbev_conv_dims* d, int mode, hipStream_t st) {
  const int N = mode == 0 ? d->Cout : d->Cin, mt = (9 * N + 7) / 8;
  hipLaunchKernelGGL(pack_thinout_kernel, dim3(cdiv((size_t)mt * 16 * 64, 256)), dim3(256), 0, st, w, wp, N, mt, mode);
  return ssbev_launch_status();
}

template <int NOUT>
static void thinout_run(const float* x, const float* wp, const float* bias, float* y, float* u, ThinGeom g, hipStream_t st) {
  ThinGeom gu = g;
  gu.ntw = (g.W + 29) / 30;
  gu.ntiles = g.B * g.D * g.H * gu.ntw;
  const unsigned wgs = (unsigned)std::min<long>((gu.ntiles + 3) / 4, 2048);
  hipLaunchKernelGGL(conv_thinout_u_kernel<NOUT>, dim3(wgs), dim3(256), 0, st, x, wp, u, gu);
  const long plane = (long)g.B * g.D * g.H * g.W;
  hipLaunchKernelGGL(conv_thinout_sum_kernel<NOUT>, dim3(cdiv((size_t)plane, 256)), dim3(256), 0, st, u, bias, y, g);
}

int thinout_launch(const float* x, const float* wp, const float* bias, float* y, const ssbev_conv_dims* d, int mode,
                   void* ws, size_t ws_bytes, hipStream_t st) {
  if (!ws || ws_bytes < thinout_workspace(d, mode)) return SSBEV_EWORKSPACE;
  ThinGeom g = make_geom(d, mode);
  g.has_bias = (mode == 0 && bias) ? 1 : 0;
  const int N = mode == 0 ? d->Cout : d->Cin;
  float* u = static_cast<float*>(ws);
  if (N == 1) thinout_run<1>(x, wp, bias, y, u, g, st);
  else if (N == 2) thinout_run<2>(x, wp, bias, y, u, g, st);
  else thinout_run<4>(x, wp, bias, y, u, g, st);
  return ssbev_launch_status();
}

static ThinWgGeom make_wg_geom(const ssbev_conv_dims* d) {
  ThinWgGeom g;
  g.B = d->B; g.D = d->Do; g.H = d->Ho; g.W = d->Wo;
  const int nrows = g.B * g.D * g.H;
  g.rows_per_wave = std::max(1, (nrows + 4607) / 4608);
  g.nchunks = (nrows + g.rows_per_wave - 1) / g.rows_per_wave;
  return g;
}

bool wgrad_applicable(const ssbev_conv_dims* d) {
  if (!same_3x3x3(d) || d->tile_hint == 7) return false;
  const int a = d->Cin, b = d->Cout;
  return (a == 32 && (b == 1 || b == 2 || b == 4)) || (b == 32 && (a == 1 || a == 2 || a == 4));
}

size_t wgrad_workspace(const ssbev_conv_dims* d) {
  const ThinWgGeom g = make_wg_geom(d);
  const int nt = d->Cin == 32 ? d->Cout : d->Cin, mt = (27 * nt + 31) / 32;
  return (size_t)g.nchunks * mt * 1024 * sizeof(float);
}

template <int NT, int SGN>
static void wgrad_run(const float* wide, const float* thin, float* ws, const ThinWgGeom& g, hipStream_t st) {
  hipLaunchKernelGGL((wgrad_thinside_kernel<NT, SGN>), dim3((g.nchunks + 3) / 4), dim3(256), 0, st, wide, thin, ws, g);
}

int wgrad_launch(const float* x, const float* gy, float* gw, const ssbev_conv_dims* d, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!ws || ws_bytes < wgrad_workspace(d)) return SSBEV_EWORKSPACE;
  const ThinWgGeom g = make_wg_geom(d);
  const bool thin_out = d->Cin == 32 && d->Cout != 32;
  const int nt = thin_out ? d->Cout : d->Cin, mt = (27 * nt + 31) / 32;
  float* wsf = static_cast<float*>(ws);
  if (thin_out) {
    if (nt == 1) wgrad_run<1, -1>(x, gy, wsf, g, st);
    else if (nt == 2) wgrad_run<2, -1>(x, gy, wsf, g, st);
    else wgrad_run<4, -1>(x, gy, wsf, g, st);
  } else {
    if (nt == 1) wgrad_run<1, 1>(gy, x, wsf, g, st);
    else if (nt == 2) wgrad_run<2, 1>(gy, x, wsf, g, st);
    else wgrad_run<4, 1>(gy, x, wsf, g, st);
  }
  hipLaunchKernelGGL(wgrad_thinside_reduce_kernel, dim3(27 * nt), dim3(1024), 0, st, wsf, gw, g.nchunks, nt, mt,
                     thin_out ? 1 : 0);
  return ssbev_launch_status();
}

}  // namespace ssbev_thin

extern "C" {

size_t ssbev_conv_thin_workspace(const ssbev_conv_dims* d, int mode) {
  if (!d || (mode != 0 && mode != 1) || !ssbev_thin::thinout_applicable(d, mode)) return 0;
  return ssbev_thin::thinout_workspace(d, mode);
}

size_t ssbev_conv_thin_packed_elems(const ssbev_conv_dims* d, int mode) {
  if (!d || (mode != 0 && mode != 1) || !ssbev_thin::thinout_applicable(d, mode)) return 0;
  return ssbev_thin::thinout_packed_elems(d, mode);
}

int ssbev_conv_thin_pack(const float* w_src, float* w_packed, const ssbev_conv_dims* d, int mode, ssbev_stream_t stream) {
  if (!d || !w_src || !w_packed || (mode != 0 && mode != 1) || !ssbev_thin::thinout_applicable(d, mode)) return SSBEV_EINVAL;
  return ssbev_thin::thinout_pack(w_src, w_packed, d, mode, as_stream(stream));
}

int ssbev_conv_thin_run(const float* x, const float* w_packed, const float* bias, float* y, const ssbev_conv_dims* d,
                        int mode, void* workspace, size_t ws_bytes, ssbev_stream_t stream) {
  if (!d || !x || !w_packed || !y || (mode != 0 && mode != 1) || !ssbev_thin::thinout_applicable(d, mode)) return SSBEV_EINVAL;
  return ssbev_thin::thinout_launch(x, w_packed, bias, y, d, mode, workspace, ws_bytes, as_stream(stream));
}

}  // extern "C"
