// Winograd F(2x2x2, 3x3x3) transforms for the wide stride-1 3x3x3 convolutions (voxel encoder 128..512 channels, the
// 384 -> 192 occupancy-head conv).  y = A^T [ (G g G^T) (.) (B^T d B) ] A along each of the three axes: 64 multiplies
// per 8 outputs instead of 216 (3.375x fewer MACs); the element-wise stage is 64 independent [tiles x Cin] x [Cin x Cout]
// GEMMs.  The transforms use only additions (B, A have entries 0 / +-1), so fp32 results agree with the direct
// convolution to ~1e-6 relative.  All three kernels are pure streaming passes: one thread per (tile, channel), lanes
// along the channel axis (coalesced at every one of the 64 positions / frequencies).
//   input transform   x [B,D,H,W,C]      -> V [64][T][C]      V = B^T d B,  d = 4x4x4 input tile at (2i-1, 2j-1, 2k-1)
//   output transform  M [64][T][C]       -> y [B,D,H,W,C]     y = A^T M A   (2x2x2 outputs per tile)
//   output adjoint    g [B,D,H,W,C]      -> Z [64][T][C]      Z = A g A^T   (weight gradient: gU = sum_t V (.) Z)
// T = B * D/2 * H/2 * W/2 (even extents).
#include "common.h"

#include <algorithm>
#include <cstdlib>

namespace {

struct WinoGeom { int B, D, H, W, C; int acc = 0; };     // acc: the output transforms ADD to y (gradient slots, functional.fork)
__device__ const float kWinoZeros[4] = {0.f, 0.f, 0.f, 0.f};

// Storage type of the transformed-domain tensors (V, M, Z): float, or bf16 bit patterns (the `_bf16` entry points: half
// the HBM traffic of the streaming passes, and the frequency GEMMs then run on the bf16 matrix pipe with fp32
// accumulation; round-to-nearest-even on store via v_cvt_pk_bf16_f32).
typedef unsigned short bf16_bits;
__device__ __forceinline__ float fload(const float* p) { return *p; }
__device__ __forceinline__ float fload(const bf16_bits* p) { return __uint_as_float((unsigned)*p << 16); }
__device__ __forceinline__ void fstore(float* p, float v) { *p = v; }
__device__ __forceinline__ void fstore(bf16_bits* p, float v) { *p = __builtin_bit_cast(bf16_bits, (__bf16)v); }

// 1-D F(2,3) transforms, in place on 4 values with stride `s`
__device__ __forceinline__ void bt4(float* v, int s) {        // B^T d
  const float d0 = v[0], d1 = v[s], d2 = v[2 * s], d3 = v[3 * s];
  v[0] = d0 - d2; v[s] = d1 + d2; v[2 * s] = d2 - d1; v[3 * s] = d1 - d3;
}
__device__ __forceinline__ void a4(float* v, int s) {         // A g : 2 -> 4 values (input in v[0], v[s])
  const float g0 = v[0], g1 = v[s];
  v[0] = g0; v[s] = g0 + g1; v[2 * s] = g0 - g1; v[3 * s] = -g1;
}

// NV channels per thread (consecutive: one packed access).  NV = 2 is used by the all-bf16 variants: a thread moves 4 bytes per
// access like the fp32 kernels do with NV = 1 (2-byte lanes left the streaming passes instruction-bound: the bf16 input transform
// took 346 us where the fp32 one takes 185 -- profiles/r4c_summary_bf16_storage_b2_first.txt).
template <int NV, typename T>
__device__ __forceinline__ void ldv(const T* p, float (&o)[NV]) {
  if constexpr (NV == 1) {
    o[0] = ld1(p);
  } else if constexpr (sizeof(T) == 2) {
    const unsigned u = *reinterpret_cast<const unsigned*>(p);
    o[0] = __uint_as_float(u << 16); o[1] = __uint_as_float(u & 0xffff0000u);
  } else {
    const float2 f = *reinterpret_cast<const float2*>(p);
    o[0] = f.x; o[1] = f.y;
  }
}
template <int NV, typename T>
__device__ __forceinline__ void stv(T* p, const float (&v)[NV]) {
  if constexpr (NV == 1) {
    st1(p, v[0]);
  } else if constexpr (sizeof(T) == 2) {
    *reinterpret_cast<unsigned*>(p) = pack_bf2(v[0], v[1]);
  } else {
    *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
  }
}

// decode thread i -> (channel c, tile, td, th, tw, b) for the 3-D tilings / (c, tile, th, tw, bd) for the 2-D ones
template <int NV>
__device__ __forceinline__ void wino_decode3(long i, const WinoGeom& g, int& c, long& tile, int& tw, int& th, int& td, int& b) {
  const int cq = g.C / NV;
  c = NV * (int)(i % cq);
  long t = i / cq;
  tile = t;
  tw = (int)(t % (g.W / 2)); t /= g.W / 2;
  th = (int)(t % (g.H / 2)); t /= g.H / 2;
  td = (int)(t % (g.D / 2));
  b = (int)(t / (g.D / 2));
}
template <int NV>
__device__ __forceinline__ void wino_decode2(long i, const WinoGeom& g, int& c, long& tile, int& tw, int& th, long& bd) {
  const int cq = g.C / NV;
  c = NV * (int)(i % cq);
  long t = i / cq;
  tile = t;
  tw = (int)(t % (g.W / 2)); t /= g.W / 2;
  th = (int)(t % (g.H / 2));
  bd = t / (g.H / 2);
}

template <typename TF, typename TA = float, int NV = 1>
__global__ void __launch_bounds__(256)
wino_input_kernel(const TA* __restrict__ x, TF* __restrict__ V, WinoGeom g, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int c, tw, th, td, b;
  long tile;
  wino_decode3<NV>(i, g, c, tile, tw, th, td, b);
  float v[NV][64];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int d = 2 * td - 1 + a;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int h = 2 * th - 1 + e;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const int w = 2 * tw - 1 + f;
        const bool ok = d >= 0 && d < g.D && h >= 0 && h < g.H && w >= 0 && w < g.W;
        float t[NV];
#pragma unroll
        for (int n = 0; n < NV; ++n) t[n] = 0.0f;
        if (ok) ldv<NV>(x + ((((long)b * g.D + d) * g.H + h) * g.W + w) * g.C + c, t);
#pragma unroll
        for (int n = 0; n < NV; ++n) v[n][(a * 4 + e) * 4 + f] = t[n];
      }
    }
  }
#pragma unroll
  for (int n = 0; n < NV; ++n) {
#pragma unroll
    for (int p = 0; p < 16; ++p) bt4(v[n] + p * 4, 1);                            // along w
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int f = 0; f < 4; ++f) bt4(v[n] + a * 16 + f, 4);                        // along h
#pragma unroll
    for (int p = 0; p < 16; ++p) bt4(v[n] + p, 16);                               // along d
  }
  const long T = total / (g.C / NV);
#pragma unroll
  for (int xi = 0; xi < 64; ++xi) {
    float t[NV];
#pragma unroll
    for (int n = 0; n < NV; ++n) t[n] = v[n][xi];
    stv<NV>(V + ((long)xi * T + tile) * g.C + c, t);
  }
}

template <typename TF, typename TA = float, int NV = 1>
__global__ void __launch_bounds__(256)
wino_output_kernel(const TF* __restrict__ M, TA* __restrict__ y, WinoGeom g, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int c, tw, th, td, b;
  long tile;
  wino_decode3<NV>(i, g, c, tile, tw, th, td, b);
  const long T = total / (g.C / NV);
  float m[NV][64];
#pragma unroll
  for (int xi = 0; xi < 64; ++xi) {
    float t[NV];
    ldv<NV>(M + ((long)xi * T + tile) * g.C + c, t);
#pragma unroll
    for (int n = 0; n < NV; ++n) m[n][xi] = t[n];
  }
  float out[NV][8];                                  // (dz, e, f)
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    // A^T along w: 4 -> 2   (y0 = m0 + m1 + m2, y1 = m1 - m2 - m3)
    float r1[32];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      r1[p * 2 + 0] = m[n][p * 4] + m[n][p * 4 + 1] + m[n][p * 4 + 2];
      r1[p * 2 + 1] = m[n][p * 4 + 1] - m[n][p * 4 + 2] - m[n][p * 4 + 3];
    }
    float r2[16];                                  // along h: index (a*4 + e)*2 + f -> (a*2 + e')*2 + f
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const float m0 = r1[(a * 4 + 0) * 2 + f], m1 = r1[(a * 4 + 1) * 2 + f], m2 = r1[(a * 4 + 2) * 2 + f],
                    m3 = r1[(a * 4 + 3) * 2 + f];
        r2[(a * 2 + 0) * 2 + f] = m0 + m1 + m2;
        r2[(a * 2 + 1) * 2 + f] = m1 - m2 - m3;
      }
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const float m0 = r2[(0 * 2 + e) * 2 + f], m1 = r2[(1 * 2 + e) * 2 + f], m2 = r2[(2 * 2 + e) * 2 + f],
                    m3 = r2[(3 * 2 + e) * 2 + f];
        out[n][(0 * 2 + e) * 2 + f] = m0 + m1 + m2;
        out[n][(1 * 2 + e) * 2 + f] = m1 - m2 - m3;
      }
  }
#pragma unroll
  for (int dz = 0; dz < 2; ++dz)
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        float t[NV];
#pragma unroll
        for (int n = 0; n < NV; ++n) t[n] = out[n][(dz * 2 + e) * 2 + f];
        stv<NV>(y + ((((long)b * g.D + 2 * td + dz) * g.H + 2 * th + e) * g.W + 2 * tw + f) * g.C + c, t);
      }
}

template <typename TF, typename TA = float, int NV = 1>
__global__ void __launch_bounds__(256)
wino_output_adjoint_kernel(const TA* __restrict__ gy, TF* __restrict__ Z, WinoGeom g, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int c, tw, th, td, b;
  long tile;
  wino_decode3<NV>(i, g, c, tile, tw, th, td, b);
  float v[NV][64];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        float t[NV];
        ldv<NV>(gy + ((((long)b * g.D + 2 * td + a) * g.H + 2 * th + e) * g.W + 2 * tw + f) * g.C + c, t);
#pragma unroll
        for (int n = 0; n < NV; ++n) v[n][(a * 4 + e) * 4 + f] = t[n];
      }
#pragma unroll
  for (int n = 0; n < NV; ++n) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int e = 0; e < 2; ++e) a4(v[n] + (a * 4 + e) * 4, 1);                    // along w: 2 -> 4
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int f = 0; f < 4; ++f) a4(v[n] + a * 16 + f, 4);                         // along h
#pragma unroll
    for (int p = 0; p < 16; ++p) a4(v[n] + p, 16);                                // along d
  }
  const long T = total / (g.C / NV);
#pragma unroll
  for (int xi = 0; xi < 64; ++xi) {
    float t[NV];
#pragma unroll
    for (int n = 0; n < NV; ++n) t[n] = v[n][xi];
    stv<NV>(Z + ((long)xi * T + tile) * g.C + c, t);
  }
}

// ---- 2-D variant, F(2x2, 3x3): 16 frequencies, tiles over (h, w); the D axis of the dims struct is a batch axis ----
template <typename TF, typename TA = float, int NV = 1>
__global__ void __launch_bounds__(256)
wino2d_input_kernel(const TA* __restrict__ x, TF* __restrict__ V, WinoGeom g, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int c, tw, th;
  long tile, bd;
  wino_decode2<NV>(i, g, c, tile, tw, th, bd);
  float v[NV][16];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int h = 2 * th - 1 + e;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int w = 2 * tw - 1 + f;
      const bool ok = h >= 0 && h < g.H && w >= 0 && w < g.W;
      float t[NV];
#pragma unroll
      for (int n = 0; n < NV; ++n) t[n] = 0.0f;
      if (ok) ldv<NV>(x + ((bd * g.H + h) * g.W + w) * g.C + c, t);
#pragma unroll
      for (int n = 0; n < NV; ++n) v[n][e * 4 + f] = t[n];
    }
  }
#pragma unroll
  for (int n = 0; n < NV; ++n) {
#pragma unroll
    for (int e = 0; e < 4; ++e) bt4(v[n] + e * 4, 1);
#pragma unroll
    for (int f = 0; f < 4; ++f) bt4(v[n] + f, 4);
  }
  const long T = total / (g.C / NV);
#pragma unroll
  for (int xi = 0; xi < 16; ++xi) {
    float t[NV];
#pragma unroll
    for (int n = 0; n < NV; ++n) t[n] = v[n][xi];
    stv<NV>(V + ((long)xi * T + tile) * g.C + c, t);
  }
}

template <typename TF, typename TA = float, int NV = 1>
__global__ void __launch_bounds__(256)
wino2d_output_kernel(const TF* __restrict__ M, TA* __restrict__ y, WinoGeom g, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int c, tw, th;
  long tile, bd;
  wino_decode2<NV>(i, g, c, tile, tw, th, bd);
  const long T = total / (g.C / NV);
  float m[NV][16];
#pragma unroll
  for (int xi = 0; xi < 16; ++xi) {
    float t[NV];
    ldv<NV>(M + ((long)xi * T + tile) * g.C + c, t);
#pragma unroll
    for (int n = 0; n < NV; ++n) m[n][xi] = t[n];
  }
  float oldv[NV][4];
#pragma unroll
  for (int n = 0; n < NV; ++n)
#pragma unroll
    for (int k = 0; k < 4; ++k) oldv[n][k] = 0.0f;
  if (g.acc) {                                       // y += result: old values first, then the stores
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const long base = ((bd * g.H + 2 * th) * g.W + 2 * tw + f) * g.C + c;
      float t0[NV], t1[NV];
      ldv<NV>(y + base, t0);
      ldv<NV>(y + base + (long)g.W * g.C, t1);
#pragma unroll
      for (int n = 0; n < NV; ++n) { oldv[n][2 * f] = t0[n]; oldv[n][2 * f + 1] = t1[n]; }
    }
  }
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    float y0[NV], y1[NV];
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      float r[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        r[e * 2 + 0] = m[n][e * 4] + m[n][e * 4 + 1] + m[n][e * 4 + 2];
        r[e * 2 + 1] = m[n][e * 4 + 1] - m[n][e * 4 + 2] - m[n][e * 4 + 3];
      }
      y0[n] = r[0 * 2 + f] + r[1 * 2 + f] + r[2 * 2 + f] + oldv[n][2 * f];
      y1[n] = r[1 * 2 + f] - r[2 * 2 + f] - r[3 * 2 + f] + oldv[n][2 * f + 1];
    }
    const long base = ((bd * g.H + 2 * th) * g.W + 2 * tw + f) * g.C + c;
    stv<NV>(y + base, y0);
    stv<NV>(y + base + (long)g.W * g.C, y1);
  }
}

template <typename TF, typename TA = float, int NV = 1>
__global__ void __launch_bounds__(256)
wino2d_output_adjoint_kernel(const TA* __restrict__ gy, TF* __restrict__ Z, WinoGeom g, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int c, tw, th;
  long tile, bd;
  wino_decode2<NV>(i, g, c, tile, tw, th, bd);
  float v[NV][16];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      float t[NV];
      ldv<NV>(gy + ((bd * g.H + 2 * th + e) * g.W + 2 * tw + f) * g.C + c, t);
#pragma unroll
      for (int n = 0; n < NV; ++n) v[n][e * 4 + f] = t[n];
    }
#pragma unroll
  for (int n = 0; n < NV; ++n) {
#pragma unroll
    for (int e = 0; e < 2; ++e) a4(v[n] + e * 4, 1);
#pragma unroll
    for (int f = 0; f < 4; ++f) a4(v[n] + f, 4);
  }
  const long T = total / (g.C / NV);
#pragma unroll
  for (int xi = 0; xi < 16; ++xi) {
    float t[NV];
#pragma unroll
    for (int n = 0; n < NV; ++n) t[n] = v[n][xi];
    stv<NV>(Z + ((long)xi * T + tile) * g.C + c, t);
  }
}

bool wino2d_ok(const ssbev_wino_dims* d) {
  return d && d->B > 0 && d->C > 0 && d->D > 0 && d->H > 0 && d->W > 0 && d->H % 2 == 0 && d->W % 2 == 0;
}

// ---- weight transforms (tiny: one thread per (cin, cout) pair) ---------------------------------------------------
// G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]];  U = G g G^T per axis.  w is the torch layout [Cout][Cin][taps].
__device__ __forceinline__ void g4(const float* g, int s, float* o, int so) {      // 3 -> 4
  const float g0 = g[0], g1 = g[s], g2 = g[2 * s];
  o[0] = g0; o[so] = 0.5f * (g0 + g1 + g2); o[2 * so] = 0.5f * (g0 - g1 + g2); o[3 * so] = g2;
}
__device__ __forceinline__ void gt3(const float* u, int s, float* o, int so) {     // G^T u : 4 -> 3
  const float u0 = u[0], u1 = u[s], u2 = u[2 * s], u3 = u[3 * s];
  o[0] = u0 + 0.5f * (u1 + u2); o[so] = 0.5f * (u1 - u2); o[2 * so] = 0.5f * (u1 + u2) + u3;
}

// mode 0: U[xi][ci][co] = G w[co][ci] G^T          (forward)            U is [NF][Cin][Cout]
// mode 1: U[xi][co][ci] = G flip(w[co][ci]) G^T    (data gradient)      U is [NF][Cout][Cin]
template <int ND>
__global__ void __launch_bounds__(256)
wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin, int mode) {
  constexpr int TAPS = ND == 3 ? 27 : 9, NF = ND == 3 ? 64 : 16;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Cout * Cin) return;
  const int co = mode == 0 ? i % Cout : i / Cin, ci = mode == 0 ? i / Cout : i % Cin;   // fastest index = U's last axis
  float g[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t) g[t] = w[((size_t)co * Cin + ci) * TAPS + (mode == 0 ? t : TAPS - 1 - t)];
  float u[NF];
  if (ND == 3) {
    float a[36], b[48];                       // [3][3][4], [3][4][4]
#pragma unroll
    for (int p = 0; p < 9; ++p) g4(g + p * 3, 1, a + p * 4, 1);                              // w axis
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int f = 0; f < 4; ++f) g4(a + d * 12 + f, 4, b + d * 16 + f, 4);                   // h axis
#pragma unroll
    for (int p = 0; p < 16; ++p) g4(b + p, 16, u + p, 16);                                  // d axis
  } else {
    float a[12];
#pragma unroll
    for (int p = 0; p < 3; ++p) g4(g + p * 3, 1, a + p * 4, 1);
#pragma unroll
    for (int f = 0; f < 4; ++f) g4(a + f, 4, u + f, 4);
  }
  const size_t plane = (size_t)Cout * Cin;
  const size_t pos = mode == 0 ? (size_t)ci * Cout + co : (size_t)co * Cin + ci;
#pragma unroll
  for (int xi = 0; xi < NF; ++xi) U[xi * plane + pos] = u[xi];
}

// gw[co][ci][taps] = G^T gU[.][ci][co] G  (weight gradient; gU is [NF][Cin][Cout])
template <int ND>
__global__ void __launch_bounds__(256)
wino_weight_grad_kernel(const float* __restrict__ gU, float* __restrict__ gw, int Cout, int Cin) {
  constexpr int TAPS = ND == 3 ? 27 : 9, NF = ND == 3 ? 64 : 16;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Cout * Cin) return;
  const int co = i % Cout, ci = i / Cout;
  const size_t plane = (size_t)Cout * Cin;
  float u[NF];
#pragma unroll
  for (int xi = 0; xi < NF; ++xi) u[xi] = gU[xi * plane + (size_t)ci * Cout + co];
  float g[TAPS];
  if (ND == 3) {
    float a[48], b[36];                       // [4][4][3] after w, [4][3][3] after h
#pragma unroll
    for (int p = 0; p < 16; ++p) gt3(u + p * 4, 1, a + p * 3, 1);
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int f = 0; f < 3; ++f) gt3(a + d * 12 + f, 3, b + d * 9 + f, 3);
#pragma unroll
    for (int p = 0; p < 9; ++p) gt3(b + p, 9, g + p, 9);
  } else {
    float a[12];
#pragma unroll
    for (int p = 0; p < 4; ++p) gt3(u + p * 4, 1, a + p * 3, 1);
#pragma unroll
    for (int f = 0; f < 3; ++f) gt3(a + f, 3, g + f, 3);
  }
#pragma unroll
  for (int t = 0; t < TAPS; ++t) gw[((size_t)co * Cin + ci) * TAPS + t] = g[t];
}

// ---- F(4,3) along h and w, F(2,3) along d: "F(2x4x4, 3x3x3)" and its 2-D sibling F(4x4, 3x3) -------------------------
// The F(2,3)^3 pipeline above is bound by the 8x larger transformed tensors.  With 4-wide tiles along h and w the
// transformed domain is only 4.5x (2-D: 2.25x) the activation and the GEMM stage does 6x (2-D: 4x) fewer multiply-adds
// than the direct convolution: per 2x4x4 = 32 outputs, 4*6*6 = 144 frequencies.  Lavin & Gray's F(4,3) matrices; the
// transforms involve the constants 2, 4, 5, 8 and 1/4 .. 1/24, so fp32 results carry ~1e-6 .. 1e-5 relative error against
// the direct kernel (tests: 2e-4 gate; the 1e-3 logits contract holds with margin).
//   V / M / Z: [NF][T][C], frequency index xi = (a * 6 + e) * 6 + f  (a: d-axis F(2,3), e: h, f: w);
//   T = B * D/2 * H/4 * W/4 (3-D) or B * D * H/4 * W/4 (2-D, NF = 36, D a batch axis).
__device__ __forceinline__ void bt6(float* v, int s) {        // B^T d, 6 -> 6
  const float d0 = v[0], d1 = v[s], d2 = v[2 * s], d3 = v[3 * s], d4 = v[4 * s], d5 = v[5 * s];
  v[0] = 4.0f * d0 - 5.0f * d2 + d4;
  v[s] = -4.0f * d1 - 4.0f * d2 + d3 + d4;
  v[2 * s] = 4.0f * d1 - 4.0f * d2 - d3 + d4;
  v[3 * s] = -2.0f * d1 - d2 + 2.0f * d3 + d4;
  v[4 * s] = 2.0f * d1 - d2 - 2.0f * d3 + d4;
  v[5 * s] = 4.0f * d1 - 5.0f * d3 + d5;
}
__device__ __forceinline__ void at6(const float* m, int s, float* o, int so) {     // A^T m, 6 -> 4
  const float m0 = m[0], m1 = m[s], m2 = m[2 * s], m3 = m[3 * s], m4 = m[4 * s], m5 = m[5 * s];
  const float p = m1 + m2, q = m1 - m2, r = m3 + m4, t = m3 - m4;
  o[0] = m0 + p + r;
  o[so] = q + 2.0f * t;
  o[2 * so] = p + 4.0f * r;
  o[3 * so] = q + 8.0f * t + m5;
}
__device__ __forceinline__ void a6(const float* g, int s, float* o, int so) {      // A g, 4 -> 6 (adjoint of at6)
  const float g0 = g[0], g1 = g[s], g2 = g[2 * s], g3 = g[3 * s];
  o[0] = g0;
  o[so] = g0 + g1 + g2 + g3;
  o[2 * so] = g0 - g1 + g2 - g3;
  o[3 * so] = g0 + 2.0f * g1 + 4.0f * g2 + 8.0f * g3;
  o[4 * so] = g0 - 2.0f * g1 + 4.0f * g2 - 8.0f * g3;
  o[5 * so] = g3;
}
__device__ __forceinline__ void g6(const float* g, int s, float* o, int so) {      // G g, 3 -> 6
  const float g0 = g[0], g1 = g[s], g2 = g[2 * s];
  o[0] = 0.25f * g0;
  o[so] = -(g0 + g1 + g2) * (1.0f / 6.0f);
  o[2 * so] = -(g0 - g1 + g2) * (1.0f / 6.0f);
  o[3 * so] = g0 * (1.0f / 24.0f) + g1 * (1.0f / 12.0f) + g2 * (1.0f / 6.0f);
  o[4 * so] = g0 * (1.0f / 24.0f) - g1 * (1.0f / 12.0f) + g2 * (1.0f / 6.0f);
  o[5 * so] = g2;
}
__device__ __forceinline__ void gt6(const float* u, int s, float* o, int so) {     // G^T u, 6 -> 3
  const float u0 = u[0], u1 = u[s], u2 = u[2 * s], u3 = u[3 * s], u4 = u[4 * s], u5 = u[5 * s];
  o[0] = 0.25f * u0 - (u1 + u2) * (1.0f / 6.0f) + (u3 + u4) * (1.0f / 24.0f);
  o[so] = (u2 - u1) * (1.0f / 6.0f) + (u3 - u4) * (1.0f / 12.0f);
  o[2 * so] = -(u1 + u2) * (1.0f / 6.0f) + (u3 + u4) * (1.0f / 6.0f) + u5;
}

// decode (tile, channel).  DA = outputs per tile along d: 0 (2-D: every plane is its own "tile"), 2 (F(2,3)) or 4 (F(4,3):
// F(4x4x4, 3x3x3), 216 frequencies per 64 outputs, 8x fewer multiply-adds, 3.375x transformed domain)
template <int DA>
__device__ __forceinline__ void tile43(long t, const WinoGeom& g, int& b_or_bd, int& td, int& th, int& tw) {
  tw = (int)(t % (g.W / 4)); t /= g.W / 4;
  th = (int)(t % (g.H / 4)); t /= g.H / 4;
  if (DA) { td = (int)(t % (g.D / DA)); b_or_bd = (int)(t / (g.D / DA)); }
  else { td = 0; b_or_bd = (int)t; }
}

template <int DA, typename TF>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, DA == 4 ? 1 : (DA == 2 ? 2 : 8))))
wino43_input_kernel(const float* __restrict__ x, TF* __restrict__ V, WinoGeom g, long total) {
  constexpr int NA = DA ? DA + 2 : 1, NF = NA * 36;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % g.C);
  const long tile = i / g.C;
  int b, td, th, tw;
  tile43<DA>(tile, g, b, td, th, tw);
  float v[NF];
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    const int d = DA ? DA * td - 1 + a : 0;
    const long plane = DA ? (long)b * g.D + d : (long)b;
    const bool dok = !DA || (d >= 0 && d < g.D);
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      const int h = 4 * th - 1 + e;
#pragma unroll
      for (int f = 0; f < 6; ++f) {
        const int w = 4 * tw - 1 + f;
        const bool ok = dok && h >= 0 && h < g.H && w >= 0 && w < g.W;
        v[(a * 6 + e) * 6 + f] = ok ? x[((plane * g.H + h) * g.W + w) * g.C + c] : 0.0f;
      }
    }
  }
#pragma unroll
  for (int p = 0; p < NA * 6; ++p) bt6(v + p * 6, 1);                            // along w
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int f = 0; f < 6; ++f) bt6(v + a * 36 + f, 6);                          // along h
  if (DA == 2) {
#pragma unroll
    for (int p = 0; p < 36; ++p) bt4(v + p, 36);                                 // along d (F(2,3))
  } else if (DA == 4) {
#pragma unroll
    for (int p = 0; p < 36; ++p) bt6(v + p, 36);                                 // along d (F(4,3))
  }
  const long T = total / g.C;
#pragma unroll
  for (int xi = 0; xi < NF; ++xi) fstore(V + ((long)xi * T + tile) * g.C + c, v[xi]);
}

template <int DA, typename TF>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, DA == 4 ? 1 : (DA == 2 ? 2 : 8))))
wino43_output_kernel(const TF* __restrict__ M, float* __restrict__ y, WinoGeom g, long total) {
  constexpr int NA = DA ? DA + 2 : 1, NF = NA * 36;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % g.C);
  const long tile = i / g.C;
  int b, td, th, tw;
  tile43<DA>(tile, g, b, td, th, tw);
  const long T = total / g.C;
  float m[NF];
#pragma unroll
  for (int xi = 0; xi < NF; ++xi) m[xi] = fload(M + ((long)xi * T + tile) * g.C + c);
  float r1[NA * 6 * 4];                              // [a][e][4] after w
#pragma unroll
  for (int p = 0; p < NA * 6; ++p) at6(m + p * 6, 1, r1 + p * 4, 1);
  float r2[NA * 4 * 4];                              // [a][4][4] after h
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int f = 0; f < 4; ++f) at6(r1 + a * 24 + f, 4, r2 + a * 16 + f, 4);
  if (DA == 0 && g.acc) {                            // y += result: every old value is requested before the first store
    float oldv[16];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int f = 0; f < 4; ++f) oldv[e * 4 + f] = y[(((long)b * g.H + 4 * th + e) * g.W + 4 * tw + f) * g.C + c];
#pragma unroll
    for (int k = 0; k < 16; ++k) r2[k] += oldv[k];
  }
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      if (DA == 2) {
        const float m0 = r2[0 * 16 + e * 4 + f], m1 = r2[1 * 16 + e * 4 + f], m2 = r2[2 * 16 + e * 4 + f], m3 = r2[3 * 16 + e * 4 + f];
        const long base = ((((long)b * g.D + 2 * td) * g.H + 4 * th + e) * g.W + 4 * tw + f) * g.C + c;
        y[base] = m0 + m1 + m2;
        y[base + (long)g.H * g.W * g.C] = m1 - m2 - m3;
      } else if (DA == 4) {
        float o[4];
        at6(r2 + e * 4 + f, 16, o, 1);
        const long base = ((((long)b * g.D + 4 * td) * g.H + 4 * th + e) * g.W + 4 * tw + f) * g.C + c;
#pragma unroll
        for (int a = 0; a < 4; ++a) y[base + (long)a * g.H * g.W * g.C] = o[a];
      } else {
        y[(((long)b * g.H + 4 * th + e) * g.W + 4 * tw + f) * g.C + c] = r2[e * 4 + f];
      }
    }
}

template <int DA, typename TF>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, DA == 4 ? 1 : (DA == 2 ? 2 : 8))))
wino43_output_adjoint_kernel(const float* __restrict__ gy, TF* __restrict__ Z, WinoGeom g, long total) {
  constexpr int NA = DA ? DA + 2 : 1, NG = DA ? DA : 1, NF = NA * 36;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % g.C);
  const long tile = i / g.C;
  int b, td, th, tw;
  tile43<DA>(tile, g, b, td, th, tw);
  float gin[NG * 16];
#pragma unroll
  for (int a = 0; a < NG; ++a) {
    const long plane = DA ? (long)b * g.D + DA * td + a : (long)b;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int f = 0; f < 4; ++f) gin[(a * 4 + e) * 4 + f] = gy[((plane * g.H + 4 * th + e) * g.W + 4 * tw + f) * g.C + c];
  }
  float r1[NG * 4 * 6];                              // [a][4][6] after w
#pragma unroll
  for (int p = 0; p < NG * 4; ++p) a6(gin + p * 4, 1, r1 + p * 6, 1);
  float v[NF];                                       // [a'][6][6]; 3-D: filled at a' = 0, 1 then expanded along d
#pragma unroll
  for (int a = 0; a < NG; ++a)
#pragma unroll
    for (int f = 0; f < 6; ++f) a6(r1 + a * 24 + f, 6, v + a * 36 + f, 6);
  if (DA == 2) {
#pragma unroll
    for (int p = 0; p < 36; ++p) a4(v + p, 36);                                  // along d: 2 -> 4
  } else if (DA == 4) {
#pragma unroll
    for (int p = 0; p < 36; ++p) {
      float t6[6];
      a6(v + p, 36, t6, 1);
#pragma unroll
      for (int a = 0; a < 6; ++a) v[a * 36 + p] = t6[a];
    }
  }
  const long T = total / g.C;
#pragma unroll
  for (int xi = 0; xi < NF; ++xi) fstore(Z + ((long)xi * T + tile) * g.C + c, v[xi]);
}

// weights: w [Cout][Cin][taps] -> U [NF][Cin][Cout] (mode 0) or [NF][Cout][Cin] with mirrored taps (mode 1)
template <int DA>
__global__ void __launch_bounds__(256)
wino43_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin, int mode) {
  constexpr int ND_ = DA ? 3 : 1, TAPS = ND_ * 9, NA = DA ? DA + 2 : 1, NF = NA * 36;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Cout * Cin) return;
  const int co = mode == 0 ? i % Cout : i / Cin, ci = mode == 0 ? i / Cout : i % Cin;
  float g[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t) g[t] = w[((size_t)co * Cin + ci) * TAPS + (mode == 0 ? t : TAPS - 1 - t)];
  float a1[ND_ * 3 * 6];                             // [kd][kh][6] after w
#pragma unroll
  for (int p = 0; p < ND_ * 3; ++p) g6(g + p * 3, 1, a1 + p * 6, 1);
  float a2[ND_ * 36];                                // [kd][6][6] after h
#pragma unroll
  for (int d = 0; d < ND_; ++d)
#pragma unroll
    for (int f = 0; f < 6; ++f) g6(a1 + d * 18 + f, 6, a2 + d * 36 + f, 6);
  float u[NF];
  if (DA == 2) {
#pragma unroll
    for (int p = 0; p < 36; ++p) g4(a2 + p, 36, u + p, 36);                      // along d: 3 -> 4
  } else if (DA == 4) {
#pragma unroll
    for (int p = 0; p < 36; ++p) g6(a2 + p, 36, u + p, 36);                      // along d: 3 -> 6
  } else {
#pragma unroll
    for (int p = 0; p < 36; ++p) u[p] = a2[p];
  }
  const size_t plane = (size_t)Cout * Cin;
  const size_t pos = mode == 0 ? (size_t)ci * Cout + co : (size_t)co * Cin + ci;
#pragma unroll
  for (int xi = 0; xi < NF; ++xi) U[xi * plane + pos] = u[xi];
}

template <int DA>
__global__ void __launch_bounds__(256)
wino43_weight_grad_kernel(const float* __restrict__ gU, float* __restrict__ gw, int Cout, int Cin) {
  constexpr int ND_ = DA ? 3 : 1, TAPS = ND_ * 9, NA = DA ? DA + 2 : 1, NF = NA * 36;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Cout * Cin) return;
  const int co = i % Cout, ci = i / Cout;
  const size_t plane = (size_t)Cout * Cin;
  float u[NF];
#pragma unroll
  for (int xi = 0; xi < NF; ++xi) u[xi] = gU[xi * plane + (size_t)ci * Cout + co];
  float a1[NA * 6 * 3];                              // [a][6][3] after w
#pragma unroll
  for (int p = 0; p < NA * 6; ++p) gt6(u + p * 6, 1, a1 + p * 3, 1);
  float a2[NA * 9];                                  // [a][3][3] after h
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int f = 0; f < 3; ++f) gt6(a1 + a * 18 + f, 3, a2 + a * 9 + f, 3);
  float g[TAPS];
  if (DA == 2) {
#pragma unroll
    for (int p = 0; p < 9; ++p) gt3(a2 + p, 9, g + p, 9);                        // along d: 4 -> 3
  } else if (DA == 4) {
#pragma unroll
    for (int p = 0; p < 9; ++p) gt6(a2 + p, 9, g + p, 9);                        // along d: 6 -> 3
  } else {
#pragma unroll
    for (int p = 0; p < 9; ++p) g[p] = a2[p];
  }
#pragma unroll
  for (int t = 0; t < TAPS; ++t) gw[((size_t)co * Cin + ci) * TAPS + t] = g[t];
}

bool wino43_ok(const ssbev_wino_dims* d, int da) {
  return d && d->B > 0 && d->C > 0 && d->D > 0 && d->H > 0 && d->W > 0 && d->H % 4 == 0 && d->W % 4 == 0 &&
         (da == 0 || d->D % da == 0);
}

// ---- depth-fused frequency GEMM -------------------------------------------------------------------------------------
// The plain pipeline materialises the 3-D transformed input V (8x the activation) and the 3-D transformed output M (8x)
// in HBM and is bound by exactly that traffic.  This kernel works on tensors transformed over (h, w) only (4x):
//   P  [16][B*D*Thw][Cin]   = wino2d_input(x)           Mo [16][B*D*Thw][Cout] -> wino2d_output -> y
// and does the depth axis of F(2,3) in registers: for a depth tile i (outputs 2i, 2i+1) a wave loads the four planes
// 2i-1 .. 2i+2 of P as MFMA A operands, forms the four depth frequencies v0 = p0 - p2, v1 = p1 + p2, v2 = p2 - p1,
// v3 = p1 - p3 with three adds per element, multiplies each with its own weight matrix U[xi_d, xi_hw] (4 accumulator
// groups), and combines them in the epilogue (o0 = m0 + m1 + m2, o1 = m1 - m2 - m3).  Same 3.375x MAC reduction, half the
// HBM traffic in the GEMM stage, no separate depth transform passes.
// One wave: 32 (h,w)-tiles of one (b, depth tile, xi_hw) x NT*32 output channels; A = one float4 of a tile's channels per
// lane (as in conv_gather_kernel), B = packed weights Wp[xi][q][kh][n][4].
typedef float wf32x16 __attribute__((ext_vector_type(16)));

struct WinoGemmGeom { int B, D, Thw, Cin, Cout, CoutPad; };

template <int MT, int NT>
__global__ void __launch_bounds__(256)
wino_dgemm_kernel(const float* __restrict__ P, const float* __restrict__ Wp, float* __restrict__ Mo, WinoGemmGeom g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const int tgroups = (g.Thw + 32 * MT - 1) / (32 * MT), tg4 = ((tgroups + 3) >> 2) << 2;
  // blockIdx.x: (b, depth tile, 4 tile groups) ; blockIdx.y: column group ; blockIdx.z: xi_hw
  const int tg = (blockIdx.x * 4 + wave) % tg4;
  const int bi = (blockIdx.x * 4 + wave) / tg4;
  if (tg >= tgroups) return;
  const int i = bi % (g.D / 2), b = bi / (g.D / 2);
  const int xhw = blockIdx.z, n0 = blockIdx.y * (NT * 32);
  const long R = (long)g.B * g.D * g.Thw;
  const float* pa[4][MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int tq = (tg * MT + mt) * 32 + li;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int d = 2 * i - 1 + a;
      pa[a][mt] = (tq < g.Thw && d >= 0 && d < g.D)
                      ? P + ((long)xhw * R + ((long)b * g.D + d) * g.Thw + tq) * g.Cin + 4 * lk : nullptr;
    }
  }
  wf32x16 acc[4][MT][NT];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][mt][nt][r] = 0.0f;
  const int Q = (g.Cin + 7) >> 3;
  const size_t fstride = (size_t)Q * 2 * g.CoutPad * 4;               // floats per frequency in Wp
  const float* wl = Wp + ((size_t)lk * g.CoutPad + n0 + li) * 4 + (size_t)xhw * fstride;
  for (int q = 0; q < Q; ++q) {
    const bool cok = (8 * q + 4 * lk) < g.Cin;
    float4 v[4][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float4 p[4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
        p[a] = (pa[a][mt] && cok) ? *reinterpret_cast<const float4*>(pa[a][mt] + 8 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      v[0][mt] = make_float4(p[0].x - p[2].x, p[0].y - p[2].y, p[0].z - p[2].z, p[0].w - p[2].w);
      v[1][mt] = make_float4(p[1].x + p[2].x, p[1].y + p[2].y, p[1].z + p[2].z, p[1].w + p[2].w);
      v[2][mt] = make_float4(p[2].x - p[1].x, p[2].y - p[1].y, p[2].z - p[1].z, p[2].w - p[1].w);
      v[3][mt] = make_float4(p[1].x - p[3].x, p[1].y - p[3].y, p[1].z - p[3].z, p[1].w - p[3].w);
    }
    float4 wv[4][NT];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        wv[f][nt] = (n0 + nt * 32 < g.CoutPad)
                        ? *reinterpret_cast<const float4*>(wl + (size_t)f * 16 * fstride + ((size_t)q * 2 * g.CoutPad + nt * 32) * 4)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
#define SSBEV_WG_COMP(COMP)                                                                   \
    _Pragma("unroll") for (int f = 0; f < 4; ++f)                                             \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                         \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                         \
      acc[f][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[f][mt].COMP, wv[f][nt].COMP, acc[f][mt][nt], 0, 0, 0);
    SSBEV_WG_COMP(x) SSBEV_WG_COMP(y) SSBEV_WG_COMP(z) SSBEV_WG_COMP(w)
#undef SSBEV_WG_COMP
  }
  // epilogue: depth output transform; row = (r&3) + 8(r>>2) + 4 lk is the tile inside the group, column li the channel
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int t0 = (tg * MT + mt) * 32;
    float* o0 = Mo + ((long)xhw * R + ((long)b * g.D + 2 * i) * g.Thw + t0) * g.Cout;
    float* o1 = o0 + (long)g.Thw * g.Cout;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = n0 + nt * 32 + li;
      if (co >= g.Cout) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (t0 + row < g.Thw) {
          const float m0 = acc[0][mt][nt][r], m1 = acc[1][mt][nt][r], m2 = acc[2][mt][nt][r], m3 = acc[3][mt][nt][r];
          o0[(long)row * g.Cout + co] = m0 + m1 + m2;
          o1[(long)row * g.Cout + co] = m1 - m2 - m3;
        }
      }
    }
  }
}

// ---- batched frequency GEMM  C[xi][T][N] = A[xi][T][K] x B[xi][K][N]  (the element-wise stage of the plain pipeline) ----
// HBM-streaming design: the big operand A (transformed activations, read exactly once) is staged through LDS with
// global_load_lds_dwordx4 -- whole kilobytes per instruction, 16-byte XOR swizzle on the global side so that the MFMA
// A-operand reads (one ds_read_b128 = 4 k-values of a row per lane) are conflict-free -- in a ring of 3 stages of BK = 64
// columns; the small operand B (transformed weights, Wp[xi][q][kh][n][4], L2-resident) is read as coalesced float4 like
// the convolution kernels' weights.  A workgroup owns 64 rows and up to 4*NT*32 columns: its 4 waves share the A tile
// (wave w -> columns [w*NT*32, ...)), so A leaves LDS four times per HBM read.  Light waves (32*NT accumulators): three
// workgroups per CU.
struct WinoBgemmGeom { long T; int K, N, NPad; };

constexpr int kBgBM = 64, kBgBK = 64, kBgStages = 3;

template <int NT>
__global__ void __launch_bounds__(256)
wino_bgemm_kernel(const float* __restrict__ A, const float* __restrict__ Wp, float* __restrict__ Cm, WinoBgemmGeom g) {
  extern __shared__ __align__(16) float al[];            // [stages][64 rows][64 k], float4 slots swizzled by (row & 7)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const int xi = blockIdx.z;
  const int n0 = (blockIdx.y * 4 + wave) * (NT * 32);
  const bool col_active = n0 < g.N;
  const float* Ax = A + (long)xi * g.T * g.K;
  float* Cx = Cm + (long)xi * g.T * g.N;
  const int Q = (g.K + 7) >> 3, nstage = (g.K + kBgBK - 1) / kBgBK;
  const size_t fstride = (size_t)Q * 2 * g.NPad * 4;
  const float* wl = Wp + (size_t)xi * fstride + ((size_t)lk * g.NPad + n0 + li) * 4;
  // persistent walk: this workgroup owns row blocks rb0, rb0+1, ..., the LDS ring runs across row-block boundaries so
  // that the loads of the next block are in flight while this one is multiplied
  const long nrb = (g.T + kBgBM - 1) / kBgBM;
  const long rb_per = (nrb + gridDim.x - 1) / gridDim.x;
  const long rb_begin = (long)blockIdx.x * rb_per, rb_end = min(nrb, rb_begin + rb_per);
  const long total = (rb_end - rb_begin) * nstage;       // flat (row block, k stage) sequence
  if (total <= 0) return;

  // staging: a stage = 64 rows x 16 float4 slots = 1024 items = 16 wave instructions, 4 per wave (the LDS destination of a
  // global_load_lds is wave-uniform base + lane * 16: the 64 items of one instruction are consecutive)
  auto issue = [&](long s) {
    const long rb = rb_begin + s / nstage;
    const int k0 = (int)(s % nstage) * kBgBK, buf = (int)(s % kBgStages);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int jbase = (wave + 4 * e) * 64;
      const int j = jbase + lane, row = j >> 4, slot = j & 15;
      const int src_slot = (slot & 8) | ((slot & 7) ^ (row & 7));
      const long r = rb * kBgBM + row;
      const int k = k0 + src_slot * 4;
      const float* src = (r < g.T && k < g.K) ? Ax + r * g.K + k : kWinoZeros;
      __builtin_amdgcn_global_load_lds(src, al + buf * (kBgBM * kBgBK) + jbase * 4, 16, 0, 0);
    }
  };
  wf32x16 acc[2][NT];
  // weights of a stage: 8 coalesced float4 per column tile; double-buffered in registers and requested BEFORE the A tile
  // of the stage after next, so that (in-order return) waiting for them never waits for the newest A tile
  float4 bvA[kBgBK / 8][NT], bvB[kBgBK / 8][NT];
  auto load_b = [&](long s, float4 (&bv)[kBgBK / 8][NT]) {
    const int st = (int)(s % nstage);
#pragma unroll
    for (int qq = 0; qq < kBgBK / 8; ++qq) {
      const int q = st * (kBgBK / 8) + qq;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        bv[qq][nt] = (col_active && q < Q && n0 + nt * 32 < g.NPad)
                         ? *reinterpret_cast<const float4*>(wl + ((size_t)q * 2 * g.NPad + nt * 32) * 4)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  issue(0);
  load_b(0, bvA);
  if (total > 1) issue(1);
  for (long s = 0; s < total; ++s) {
    const int st = (int)(s % nstage), buf = (int)(s % kBgStages);
    float4 (&bv)[kBgBK / 8][NT] = (s & 1) ? bvB : bvA;
    if (st == 0) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
    }
    // outstanding, oldest first: A(s), B(s), A(s+1) -> stage s is complete once at most A(s+1)'s four loads remain
    if (s + 1 < total) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < total) { if (s & 1) load_b(s + 1, bvA); else load_b(s + 1, bvB); }
    if (s + 2 < total) issue(s + 2);
    if (col_active) {
      const float* ab = al + buf * (kBgBM * kBgBK);
#pragma unroll
      for (int qq = 0; qq < kBgBK / 8; ++qq) {
        float4 av[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int row = mt * 32 + li;
          const int slot = 2 * qq + lk;                              // logical float4 slot (k = 8 qq + 4 lk ..+3)
          av[mt] = *reinterpret_cast<const float4*>(ab + row * kBgBK + (((slot & 8) | ((slot & 7) ^ (row & 7))) << 2));
        }
#define SSBEV_BG_COMP(COMP)                                                                 \
        _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                    \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                   \
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt].COMP, bv[qq][nt].COMP, acc[mt][nt], 0, 0, 0);
        SSBEV_BG_COMP(x) SSBEV_BG_COMP(y) SSBEV_BG_COMP(z) SSBEV_BG_COMP(w)
#undef SSBEV_BG_COMP
      }
      if (st == nstage - 1) {                                        // row block finished: store its tile
        const long r0 = (rb_begin + s / nstage) * kBgBM;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int co = n0 + nt * 32 + li;
            if (co >= g.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const long row = r0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
              if (row < g.T) Cx[row * g.N + co] = acc[mt][nt][r];
            }
          }
      }
    }
  }
}

// packed weights of the depth-fused GEMM: Wp[xi = xi_d*16 + xi_hw][q][kh][n][t] = U[xi][k = 8q+4kh+t][n]
//   mode 0: U = G w[n][k] G^T (forward, K = Cin, N = Cout)   mode 1: U = G flip(w[k][n]) G^T (data gradient, K = Cout, N = Cin)
__global__ void __launch_bounds__(256)
wino_weight_packed_kernel(const float* __restrict__ w, float* __restrict__ Wp, int Cout, int Cin, int mode) {
  const int K = mode == 0 ? Cin : Cout, N = mode == 0 ? Cout : Cin;
  const int KPad = (K + 7) & ~7, NPad = (N + 31) & ~31;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= KPad * NPad) return;
  const int n = i % NPad, k = i / NPad;
  float u[64];
  if (k < K && n < N) {
    float gk[27];
    const int co = mode == 0 ? n : k, ci = mode == 0 ? k : n;
#pragma unroll
    for (int t = 0; t < 27; ++t) gk[t] = w[((size_t)co * Cin + ci) * 27 + (mode == 0 ? t : 26 - t)];
    float a[36], bb[48];
#pragma unroll
    for (int p = 0; p < 9; ++p) g4(gk + p * 3, 1, a + p * 4, 1);
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int f = 0; f < 4; ++f) g4(a + d * 12 + f, 4, bb + d * 16 + f, 4);
#pragma unroll
    for (int p = 0; p < 16; ++p) g4(bb + p, 16, u + p, 16);
  } else {
#pragma unroll
    for (int xi = 0; xi < 64; ++xi) u[xi] = 0.0f;
  }
  const int Q = KPad >> 3, q = k >> 3, kh = (k >> 2) & 1, t = k & 3;
#pragma unroll
  for (int xi = 0; xi < 64; ++xi) Wp[((((size_t)xi * Q + q) * 2 + kh) * NPad + n) * 4 + t] = u[xi];
}

bool wino_ok(const ssbev_wino_dims* d) {
  return d && d->B > 0 && d->C > 0 && d->D > 0 && d->H > 0 && d->W > 0 && d->D % 2 == 0 && d->H % 2 == 0 && d->W % 2 == 0;
}

}  // namespace

extern "C" {

#define SSBEV_WINO_ENTRY(NAME, KERNEL, TSRC, TDST)                                                                 \
  int NAME(const TSRC* src, TDST* dst, const ssbev_wino_dims* d, ssbev_stream_t stream) {                         \
    if (!wino_ok(d) || !src || !dst) return SSBEV_EINVAL;                                                          \
    const long total = (long)d->B * (d->D / 2) * (d->H / 2) * (d->W / 2) * d->C;                                   \
    const WinoGeom g{d->B, d->D, d->H, d->W, d->C};                                                                \
    hipLaunchKernelGGL(KERNEL, dim3(cdiv((size_t)total, 256)), dim3(256), 0, as_stream(stream), src, dst, g, total); \
    return ssbev_launch_status();                                                                                  \
  }

#define SSBEV_WINO2D_ENTRY(NAME, KERNEL, TSRC, TDST)                                                               \
  int NAME(const TSRC* src, TDST* dst, const ssbev_wino_dims* d, ssbev_stream_t stream) {                         \
    if (!wino2d_ok(d) || !src || !dst) return SSBEV_EINVAL;                                                        \
    const long total = (long)d->B * d->D * (d->H / 2) * (d->W / 2) * d->C;                                         \
    const WinoGeom g{d->B, d->D, d->H, d->W, d->C};                                                                \
    hipLaunchKernelGGL(KERNEL, dim3(cdiv((size_t)total, 256)), dim3(256), 0, as_stream(stream), src, dst, g, total); \
    return ssbev_launch_status();                                                                                  \
  }

int ssbev_wino_weight_transform(const float* w, float* U, int Cout, int Cin, int ndim, int mode, ssbev_stream_t stream) {
  if (!w || !U || Cout <= 0 || Cin <= 0 || (ndim != 2 && ndim != 3) || (mode != 0 && mode != 1)) return SSBEV_EINVAL;
  const dim3 grid(cdiv((size_t)Cout * Cin, 256)), block(256);
  if (ndim == 3) hipLaunchKernelGGL(wino_weight_kernel<3>, grid, block, 0, as_stream(stream), w, U, Cout, Cin, mode);
  else hipLaunchKernelGGL(wino_weight_kernel<2>, grid, block, 0, as_stream(stream), w, U, Cout, Cin, mode);
  return ssbev_launch_status();
}

int ssbev_wino_weight_grad(const float* gU, float* gw, int Cout, int Cin, int ndim, ssbev_stream_t stream) {
  if (!gU || !gw || Cout <= 0 || Cin <= 0 || (ndim != 2 && ndim != 3)) return SSBEV_EINVAL;
  const dim3 grid(cdiv((size_t)Cout * Cin, 256)), block(256);
  if (ndim == 3) hipLaunchKernelGGL(wino_weight_grad_kernel<3>, grid, block, 0, as_stream(stream), gU, gw, Cout, Cin);
  else hipLaunchKernelGGL(wino_weight_grad_kernel<2>, grid, block, 0, as_stream(stream), gU, gw, Cout, Cin);
  return ssbev_launch_status();
}

size_t ssbev_wino_dgemm_packed_elems(int Cout, int Cin) {
  const size_t a = (size_t)((Cin + 7) & ~7) * ((Cout + 31) & ~31), b = (size_t)((Cout + 7) & ~7) * ((Cin + 31) & ~31);
  return 64 * (a > b ? a : b);
}

int ssbev_wino_dgemm_pack(const float* w, float* Wp, int Cout, int Cin, int mode, ssbev_stream_t stream) {
  if (!w || !Wp || Cout <= 0 || Cin <= 0 || (mode != 0 && mode != 1)) return SSBEV_EINVAL;
  const int K = mode == 0 ? Cin : Cout, N = mode == 0 ? Cout : Cin;
  const size_t total = (size_t)((K + 7) & ~7) * ((N + 31) & ~31);
  hipLaunchKernelGGL(wino_weight_packed_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), w, Wp, Cout, Cin,
                     mode);
  return ssbev_launch_status();
}

// P [16][B*D*Thw][K] (hw-transformed input), Wp from ssbev_wino_dgemm_pack, Mo [16][B*D*Thw][N]; d->C = K, N given
int ssbev_wino_dgemm(const float* P, const float* Wp, float* Mo, const ssbev_wino_dims* d, int N, ssbev_stream_t stream) {
  if (!wino_ok(d) || !P || !Wp || !Mo || N <= 0 || d->C % 4 != 0) return SSBEV_EINVAL;
  const WinoGemmGeom g{d->B, d->D, (d->H / 2) * (d->W / 2), d->C, N, (N + 31) & ~31};
  // tiling: env SSBEV_WINO_TILE = MT*10 + NT (tuning), default <1,2> (two to three waves per SIMD)
  static const int forced = [] { const char* e = ssbev_tune("SSBEV_WINO_TILE"); return e ? atoi(e) : 0; }();
  int mt = 1, nt = 2;
  if (forced) { mt = forced / 10; nt = forced % 10; }
  const int tgroups = (g.Thw + 32 * mt - 1) / (32 * mt), tg4 = (tgroups + 3) >> 2;
  dim3 grid((unsigned)((long)g.B * (g.D / 2) * tg4), cdiv(N, nt * 32), 16), block(256);
  hipStream_t st = as_stream(stream);
#define SSBEV_WG_LAUNCH(M_, N_) hipLaunchKernelGGL((wino_dgemm_kernel<M_, N_>), grid, block, 0, st, P, Wp, Mo, g)
  switch (mt * 10 + nt) {
    case 14: SSBEV_WG_LAUNCH(1, 4); break;
    case 13: SSBEV_WG_LAUNCH(1, 3); break;
    case 12: SSBEV_WG_LAUNCH(1, 2); break;
    case 22: SSBEV_WG_LAUNCH(2, 2); break;
    case 21: SSBEV_WG_LAUNCH(2, 1); break;
    case 11: SSBEV_WG_LAUNCH(1, 1); break;
    default: return SSBEV_EINVAL;
  }
#undef SSBEV_WG_LAUNCH
  return ssbev_launch_status();
}

// A [64][T][K], Wp = ssbev_wino_dgemm_pack(...) ([64][K/8][2][NPad][4]), Cm [64][T][N]
int ssbev_wino_bgemm(const float* A, const float* Wp, float* Cm, int64_t T, int K, int N, ssbev_stream_t stream) {
  if (!A || !Wp || !Cm || T <= 0 || K <= 0 || N <= 0 || K % 4 != 0) return SSBEV_EINVAL;
  const WinoBgemmGeom g{(long)T, K, N, (N + 31) & ~31};
  const int nt = N > 128 ? 2 : 1;
  // persistent workgroups: ~3 per CU over the 64 frequencies and column groups, each walking >= 4 row blocks
  const long nrb = ((long)T + kBgBM - 1) / kBgBM;
  const int ycols = cdiv(N, 4 * nt * 32);
  long gx = std::max(1L, (768L + 64 * ycols - 1) / (64 * ycols));
  gx = std::min(gx, std::max(1L, nrb / 4));
  dim3 grid((unsigned)gx, ycols, 64), block(256);
  const size_t lds = (size_t)kBgStages * kBgBM * kBgBK * sizeof(float);
  hipStream_t st = as_stream(stream);
  if (nt == 2) hipLaunchKernelGGL(wino_bgemm_kernel<2>, grid, block, lds, st, A, Wp, Cm, g);
  else hipLaunchKernelGGL(wino_bgemm_kernel<1>, grid, block, lds, st, A, Wp, Cm, g);
  return ssbev_launch_status();
}

SSBEV_WINO2D_ENTRY(ssbev_wino2d_input_transform, wino2d_input_kernel<float>, float, float)
SSBEV_WINO2D_ENTRY(ssbev_wino2d_output_transform, wino2d_output_kernel<float>, float, float)
SSBEV_WINO2D_ENTRY(ssbev_wino2d_output_adjoint, wino2d_output_adjoint_kernel<float>, float, float)

SSBEV_WINO_ENTRY(ssbev_wino_input_transform, wino_input_kernel<float>, float, float)
SSBEV_WINO_ENTRY(ssbev_wino_output_transform, wino_output_kernel<float>, float, float)
SSBEV_WINO_ENTRY(ssbev_wino_output_adjoint, wino_output_adjoint_kernel<float>, float, float)

#define SSBEV_WINO43_ENTRY(NAME, KERNEL, DA, TF, TSRC, TDST)                                                        \
  int NAME(const TSRC* src, TDST* dst, const ssbev_wino_dims* d, ssbev_stream_t stream) {                         \
    if (!wino43_ok(d, DA) || !src || !dst) return SSBEV_EINVAL;                                                     \
    const long total = (long)d->B * (DA ? d->D / DA : d->D) * (d->H / 4) * (d->W / 4) * d->C;                      \
    const WinoGeom g{d->B, d->D, d->H, d->W, d->C};                                                                \
    hipLaunchKernelGGL((KERNEL<DA, TF>), dim3(cdiv((size_t)total, 256)), dim3(256), 0, as_stream(stream), src, dst, g, \
                       total);                                                                                     \
    return ssbev_launch_status();                                                                                  \
  }

SSBEV_WINO43_ENTRY(ssbev_wino43_input_transform, wino43_input_kernel, 2, float, float, float)
SSBEV_WINO43_ENTRY(ssbev_wino43_output_transform, wino43_output_kernel, 2, float, float, float)
SSBEV_WINO43_ENTRY(ssbev_wino43_output_adjoint, wino43_output_adjoint_kernel, 2, float, float, float)
SSBEV_WINO43_ENTRY(ssbev_wino43_2d_input_transform, wino43_input_kernel, 0, float, float, float)
SSBEV_WINO43_ENTRY(ssbev_wino43_2d_output_transform, wino43_output_kernel, 0, float, float, float)
SSBEV_WINO43_ENTRY(ssbev_wino43_2d_output_adjoint, wino43_output_adjoint_kernel, 0, float, float, float)
SSBEV_WINO43_ENTRY(ssbev_wino43_input_transform_bf16, wino43_input_kernel, 2, bf16_bits, float, uint16_t)
SSBEV_WINO43_ENTRY(ssbev_wino43_output_transform_bf16, wino43_output_kernel, 2, bf16_bits, uint16_t, float)
SSBEV_WINO43_ENTRY(ssbev_wino43_output_adjoint_bf16, wino43_output_adjoint_kernel, 2, bf16_bits, float, uint16_t)
SSBEV_WINO43_ENTRY(ssbev_wino43_2d_input_transform_bf16, wino43_input_kernel, 0, bf16_bits, float, uint16_t)
SSBEV_WINO43_ENTRY(ssbev_wino43_2d_output_transform_bf16, wino43_output_kernel, 0, bf16_bits, uint16_t, float)
SSBEV_WINO43_ENTRY(ssbev_wino43_2d_output_adjoint_bf16, wino43_output_adjoint_kernel, 0, bf16_bits, float, uint16_t)

SSBEV_WINO43_ENTRY(ssbev_wino444_input_transform, wino43_input_kernel, 4, float, float, float)
SSBEV_WINO43_ENTRY(ssbev_wino444_output_transform, wino43_output_kernel, 4, float, float, float)
SSBEV_WINO43_ENTRY(ssbev_wino444_output_adjoint, wino43_output_adjoint_kernel, 4, float, float, float)

// y += output transform (the data gradient of a second consumer lands in the first one's buffer)
int ssbev_wino43_2d_output_transform_acc(const float* src, float* dst, const ssbev_wino_dims* d, ssbev_stream_t stream) {
  if (!wino43_ok(d, 0) || !src || !dst) return SSBEV_EINVAL;
  const long total = (long)d->B * d->D * (d->H / 4) * (d->W / 4) * d->C;
  WinoGeom g{d->B, d->D, d->H, d->W, d->C};
  g.acc = 1;
  hipLaunchKernelGGL((wino43_output_kernel<0, float>), dim3(cdiv((size_t)total, 256)), dim3(256), 0, as_stream(stream), src, dst, g, total);
  return ssbev_launch_status();
}

int ssbev_wino2d_output_transform_acc(const float* src, float* dst, const ssbev_wino_dims* d, ssbev_stream_t stream) {
  if (!wino2d_ok(d) || !src || !dst) return SSBEV_EINVAL;
  const long total = (long)d->B * d->D * (d->H / 2) * (d->W / 2) * d->C;
  WinoGeom g{d->B, d->D, d->H, d->W, d->C};
  g.acc = 1;
  hipLaunchKernelGGL(wino2d_output_kernel<float>, dim3(cdiv((size_t)total, 256)), dim3(256), 0, as_stream(stream), src, dst, g, total);
  return ssbev_launch_status();
}

int ssbev_wino43_weight_transform(const float* w, float* U, int Cout, int Cin, int ndim, int mode, ssbev_stream_t stream) {
  if (!w || !U || Cout <= 0 || Cin <= 0 || (ndim != 2 && ndim != 3 && ndim != 4) || (mode != 0 && mode != 1)) return SSBEV_EINVAL;
  const dim3 grid(cdiv((size_t)Cout * Cin, 256)), block(256);
  if (ndim == 4) hipLaunchKernelGGL(wino43_weight_kernel<4>, grid, block, 0, as_stream(stream), w, U, Cout, Cin, mode);
  else if (ndim == 3) hipLaunchKernelGGL(wino43_weight_kernel<2>, grid, block, 0, as_stream(stream), w, U, Cout, Cin, mode);
  else hipLaunchKernelGGL(wino43_weight_kernel<0>, grid, block, 0, as_stream(stream), w, U, Cout, Cin, mode);
  return ssbev_launch_status();
}

int ssbev_wino43_weight_grad(const float* gU, float* gw, int Cout, int Cin, int ndim, ssbev_stream_t stream) {
  if (!gU || !gw || Cout <= 0 || Cin <= 0 || (ndim != 2 && ndim != 3 && ndim != 4)) return SSBEV_EINVAL;
  const dim3 grid(cdiv((size_t)Cout * Cin, 256)), block(256);
  if (ndim == 4) hipLaunchKernelGGL(wino43_weight_grad_kernel<4>, grid, block, 0, as_stream(stream), gU, gw, Cout, Cin);
  else if (ndim == 3) hipLaunchKernelGGL(wino43_weight_grad_kernel<2>, grid, block, 0, as_stream(stream), gU, gw, Cout, Cin);
  else hipLaunchKernelGGL(wino43_weight_grad_kernel<0>, grid, block, 0, as_stream(stream), gU, gw, Cout, Cin);
  return ssbev_launch_status();
}

// bf16 storage of the transformed-domain tensor (uint16_t = bf16 bit pattern); activations / gradients stay fp32
SSBEV_WINO2D_ENTRY(ssbev_wino2d_input_transform_bf16, wino2d_input_kernel<bf16_bits>, float, uint16_t)
SSBEV_WINO2D_ENTRY(ssbev_wino2d_output_transform_bf16, wino2d_output_kernel<bf16_bits>, uint16_t, float)
SSBEV_WINO2D_ENTRY(ssbev_wino2d_output_adjoint_bf16, wino2d_output_adjoint_kernel<bf16_bits>, float, uint16_t)

// bf16 ACTIVATIONS on the tensor side as well (round 4, bf16 storage mode): x / gy / y are bf16 channels-last tensors; two
// channels per thread (packed 4-byte accesses) whenever C is even
#define SSBEV_WINO16_ENTRY(NAME, KERNEL, DIV_D, OK)                                                                   \
  int NAME(const uint16_t* src, uint16_t* dst, const ssbev_wino_dims* d, ssbev_stream_t stream) {                    \
    if (!OK(d) || !src || !dst) return SSBEV_EINVAL;                                                                  \
    const int nv = d->C % 2 == 0 ? 2 : 1;                                                                             \
    const long total = (long)d->B * (d->D / DIV_D) * (d->H / 2) * (d->W / 2) * (d->C / nv);                           \
    const WinoGeom g{d->B, d->D, d->H, d->W, d->C};                                                                   \
    if (nv == 2)                                                                                                      \
      hipLaunchKernelGGL((KERNEL<bf16_bits, bf16_bits, 2>), dim3(cdiv((size_t)total, 256)), dim3(256), 0, as_stream(stream), src, dst, g, total); \
    else                                                                                                              \
      hipLaunchKernelGGL((KERNEL<bf16_bits, bf16_bits, 1>), dim3(cdiv((size_t)total, 256)), dim3(256), 0, as_stream(stream), src, dst, g, total); \
    return ssbev_launch_status();                                                                                     \
  }
SSBEV_WINO16_ENTRY(ssbev_wino_input_transform_bf16a, wino_input_kernel, 2, wino_ok)
SSBEV_WINO16_ENTRY(ssbev_wino_output_transform_bf16a, wino_output_kernel, 2, wino_ok)
SSBEV_WINO16_ENTRY(ssbev_wino_output_adjoint_bf16a, wino_output_adjoint_kernel, 2, wino_ok)
SSBEV_WINO16_ENTRY(ssbev_wino2d_input_transform_bf16a, wino2d_input_kernel, 1, wino2d_ok)
SSBEV_WINO16_ENTRY(ssbev_wino2d_output_transform_bf16a, wino2d_output_kernel, 1, wino2d_ok)
SSBEV_WINO16_ENTRY(ssbev_wino2d_output_adjoint_bf16a, wino2d_output_adjoint_kernel, 1, wino2d_ok)

SSBEV_WINO_ENTRY(ssbev_wino_input_transform_bf16, wino_input_kernel<bf16_bits>, float, uint16_t)
SSBEV_WINO_ENTRY(ssbev_wino_output_transform_bf16, wino_output_kernel<bf16_bits>, uint16_t, float)
SSBEV_WINO_ENTRY(ssbev_wino_output_adjoint_bf16, wino_output_adjoint_kernel<bf16_bits>, float, uint16_t)

}  // extern "C"
