// Dense N-d convolution family as im2col-free implicit GEMM on the gfx950 matrix cores.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate, 157 TF/s chip peak).
// Layout: channels-last activations [B, D, H, W, C].  Kernels in this file:
//
//   conv_gather_kernel<MT,NT,QU>   forward / data gradient of every conv, deconv, strided, dilated, 1x1 layer that has no
//                                  more specialised kernel.  GEMM rows M = output voxels, columns N = output channels,
//                                  K = taps x input channels.  A operand: lane (i = lane&31, kh = lane>>5) reads ONE
//                                  float4 = x[voxel_i + tap][8q + 4kh .. +3] straight from global/L2 (a voxel's channel
//                                  vector is contiguous); the 4 components feed 4 consecutive MFMAs.  No LDS, no im2col
//                                  buffer: the "patch matrix" only ever exists as addresses.  B operand: weights
//                                  pre-packed so that lane (j, kh) reads one float4 Wp[tap][q][kh][j][0..3] (512 B
//                                  contiguous per half wave, L1/L2 resident).  Two gather forms: form 0 (conv)
//                                  in = o*stride - pad + k*dil; form 1 (deconv) in = (o + pad - k*dil)/stride with the
//                                  outputs enumerated per PARITY CLASS (o mod stride) so that the valid taps are uniform
//                                  over a tile: no wasted MFMAs on transposed convolutions.  Register tilings chosen per
//                                  problem from measured sweeps (dispatch_gather); <1,5> is software pipelined.
//   conv_tap_kernel                forward / data gradient of the <= 32-channel stride-1 3x3x3 layers: input rows in an
//                                  LDS ring (global_load_lds), weights in registers, taps split over the waves.
//   wgrad_lds_kernel<...>          weight gradient of the 3x3(x3) layers (stride 1, stride 2, transposed): both operands
//                                  staged through LDS, 9 accumulators per wave.
//   wgrad_1x1_kernel, wgrad_cf_kernel, wgrad_kernel   weight gradients outside that family (1x1 streaming; dilated via
//                                  channel-major copies; k = s deconvs), wgrad_reduce(_tiled)_kernel folds the split-K
//                                  partial tiles in a fixed order (deterministic, no float atomics).
// The wide (>= 64 channel) stride-1 3x3(x3) layers normally do not come here at all: functional.conv3d / conv2d route them
// through the Winograd path (winograd.hip).
#include "common.h"
#include "conv_bf16.h"
#include "conv_thin_mfma.h"

#include <algorithm>
#include <cstdio>
#include <vector>
#include <array>
#include <map>
#include <mutex>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvGeom {
  int B, Cin, Cout, CinPad, CoutPad;   // channel roles of THIS gather (K = Cin, N = Cout)
  int Di, Hi, Wi, Do, Ho, Wo;          // source grid (i) and destination grid (o) of THIS gather
  int kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw;
  int form;                            // 0 conv gather, 1 deconv gather
  int relu, accumulate;
  int hint;                            // 0 or MT*100+NT*10+QU
  int chunk_taps;                      // LDSB variant: taps staged in LDS per pass
  int bf16;                            // operands rounded to bf16, v_mfma_f32_32x32x16_bf16 (precision = 1)
};

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// bf16 mode (ssbev_conv_dims.precision = 1, BASELINE configs[3]): tensors stay fp32 in memory; operands are rounded to
// bf16 (nearest even, v_cvt_pk_bf16_f32) in registers and contracted with v_mfma_f32_32x32x16_bf16 (fp32 accumulate):
// one instruction covers the 16 input channels that take eight fp32 MFMAs.  Lane (l & 31, l >> 5) supplies k slots
// 8 * (l >> 5) .. + 7 of both operands; any assignment of channels to slots is valid as long as A and B agree, and both
// use [quad of k-step 2j | quad of k-step 2j + 1] of the fp32 layouts.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 to_bf16x8(const float4& a, const float4& b) {
  bf16x8 r;
  r[0] = (__bf16)a.x; r[1] = (__bf16)a.y; r[2] = (__bf16)a.z; r[3] = (__bf16)a.w;
  r[4] = (__bf16)b.x; r[5] = (__bf16)b.y; r[6] = (__bf16)b.z; r[7] = (__bf16)b.w;
  return r;
}
__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// One wave computes an (MT*32 voxels) x (NT*32 channels) tile.  QU consecutive 8-channel k-steps are
// unrolled as straight-line code so that all their operand loads are in flight before the first MFMA
// of the group issues (the compiler then places counted vmcnt waits between the MFMA clusters).
// LDSB: the packed weights of this N tile (all taps) are copied ONCE per workgroup into LDS and the B operand
// is read with ds_read_b128.  For the 32-channel cost-volume layers the 110 KB of weights do not fit the
// 32 KB L1, so without this every wave streams them from L2 for each of its 64 voxels (as much L2 traffic
// as the activations themselves); a 16-wave workgroup amortises one copy over 1024 voxels.
template <int MT, int NT, int QU, int WPB = 4, bool LDSB = false, bool PIPE = false, bool BF16 = false>
__global__ void __launch_bounds__(WPB * 64)
conv_gather_kernel(const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
                   float* __restrict__ y, ConvGeom g) {
  extern __shared__ __align__(16) float wlds[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int li = lane & 31, lk = lane >> 5;

  // ---- parity class (form 1) --------------------------------------------------------------
  int par_d = 0, par_h = 0, par_w = 0;
  int Dc = g.Do, Hc = g.Ho, Wc = g.Wo;   // extent of the enumerated (per-class) output grid
  if (g.form == 1) {
    // heaviest parity class first: class (1,1,1) of a k3 s2 problem walks 8 taps, class (0,0,0) one; workgroups are
    // dispatched in blockIdx order, so the light classes fill the tail of the heavy ones instead of the other way round
    int cls = (int)gridDim.z - 1 - (int)blockIdx.z;
    par_w = cls % g.sw; cls /= g.sw;
    par_h = cls % g.sh; cls /= g.sh;
    par_d = cls;
    // class extent = number of o in [0, Do) with o % s == par
    Dc = (g.Do - par_d + g.sd - 1) / g.sd; Hc = (g.Ho - par_h + g.sh - 1) / g.sh; Wc = (g.Wo - par_w + g.sw - 1) / g.sw;
  }
  const long Mtot = (long)g.B * Dc * Hc * Wc;
  // XCD-aware remap (workgroup L runs on XCD L % 8, each XCD has its own L2): give every XCD one contiguous
  // range of tiles, N-tile index fastest, so that neighbouring voxel tiles -- which share their kd/kh halo
  // rows and their whole A operand across N tiles -- hit the SAME L2 instead of eight different ones.
  int bx, by;
  {
    const unsigned n = gridDim.x * gridDim.y;
    const unsigned L = blockIdx.x + gridDim.x * blockIdx.y;
    const unsigned xcd = L & 7, q = n >> 3, r = n & 7;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const unsigned Lp = base + (L >> 3);
    bx = (int)(Lp / gridDim.y);
    by = (int)(Lp % gridDim.y);
  }
  const long m_wave = ((long)bx * WPB + wave) * (MT * 32);
  const int n0 = by * (NT * 32);
  const bool active = m_wave < Mtot;
  if (!LDSB && !active) return;        // (LDSB: idle waves still take part in the staging barriers)

  // ---- decode this lane's voxel for every M sub-tile ----------------------------------------
  int ob[MT], od[MT], oh[MT], ow[MT];
  bool mok[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    long m = m_wave + mt * 32 + li;
    mok[mt] = m < Mtot;
    if (!mok[mt]) m = 0;
    ow[mt] = (int)(m % Wc); m /= Wc;
    oh[mt] = (int)(m % Hc); m /= Hc;
    od[mt] = (int)(m % Dc);
    ob[mt] = (int)(m / Dc);
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

  const int Q = g.CinPad >> 3;
  const size_t w_tap_stride = (size_t)Q * 2 * g.CoutPad * 4;
  const float* wlane = wp + ((size_t)lk * g.CoutPad + n0 + li) * 4;

  // tap ranges: form 0 walks all taps; form 1 only those congruent with the parity class
  int kd0 = 0, kh0 = 0, kw0 = 0, kds = 1, khs = 1, kws = 1;
  if (g.form == 1) {
    kd0 = (par_d + g.pd) % g.sd; kh0 = (par_h + g.ph) % g.sh; kw0 = (par_w + g.pw) % g.sw;
    kds = g.sd; khs = g.sh; kws = g.sw;
  }
  // One flat loop over the (wave-uniform) tap list.  Per-tap lane work is reduced to a mask test and one
  // pointer add: along each axis the source coordinate is  base + i*step  (i = tap ordinal), so validity
  // is a per-axis bit mask computed once, and the tap's address offset is wave-uniform (scalar registers).
  const int nkd = (g.kd - kd0 + kds - 1) / kds, nkh = (g.kh - kh0 + khs - 1) / khs, nkw = (g.kw - kw0 + kws - 1) / kws;
  const int ntaps = (kd0 < g.kd && kh0 < g.kh && kw0 < g.kw) ? nkd * nkh * nkw : 0;
  int step_d, step_h, step_w;
  if (g.form == 0) { step_d = g.dd; step_h = g.dh; step_w = g.dw; }
  else { step_d = -(kds * g.dd) / g.sd; step_h = -(khs * g.dh) / g.sh; step_w = -(kws * g.dw) / g.sw; }
  const float* pbase[MT];
  unsigned vmask[MT];          // bit (ia*8+ib)*8+ic ... packed as three per-axis masks: d | h<<8 | w<<16
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int bd, bh, bw;
    if (g.form == 0) {
      bd = od[mt] * g.sd - g.pd; bh = oh[mt] * g.sh - g.ph; bw = ow[mt] * g.sw - g.pw;
    } else {
      bd = od[mt] + (par_d + g.pd - kd0 * g.dd) / g.sd;
      bh = oh[mt] + (par_h + g.ph - kh0 * g.dh) / g.sh;
      bw = ow[mt] + (par_w + g.pw - kw0 * g.dw) / g.sw;
    }
    unsigned m = 0;
    for (int i = 0; i < nkd && i < 8; ++i) { const int v = bd + i * step_d; m |= (v >= 0 && v < g.Di) ? (1u << i) : 0u; }
    for (int i = 0; i < nkh && i < 8; ++i) { const int v = bh + i * step_h; m |= (v >= 0 && v < g.Hi) ? (1u << (8 + i)) : 0u; }
    for (int i = 0; i < nkw && i < 8; ++i) { const int v = bw + i * step_w; m |= (v >= 0 && v < g.Wi) ? (1u << (16 + i)) : 0u; }
    vmask[mt] = mok[mt] ? m : 0u;
    // may point outside the tensor for border voxels; only dereferenced under the mask
    pbase[mt] = x + ((((long)ob[mt] * g.Di + bd) * g.Hi + bh) * g.Wi + bw) * (long)g.Cin + 4 * lk;
  }
  if constexpr (!LDSB && PIPE) {
    // Software-pipelined walk over the flat (tap, channel-group) sequence: the operands of group n+1 are requested
    // right after the first MFMA of group n and consumed one group later.  Used for the <1,5> tiling of the 48x160
    // maps (one wave per SIMD, nothing else hides the L1/L2 latency: 98 -> 110 TF/s); the tilings that run several
    // waves per SIMD lose more from the doubled operand registers than they gain (measured), so they keep the
    // plain loop below.
    const int ngq = Q / QU, G = ntaps * ngq;
    int l_ti = 0, l_q0 = 0;
    const float* l_ap[MT];
    const float* l_wt = wlane;
    auto tap_state = [&](int t) {
      const int ic = t % nkw, ib = (t / nkw) % nkh, ia = t / (nkw * nkh);
      const int c = kw0 + ic * kws, bq = kh0 + ib * khs, a = kd0 + ia * kds;
      const int tap = (a * g.kh + bq) * g.kw + c;
      const long toff = (((long)ia * step_d * g.Hi + (long)ib * step_h) * g.Wi + (long)ic * step_w) * g.Cin;   // uniform
      const unsigned need = (1u << ia) | (1u << (8 + ib)) | (1u << (16 + ic));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) l_ap[mt] = ((vmask[mt] & need) == need) ? pbase[mt] + toff : nullptr;
      l_wt = wlane + (size_t)tap * w_tap_stride;
    };
    auto load = [&](float4 (&av)[QU][MT], float4 (&bv)[QU][NT]) {
#pragma unroll
      for (int u = 0; u < QU; ++u) {
        const int q = l_q0 + u;
        const bool cok = (8 * q + 4 * lk) < g.Cin;   // Cin is padded to 8 only in the packed weights
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          av[u][mt] = (l_ap[mt] && cok) ? *reinterpret_cast<const float4*>(l_ap[mt] + 8 * q) : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          bv[u][nt] = (n0 + nt * 32 < g.CoutPad)      // column tiles past the padded weights (forced tilings): zeros
                          ? *reinterpret_cast<const float4*>(l_wt + ((size_t)q * 2 * g.CoutPad + nt * 32) * 4)
                          : make_float4(0, 0, 0, 0);
      }
      l_q0 += QU;
      if (l_q0 >= Q) {
        l_q0 = 0;
        if (++l_ti < ntaps) tap_state(l_ti);
      }
    };
    auto mm_head = [&](const float4 (&av)[QU][MT], const float4 (&bv)[QU][NT]) {
      if constexpr (BF16)
        acc[0][0] = mfma_bf16(to_bf16x8(av[0][0], av[QU > 1 ? 1 : 0][0]), to_bf16x8(bv[0][0], bv[QU > 1 ? 1 : 0][0]), acc[0][0]);
      else
        acc[0][0] = mfma32(av[0][0].x, bv[0][0].x, acc[0][0]);
    };
    auto mm_tail = [&](const float4 (&av)[QU][MT], const float4 (&bv)[QU][NT]) {
      if constexpr (BF16) {
        static_assert(!BF16 || !PIPE || QU % 2 == 0, "pipelined bf16 walk pairs the k-steps");
#pragma unroll
        for (int u = 0; u < QU; u += 2) {
          bf16x8 ab[MT], bb[NT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) ab[mt] = to_bf16x8(av[u][mt], av[u + 1 < QU ? u + 1 : u][mt]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bb[nt] = to_bf16x8(bv[u][nt], bv[u + 1 < QU ? u + 1 : u][nt]);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              if (u + mt + nt != 0) acc[mt][nt] = mfma_bf16(ab[mt], bb[nt], acc[mt][nt]);
        }
        return;
      }
#pragma unroll
      for (int u = 0; u < QU; ++u) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            if (u + mt + nt != 0) acc[mt][nt] = mfma32(av[u][mt].x, bv[u][nt].x, acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma32(av[u][mt].y, bv[u][nt].y, acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma32(av[u][mt].z, bv[u][nt].z, acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma32(av[u][mt].w, bv[u][nt].w, acc[mt][nt]);
      }
    };
    float4 avA[QU][MT], bvA[QU][NT], avB[QU][MT], bvB[QU][NT];
    if (G > 0) {
      tap_state(0);
      load(avA, bvA);
    }
    for (int gi = 0; gi < G; gi += 2) {
      mm_head(avA, bvA);
      __builtin_amdgcn_sched_barrier(0);
      if (gi + 1 < G) load(avB, bvB);
      __builtin_amdgcn_sched_barrier(0);
      mm_tail(avA, bvA);
      __builtin_amdgcn_sched_barrier(0);
      if (gi + 1 >= G) break;
      mm_head(avB, bvB);
      __builtin_amdgcn_sched_barrier(0);
      if (gi + 2 < G) load(avA, bvA);
      __builtin_amdgcn_sched_barrier(0);
      mm_tail(avB, bvB);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
  const int taps_total = g.kd * g.kh * g.kw;
  const int chunk_taps = LDSB ? g.chunk_taps : taps_total;
  int ti = 0;
  for (int t0 = 0; t0 < taps_total; t0 += chunk_taps) {
  if (LDSB) {   // stage taps [t0, t0+chunk): rows [tap][q][kh] x (NT*32 couts) x float4 of the packed weights
    const int rows = min(chunk_taps, taps_total - t0) * Q * 2;
    __syncthreads();
    for (int i = threadIdx.x; i < rows * NT * 32; i += WPB * 64) {
      const int r = i / (NT * 32), cc = i - r * (NT * 32);
      reinterpret_cast<float4*>(wlds)[i] =
          reinterpret_cast<const float4*>(wp)[((size_t)t0 * Q * 2 + r) * g.CoutPad + n0 + cc];
    }
    __syncthreads();
  }
  for (; ti < ntaps; ++ti) {
    const int ic = ti % nkw, ib = (ti / nkw) % nkh, ia = ti / (nkw * nkh);
    const int c = kw0 + ic * kws, bq = kh0 + ib * khs, a = kd0 + ia * kds;
    const int tap = (a * g.kh + bq) * g.kw + c;
    if (tap >= t0 + chunk_taps) break;         // belongs to the next staged chunk
    if (LDSB && !active) continue;
    const long toff = (((long)ia * step_d * g.Hi + (long)ib * step_h) * g.Wi + (long)ic * step_w) * g.Cin;   // uniform
    const unsigned need = (1u << ia) | (1u << (8 + ib)) | (1u << (16 + ic));
    const float* ap[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) ap[mt] = ((vmask[mt] & need) == need) ? pbase[mt] + toff : nullptr;
    const float* wt = wlane + (size_t)tap * w_tap_stride;
    for (int q0 = 0; q0 < Q; q0 += QU) {
      float4 av[QU][MT], bv[QU][NT];
#pragma unroll
      for (int u = 0; u < QU; ++u) {
        const int q = q0 + u;
        const bool cok = (8 * q + 4 * lk) < g.Cin;   // Cin is padded to 8 only in the packed weights
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          av[u][mt] = (ap[mt] && cok) ? *reinterpret_cast<const float4*>(ap[mt] + 8 * q) : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          bv[u][nt] = LDSB ? reinterpret_cast<const float4*>(wlds)[(((tap - t0) * Q + q) * 2 + lk) * (NT * 32) + nt * 32 + li]
                      : (n0 + nt * 32 < g.CoutPad)
                            ? *reinterpret_cast<const float4*>(wt + ((size_t)q * 2 * g.CoutPad + nt * 32) * 4)
                            : make_float4(0, 0, 0, 0);
      }
      // component-major: consecutive MFMAs go to different accumulators (no back-to-back dependent issue)
#define SSBEV_GATHER_STEP(COMP)                                                              \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                      \
      _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                      \
        acc[mt][nt] = mfma32(av[u][mt].COMP, bv[u][nt].COMP, acc[mt][nt]);
      if constexpr (BF16) {       // k-steps in pairs: 16 channels per bf16 MFMA (an odd last step pairs with zeros)
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < QU; u += 2) {
          bf16x8 ab[MT], bb[NT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) ab[mt] = to_bf16x8(av[u][mt], u + 1 < QU ? av[u + 1 < QU ? u + 1 : u][mt] : z4);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bb[nt] = to_bf16x8(bv[u][nt], u + 1 < QU ? bv[u + 1 < QU ? u + 1 : u][nt] : z4);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma_bf16(ab[mt], bb[nt], acc[mt][nt]);
        }
      } else {
#pragma unroll
      for (int u = 0; u < QU; ++u) {
        SSBEV_GATHER_STEP(x) SSBEV_GATHER_STEP(y) SSBEV_GATHER_STEP(z) SSBEV_GATHER_STEP(w)
      }
      }
#undef SSBEV_GATHER_STEP
    }
  }
  }   // staged tap chunks
  }   // LDSB path
  if (LDSB && !active) return;

  // ---- epilogue: C/D layout row = (r&3) + 8*(r>>2) + 4*lk, col = li ---------------------------
  // voxel of accumulator row r of sub-tile mt (its coordinates live in lane `row`, any lk: fetched with shuffles)
  auto row_vox = [&](int mt, int r, bool& rok) -> size_t {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
    const int rb = __shfl(ob[mt], row, 64), rd = __shfl(od[mt], row, 64);
    const int rh = __shfl(oh[mt], row, 64), rw = __shfl(ow[mt], row, 64);
    rok = __shfl((int)mok[mt], row, 64) != 0;
    if (g.form == 0) return (((size_t)rb * g.Do + rd) * g.Ho + rh) * g.Wo + rw;
    return (((size_t)rb * g.Do + (rd * g.sd + par_d)) * g.Ho + (rh * g.sh + par_h)) * g.Wo + (rw * g.sw + par_w);
  };
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if (g.accumulate) {
      // y += result (gradient slots): the old values of EIGHT rows are requested before any of them is needed -- two
      // memory latencies per sub-tile; a load placed between the stores waits for each store in turn (+29 % on 32 -> 64)
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += 8) {
        float oldv[8][NT];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          bool rok;
          const size_t vox = row_vox(mt, r0 + r, rok);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int co = n0 + nt * 32 + li;
            oldv[r][nt] = (rok && co < g.Cout) ? y[vox * g.Cout + co] : 0.0f;
          }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt][r0 + r] += oldv[r][nt];
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      bool rok;
      const size_t vox = row_vox(mt, r, rok);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int co = n0 + nt * 32 + li;
        if (rok && co < g.Cout) {
          float v = acc[mt][nt][r];
          if (bias) v += bias[co];
          if (g.relu) v = fmaxf(v, 0.0f);
          y[vox * g.Cout + co] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------- weight packing
// dst[tap][q][kh][n][t] = W[k_ch = 8q+4kh+t][n_ch = n][tap]   (zero beyond Cin / Cout)
// src index depends on the torch layout and on which tensor axis plays K / N:
//   layout 0: [A0, A1, taps] with K = A1, N = A0   (conv fwd: [Cout,Cin,k])
//   layout 1: [A0, A1, taps] with K = A0, N = A1   (deconv fwd: [Cin,Cout,k]; conv bwd-data: [Cout,Cin,k])
__global__ void pack_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int K, int N, int KPad,
                                   int NPad, int taps, int layout, long total) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  long r = i;
  const int t = (int)(r & 3); r >>= 2;
  const int n = (int)(r % NPad); r /= NPad;
  const int kh = (int)(r & 1); r >>= 1;
  const int Q = KPad >> 3;
  const int q = (int)(r % Q);
  const int tap = (int)(r / Q);
  const int k = 8 * q + 4 * kh + t;
  float v = 0.0f;
  if (k < K && n < N) {
    const size_t a0 = layout == 0 ? n : k, a1 = layout == 0 ? k : n;
    const size_t A1 = layout == 0 ? K : N;
    v = src[(a0 * A1 + a1) * taps + tap];
  }
  dst[i] = v;
}

// ---------------------------------------------------------------- weight gradient
// T[tap][qc][pc] = sum_m Qt[pos(m, tap)][qc] * Pt[m][pc]     (m over the "small" grid)
//   conv  : Pt = gy [B,Do,Ho,Wo,Cout] (small), Qt = x  [B,Di,Hi,Wi,Cin] sampled at m*s - p + k*dil
//   deconv: Pt = x  [B,Di,Hi,Wi,Cin ] (small), Qt = gy [B,Do,Ho,Wo,Cout] sampled at m*s - p + k
// MFMA roles: rows i = q-channel, cols j = p-channel, k = 2 voxels per instruction.
// One wave: (32*MQ q-channels) x (32 p-channels) x TW taps along the innermost kernel axis, over one
// chunk of voxels.  Partials go to ws[chunk][...]; wgrad_reduce_kernel sums them in chunk order.
struct WgradGeom {
  int B, Cp, Cq;                 // channels of the small-grid tensor (P) and the sampled tensor (Q)
  int Ds, Hs, Ws;                // small grid
  int Dq, Hq, Wq;                // sampled grid
  int kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw;
  int chunk;                     // voxels per chunk (multiple of 2)
  int nchunks;
};

// One wave: (32*MQ q-channels) x (32*MP p-channels) x (TH x TW taps of one kd slice), over one chunk of
// voxels.  Every loaded P value feeds MQ*TH*TW MFMAs and every Q value MP of them; the voxel walk is an
// incremental (w,h,d,b) counter (no integer division in the loop).
template <int MQ, int MP, int TH, int TW>
__global__ void __launch_bounds__(256)
wgrad_kernel(const float* __restrict__ P, const float* __restrict__ Qt, float* __restrict__ ws, WgradGeom g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const int chunk_id = blockIdx.x * 4 + wave;
  if (chunk_id >= g.nchunks) return;
  const int nqt = (g.Cq + 32 * MQ - 1) / (32 * MQ);
  const int qt = blockIdx.y % nqt, pt = blockIdx.y / nqt;
  const int kw_groups = (g.kw + TW - 1) / TW, kh_groups = (g.kh + TH - 1) / TH;
  int tg = blockIdx.z;
  const int kwg = tg % kw_groups; tg /= kw_groups;
  const int khg = tg % kh_groups;
  const int kdi = tg / kh_groups;

  f32x16 acc[MQ][MP][TH][TW];
#pragma unroll
  for (int a = 0; a < MQ; ++a)
#pragma unroll
    for (int b = 0; b < MP; ++b)
#pragma unroll
      for (int c = 0; c < TH; ++c)
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][c][t][r] = 0.0f;

  const int Mtot = g.B * g.Ds * g.Hs * g.Ws;
  const int m_begin = chunk_id * g.chunk;
  const int m_end = min(Mtot, m_begin + g.chunk);
  // this half-wave's voxel walks m_begin + lk, +2, +4, ...
  int m = m_begin + lk;
  int w, h, d, b;
  {
    int r = min(m, Mtot - 1);
    w = r % g.Ws; r /= g.Ws;
    h = r % g.Hs; r /= g.Hs;
    d = r % g.Ds;
    b = r / g.Ds;
  }
  int qoff[MQ], poff[MP];
  bool qok[MQ], pok[MP];
#pragma unroll
  for (int a = 0; a < MQ; ++a) { qoff[a] = (qt * MQ + a) * 32 + li; qok[a] = qoff[a] < g.Cq; }
#pragma unroll
  for (int a = 0; a < MP; ++a) { poff[a] = (pt * MP + a) * 32 + li; pok[a] = poff[a] < g.Cp; }

  // U voxel pairs per trip: all (MP + MQ*TH*TW) * U operand loads are issued before the MFMA cluster
  constexpr int U = (MQ * MP * TH * TW >= 9) ? 2 : 4;
  for (; m < m_end + lk; m += 2 * U) {   // both halves run the same trip count; tails are masked
    float pv[U][MP], qv[U][TH][TW][MQ];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int mu = m + 2 * u;
      const bool mok = mu < m_end;
#pragma unroll
      for (int a = 0; a < MP; ++a) pv[u][a] = (mok && pok[a]) ? P[(size_t)mu * g.Cp + poff[a]] : 0.0f;
      const int id = d * g.sd - g.pd + kdi * g.dd;
      const bool dok = mok && id >= 0 && id < g.Dq;
      const int iw0 = w * g.sw - g.pw + (kwg * TW) * g.dw;
#pragma unroll
      for (int c = 0; c < TH; ++c) {
        const int khi = khg * TH + c;
        const int ih = h * g.sh - g.ph + khi * g.dh;
        const bool rok = dok && khi < g.kh && ih >= 0 && ih < g.Hq;
        const size_t rowbase = (((size_t)b * g.Dq + id) * g.Hq + ih) * g.Wq;
#pragma unroll
        for (int t = 0; t < TW; ++t) {
          const int iw = iw0 + t * g.dw;
          const bool ok = rok && (kwg * TW + t) < g.kw && iw >= 0 && iw < g.Wq;
#pragma unroll
          for (int a = 0; a < MQ; ++a) qv[u][c][t][a] = (ok && qok[a]) ? Qt[(rowbase + iw) * g.Cq + qoff[a]] : 0.0f;
        }
      }
      // advance the voxel counter by 2
      w += 2;
      while (w >= g.Ws) {
        w -= g.Ws;
        if (++h >= g.Hs) { h = 0; if (++d >= g.Ds) { d = 0; ++b; } }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < TH; ++c)
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
          for (int a = 0; a < MQ; ++a)
#pragma unroll
            for (int e = 0; e < MP; ++e) acc[a][e][c][t] = mfma32(qv[u][c][t][a], pv[u][e], acc[a][e][c][t]);
  }
  // partial tiles -> ws[chunk][tap][q-channel][p-channel]
  const int taps = g.kd * g.kh * g.kw;
#pragma unroll
  for (int c = 0; c < TH; ++c)
#pragma unroll
    for (int t = 0; t < TW; ++t) {
      const int khi = khg * TH + c, kwi = kwg * TW + t;
      if (khi >= g.kh || kwi >= g.kw) continue;
      const int tap = (kdi * g.kh + khi) * g.kw + kwi;
      float* dst = ws + (((size_t)chunk_id * taps + tap) * g.Cq) * g.Cp;
#pragma unroll
      for (int a = 0; a < MQ; ++a)
#pragma unroll
        for (int e = 0; e < MP; ++e)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (qt * MQ + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (row < g.Cq && pok[e]) dst[(size_t)row * g.Cp + poff[e]] = acc[a][e][c][t][r];
          }
    }
}

// gw (torch layout) = sum over chunks in a fixed order (deterministic).
//   conv  : gw[co = p][ci = q][tap];  deconv: gw[ci = p][co = q][tap]   -> both are [p][q][tap]
// A workgroup owns 32 consecutive slab elements; its 8 thread rows stride over the chunks (coalesced 128-B
// reads per chunk) and are folded through LDS in row order.
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ gw, int nchunks, int taps, int Cq, int Cp,
                    long total, long cstride) {
  __shared__ float part[8][33];
  const int e = threadIdx.x & 31, row = threadIdx.x >> 5;
  const long i = (long)blockIdx.x * 32 + e;                 // i over ws-slab order [tap][q][p]
  float s = 0.0f;
  if (i < total)
    for (int c = row; c < nchunks; c += 8) s += ws[(size_t)c * cstride + i];
  part[row][e] = s;
  __syncthreads();
  if (row == 0 && i < total) {
    float t = 0.0f;
#pragma unroll
    for (int r = 0; r < 8; ++r) t += part[r][e];
    const int p = (int)(i % Cp);
    const int q = (int)((i / Cp) % Cq);
    const int tap = (int)(i / ((long)Cp * Cq));
    gw[((size_t)p * Cq + q) * taps + tap] = t;
  }
}

// Same reduction for layers with many channels (where the OUTPUT is the big object: 28 MB for 512x512x27): the kernel
// above writes gw[p][q][tap] one float at a time (a 108-byte stride between the lanes of a wave: every 4-byte store
// dirties its own sector).  Here a workgroup owns 32 p x 8 q x all taps: the chunk sums are taken with coalesced 128-byte
// reads, transposed through LDS, and written as 32 runs of 8*taps contiguous floats.
__global__ void __launch_bounds__(256)
wgrad_reduce_tiled_kernel(const float* __restrict__ ws, float* __restrict__ gw, int nchunks, int taps, int Cq, int Cp,
                          long total, long cstride) {
  extern __shared__ float tile[];                 // [32 p][8 q * taps] with an odd pitch
  const int P = 8 * taps + 1;
  const int e = threadIdx.x & 31, row = threadIdx.x >> 5;
  const int p0 = blockIdx.x * 32, q0 = blockIdx.y * 8;
  const bool ok = p0 + e < Cp && q0 + row < Cq;
  for (int tap = 0; tap < taps; ++tap) {
    float s = 0.0f;
    if (ok) {
      const float* src = ws + ((size_t)tap * Cq + q0 + row) * Cp + p0 + e;
#pragma unroll 4
      for (int c = 0; c < nchunks; ++c) s += src[(size_t)c * cstride];
    }
    tile[e * P + row * taps + tap] = s;
  }
  __syncthreads();
  const int run = 8 * taps;
  for (int j = threadIdx.x; j < 32 * run; j += 256) {
    const int pe = j / run, r = j - pe * run;
    const int q = q0 + r / taps, tap = r % taps, pp = p0 + pe;
    if (pp < Cp && q < Cq) gw[((size_t)pp * Cq + q) * taps + tap] = tile[pe * P + r];
  }
}

// First stage for the narrow layers: there the partials are the big object (e.g. 27 x 32 x 32 floats x ~700 chunks = 73 MB)
// and the kernels above walk the chunk axis with a few hundred workgroups of 4-byte lanes -- 53 us per layer at 1.4 TB/s,
// 26 layers per step.  Here the chunk axis is cut into `nsl` slices: workgroup (x, y) sums slice y of 256 float4 columns,
// eight independent loads in flight per thread, and writes the result over the FIRST slab of its own slice (the only
// reader of those bytes is the thread that writes them).  The second stage then folds nsl slabs, per * total apart.
// Fixed order everywhere: deterministic.
typedef float wr_f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256)
wgrad_reduce_stage_kernel(float* __restrict__ ws, int nchunks, int per, long total4) {
  const long j = (long)blockIdx.x * 256 + threadIdx.x;
  if (j >= total4) return;
  const int c0 = blockIdx.y * per, c1 = min(nchunks, c0 + per);
  wr_f4* base = reinterpret_cast<wr_f4*>(ws) + (size_t)c0 * total4 + j;
  const wr_f4* p = base;
  wr_f4 a[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) a[u] = wr_f4{0.0f, 0.0f, 0.0f, 0.0f};
  for (int c = c0; c < c1; c += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (c + u < c1) a[u] += p[(size_t)u * total4];
    p += 8 * total4;
  }
  *base = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}

void launch_wgrad_reduce(float* partial, float* gw, int nchunks, int taps, int Cq, int Cp, hipStream_t st) {
  const long total = (long)taps * Cq * Cp;
  long cstride = total;
  static const int two_stage = ssbev_tune("SSBEV_WGRAD_REDUCE2") ? atoi(ssbev_tune("SSBEV_WGRAD_REDUCE2")) : 1;
  if (two_stage && nchunks >= 64 && total % 4 == 0) {
    const long total4 = total / 4, bx = cdiv(total4, 256);
    int nsl = (int)std::min<long>(std::max<long>(cdiv(2048, bx), 1), nchunks / 8);
    const int per = cdiv(nchunks, nsl);
    nsl = cdiv(nchunks, per);
    hipLaunchKernelGGL(wgrad_reduce_stage_kernel, dim3((unsigned)bx, (unsigned)nsl), dim3(256), 0, st, partial, nchunks, per,
                       total4);
    nchunks = nsl;
    cstride = (long)per * total;
  }
  if (taps <= 32 && (long)cdiv(Cp, 32) * cdiv(Cq, 8) >= 128) {
    hipLaunchKernelGGL(wgrad_reduce_tiled_kernel, dim3(cdiv(Cp, 32), cdiv(Cq, 8)), dim3(256),
                       (size_t)32 * (8 * taps + 1) * sizeof(float), st, partial, gw, nchunks, taps, Cq, Cp, total, cstride);
  } else {
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(total, 32)), dim3(256), 0, st, partial, gw, nchunks, taps, Cq, Cp,
                       total, cstride);
  }
}

// ---------------------------------------------------------------- host side
bool conv_dims_ok(const ssbev_conv_dims* d) {
  if (!d) return false;
  if (d->B <= 0 || d->Cin <= 0 || d->Cout <= 0) return false;
  if (d->Di <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->Do <= 0 || d->Ho <= 0 || d->Wo <= 0) return false;
  if (d->kd <= 0 || d->kh <= 0 || d->kw <= 0 || d->sd <= 0 || d->sh <= 0 || d->sw <= 0) return false;
  if (d->dd <= 0 || d->dh <= 0 || d->dw <= 0 || d->pd < 0 || d->ph < 0 || d->pw < 0) return false;
  if (d->kd > 8 || d->kh > 8 || d->kw > 8) return false;
  const bool strided = d->sd > 1 || d->sh > 1 || d->sw > 1;
  const bool dilated = d->dd > 1 || d->dh > 1 || d->dw > 1;
  if (strided && dilated) return false;
  return true;
}

int pad8(int c) { return (c + 7) & ~7; }
int pad32(int c) { return (c + 31) & ~31; }

template <int MT, int NT, int QU>
int launch_gather(const float* x, const float* wp, const float* bias, float* y, const ConvGeom& g, hipStream_t st) {
  long Mtot = (long)g.B * g.Do * g.Ho * g.Wo;
  int classes = 1;
  if (g.form == 1) {   // grid sized for the largest parity class (parity 0)
    classes = g.sd * g.sh * g.sw;
    Mtot = (long)g.B * ((g.Do + g.sd - 1) / g.sd) * ((g.Ho + g.sh - 1) / g.sh) * ((g.Wo + g.sw - 1) / g.sw);
  }
  dim3 grid(cdiv(Mtot, 4 * MT * 32), cdiv(g.Cout, NT * 32), classes), block(256);
  constexpr bool PIPE = NT == 5;
  if (g.bf16) {                    // k-steps are consumed in pairs: at least two per group whenever Cin allows
    constexpr int QB = QU < 2 ? 2 : QU;
    if ((g.CinPad >> 3) % QB == 0)
      hipLaunchKernelGGL((conv_gather_kernel<MT, NT, QB, 4, false, PIPE, true>), grid, block, 0, st, x, wp, bias, y, g);
    else if ((g.CinPad >> 3) % 2 == 0)
      hipLaunchKernelGGL((conv_gather_kernel<MT, NT, 2, 4, false, PIPE, true>), grid, block, 0, st, x, wp, bias, y, g);
    else
      hipLaunchKernelGGL((conv_gather_kernel<MT, NT, 1, 4, false, false, true>), grid, block, 0, st, x, wp, bias, y, g);
    return ssbev_launch_status();
  }
  if ((g.CinPad >> 3) % QU == 0)
    hipLaunchKernelGGL((conv_gather_kernel<MT, NT, QU, 4, false, PIPE>), grid, block, 0, st, x, wp, bias, y, g);
  else
    hipLaunchKernelGGL((conv_gather_kernel<MT, NT, 1, 4, false, PIPE>), grid, block, 0, st, x, wp, bias, y, g);
  return ssbev_launch_status();
}

// 16-wave workgroups with LDS-resident weights (small-channel layers)
template <int MT, int QU>
int launch_gather_ldsb(const float* x, const float* wp, const float* bias, float* y, const ConvGeom& g, hipStream_t st) {
  constexpr int WPB = 16;
  long Mtot = (long)g.B * g.Do * g.Ho * g.Wo;
  int classes = 1;
  if (g.form == 1) {
    classes = g.sd * g.sh * g.sw;
    Mtot = (long)g.B * ((g.Do + g.sd - 1) / g.sd) * ((g.Ho + g.sh - 1) / g.sh) * ((g.Wo + g.sw - 1) / g.sw);
  }
  const size_t per_tap = (size_t)(g.CinPad >> 3) * 2 * 32 * 16;
  const int taps_total = g.kd * g.kh * g.kw;
  int chunk = (int)((144 * 1024) / per_tap);
  if (chunk < 1) return SSBEV_EINVAL;
  if (chunk > taps_total) chunk = taps_total;
  const int passes = (taps_total + chunk - 1) / chunk;
  chunk = (taps_total + passes - 1) / passes;            // equalise the passes
  ConvGeom gg = g;
  gg.chunk_taps = chunk;
  const size_t lds = per_tap * chunk;
  auto kern = g.bf16 ? conv_gather_kernel<MT, 1, QU, WPB, true, false, true> : conv_gather_kernel<MT, 1, QU, WPB, true>;
  if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return SSBEV_ELAUNCH;
  dim3 grid(cdiv(Mtot, WPB * MT * 32), cdiv(g.Cout, 32), classes), block(WPB * 64);
  hipLaunchKernelGGL(kern, grid, block, lds, st, x, wp, bias, y, gg);
  return ssbev_launch_status();
}

template <int MT, int NT>
int launch_gather_qu(int qu, const float* x, const float* wp, const float* bias, float* y, const ConvGeom& g,
                     hipStream_t st) {
  if (qu == 4) return launch_gather<MT, NT, 4>(x, wp, bias, y, g, st);
  if (qu == 2) return launch_gather<MT, NT, 2>(x, wp, bias, y, g, st);
  return launch_gather<MT, NT, 1>(x, wp, bias, y, g, st);
}

int launch_gather_cfg(int mt, int nt, int qu, const float* x, const float* wp, const float* bias, float* y,
                      const ConvGeom& g, hipStream_t st) {
  switch (mt * 10 + nt) {
    case 41: return launch_gather_qu<4, 1>(qu, x, wp, bias, y, g, st);
    case 21: return launch_gather_qu<2, 1>(qu, x, wp, bias, y, g, st);
    case 11: return launch_gather_qu<1, 1>(qu, x, wp, bias, y, g, st);
    case 24: return launch_gather_qu<2, 4>(qu, x, wp, bias, y, g, st);
    case 22: return launch_gather_qu<2, 2>(qu, x, wp, bias, y, g, st);
    case 12: return launch_gather_qu<1, 2>(qu, x, wp, bias, y, g, st);
    case 15: return launch_gather_qu<1, 5>(qu, x, wp, bias, y, g, st);
    case 14: return launch_gather_qu<1, 4>(qu, x, wp, bias, y, g, st);
    case 31: return launch_gather_qu<3, 1>(qu, x, wp, bias, y, g, st);
    case 23: return launch_gather_qu<2, 3>(qu, x, wp, bias, y, g, st);
    case 26: return launch_gather_qu<2, 6>(qu, x, wp, bias, y, g, st);
    case 16: return launch_gather_qu<1, 6>(qu, x, wp, bias, y, g, st);
    case 13: return launch_gather_qu<1, 3>(qu, x, wp, bias, y, g, st);
    case 91: return launch_gather_ldsb<2, 2>(x, wp, bias, y, g, st);      // tuning hook: LDS-resident weights
    case 92: return launch_gather_ldsb<1, 2>(x, wp, bias, y, g, st);
    default: return SSBEV_EINVAL;
  }
}

long gather_blocks(const ConvGeom& g, int MT, int NT) {
  long M = (long)g.B * g.Do * g.Ho * g.Wo;
  long classes = 1;
  if (g.form == 1) {
    classes = (long)g.sd * g.sh * g.sw;
    M = (long)g.B * ((g.Do + g.sd - 1) / g.sd) * ((g.Ho + g.sh - 1) / g.sh) * ((g.Wo + g.sw - 1) / g.sw);
  }
  return ((M + 4 * MT * 32 - 1) / (4 * MT * 32)) * ((g.Cout + NT * 32 - 1) / (NT * 32)) * classes;
}

// Register-tiling heuristic, fitted to a sweep of all (MT, NT, QU) variants over the hot-path layer
// shapes on MI355X (tools/sweep_tiles.py, profiles/r1_tile_sweep.txt):
//   * NT: 4 when Cout is a multiple of 128 and the grid still fills the chip, else 2 (Cout > 32), else 1;
//   * MT: 2 unless that leaves fewer than ~160 workgroups;
//   * QU (k-steps whose operand loads are issued together): deep prefetch (4) for wide tiles and for
//     small grids (latency-bound), 1-2 when many waves per SIMD already hide the latency.
__device__ const float kGatherZeros[4] __attribute__((aligned(16))) = {0.f, 0.f, 0.f, 0.f};

// ------------------------------------------------------------------------------------------------
// conv_igemm_kernel (round 5): LDS-staged implicit GEMM for the layers conv_gather_kernel still served at 59-97 TF/s -- the
// stride-2 3x3x3 convolutions and transposed convolutions between the hourglass levels 64 <-> 128 (VT:75-88), the encoder's
// downsamplers 128 -> 256 -> 512 (resnet3d.py) and their data gradients, any layer with Cin % 32 == 0 of the gather's source.
//     y[m][n] = sum over (tap, k) of  x[src(m, tap)][k] * W[tap][k][n]            m = destination voxel, rows of a GEMM
// conv_gather_kernel lets every lane fetch the float4s of its voxel from L1/L2 for every MFMA group (one wave per SIMD
// waits for each of them); here a workgroup owns a BM x BN tile like gemm_nn_kernel (csrc/gemm.hip): per stage (= one tap x
// 32 source channels) the BM gathered rows and the 32 x BN weight block travel global -> LDS by global_load_lds_dwordx4,
// double buffered, and the fragments are ds_read_b128 on both sides -- the packed weight layout of pack_weight_kernel
// ([tap][q][kh][n][4]) IS the B fragment layout, so one 16-byte read feeds four MFMAs.  Padding = the zero line (per row and
// tap: a 24-bit per-axis validity mask, as in conv_gather_kernel); parity classes of the transposed forms = blockIdx.z.
// Rows are decoded ONCE per workgroup into an LDS table (source offset, mask, destination offset).
template <int WN, int MW, int WGN>
__global__ void __launch_bounds__(256, 2)
conv_igemm_kernel(const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
                  float* __restrict__ y, ConvGeom g, int mblocks, int nblocks) {
  constexpr int WGM = 4 / WGN;
  constexpr int BM = 32 * MW * WGM, BN = 32 * WN * WGN, BK = 32;
  constexpr int AF = BM * BK, BF = BK * BN, SF = AF + BF;
  constexpr int AI = AF / 256, BI = BF / 256;
  constexpr int AE = (AI + 3) / 4, BE = (BI + 3) / 4;
  extern __shared__ __align__(16) float lds[];                    // [2][A slab | B slab] | row table
  long* t_src = reinterpret_cast<long*>(lds + 2 * SF);            // [BM] source offset (floats) of the row's base voxel
  long* t_dst = t_src + BM;                                       // [BM] destination offset (floats), -1 = no such row
  unsigned* t_msk = reinterpret_cast<unsigned*>(t_dst + BM);      // [BM] per-axis tap validity: d | h << 8 | w << 16
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lk = lane >> 5;
  const int wm = wave / WGN, wn = wave % WGN;

  // ---- parity class (form 1), as conv_gather_kernel: heaviest class first
  int par_d = 0, par_h = 0, par_w = 0;
  int Dc = g.Do, Hc = g.Ho, Wc = g.Wo;
  if (g.form == 1) {
    int cls = (int)gridDim.z - 1 - (int)blockIdx.z;
    par_w = cls % g.sw; cls /= g.sw;
    par_h = cls % g.sh; cls /= g.sh;
    par_d = cls;
    Dc = (g.Do - par_d + g.sd - 1) / g.sd; Hc = (g.Ho - par_h + g.sh - 1) / g.sh; Wc = (g.Wo - par_w + g.sw - 1) / g.sw;
  }
  const long Mtot = (long)g.B * Dc * Hc * Wc;
  int mb, nb;
  {
    const unsigned n = gridDim.x, L = blockIdx.x;
    const unsigned xcd = L & 7, q = n >> 3, r = n & 7;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const unsigned Lp = base + (L >> 3);
    nb = (int)(Lp % (unsigned)nblocks);
    mb = (int)(Lp / (unsigned)nblocks);
  }
  const long m0 = (long)mb * BM;
  const int n0 = nb * BN;
  if (m0 >= Mtot) return;                                         // (class extents differ: the grid is sized for the largest)

  int kd0 = 0, kh0 = 0, kw0 = 0, kds = 1, khs = 1, kws = 1;
  if (g.form == 1) {
    kd0 = (par_d + g.pd) % g.sd; kh0 = (par_h + g.ph) % g.sh; kw0 = (par_w + g.pw) % g.sw;
    kds = g.sd; khs = g.sh; kws = g.sw;
  }
  const int nkd = (g.kd - kd0 + kds - 1) / kds, nkh = (g.kh - kh0 + khs - 1) / khs, nkw = (g.kw - kw0 + kws - 1) / kws;
  const int ntaps = (kd0 < g.kd && kh0 < g.kh && kw0 < g.kw) ? nkd * nkh * nkw : 0;
  int step_d, step_h, step_w;
  if (g.form == 0) { step_d = g.dd; step_h = g.dh; step_w = g.dw; }
  else { step_d = -(kds * g.dd) / g.sd; step_h = -(khs * g.dh) / g.sh; step_w = -(kws * g.dw) / g.sw; }

  // ---- row table: thread r decodes row r of the tile
  if (tid < BM) {
    long m = m0 + tid;
    const bool ok = m < Mtot;
    if (!ok) m = 0;
    const int ow = (int)(m % Wc); m /= Wc;
    const int oh = (int)(m % Hc); m /= Hc;
    const int od = (int)(m % Dc);
    const int ob = (int)(m / Dc);
    int bd, bh, bw;
    if (g.form == 0) {
      bd = od * g.sd - g.pd; bh = oh * g.sh - g.ph; bw = ow * g.sw - g.pw;
    } else {
      bd = od + (par_d + g.pd - kd0 * g.dd) / g.sd;
      bh = oh + (par_h + g.ph - kh0 * g.dh) / g.sh;
      bw = ow + (par_w + g.pw - kw0 * g.dw) / g.sw;
    }
    unsigned msk = 0;
    for (int i = 0; i < nkd && i < 8; ++i) { const int v = bd + i * step_d; msk |= (v >= 0 && v < g.Di) ? (1u << i) : 0u; }
    for (int i = 0; i < nkh && i < 8; ++i) { const int v = bh + i * step_h; msk |= (v >= 0 && v < g.Hi) ? (1u << (8 + i)) : 0u; }
    for (int i = 0; i < nkw && i < 8; ++i) { const int v = bw + i * step_w; msk |= (v >= 0 && v < g.Wi) ? (1u << (16 + i)) : 0u; }
    t_msk[tid] = ok ? msk : 0u;
    t_src[tid] = ((((long)ob * g.Di + bd) * g.Hi + bh) * g.Wi + bw) * (long)g.Cin;      // may lie outside: only used under the mask
    long dst;
    if (g.form == 0) dst = (((long)ob * g.Do + od) * g.Ho + oh) * g.Wo + ow;
    else dst = (((long)ob * g.Do + (od * g.sd + par_d)) * g.Ho + (oh * g.sh + par_h)) * g.Wo + (ow * g.sw + par_w);
    t_dst[tid] = ok ? dst * (long)g.Cout : -1;
  }
  __syncthreads();

  // ---- per-lane copy sources of the A slab: BM rows x 8 slots of 16 B, slots XOR-swizzled by ((row >> 1) & 7)
  const float* ap[AE]; unsigned am[AE];
#pragma unroll
  for (int e = 0; e < AE; ++e) {
    const int item = (wave + 4 * e) * 64 + lane, row = item >> 3, slot = item & 7;
    const bool on = wave + 4 * e < AI;
    ap[e] = x + (on ? t_src[row] : 0) + ((slot ^ ((row >> 1) & 7)) << 2);
    am[e] = on ? t_msk[row] : 0u;
  }
  // B slab: the 8 (q, kh) rows of this stage's 32 source channels x BN columns x 4 floats, contiguous per row in the packed
  // weights; piece j = wave + 4 e covers slab floats [256 j, 256 j + 256)
  const int Q = g.CinPad >> 3;
  long boff[BE];
#pragma unroll
  for (int e = 0; e < BE; ++e) {
    const int j = wave + 4 * e, f = j * 256 + 4 * lane;
    const int r8 = f / (BN * 4), col = f % (BN * 4);
    boff[e] = n0 + col / 4 < g.CoutPad ? (long)r8 * g.CoutPad * 4 + (long)n0 * 4 + col : -1;     // columns past the padded weights: zeros
  }
  const int cq_n = g.Cin >> 5;                                    // 32-channel stages per tap
  const int nst = ntaps * cq_n;
  // tap state of the stage being ISSUED (wave-uniform, advanced incrementally)
  int i_ia = 0, i_ib = 0, i_ic = 0, i_cq = 0;
  auto issue = [&](int buf) {
    const int tap = ((kd0 + i_ia * kds) * g.kh + (kh0 + i_ib * khs)) * g.kw + (kw0 + i_ic * kws);
    const long toff = (((long)i_ia * step_d * g.Hi + (long)i_ib * step_h) * g.Wi + (long)i_ic * step_w) * g.Cin + i_cq * 32;
    const unsigned need = (1u << i_ia) | (1u << (8 + i_ib)) | (1u << (16 + i_ic));
#pragma unroll
    for (int e = 0; e < AE; ++e) {
      if (AI % 4 != 0 && wave + 4 * e >= AI) break;
      const float* src = ((am[e] & need) == need) ? ap[e] + toff : kGatherZeros;
      __builtin_amdgcn_global_load_lds(src, lds + buf * SF + (wave + 4 * e) * 256, 16, 0, 0);
    }
    const float* wb = wp + ((size_t)tap * Q + 4 * i_cq) * 2 * g.CoutPad * 4;
#pragma unroll
    for (int e = 0; e < BE; ++e) {
      if (BI % 4 != 0 && wave + 4 * e >= BI) break;
      // (a named pointer: with the sum as the builtin's argument hipcc 7.2 drops the HOST stub of the kernel)
      const float* bsrc = boff[e] >= 0 ? wb + boff[e] : kGatherZeros;
      __builtin_amdgcn_global_load_lds(bsrc, lds + buf * SF + AF + (wave + 4 * e) * 256, 16, 0, 0);
    }
    if (++i_cq == cq_n) {
      i_cq = 0;
      if (++i_ic == nkw) { i_ic = 0; if (++i_ib == nkh) { i_ib = 0; ++i_ia; } }
    }
  };

  f32x16 acc[MW][WN];
#pragma unroll
  for (int mt = 0; mt < MW; ++mt)
#pragma unroll
    for (int nt = 0; nt < WN; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

  typedef float v4f __attribute__((ext_vector_type(4)));
  if (nst > 0) issue(0);
  for (int st = 0; st < nst; ++st) {
    const int buf = st & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (st + 1 < nst) issue(buf ^ 1);
    const float* as = lds + buf * SF;
    const float* bs = as + AF;
    v4f ac[MW], bc[WN], an[MW], bn[WN];
    auto fetch = [&](int q, v4f (&a)[MW], v4f (&bf)[WN]) {
#pragma unroll
      for (int mt = 0; mt < MW; ++mt) {
        const int row = (wm * MW + mt) * 32 + li;
        a[mt] = *reinterpret_cast<const v4f*>(as + row * BK + (((2 * q + lk) ^ ((row >> 1) & 7)) << 2));
      }
#pragma unroll
      for (int nt = 0; nt < WN; ++nt)
        bf[nt] = *reinterpret_cast<const v4f*>(bs + ((2 * q + lk) * BN + (wn * WN + nt) * 32 + li) * 4);
    };
    fetch(0, ac, bc);
#pragma unroll
    for (int q = 0; q < BK / 8; ++q) {
      if (q + 1 < BK / 8) fetch(q + 1, an, bn);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mt = 0; mt < MW; ++mt)
#pragma unroll
          for (int nt = 0; nt < WN; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[mt][t], bc[nt][t], acc[mt][nt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MW; ++mt) ac[mt] = an[mt];
#pragma unroll
      for (int nt = 0; nt < WN; ++nt) bc[nt] = bn[nt];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  // ---- epilogue: accumulator row = (r & 3) + 8 (r >> 2) + 4 lk, column li; every load before the first store
  float bv[WN];
#pragma unroll
  for (int nt = 0; nt < WN; ++nt) {
    const int co = min(n0 + (wn * WN + nt) * 32 + li, g.Cout - 1);
    bv[nt] = bias ? bias[co] : 0.0f;
  }
  long rowbase[MW][16];
#pragma unroll
  for (int mt = 0; mt < MW; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) rowbase[mt][r] = t_dst[(wm * MW + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk];
  if (g.accumulate) {
#pragma unroll
    for (int mt = 0; mt < MW; ++mt)
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += 8) {
        float oldv[8][WN];
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int nt = 0; nt < WN; ++nt) {
            const int co = n0 + (wn * WN + nt) * 32 + li;
            oldv[r][nt] = (rowbase[mt][r0 + r] >= 0 && co < g.Cout) ? y[rowbase[mt][r0 + r] + co] : 0.0f;
          }
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int nt = 0; nt < WN; ++nt) acc[mt][nt][r0 + r] += oldv[r][nt];
      }
  }
#pragma unroll
  for (int nt = 0; nt < WN; ++nt) asm volatile("" : "+v"(bv[nt]));
#pragma unroll
  for (int nt = 0; nt < WN; ++nt) {
    const int co = n0 + (wn * WN + nt) * 32 + li;
    if (co >= g.Cout) continue;
#pragma unroll
    for (int mt = 0; mt < MW; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (rowbase[mt][r] >= 0) {
          float v = acc[mt][nt][r] + bv[nt];
          if (g.relu) v = fmaxf(v, 0.0f);
          y[rowbase[mt][r] + co] = v;
        }
      }
  }
}

// the layers conv_igemm_kernel takes (SSBEV_IGEMM=0 or tile_hint >= 10 keep conv_gather_kernel; the Winograd / tap / thin kernels are
// chosen before dispatch_gather is reached): fp32 gathers whose source
// has a multiple of 32 channels, more than one tap or stride, <= 8 taps per axis, and enough rows to fill tiles
bool conv_igemm_applicable(const ConvGeom& g) {
  const char* env = ssbev_env("SSBEV_IGEMM");                        // (read per call: the tests switch it inside one process)
  const int mode = env ? atoi(env) : 2;
  if (mode == 0 || g.bf16 || g.hint) return false;
  if (g.Cin % 32 != 0 || g.Cout % 4 != 0 || g.Cout < 64) return false;
  if (g.kd > 8 || g.kh > 8 || g.kw > 8) return false;
  const int taps = g.kd * g.kh * g.kw;
  if (taps == 1) return false;                                    // pointwise layers: the GEMM family / conv_pw32
  if (g.form == 0 && g.sd * g.sh * g.sw == 1 && mode != 2)
    return false;                                                 // 1 = strided / transposed forms only; 2 (default) = also the stride-1
                                                                  // layers no Winograd / tap kernel took (DepthNet's dilated 640 -> 640: -0.25 ms)
  const long M = (long)g.B * g.Do * g.Ho * g.Wo;
  return M >= 2048;
}

template <int WN, int MW, int WGN>
int launch_igemm_t(const float* x, const float* wp, const float* bias, float* y, const ConvGeom& g, hipStream_t st) {
  constexpr int BM = 32 * MW * (4 / WGN), BN = 32 * WN * WGN;
  long Mtot = (long)g.B * g.Do * g.Ho * g.Wo;
  int classes = 1;
  if (g.form == 1) {
    classes = g.sd * g.sh * g.sw;
    Mtot = (long)g.B * ((g.Do + g.sd - 1) / g.sd) * ((g.Ho + g.sh - 1) / g.sh) * ((g.Wo + g.sw - 1) / g.sw);
  }
  const int mblocks = (int)((Mtot + BM - 1) / BM), nblocks = (g.Cout + BN - 1) / BN;
  const size_t lds = (size_t)2 * (BM * 32 + 32 * BN) * sizeof(float) + (size_t)BM * (2 * sizeof(long) + sizeof(unsigned));
  auto kern = conv_igemm_kernel<WN, MW, WGN>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return SSBEV_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3((unsigned)(mblocks * nblocks), 1, classes), dim3(256), lds, st, x, wp, bias, y, g, mblocks, nblocks);
  return ssbev_launch_status();
}

int launch_igemm(const float* x, const float* wp, const float* bias, float* y, const ConvGeom& g, hipStream_t st) {
  long Mtot = (long)g.B * g.Do * g.Ho * g.Wo;
  long classes = 1;
  if (g.form == 1) {
    classes = (long)g.sd * g.sh * g.sw;
    Mtot = (long)g.B * ((g.Do + g.sd - 1) / g.sd) * ((g.Ho + g.sh - 1) / g.sh) * ((g.Wo + g.sw - 1) / g.sw);
  }
  // the largest tile that still leaves enough workgroups: 512 for the plain form, 1024 for the parity-class form (its classes walk
  // 1 .. 8 taps: the heavy ones must fill the chip on their own).  Measured (tools/igemm_probe.py, profiles/r5_igemm_probe.txt):
  // 64 -> 128 s2 at 23 040 output voxels: 128x128 = 180 workgroups 134 us, 64x64 = 720 workgroups 113 us.
  const int force = ssbev_tune("SSBEV_IGEMM_TILE") ? atoi(ssbev_tune("SSBEV_IGEMM_TILE")) : 0;             // probing: bm * 1000 + bn
  const bool wide = g.CoutPad % 128 == 0;
  const long need = g.form == 1 ? 1024 : 512;
  const int cand[3][2] = {{128, wide ? 128 : 64}, {64, wide ? 128 : 64}, {64, 64}};
  int bm = 64, bn = 64;
  for (int i = 0; i < 3; ++i) {
    const long blocks = ((Mtot + cand[i][0] - 1) / cand[i][0]) * ((g.Cout + cand[i][1] - 1) / cand[i][1]) * classes;
    if (blocks >= need) { bm = cand[i][0]; bn = cand[i][1]; break; }
  }
  if (force) { bm = force / 1000; bn = force % 1000; }
  if (bm == 128 && bn == 128) return launch_igemm_t<2, 2, 2>(x, wp, bias, y, g, st);
  if (bm == 128 && bn == 64) return launch_igemm_t<1, 2, 2>(x, wp, bias, y, g, st);
  if (bm == 64 && bn == 128) return launch_igemm_t<2, 1, 2>(x, wp, bias, y, g, st);
  if (bm == 64 && bn == 64) return launch_igemm_t<1, 1, 2>(x, wp, bias, y, g, st);
  return SSBEV_EINVAL;
}

int dispatch_gather(const float* x, const float* wp, const float* bias, float* y, const ConvGeom& g, hipStream_t st) {
  if (g.Cin % 4 != 0) return SSBEV_EINVAL;
  if (conv_igemm_applicable(g)) return launch_igemm(x, wp, bias, y, g, st);
  if (g.hint) return launch_gather_cfg(g.hint / 100, (g.hint / 10) % 10, g.hint % 10, x, wp, bias, y, g, st);
  const long Mtot = (long)g.B * g.Do * g.Ho * g.Wo;
  // 48x160 feature maps (7680 pixels = 240 row tiles): <1,5> makes 240 x (Cout/160) waves -- 960 of the chip's 1024
  // SIMDs busy in ONE round for Cout = 640, each A operand feeding five column tiles (sweep: 98 vs 81 TF/s)
  if (Mtot <= 8192 && g.Cout % 160 == 0 && g.Cout >= 640 && g.sd * g.sh * g.sw == 1)
    return launch_gather_cfg(1, 5, 4, x, wp, bias, y, g, st);
  {
    // other small maps (per parity class for the deconv form): many small tiles; deep operand prefetch for the plain
    // form (256 -> 512 s2 at 32x32x4 outputs: 0.53 -> 0.34 ms), shallow for the parity-class form (0.73 -> 0.37 ms)
    const long classes = g.form == 1 ? (long)g.sd * g.sh * g.sw : 1;
    const long keff = g.form == 1 ? (long)cdiv(g.kd, g.sd) * cdiv(g.kh, g.sh) * cdiv(g.kw, g.sw) * g.Cin
                                  : (long)g.kd * g.kh * g.kw * g.Cin;
    if (Mtot / classes <= 8192 && keff <= 8192 && g.Cout > 64)
      return launch_gather_cfg(1, 1, g.form == 1 ? 1 : 4, x, wp, bias, y, g, st);
  }
  {
    const size_t wbytes = (size_t)g.kd * g.kh * g.kw * (g.CinPad >> 3) * 2 * 32 * 16;
    if (g.Cout <= 32 && wbytes <= 144 * 1024 && wbytes >= 32 * 1024 && (g.CinPad >> 3) % 2 == 0 &&
        gather_blocks(g, 2, 1) >= 2048)
      return launch_gather_ldsb<2, 2>(x, wp, bias, y, g, st);
  }
  // 192-wide outputs (the occupancy head's 384 -> 192 conv and its data gradient): six column tiles per wave, the A
  // operand is fetched once per 192 output channels (sweep: 133 vs 118 TF/s for <2,2>)
  if (g.Cout % 192 == 0 && gather_blocks(g, 2, 6) >= 256) return launch_gather_cfg(2, 6, 4, x, wp, bias, y, g, st);
  int nt = (g.Cout % 128 == 0) ? 4 : (g.Cout > 32 ? 2 : 1);
  if (nt == 4 && gather_blocks(g, 2, 4) < 256) nt = 2;
  const int mt = gather_blocks(g, 2, nt) >= 160 ? 2 : 1;
  int qu = gather_blocks(g, mt, nt) < 512 ? 4 : (nt == 4 ? 4 : (nt == 2 ? 1 : 2));
  if (g.form == 1 && nt == 1 && gather_blocks(g, mt, nt) >= 512) qu = 1;     // parity-class gathers into 32 channels: 0.35 vs 0.46 ms
  if (g.form == 1 && g.sd * g.sh * g.sw > 1) {
    // parity-class gathers (1 .. 8 taps per class): short per-tile loops favour many small tiles and a shallow prefetch
    // (sweep over the stride-2 layers: <1,min(nt,2),1> is 5-20 % faster than the plain-form choice in every case)
    return launch_gather_cfg(1, nt > 2 ? 2 : nt, 1, x, wp, bias, y, g, st);
  }
  return launch_gather_cfg(mt, nt, qu, x, wp, bias, y, g, st);
}

struct WgradCfg { int MQ, MP, TH, TW; };

// ---------------------------------------------------------------- weight gradient, channels-first path
// For stride-1 "same" convolutions (the bulk of the FLOPs) the weight gradient is computed from two
// scratch copies made on the fly:
//   Pt  [B][D][H][Cp][W]              gy with every image row transposed to channel-major
//   Qp  [B][D+2pd][H+2ph][Cq][Wp]     x likewise AND zero-padded by the conv padding
// (row-blocked: the 32 channel rows a wave touches per trip lie within a few KB -- a fully channel-major
//  [C][N] copy puts them megabytes apart and the loads become translation-bound)
// With the reduction axis (voxels) contiguous, a lane loads FOUR k-values per operand with one float4
// (the channels-last kernel above needs one dword load per MFMA operand), every tap is a constant
// offset into Qp, and the zero padding removes all bounds tests from the loop.  Each 16-byte load
// feeds 4*(MQ or MP)*taps MFMAs: the same operand economy as the forward gather kernel.
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };

__global__ void __launch_bounds__(256)
transpose_pad_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int B, int D, int H, int W,
                     int pd, int ph, int pw, int Dp, int Hp, int Wp) {
  // src [B*D*H*W][C] -> dst [B][Dp][Hp][C][Wp] (interior; a padded ROW is channel-major), 64 x 32 tiles through LDS
  __shared__ float tile[32][65];
  const long V = (long)B * D * H * W;
  const long v0 = (long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 32;
  for (int i = threadIdx.x; i < 64 * 8; i += 256) {
    const int r = i >> 3, c4 = (i & 7) * 4;
    const long v = v0 + r;
    float4 t = make_float4(0, 0, 0, 0);
    if (v < V && c0 + c4 < C) t = *reinterpret_cast<const float4*>(src + (size_t)v * C + c0 + c4);
    tile[c4 + 0][r] = t.x; tile[c4 + 1][r] = t.y; tile[c4 + 2][r] = t.z; tile[c4 + 3][r] = t.w;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 32 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    const long v = v0 + r;
    if (v >= V || c0 + c >= C) continue;
    long t = v;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H); t /= H;
    const int d = (int)(t % D);
    const int b = (int)(t / D);
    dst[((((size_t)b * Dp + d + pd) * Hp + h + ph) * C + c0 + c) * Wp + w + pw] = tile[c][r];
  }
}

struct WgradCfGeom {
  int B, Cp, Cq, D, H, W;
  int kd, kh, kw, dd, dh, dw;
  int Dp, Hp, Wp;          // padded extents of Qp
  int chunk, nchunks;      // voxels per chunk (multiple of 8)
};

template <int MQ, int MP, int TH, int TW>
__global__ void __launch_bounds__(256)
wgrad_cf_kernel(const float* __restrict__ Pt, const float* __restrict__ Qp, float* __restrict__ ws, WgradCfGeom g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lk = lane >> 5;
  // XCD-aware remap: the gridDim.y*gridDim.z workgroups that walk the SAME voxel chunks (different channel
  // tiles / tap groups) get consecutive slots of ONE XCD, so they stream the chunk through one shared L2.
  unsigned bx, byy, bz;
  {
    const unsigned per = gridDim.y * gridDim.z, n = gridDim.x * per;
    const unsigned L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned xcd = L & 7, q = n >> 3, r = n & 7;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const unsigned Lp = base + (L >> 3);
    bx = Lp / per;
    byy = (Lp % per) % gridDim.y;
    bz = (Lp % per) / gridDim.y;
  }
  const int chunk_id = bx * 4 + wave;
  if (chunk_id >= g.nchunks) return;
  const int nqt = (g.Cq + 32 * MQ - 1) / (32 * MQ);
  const int qt = byy % nqt, pt = byy / nqt;
  const int kw_groups = (g.kw + TW - 1) / TW, kh_groups = (g.kh + TH - 1) / TH;
  int tg = bz;
  const int kwg = tg % kw_groups; tg /= kw_groups;
  const int khg = tg % kh_groups;
  const int kdi = tg / kh_groups;

  f32x16 acc[MQ][MP][TH][TW];
#pragma unroll
  for (int a = 0; a < MQ; ++a)
#pragma unroll
    for (int b = 0; b < MP; ++b)
#pragma unroll
      for (int c = 0; c < TH; ++c)
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][c][t][r] = 0.0f;

  const long N = (long)g.B * g.D * g.H * g.W;
  const long m_begin = (long)chunk_id * g.chunk;
  const long m_end = min(N, m_begin + g.chunk);
  // channel of this lane in every tile (clamped: out-of-range channels are discarded at the store)
  int pch[MP], qch[MQ];
#pragma unroll
  for (int a = 0; a < MP; ++a) pch[a] = min((pt * MP + a) * 32 + li, g.Cp - 1);
#pragma unroll
  for (int a = 0; a < MQ; ++a) qch[a] = min((qt * MQ + a) * 32 + li, g.Cq - 1);
  // One trip covers 32 voxels: the lane pair (li,0),(li,1) of a channel row consumes one whole 128-B line
  // (J = 4 float4 per lane), so no load relies on the line surviving in L1 until a later trip.
  // Each float4 group is 4 voxels of one row (W % 4 == 0); its (w,h,d,b) is tracked incrementally.
  constexpr int J = 4;
  long m = m_begin + 4 * J * lk;
  int w, h, d, b;
  {
    long r = min(m, N - 4);
    w = (int)(r % g.W); r /= g.W;
    h = (int)(r % g.H); r /= g.H;
    d = (int)(r % g.D);
    b = (int)(r / g.D);
  }
  auto advance = [&](int step) {
    w += step;
    while (w >= g.W) {
      w -= g.W;
      if (++h >= g.H) { h = 0; if (++d >= g.D) { d = 0; ++b; } }
    }
  };
  for (; m < m_end + 4 * J * lk; m += 8 * J) {      // same trip count in both halves; tails are masked
    float4 pv[J][MP];
    f4u qv[J][TH][TW][MQ];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const long mj = m + 4 * j;
      const bool ok = mj < m_end;
      const size_t prow = ok ? (((size_t)b * g.D + d) * g.H + h) * g.Cp : 0;     // row (b,d,h) of Pt
#pragma unroll
      for (int a = 0; a < MP; ++a)
        pv[j][a] = ok ? *reinterpret_cast<const float4*>(Pt + (prow + pch[a]) * g.W + w) : make_float4(0, 0, 0, 0);
      // padded coordinates of tap (kdi, khg*TH, kwg*TW): the conv padding is already inside Qp
      const size_t qrow = ok ? ((size_t)b * g.Dp + d + kdi * g.dd) * g.Hp + h + khg * TH * g.dh : 0;
      const int qw = ok ? w + kwg * TW * g.dw : 0;
#pragma unroll
      for (int c = 0; c < TH; ++c)
#pragma unroll
        for (int t = 0; t < TW; ++t) {
          const bool tok = ok && (khg * TH + c) < g.kh && (kwg * TW + t) < g.kw;
          const size_t row = tok ? qrow + (size_t)c * g.dh : qrow;
          const int ww = tok ? qw + t * g.dw : qw;
#pragma unroll
          for (int a = 0; a < MQ; ++a)
            qv[j][c][t][a] = *reinterpret_cast<const f4u*>(Qp + (row * g.Cq + qch[a]) * g.Wp + ww);
        }
      advance(4);
    }
    advance(4 * J);      // skip the other half-wave's 16 voxels
    // component-major order: consecutive MFMAs go to DIFFERENT accumulators (no back-to-back dependent issue)
#define SSBEV_WG_STEP(COMP)                                                                              \
    _Pragma("unroll") for (int c = 0; c < TH; ++c)                                                         \
    _Pragma("unroll") for (int t = 0; t < TW; ++t)                                                         \
    _Pragma("unroll") for (int a = 0; a < MQ; ++a)                                                         \
    _Pragma("unroll") for (int e = 0; e < MP; ++e)                                                         \
      acc[a][e][c][t] = mfma32(qv[j][c][t][a].COMP, pv[j][e].COMP, acc[a][e][c][t]);
#pragma unroll
    for (int j = 0; j < J; ++j) {
      SSBEV_WG_STEP(x) SSBEV_WG_STEP(y) SSBEV_WG_STEP(z) SSBEV_WG_STEP(w)
    }
#undef SSBEV_WG_STEP
  }
  const int taps = g.kd * g.kh * g.kw;
#pragma unroll
  for (int c = 0; c < TH; ++c)
#pragma unroll
    for (int t = 0; t < TW; ++t) {
      const int khi = khg * TH + c, kwi = kwg * TW + t;
      if (khi >= g.kh || kwi >= g.kw) continue;
      const int tap = (kdi * g.kh + khi) * g.kw + kwi;
      float* dst = ws + (((size_t)chunk_id * taps + tap) * g.Cq) * g.Cp;
#pragma unroll
      for (int a = 0; a < MQ; ++a)
#pragma unroll
        for (int e = 0; e < MP; ++e) {
          const int pc = (pt * MP + e) * 32 + li;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (qt * MQ + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (row < g.Cq && pc < g.Cp) dst[(size_t)row * g.Cp + pc] = acc[a][e][c][t][r];
          }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of a stride-1 1x1(x1) convolution: a plain [Cq x N] x [N x Cp] GEMM whose reduction axis is
// the voxel axis.  HBM-bound (both activations are read once, 2 flop per loaded byte at C = 32), so no staging:
// lane (li, lk) reads channel li of voxel v + lk straight from the channels-last rows (each half wave = one
// 128-byte line), U k-steps of loads in flight per wave, many light waves per SIMD.
template <int MQ, int MP>
__global__ void __launch_bounds__(1024)
wgrad_1x1_kernel(const float* __restrict__ P, const float* __restrict__ Q, float* __restrict__ ws, long N, int Cq,
                 int Cp, int chunk, int nchunks) {
  constexpr int U = 8;
  __shared__ float red[8][16 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const int chunk_id = blockIdx.x * 16 + wave;          // chunks past the end are empty ranges
  const int nqt = (Cq + 32 * MQ - 1) / (32 * MQ);
  const int qt = blockIdx.y % nqt, pt = blockIdx.y / nqt;
  int pch[MP], qch[MQ];
#pragma unroll
  for (int e = 0; e < MP; ++e) pch[e] = min((pt * MP + e) * 32 + li, Cp - 1);
#pragma unroll
  for (int a = 0; a < MQ; ++a) qch[a] = min((qt * MQ + a) * 32 + li, Cq - 1);
  f32x16 acc[MQ][MP];
#pragma unroll
  for (int a = 0; a < MQ; ++a)
#pragma unroll
    for (int e = 0; e < MP; ++e)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][e][r] = 0.0f;
  const long v_begin = min(N, (long)chunk_id * chunk), v_end = min(N, v_begin + chunk);
  for (long v0 = v_begin; v0 < v_end; v0 += 2 * U) {
    float pv[U][MP], qv[U][MQ];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long v = v0 + 2 * u + lk;
      const bool ok = v < v_end;
      const long vs = ok ? v : v_begin;
#pragma unroll
      for (int e = 0; e < MP; ++e) { const float t = P[vs * Cp + pch[e]]; pv[u][e] = ok ? t : 0.0f; }
#pragma unroll
      for (int a = 0; a < MQ; ++a) qv[u][a] = Q[vs * Cq + qch[a]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int a = 0; a < MQ; ++a)
#pragma unroll
        for (int e = 0; e < MP; ++e) acc[a][e] = mfma32(qv[u][a], pv[u][e], acc[a][e]);
  }
  // the 16 waves of the workgroup are folded through LDS in a fixed order (two passes of 8), one 32x32 tile at
  // a time; thread t owns element t of the tile
  float* dst = ws + (size_t)blockIdx.x * Cq * Cp;
#pragma unroll
  for (int a = 0; a < MQ; ++a)
#pragma unroll
    for (int e = 0; e < MP; ++e) {
      float t = 0.0f;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        if ((wave >> 3) == pass) {
#pragma unroll
          for (int r = 0; r < 16; ++r) red[wave & 7][r * 64 + lane] = acc[a][e][r];
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 8; ++w) t += red[w][threadIdx.x];
        __syncthreads();
      }
      const int r = threadIdx.x >> 6;
      const int row = (qt * MQ + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      const int pc = (pt * MP + e) * 32 + li;
      if (row < Cq && pc < Cp) dst[(size_t)row * Cp + pc] = t;
    }
}

struct Wgrad1x1Plan { bool ok; int MQ, MP, chunk, nchunks; long N; int Cp, Cq; };

Wgrad1x1Plan plan_wgrad_1x1(const ssbev_conv_dims* d) {
  Wgrad1x1Plan p;
  p.ok = d->kd == 1 && d->kh == 1 && d->kw == 1 && d->sd == 1 && d->sh == 1 && d->sw == 1 && d->pd == 0 && d->ph == 0 &&
         d->pw == 0 && d->Di == d->Do && d->Hi == d->Ho && d->Wi == d->Wo;
  if (!p.ok) return p;
  p.Cp = d->transposed ? d->Cin : d->Cout;
  p.Cq = d->transposed ? d->Cout : d->Cin;
  if ((long)p.Cp * p.Cq > 128L * 128L) { p.ok = false; return p; }      // wide layers are MFMA-bound: tiled kernels
  p.N = (long)d->B * d->Do * d->Ho * d->Wo;
  p.MQ = p.Cq > 32 ? 2 : 1;
  p.MP = p.Cp > 32 ? 2 : 1;
  const long tiles = (long)cdiv(p.Cq, 32 * p.MQ) * cdiv(p.Cp, 32 * p.MP);
  // ~32 waves per CU in total (16 loads in flight each), at least 256 voxels per wave, partials bounded to 64 MB
  long waves = std::max(16L, 8192 / tiles);
  waves = std::min(waves, std::max(1L, p.N / 256));
  waves = std::min(waves, std::max(1L, (64L << 20) / ((long)p.Cq * p.Cp * 4)));
  long chunk = (p.N + waves - 1) / waves;
  chunk = (chunk + 15) & ~15L;
  p.chunk = (int)chunk;
  p.nchunks = (int)cdiv((p.N + chunk - 1) / chunk, 16);       // partial tiles = workgroups (16 waves folded in LDS)
  return p;
}

// Weight gradient of 3x3 / 3x3x3 convolutions (stride 1 "same", stride-2 k3 p1, and the k3 s2 p1 op1 transposed
// convolution) with BOTH operands staged through LDS.
//
// gw[tap][cq][cp] = sum_v Q[S*v + tap - 1][cq] * P[v][cp]: P is the tensor on the coarse grid (gy of a
// convolution, x of a transposed one), Q the tensor the taps slide over, S the stride.  The reduction axis is
// the voxel axis, so the MFMA operands are Q^T and P^T.  Read from channels-last global memory that is one
// dword per lane per MFMA with every tap re-reading its shifted row through L1/L2 (PMC on the channel-major
// variant below: L1 hit rate 40 %, 5x the L2 requests of the forward kernel, one wave per SIMD, matrix pipe
// 62 % busy).  Here a workgroup walks P rows (b, d, h) of one w-segment: every step it brings S*RG new Q rows
// per kd plane (plus halo columns) and RG P rows into LDS with global_load_lds_dwordx4 (global -> LDS without
// passing through registers; border rows / halo columns / channel padding read a 16-byte zero constant
// instead, so there is no predication anywhere in the MFMA loop), and its waves take all operands from LDS:
//     P:  lds_p[row][w][cp]                           lane (li, lk) -> [w = 2ks + lk][cp = 32e + li]
//     Q:  lds_q[plane][ring slot][S*w + j][cq]        the 3x3 (i, j) taps = 3 ring slots x 3 column shifts
// At stride 1 a k-step costs 4 conflict-free LDS instructions (one ds_read_b32 + three ds_read2_b32; the third
// column of step ks is the first of step ks+1) for 9 MFMAs, and one global 16-byte load feeds 9 * 16 MFMAs.
// Q rows live in a ring of slots per plane (a row is loaded once and used by up to three P rows); the loads of
// step s+1 are issued before the MFMAs of step s, so one barrier per step suffices.
// Wave roles: wave = ((ksp * KDB + z) * MPB + e) * MQB + a -> 32 cq (a) x 32 cp (e) x kd plane z, 9 accumulators;
// KSPLIT > 1 splits the k-steps of a row between waves (their partial tiles go to separate workspace slabs).
struct WgradLdsGeom {
  int B, Dp, Hp, Wp;              // grid of P
  int Dq, Hq, Wq;                 // grid of Q
  int Cp, Cq, kd, pd, SD;         // SD = stride along d (1 for 2-D problems)
  int RG, Wseg, nseg, nrg;        // P rows per step, w-segment length (P voxels), segments per row, row groups per plane
  int NG;                         // B * Dp * nrg row groups in total
  int gpc, nranges;               // row groups per chunk, chunks per segment
  int nslot;                      // ring slots per plane
};

constexpr int kWgLdsMaxX = 8, kWgLdsMaxG = 4;    // global->LDS wave instructions (Q rows, P rows) per wave and step
__device__ const float kWgZeros[4] = {0.f, 0.f, 0.f, 0.f};

template <int MQB, int MPB, int KDB, int KSPLIT, int S, bool WINO = false, int NKS = 0>
__global__ void __launch_bounds__(64 * MQB * MPB * KDB * KSPLIT) __attribute__((amdgpu_waves_per_eu(2, 2)))
wgrad_lds_kernel(const float* __restrict__ Pg, const float* __restrict__ Qg, float* __restrict__ ws, WgradLdsGeom g) {
  constexpr int NW = MQB * MPB * KDB * KSPLIT, CQ = 32 * MQB, CP = 32 * MPB, MAXX = kWgLdsMaxX, MAXG = kWgLdsMaxG;
  extern __shared__ __align__(16) float wl[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lk = lane >> 5;
  unsigned bx, byy, bz;
  {   // XCD-aware remap: the workgroups that walk the SAME chunk (other channel tiles / kd) share one L2
    const unsigned per = gridDim.y * gridDim.z, n = gridDim.x * per;
    const unsigned L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned xcd = L & 7, q = n >> 3, r = n & 7;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const unsigned Lp = base + (L >> 3);
    bx = Lp / per;
    byy = (Lp % per) % gridDim.y;
    bz = (Lp % per) / gridDim.y;
  }
  const int chunk_id = bx;
  const int seg = chunk_id % g.nseg, range = chunk_id / g.nseg;
  const int w0 = seg * g.Wseg;
  const int nqt = (g.Cq + CQ - 1) / CQ;
  const int qt = byy % nqt, pt = byy / nqt;
  const int kd0 = bz * KDB;
  const int a = wave % MQB, e = (wave / MQB) % MPB, z = (wave / (MQB * MPB)) % KDB, ksp = wave / (MQB * MPB * KDB);

  const int ncol = S * g.Wseg + 2;               // staged Q columns: Q w = S*w0 - 1 + u
  const int xrow_f = ncol * CQ;                  // floats per staged Q row
  const int xplane_f = g.nslot * xrow_f;
  const int grow_f = g.Wseg * CP;
  const int rows_new = S * g.RG;                 // Q rows brought per step and plane
  float* xl = wl;                                // [KDB][nslot][ncol][CQ]
  float* gl = wl + KDB * xplane_f;               // [2][RG][Wseg][CP]

  // Staging work lists: one entry = one wave instruction = 64 consecutive 16-byte items of ONE row.
  // Q entries q = wave + n * NW < nxi (plane-major rows, nxc chunks per row); P entries likewise over RG * ngc.
  // Per lane: the float offset of its item relative to the step's base pointer (row, plane and column folded
  // in; -1 = read zeros, -2 = lane past the row end).  Wave-uniform: row / plane / LDS offset of the entry.
  const int xper_row = ncol * (CQ / 4), gper_row = g.Wseg * (CP / 4);
  const int nxc = (xper_row + 63) >> 6, ngc = (gper_row + 63) >> 6;
  const int nxi = KDB * rows_new * nxc, ngi = g.RG * ngc;
  const int xplane_g = g.Hq * g.Wq * g.Cq;                     // floats per (b, d) plane of Q
  int xoff[MAXX], xmeta[MAXX], goff[MAXG], gmeta[MAXG];        // meta: rr | pl << 8 | lds float offset << 10
#pragma unroll
  for (int n = 0; n < MAXX; ++n) {
    const int q = wave + n * NW;
    int off = -2, meta = -1;
    if (q < nxi) {
      const int pr = q / nxc, ch = q % nxc, pl = pr / rows_new, rr = pr % rows_new;
      const int j = ch * 64 + lane;
      if (j < xper_row) {
        const int u = j / (CQ / 4), c = qt * CQ + (j % (CQ / 4)) * 4, wsrc = S * w0 + u - 1;
        off = (wsrc >= 0 && wsrc < g.Wq && c < g.Cq) ? pl * xplane_g + (rr * g.Wq + wsrc) * g.Cq + c : -1;
      }
      meta = rr | (pl << 8) | ((pl * xplane_f + ch * 256) << 10);
    }
    xoff[n] = off;
    xmeta[n] = __builtin_amdgcn_readfirstlane(meta);
  }
#pragma unroll
  for (int n = 0; n < MAXG; ++n) {
    const int q = wave + n * NW;
    int off = -2, meta = -1;
    if (q < ngi) {
      const int rr = q / ngc, ch = q % ngc;
      const int j = ch * 64 + lane;
      if (j < gper_row) {
        const int u = j / (CP / 4), c = pt * CP + (j % (CP / 4)) * 4;
        off = c < g.Cp ? (rr * g.Wp + w0 + u) * g.Cp + c : -1;
      }
      meta = rr | ((rr * grow_f + ch * 256) << 10);
    }
    goff[n] = off;
    gmeta[n] = __builtin_amdgcn_readfirstlane(meta);
  }

  // WINO (stride 1, even RG): F(2,3) along h over PAIRS of P rows -- per k-step 4 x 3 products V_f (from four Q rows)
  // x Z_f (from two P rows) instead of 2 x 9, accumulated per frequency; the epilogue applies G^T (see conv_taph_kernel)
  static_assert(!WINO || S == 1, "the Winograd variant is stride 1");
  constexpr int NF = WINO ? 4 : 3;
  f32x16 acc[NF][3];
#pragma unroll
  for (int i = 0; i < NF; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int g_begin = range * g.gpc, g_end = min(g.NG, g_begin + g.gpc);

  // global -> LDS.  stage_q: padded Q rows [hp0, hp0 + S*RG) of planes (b, SD*d - pd + kd0 + pl) into their ring
  // slots; stage_p: P row group Gn into buffer buf.  Everything but the final address add is wave-uniform.
  auto stage_q = [&](int b, int d, int hp0) {
    const int slot0 = hp0 % g.nslot;
    const int d0 = g.SD * d + kd0 - g.pd;
    const float* base = Qg + ((long)(b * g.Dq + d0) * g.Hq + (hp0 - 1)) * (long)(g.Wq * g.Cq);
#pragma unroll
    for (int n = 0; n < MAXX; ++n) {
      const int meta = xmeta[n];
      if (meta < 0) break;
      const int rr = meta & 255, pl = (meta >> 8) & 3;
      const int h = hp0 - 1 + rr, dp = d0 + pl;
      const bool rowok = h >= 0 && h < g.Hq && dp >= 0 && dp < g.Dq;
      int slot = slot0 + rr;
      slot = slot >= g.nslot ? slot - g.nslot : slot;
      float* dst = xl + (meta >> 10) + slot * xrow_f;
      const int off = xoff[n];
      const float* src = (rowok && off >= 0) ? base + off : kWgZeros;
      if (off != -2) __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
    }
  };
  auto stage_p = [&](int Gn, int buf) {
    const float* base = Pg + (long)Gn * g.RG * (long)(g.Wp * g.Cp);
    float* dbase = gl + buf * g.RG * grow_f;
#pragma unroll
    for (int n = 0; n < MAXG; ++n) {
      const int meta = gmeta[n];
      if (meta < 0) break;
      const int off = goff[n];
      const float* src = off >= 0 ? base + off : kWgZeros;
      if (off != -2) __builtin_amdgcn_global_load_lds(src, dbase + (meta >> 10), 16, 0, 0);
    }
  };

  int cur = 0;
  bool fresh = true;                    // the Q ring does not hold this plane yet
  const int nks = (g.Wseg >> 1) / KSPLIT, ks0 = ksp * nks;     // this wave's k-steps of a row
  int hg = g_begin % g.nrg, d, b;
  {
    const int bd = g_begin / g.nrg;
    b = bd / g.Dp; d = bd % g.Dp;
  }
  for (int G = g_begin; G < g_end; ++G) {
    const int h0 = hg * g.RG;           // first P row of the step; its first padded Q row is S * h0
    if (fresh) {
      // padded Q rows [S*h0, S*h0 + S*(RG-1) + 3) in batches of S*RG; the very first step also brings its P rows
      for (int r0 = 0; r0 < S * (g.RG - 1) + 3; r0 += rows_new) stage_q(b, d, S * h0 + r0);
      if (G == g_begin) stage_p(G, cur);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    const bool has_next = G + 1 < g_end;
    const bool same_plane = has_next && hg + 1 < g.nrg;
    if (has_next) stage_p(G + 1, cur ^ 1);
    if (same_plane) stage_q(b, d, S * (h0 + g.RG) + 3 - S);

    {
      const float* gyb = gl + cur * g.RG * grow_f + e * 32 + li + (lk + 2 * ks0) * CP;
      const float* xb = xl + z * xplane_f + a * 32 + li + (lk + 2 * ks0) * S * CQ;
      int s0 = (S * h0) % g.nslot;
      if constexpr (WINO) {
        for (int rr = 0; rr < g.RG; rr += 2) {
          int s1 = s0 + 1; if (s1 >= g.nslot) s1 -= g.nslot;
          int s2 = s1 + 1; if (s2 >= g.nslot) s2 -= g.nslot;
          int s3 = s2 + 1; if (s3 >= g.nslot) s3 -= g.nslot;
          const float* q0 = xb + s0 * xrow_f;
          const float* q1 = xb + s1 * xrow_f;
          const float* q2 = xb + s2 * xrow_f;
          const float* q3 = xb + s3 * xrow_f;
          const float* pg0 = gyb + rr * grow_f;
          const float* pg1 = pg0 + grow_f;
          float c[4];
          {
            const float x0 = q0[0], x1 = q1[0], x2 = q2[0], x3 = q3[0];
            c[0] = x0 - x2; c[1] = x1 + x2; c[2] = x2 - x1; c[3] = x1 - x3;
          }
          float n1[4] = {q0[CQ], q1[CQ], q2[CQ], q3[CQ]};
          float n2[4] = {q0[2 * CQ], q1[2 * CQ], q2[2 * CQ], q3[2 * CQ]};
          float g0 = pg0[0], g1 = pg1[0];
          // NKS > 0: compile-time trip count (Wseg / 2 / KSPLIT), fully unrolled -- LDS offsets become immediates
          auto kstep = [&](int ks) {
            const float v1[4] = {n1[0] - n1[2], n1[1] + n1[2], n1[2] - n1[1], n1[1] - n1[3]};
            const float v2[4] = {n2[0] - n2[2], n2[1] + n2[2], n2[2] - n2[1], n2[1] - n2[3]};
            const float z[4] = {g0, g0 + g1, g0 - g1, -g1};
            acc[0][0] = mfma32(c[0], z[0], acc[0][0]);
            __builtin_amdgcn_sched_barrier(0);
            const int oq = 2 * ks * CQ, op = 2 * ks * CP;
            n1[0] = q0[oq + 3 * CQ]; n1[1] = q1[oq + 3 * CQ]; n1[2] = q2[oq + 3 * CQ]; n1[3] = q3[oq + 3 * CQ];
            n2[0] = q0[oq + 4 * CQ]; n2[1] = q1[oq + 4 * CQ]; n2[2] = q2[oq + 4 * CQ]; n2[3] = q3[oq + 4 * CQ];
            g0 = pg0[op + 2 * CP]; g1 = pg1[op + 2 * CP];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 1; f < 4; ++f) acc[f][0] = mfma32(c[f], z[f], acc[f][0]);
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[f][1] = mfma32(v1[f], z[f], acc[f][1]);
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[f][2] = mfma32(v2[f], z[f], acc[f][2]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < 4; ++f) c[f] = v2[f];
          };
          if constexpr (NKS > 0) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) kstep(ks);
          } else {
            for (int ks = 0; ks < nks; ++ks) kstep(ks);
          }
          s0 += 2;
          if (s0 >= g.nslot) s0 -= g.nslot;
        }
      } else
      for (int rr = 0; rr < g.RG; ++rr) {
        int s1 = s0 + 1; if (s1 >= g.nslot) s1 -= g.nslot;
        int s2 = s1 + 1; if (s2 >= g.nslot) s2 -= g.nslot;
        // All LDS addresses are (walking pointer + immediate); two k-steps per trip, the operands of the next
        // k-step in flight while the MFMAs of this one issue.  The last trip prefetches one k-step past the
        // row (never consumed; the launch reserves slack behind the P buffers for it).
        const float* p0 = xb + s0 * xrow_f;
        const float* p1 = xb + s1 * xrow_f;
        const float* p2 = xb + s2 * xrow_f;
        const float* pg = gyb + rr * grow_f;
        if constexpr (S == 1) {
          float c0 = p0[0], c1 = p1[0], c2 = p2[0];
          float pA = pg[0];
          float a01 = p0[CQ], a02 = p0[2 * CQ], a11 = p1[CQ], a12 = p1[2 * CQ], a21 = p2[CQ], a22 = p2[2 * CQ];
          for (int ks = 0; ks < nks; ks += 2) {
            acc[0][0] = mfma32(c0, pA, acc[0][0]);
            __builtin_amdgcn_sched_barrier(0);
            const float pB = pg[2 * CP];
            const float b01 = p0[3 * CQ], b02 = p0[4 * CQ];
            const float b11 = p1[3 * CQ], b12 = p1[4 * CQ];
            const float b21 = p2[3 * CQ], b22 = p2[4 * CQ];
            __builtin_amdgcn_sched_barrier(0);
            acc[1][0] = mfma32(c1, pA, acc[1][0]);
            acc[2][0] = mfma32(c2, pA, acc[2][0]);
            acc[0][1] = mfma32(a01, pA, acc[0][1]);
            acc[1][1] = mfma32(a11, pA, acc[1][1]);
            acc[2][1] = mfma32(a21, pA, acc[2][1]);
            acc[0][2] = mfma32(a02, pA, acc[0][2]);
            acc[1][2] = mfma32(a12, pA, acc[1][2]);
            acc[2][2] = mfma32(a22, pA, acc[2][2]);
            __builtin_amdgcn_sched_barrier(0);
            c0 = a02; c1 = a12; c2 = a22;
            acc[0][0] = mfma32(c0, pB, acc[0][0]);
            __builtin_amdgcn_sched_barrier(0);
            pA = pg[4 * CP];
            a01 = p0[5 * CQ]; a02 = p0[6 * CQ];
            a11 = p1[5 * CQ]; a12 = p1[6 * CQ];
            a21 = p2[5 * CQ]; a22 = p2[6 * CQ];
            __builtin_amdgcn_sched_barrier(0);
            acc[1][0] = mfma32(c1, pB, acc[1][0]);
            acc[2][0] = mfma32(c2, pB, acc[2][0]);
            acc[0][1] = mfma32(b01, pB, acc[0][1]);
            acc[1][1] = mfma32(b11, pB, acc[1][1]);
            acc[2][1] = mfma32(b21, pB, acc[2][1]);
            acc[0][2] = mfma32(b02, pB, acc[0][2]);
            acc[1][2] = mfma32(b12, pB, acc[1][2]);
            acc[2][2] = mfma32(b22, pB, acc[2][2]);
            __builtin_amdgcn_sched_barrier(0);
            c0 = b02; c1 = b12; c2 = b22;
            p0 += 4 * CQ; p1 += 4 * CQ; p2 += 4 * CQ; pg += 4 * CP;
          }
        } else {
          // stride 2: lane column = 2 * (2ks + lk) + j, consecutive k-steps share no column
          float pA = pg[0];
          float a00 = p0[0], a01 = p0[CQ], a02 = p0[2 * CQ];
          float a10 = p1[0], a11 = p1[CQ], a12 = p1[2 * CQ];
          float a20 = p2[0], a21 = p2[CQ], a22 = p2[2 * CQ];
          for (int ks = 0; ks < nks; ks += 2) {
            acc[0][0] = mfma32(a00, pA, acc[0][0]);
            __builtin_amdgcn_sched_barrier(0);
            const float pB = pg[2 * CP];
            const float b00 = p0[4 * CQ], b01 = p0[5 * CQ], b02 = p0[6 * CQ];
            const float b10 = p1[4 * CQ], b11 = p1[5 * CQ], b12 = p1[6 * CQ];
            const float b20 = p2[4 * CQ], b21 = p2[5 * CQ], b22 = p2[6 * CQ];
            __builtin_amdgcn_sched_barrier(0);
            acc[1][0] = mfma32(a10, pA, acc[1][0]);
            acc[2][0] = mfma32(a20, pA, acc[2][0]);
            acc[0][1] = mfma32(a01, pA, acc[0][1]);
            acc[1][1] = mfma32(a11, pA, acc[1][1]);
            acc[2][1] = mfma32(a21, pA, acc[2][1]);
            acc[0][2] = mfma32(a02, pA, acc[0][2]);
            acc[1][2] = mfma32(a12, pA, acc[1][2]);
            acc[2][2] = mfma32(a22, pA, acc[2][2]);
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = mfma32(b00, pB, acc[0][0]);
            __builtin_amdgcn_sched_barrier(0);
            pA = pg[4 * CP];
            a00 = p0[8 * CQ]; a01 = p0[9 * CQ]; a02 = p0[10 * CQ];
            a10 = p1[8 * CQ]; a11 = p1[9 * CQ]; a12 = p1[10 * CQ];
            a20 = p2[8 * CQ]; a21 = p2[9 * CQ]; a22 = p2[10 * CQ];
            __builtin_amdgcn_sched_barrier(0);
            acc[1][0] = mfma32(b10, pB, acc[1][0]);
            acc[2][0] = mfma32(b20, pB, acc[2][0]);
            acc[0][1] = mfma32(b01, pB, acc[0][1]);
            acc[1][1] = mfma32(b11, pB, acc[1][1]);
            acc[2][1] = mfma32(b21, pB, acc[2][1]);
            acc[0][2] = mfma32(b02, pB, acc[0][2]);
            acc[1][2] = mfma32(b12, pB, acc[1][2]);
            acc[2][2] = mfma32(b22, pB, acc[2][2]);
            __builtin_amdgcn_sched_barrier(0);
            p0 += 8 * CQ; p1 += 8 * CQ; p2 += 8 * CQ; pg += 4 * CP;
          }
        }
        s0 += S;
        if (s0 >= g.nslot) s0 -= g.nslot;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    fresh = !same_plane;
    cur ^= 1;
    if (++hg == g.nrg) {
      hg = 0;
      if (++d == g.Dp) { d = 0; ++b; }
    }
  }

  const int taps = g.kd * 9;
  const int kdi = kd0 + z;
  if (kdi < g.kd) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int tap = (kdi * 3 + i) * 3 + j;
        float* dst = ws + ((((size_t)chunk_id * KSPLIT + ksp) * taps + tap) * g.Cq) * g.Cp;
        const int pc = pt * CP + e * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = qt * CQ + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
          float v;
          if constexpr (WINO) {      // gw = G^T dU along kh
            const float hs = 0.5f * (acc[1][j][r] + acc[2][j][r]);
            v = i == 0 ? acc[0][j][r] + hs : i == 1 ? 0.5f * (acc[1][j][r] - acc[2][j][r]) : hs + acc[NF - 1][j][r];
          } else {
            v = acc[i][j][r];
          }
          if (row < g.Cq && pc < g.Cp) dst[(size_t)row * g.Cp + pc] = v;
        }
      }
  }
}

constexpr int kWgLdsRowVox = 64;                 // target P voxels (= 2 x MFMA k-steps) per step
constexpr size_t kWgLdsMaxBytes = 79 * 1024;     // LDS per workgroup (two workgroups per CU; M0-addressable)

struct WgradLdsPlan {
  WgradLdsGeom g;
  int cfg;              // 0: <2,2,1,1,1>  1: <1,1,1,4,1>  2: <1,2,1,2,2>  3: <2,2,1,1,2>
  int ksplit;
  int nchunks;          // workgroup chunks (workspace slabs = nchunks * ksplit)
  size_t lds_bytes;
  bool ok;
};

WgradLdsPlan plan_wgrad_lds_uncached(const ssbev_conv_dims* d);

// the chunking search is a few 10k iterations: done once per problem shape
WgradLdsPlan plan_wgrad_lds(const ssbev_conv_dims* d) {
  static std::mutex mu;
  static std::map<std::array<int, 21>, WgradLdsPlan> cache;
  const std::array<int, 21> key = {d->B, d->Cin, d->Cout, d->Di, d->Hi, d->Wi, d->Do, d->Ho, d->Wo, d->kd, d->kh,
                                   d->kw, d->sd, d->sh, d->sw, d->pd, d->ph, d->pw, d->dd * 64 + d->dh * 8 + d->dw,
                                   d->transposed, 0};
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  const WgradLdsPlan p = plan_wgrad_lds_uncached(d);
  cache.emplace(key, p);
  return p;
}

WgradLdsPlan plan_wgrad_lds_uncached(const ssbev_conv_dims* d) {
  WgradLdsPlan p;
  p.ok = false;
  if (d->dd != 1 || d->dh != 1 || d->dw != 1 || d->kh != 3 || d->kw != 3 || d->ph != 1 || d->pw != 1) return p;
  if ((d->kd != 1 && d->kd != 3) || d->pd != d->kd / 2 || d->Cin % 4 != 0 || d->Cout % 4 != 0) return p;
  if (d->sh != d->sw || (d->kd == 3 && d->sd != d->sh) || (d->kd == 1 && d->sd != 1)) return p;
  const int S = d->sh;
  WgradLdsGeom& g = p.g;
  g.B = d->B; g.kd = d->kd; g.pd = d->pd; g.SD = d->kd == 3 ? S : 1;
  if (S == 1) {
    if (d->transposed || d->Di != d->Do || d->Hi != d->Ho || d->Wi != d->Wo) return p;
  } else if (S == 2) {
    // conv k3 s2 p1 on even extents (out = in / 2), or its transpose k3 s2 p1 op1 (out = 2 * in)
    const int big[3] = {d->transposed ? d->Do : d->Di, d->transposed ? d->Ho : d->Hi, d->transposed ? d->Wo : d->Wi};
    const int small[3] = {d->transposed ? d->Di : d->Do, d->transposed ? d->Hi : d->Ho, d->transposed ? d->Wi : d->Wo};
    if (big[1] != 2 * small[1] || big[2] != 2 * small[2] || big[0] != g.SD * small[0]) return p;
  } else {
    return p;
  }
  if (!d->transposed) {          // P = gy on the output grid, Q = x
    g.Dp = d->Do; g.Hp = d->Ho; g.Wp = d->Wo; g.Dq = d->Di; g.Hq = d->Hi; g.Wq = d->Wi; g.Cp = d->Cout; g.Cq = d->Cin;
  } else {                       // P = x on the input grid, Q = gy
    g.Dp = d->Di; g.Hp = d->Hi; g.Wp = d->Wi; g.Dq = d->Do; g.Hq = d->Ho; g.Wq = d->Wo; g.Cp = d->Cin; g.Cq = d->Cout;
  }
  if (g.Wp % 4 != 0) return p;
  if (S == 1) p.cfg = (g.Cp <= 32 && g.Cq <= 32) ? 1 : 0;
  else p.cfg = g.Cq <= 32 ? 2 : 3;
  const int CQ = (p.cfg == 1 || p.cfg == 2) ? 32 : 64, CP = p.cfg == 1 ? 32 : 64, KDB = 1, NW = 4;
  p.ksplit = p.cfg == 1 ? 4 : (p.cfg == 2 ? 2 : 1);
  const long tiles = (long)cdiv(g.Cq, CQ) * cdiv(g.Cp, CP) * (g.kd / KDB);
  double best = 1e30;
  // cfg 1 has a Winograd-along-h variant that needs whole row pairs per step (even RG, 2/3 of the MFMAs): first look
  // for such a plan, then for any
  for (int pass = 0; pass < 2 && !p.ok; ++pass)
  for (int Wseg = 4 * p.ksplit; Wseg <= g.Wp && Wseg <= 80; Wseg += 4 * p.ksplit) {
    if (g.Wp % Wseg) continue;
    if (const char* e = ssbev_tune("SSBEV_WGL_WSEG")) { if (atoi(e) > 0 && atoi(e) != Wseg) continue; }   // tuning hook
    const int ncol = S * Wseg + 2;
    // rows per step: aim at >= 16 MFMA k-steps per wave and barrier, within the staging lists and LDS
    int RG = 1;
    while (RG * 2 <= g.Hp && g.Hp % (RG * 2) == 0 && RG * 2 <= 16 && RG * Wseg < kWgLdsRowVox * p.ksplit) RG *= 2;
    auto fits = [&](int rg) {
      return (long)KDB * S * rg * cdiv(ncol * (CQ / 4), 64) <= (long)kWgLdsMaxX * NW &&
             (long)rg * cdiv(Wseg * (CP / 4), 64) <= (long)kWgLdsMaxG * NW;
    };
    auto lds_of = [&](int rg) {
      return ((size_t)KDB * (2 * S * rg + 3 - S) * ncol * CQ + 2ul * rg * Wseg * CP) * sizeof(float);
    };
    while (RG > 1 && (!fits(RG) || lds_of(RG) > kWgLdsMaxBytes)) RG /= 2;
    const size_t lds = lds_of(RG);
    if (!fits(RG) || lds > kWgLdsMaxBytes || g.Hp % RG) continue;
    if (pass == 0 && p.cfg == 1 && RG % 2 != 0) continue;
    const int nseg = g.Wp / Wseg, nrg = g.Hp / RG;
    const long NG = (long)g.B * g.Dp * nrg;
    const long resident = 512;                       // two workgroups per CU
    const double step_cost = (double)RG * Wseg / 2 / p.ksplit + 3.0;     // MFMA k-steps + barrier / staging overhead
    for (long nr = 1; nr <= NG && nr <= 2048; ++nr) {
      const long gpc = (NG + nr - 1) / nr;
      const long nranges = (NG + gpc - 1) / gpc;
      if (nranges != nr) continue;
      const long blocks = tiles * nseg * nranges;
      const size_t wsb = (size_t)nseg * nranges * p.ksplit * g.kd * 9 * g.Cp * g.Cq * sizeof(float);
      if (wsb > (768ul << 20)) break;
      const long rounds = (blocks + resident - 1) / resident;
      const double planes = 1.0 + (double)gpc / nrg;
      const double t = rounds * (gpc * step_cost + planes * 2.5 * step_cost + 30.0) * (1.0 + 2.0 / (S * Wseg));
      if (t < best) {
        best = t;
        g.RG = RG; g.Wseg = Wseg; g.nseg = nseg; g.nrg = nrg; g.NG = (int)NG; g.gpc = (int)gpc;
        g.nranges = (int)nranges; g.nslot = 2 * S * RG + 3 - S;
        p.nchunks = (int)(nseg * nranges);
        p.lds_bytes = lds;
        p.ok = true;
      }
    }
  }
  if (ssbev_tune("SSBEV_WGL_DEBUG") && p.ok)
    fprintf(stderr, "wgrad_lds plan: cfg %d Cq %d Cp %d grid %dx%dx%d -> Wseg %d RG %d nslot %d gpc %d nranges %d nchunks %d lds %zu\n",
            p.cfg, g.Cq, g.Cp, g.Dp, g.Hp, g.Wp, g.Wseg, g.RG, g.nslot, g.gpc, g.nranges, p.nchunks, p.lds_bytes);
  return p;
}

template <int MQB, int MPB, int KDB, int KSPLIT, int S, bool WINO = false, int NKS = 0>
int launch_wgrad_lds(const float* P, const float* Q, float* ws, const WgradLdsPlan& p, hipStream_t st) {
  const WgradLdsGeom& g = p.g;
  auto kern = wgrad_lds_kernel<MQB, MPB, KDB, KSPLIT, S, WINO, NKS>;
  dim3 grid(p.nchunks, cdiv(g.Cq, 32 * MQB) * cdiv(g.Cp, 32 * MPB), g.kd / KDB), block(64 * MQB * MPB * KDB * KSPLIT);
  const size_t lds = p.lds_bytes + 1024;                                            // +1 KB: prefetch slack
  if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return SSBEV_ELAUNCH;
  hipLaunchKernelGGL(kern, grid, block, lds, st, P, Q, ws, g);
  return SSBEV_OK;
}

int run_wgrad_lds(const float* x, const float* gy, float* ws, const ssbev_conv_dims* d, const WgradLdsPlan& p,
                  hipStream_t st) {
  const float* P = d->transposed ? x : gy;
  const float* Q = d->transposed ? gy : x;
  switch (p.cfg) {
    case 0: return launch_wgrad_lds<2, 2, 1, 1, 1>(P, Q, ws, p, st);
    case 1:   // <= 32 x 32 channels: F(2,3) along h over row pairs when the step holds whole pairs (tile_hint 6: plain; 5: run-time k-step count instead of the unrolled NKS = 4 instance)
      if (p.g.RG % 2 == 0 && d->tile_hint != 6) {
        if (p.g.Wseg == 32 && d->tile_hint != 5) return launch_wgrad_lds<1, 1, 1, 4, 1, true, 4>(P, Q, ws, p, st);
        return launch_wgrad_lds<1, 1, 1, 4, 1, true>(P, Q, ws, p, st);
      }
      return launch_wgrad_lds<1, 1, 1, 4, 1>(P, Q, ws, p, st);
    case 2: return launch_wgrad_lds<1, 2, 1, 2, 2>(P, Q, ws, p, st);
    default: return launch_wgrad_lds<2, 2, 1, 1, 2>(P, Q, ws, p, st);
  }
}

// ------------------------------------------------------------------------------------------------
// 3x3x3 stride-1 "same" convolution (forward and data gradient) for <= 32 input and <= 32 output channels: the
// cost-volume layers (32 -> 32 at 192 x 48 x 160, and the 32 -> 1 / 2 -> 32 heads).  With so few channels the generic
// gather kernel above is bound by its A-operand delivery, not by the matrix pipe: every one of the 27 taps fetches
// the 128-byte voxel lines of its tile as 64 separate 16-byte L1 accesses (17 ms/step at 80 TF/s, LDS-resident
// weights included).  Here
//   * the input rows live in LDS: a workgroup walks output rows (b, d, h) of one 32-voxel w-segment and keeps a ring
//     of 4 row slots x 3 depth planes (34 voxels x 32 channels each), filled with global_load_lds_dwordx4 -- every
//     voxel line is read from L2 once per workgroup, as whole coalesced kilobytes, and used by all 27 taps;
//   * the weights live in REGISTERS: the 27 taps are dealt to the 4 waves (7/7/7/6), each wave holds its taps'
//     32 x 32 matrices as MFMA A operands (112 VGPRs) for the whole kernel: no weight traffic at all;
//   * MFMA roles: rows = output channel, columns = voxel, k = input channel.  The B operand of a tap is one
//     ds_read_b128 per 8 channels (4 k-steps); a 16-byte XOR swizzle of the channel quads (applied on the GLOBAL side
//     of the LDS load, whose LDS side must stay lane-contiguous) makes the 32 voxel lanes of a read conflict-free;
//   * the four partial 32 x 32 tiles of a row are summed through LDS in wave order (deterministic) and every wave
//     stores a quarter of the channels (+ bias, ReLU).
struct ConvTapGeom {
  int B, D, H, W, K, N;           // K input channels (multiple of 4, <= 32), N output channels (<= 32)
  int nseg, NG, gpc;              // 32-voxel segments per row, B*D*H rows, rows per chunk
  int relu, has_bias;
  int accumulate = 0;             // y += result (second consumer of a multi-consumer activation's gradient, see functional.fork)
  int Ds = 0, Hs = 0, Ws = 0;     // conv_tap2_kernel: source grid of the stride-2 gather (D / H / W are the destination grid)
  unsigned long long* dbg = nullptr;   // conv_tapdh_kernel phase clocks (tuning hook, SSBEV_TAPDH_TIMES=1)
};

constexpr int kTapWseg = 32, kTapCols = kTapWseg + 2, kTapRowF = kTapCols * 32, kTapSlots = 4;
constexpr int kTapPlaneF = kTapSlots * kTapRowF, kTapRingF = 3 * kTapPlaneF;
constexpr size_t kTapLdsBytes = (size_t)(kTapRingF + 4 * 16 * 64) * sizeof(float);
constexpr int kTapPackedElems = 4 * 7 * 16 * 64;

// w_packed[((wave * 7 + tt) * 16 + q * 4 + c) * 64 + lane] = Weff[n = lane & 31][k = 8q + 4 (lane >> 5) + c][tap = wave + 4 tt]
//   mode 0 (forward):        Weff[n][k][tap] = w[n][k][tap]            (torch layout [Cout][Cin][27])
//   mode 1 (data gradient):  Weff[n][k][tap] = w[k][n][26 - tap]       (roles swapped, taps mirrored)
__global__ void __launch_bounds__(256)
pack_tap_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int mode) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= kTapPackedElems) return;
  const int lane = i & 63, r = (i >> 6) & 15, wt = i >> 10;
  const int tt = wt % 7, wave = wt / 7;
  const int tap = wave + 4 * tt, n = lane & 31, k = 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
  const int K = mode == 0 ? Cin : Cout, N = mode == 0 ? Cout : Cin;
  float v = 0.0f;
  if (tap < 27 && n < N && k < K)
    v = mode == 0 ? w[((size_t)n * Cin + k) * 27 + tap] : w[((size_t)k * Cin + n) * 27 + (26 - tap)];
  wp[i] = v;
}

template <bool BF16>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
conv_tap_kernel(const float* __restrict__ X, const float* __restrict__ wp, const float* __restrict__ bias,
                float* __restrict__ Y, ConvTapGeom g) {
  extern __shared__ __align__(16) float tl[];
  float* ring = tl;                          // [3 planes][4 slots][34 voxels][32 channels], 16-byte swizzled
  float* red = tl + kTapRingF;               // [4 waves][16 rows][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lk = lane >> 5;

  // weights of this wave's taps: A operands, resident for the whole kernel
  float wr[BF16 ? 1 : 7][16];
  bf16x8 wb[BF16 ? 7 : 1][2];             // bf16 mode: two 16-channel operand groups per tap (56 VGPRs instead of 112)
#pragma unroll
  for (int tt = 0; tt < 7; ++tt) {
    float t16[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t16[r] = wp[((wave * 7 + tt) * 16 + r) * 64 + lane];
    if constexpr (BF16) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        wb[tt][j] = to_bf16x8(make_float4(t16[8 * j], t16[8 * j + 1], t16[8 * j + 2], t16[8 * j + 3]),
                              make_float4(t16[8 * j + 4], t16[8 * j + 5], t16[8 * j + 6], t16[8 * j + 7]));
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) wr[tt][r] = t16[r];
    }
  }
  const int ntap = wave < 3 ? 7 : 6;

  unsigned chunk_id;
  {   // XCD-aware remap: the five w-segments of a row range share one L2
    const unsigned n = gridDim.x, L = blockIdx.x;
    const unsigned xcd = L & 7, q = n >> 3, r = n & 7;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    chunk_id = base + (L >> 3);
  }
  const int seg = chunk_id % g.nseg, range = chunk_id / g.nseg;
  const int w0 = seg * kTapWseg;
  const int g_begin = range * g.gpc, g_end = min(g.NG, g_begin + g.gpc);

  // staging entries: one entry = 64 consecutive 16-byte items of one (plane, row); item = (voxel u, physical quad p4),
  // LDS offset u*32 + p4*4 (lane-contiguous), source quad p4 ^ ((u >> 1) & 7).  A voxel line is 32 floats = half the 64
  // banks, so the bank of a 16-byte read is (u & 1) * 32 + 4 * quad: the 16 lanes of one ds_read_b128 lane group
  // ({0-3, 12-15, 20-27}, ... -- MI355X_MICROARCH.md, LDS table) hold 8 even and 8 odd voxels, and the key must give the 8
  // voxels of one parity 8 different quads.  (u >> 1) & 7 does for every group and every kw shift; r2's key u & 7 takes
  // only the four values {0, 2, 4, 6} on the even voxels: a 2-way conflict on every read (PMC: SQ_LDS_BANK_CONFLICT =
  // 37.5 % of this kernel's LDS cycles, 50 % in wino_df_kernel whose slabs used the same key)
  constexpr int nxc = (kTapCols * 8 + 63) / 64;                 // 5 entries per row
  int xoff[4], xmeta[4];                                        // per wave: ceil(15 / 4) entries
  const int plane_g = g.H * g.W * g.K;
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int q = wave + n * 4;
    int off = -2, meta = -1;
    if (q < 3 * nxc) {
      const int pl = q / nxc, ch = q % nxc;
      const int j = ch * 64 + lane;
      if (j < kTapCols * 8) {
        const int u = j >> 3, c = (((j & 7) ^ ((u >> 1) & 7)) << 2), wsrc = w0 + u - 1;
        off = (wsrc >= 0 && wsrc < g.W && c < g.K) ? pl * plane_g + wsrc * g.K + c : -1;
      }
      meta = pl | ((pl * kTapPlaneF + ch * 256) << 4);
    }
    xoff[n] = off;
    xmeta[n] = __builtin_amdgcn_readfirstlane(meta);
  }
  // padded row hp (= input row hp - 1) of planes d-1, d, d+1 -> ring slot hp & 3
  auto stage_row = [&](int b, int d, int hp) {
    const float* base = X + ((long)(b * g.D + d - 1) * g.H + (hp - 1)) * (long)(g.W * g.K);
    const int h = hp - 1;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int meta = xmeta[n];
      if (meta < 0) break;
      const int pl = meta & 3, dp = d - 1 + pl;
      const bool rowok = h >= 0 && h < g.H && dp >= 0 && dp < g.D;
      float* dst = ring + (meta >> 4) + (hp & 3) * kTapRowF;
      const int off = xoff[n];
      const float* src = (rowok && off >= 0) ? base + off : kWgZeros;
      if (off != -2) glds16(src, dst);
    }
  };

  // per-tap constants: plane offset, row shift, column shift (wave-uniform)
  int tp_plane[7], tp_kh[7], tp_kw[7];
#pragma unroll
  for (int tt = 0; tt < 7; ++tt) {
    const int t = min(wave + 4 * tt, 26);
    tp_plane[tt] = (t / 9) * kTapPlaneF; tp_kh[tt] = (t / 3) % 3; tp_kw[tt] = t % 3;
  }
  const int nb = 8 * wave + 4 * lk;          // first of the 4 output channels this lane stores
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (g.has_bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = nb + i < g.N ? bias[nb + i] : 0.0f;
  }

  wait_vm0();                                 // weights and bias are in: nothing of the prologue is pending inside the loop
  bool fresh = true;
  int h = g_begin % g.H, d, b;
  {
    const int bd = g_begin / g.H;
    b = bd / g.D; d = bd % g.D;
  }
  for (int G = g_begin; G < g_end; ++G) {
    if (fresh) {
      stage_row(b, d, h); stage_row(b, d, h + 1); stage_row(b, d, h + 2);
      wait_vm0();
      __syncthreads();
    }
    const bool same_plane = G + 1 < g_end && h + 1 < g.H;
    if (same_plane) stage_row(b, d, h + 3);

    // Two accumulator chains (even / odd channel octets); the four ds_read_b128 of tap t+1 are issued right after
    // the first MFMA of tap t and consumed one tap (16 MFMAs) later.
    f32x16 acc2[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[q][r] = 0.0f;
    float4 xa[4], xb4[4];
    auto fetch = [&](int tt, float4 (&xv)[4]) {
      const int u = li + tp_kw[tt];
      const float* rowp = ring + tp_plane[tt] + ((h + tp_kh[tt]) & 3) * kTapRowF + u * 32;
      const int sw = (u >> 1) & 7;
#pragma unroll
      for (int q = 0; q < 4; ++q) xv[q] = *reinterpret_cast<const float4*>(rowp + (((2 * q + lk) ^ sw) << 2));
    };
    auto mm_head = [&](int tt, const float4 (&xv)[4]) {
      if constexpr (BF16) acc2[0] = mfma_bf16(wb[tt][0], to_bf16x8(xv[0], xv[1]), acc2[0]);
      else acc2[0] = mfma32(wr[tt][0], xv[0].x, acc2[0]);
    };
    auto mm_tail = [&](int tt, const float4 (&xv)[4]) {
      if constexpr (BF16) {
        acc2[1] = mfma_bf16(wb[tt][1], to_bf16x8(xv[2], xv[3]), acc2[1]);
        return;
      }
#pragma unroll
      for (int q = 1; q < 4; ++q) acc2[q & 1] = mfma32(wr[tt][4 * q + 0], xv[q].x, acc2[q & 1]);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc2[q & 1] = mfma32(wr[tt][4 * q + 1], xv[q].y, acc2[q & 1]);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc2[q & 1] = mfma32(wr[tt][4 * q + 2], xv[q].z, acc2[q & 1]);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc2[q & 1] = mfma32(wr[tt][4 * q + 3], xv[q].w, acc2[q & 1]);
    };
    fetch(0, xa);
#pragma unroll
    for (int tt = 0; tt < 7; tt += 2) {
      if (tt < ntap) {
        mm_head(tt, xa);
        __builtin_amdgcn_sched_barrier(0);
        if (tt + 1 < 7 && tt + 1 < ntap) fetch(tt + 1, xb4);
        __builtin_amdgcn_sched_barrier(0);
        mm_tail(tt, xa);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (tt + 1 < 7 && tt + 1 < ntap) {
        mm_head(tt + 1, xb4);
        __builtin_amdgcn_sched_barrier(0);
        if (tt + 2 < 7 && tt + 2 < ntap) fetch(tt + 2, xa);
        __builtin_amdgcn_sched_barrier(0);
        mm_tail(tt + 1, xb4);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = acc2[0][r] + acc2[1][r];
    // fold the four tap groups: every wave publishes its partial tile, then sums rows 4*wave .. 4*wave+3
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    wait_vm0();
    __syncthreads();
    {
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * wave + i;
        o[i] = ((red[(0 * 16 + r) * 64 + lane] + red[(1 * 16 + r) * 64 + lane]) + red[(2 * 16 + r) * 64 + lane]) +
               red[(3 * 16 + r) * 64 + lane] + bv[i];
        if (g.relu) o[i] = fmaxf(o[i], 0.0f);
      }
      const int wv = w0 + li;
      if (wv < g.W) {
        float* dst = Y + (((long)(b * g.D + d) * g.H + h) * g.W + wv) * g.N + nb;
        if ((g.N & 3) == 0) {
          if (nb < g.N) {
            if (g.accumulate) { const float4 t = *reinterpret_cast<const float4*>(dst); o[0] += t.x; o[1] += t.y; o[2] += t.z; o[3] += t.w; }
            *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (nb + i < g.N) dst[i] = g.accumulate ? dst[i] + o[i] : o[i];
        }
      }
    }
    barrier_lds();                         // the partial buffer is reused by the next row; the stores stay in flight
    fresh = !same_plane;
    if (++h == g.H) {
      h = 0;
      if (++d == g.D) { d = 0; ++b; }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// conv_taph_kernel: the same layers with Winograd F(2,3) along h folded into the tap walk (fp32, even H).
// conv_tap_kernel runs at ~70 % of the fp32 matrix pipe, so only fewer multiply-adds make these layers faster.  A pair
// of output rows (h0, h0+1) needs the four input rows h0-1 .. h0+2:
//     V0 = x0 - x2   V1 = x1 + x2   V2 = x2 - x1   V3 = x1 - x3        (one v_fma per operand float, sign = +-1)
//     M_f = sum over (kd, kw, k) U_f[n][k][kd][kw] * V_f               U = G w along kh (pack_taph_kernel)
//     y(h0) = M0 + M1 + M2          y(h0+1) = M1 - M2 - M3
// i.e. 4 x 9 weight matrices per row pair instead of 2 x 27: 1.5x fewer MFMAs and 3x fewer LDS operand reads.
// Eight waves: wave = (frequency f, channel half kg); each keeps its nine 32 x 16 U_f slices in 72 VGPRs, reads TWO
// ring rows per (kd, kw) and owns one 32 x 32 accumulator.  The ring holds 6 row slots x 3 planes (rows h0..h0+3 in
// use, h0+4 / h0+5 in flight); the eight partial tiles meet in LDS, where the output transform is three adds.
constexpr int kTwSlots = 6, kTwPlaneF = kTwSlots * kTapRowF, kTwRingF = 3 * kTwPlaneF;
constexpr int kTwRedF = 8 * 16 * 64;
constexpr size_t kTwLdsBytes = (size_t)(kTwRingF + 2 * kTwRedF) * sizeof(float);
constexpr int kTwPackedElems = 8 * 9 * 8 * 64;

// w_packed[((wave * 9 + c) * 8 + r) * 64 + lane] = U_f[n = lane & 31][k][kd = c / 3][kw = c % 3],  f = wave & 3,
// k = 16 (wave >> 2) + 8 (r >> 2) + 4 (lane >> 5) + (r & 3); Weff as in pack_tap_kernel (mode 1: roles swapped, mirrored)
__global__ void __launch_bounds__(256)
pack_taph_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int mode) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= kTwPackedElems) return;
  const int lane = i & 63, r = (i >> 6) & 7, wc = i >> 9;
  const int c = wc % 9, wave = wc / 9, f = wave & 3, kg = wave >> 2;
  const int kd = c / 3, kw = c % 3;
  const int n = lane & 31, k = 16 * kg + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
  const int K = mode == 0 ? Cin : Cout, N = mode == 0 ? Cout : Cin;
  float v = 0.0f;
  if (n < N && k < K) {
    float t[3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int tap = kd * 9 + kh * 3 + kw;
      t[kh] = mode == 0 ? w[((size_t)n * Cin + k) * 27 + tap] : w[((size_t)k * Cin + n) * 27 + (26 - tap)];
    }
    v = f == 0 ? t[0] : f == 1 ? 0.5f * ((t[0] + t[2]) + t[1]) : f == 2 ? 0.5f * ((t[0] + t[2]) - t[1]) : t[2];
  }
  wp[i] = v;
}

__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
conv_taph_kernel(const float* __restrict__ X, const float* __restrict__ wp, const float* __restrict__ bias,
                 float* __restrict__ Y, ConvTapGeom g) {      // g.NG / g.gpc count row PAIRS here
  extern __shared__ __align__(16) float tl[];
  float* ring = tl;                          // [3 planes][6 slots][34 voxels][32 channels], 16-byte swizzled
  float* red2 = tl + kTwRingF;               // 2 x [8 waves][16 rows][64 lanes]: double-buffered, one barrier per pair
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lk = lane >> 5;
  const int f = wave & 3, kg = wave >> 2;

  float wr[9][8];
#pragma unroll
  for (int c = 0; c < 9; ++c)
#pragma unroll
    for (int r = 0; r < 8; ++r) wr[c][r] = wp[((wave * 9 + c) * 8 + r) * 64 + lane];

  unsigned chunk_id;
  {
    const unsigned n = gridDim.x, L = blockIdx.x;
    const unsigned xcd = L & 7, q = n >> 3, r = n & 7;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    chunk_id = base + (L >> 3);
  }
  const int seg = chunk_id % g.nseg, range = chunk_id / g.nseg;
  const int H2 = g.H >> 1;
  const int g_begin = range * g.gpc, g_end = min(g.NG, g_begin + g.gpc);
  const int w0 = seg * kTapWseg;

  constexpr int nxc = (kTapCols * 8 + 63) / 64;                 // 5 staging entries per (plane, row), see conv_tap_kernel
  int xoff[2], xmeta[2];
  const int plane_g = g.H * g.W * g.K;
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int q = wave + n * 8;
    int off = -2, meta = -1;
    if (q < 3 * nxc) {
      const int pl = q / nxc, ch = q % nxc;
      const int j = ch * 64 + lane;
      if (j < kTapCols * 8) {
        const int u = j >> 3, c = (((j & 7) ^ ((u >> 1) & 7)) << 2), wsrc = w0 + u - 1;
        off = (wsrc >= 0 && wsrc < g.W && c < g.K) ? pl * plane_g + wsrc * g.K + c : -1;
      }
      meta = pl | ((pl * kTwPlaneF + ch * 256) << 4);
    }
    xoff[n] = off;
    xmeta[n] = __builtin_amdgcn_readfirstlane(meta);
  }
  auto stage_row = [&](int b, int d, int hp) {                  // padded row hp = input row hp - 1 -> slot hp % 6
    const float* base = X + ((long)(b * g.D + d - 1) * g.H + (hp - 1)) * (long)(g.W * g.K);
    const int h = hp - 1;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int meta = xmeta[n];
      if (meta < 0) break;
      const int pl = meta & 3, dp = d - 1 + pl;
      const bool rowok = h >= 0 && h < g.H && dp >= 0 && dp < g.D;
      float* dst = ring + (meta >> 4) + (hp % kTwSlots) * kTapRowF;
      const int off = xoff[n];
      const float* src = (rowok && off >= 0) ? base + off : kWgZeros;
      if (off != -2) glds16(src, dst);
    }
  };

  const int ra = f == 0 ? 0 : (f == 2 ? 2 : 1);
  const int rb = f == 3 ? 3 : (f == 2 ? 1 : 2);
  const float sgn = f == 1 ? 1.0f : -1.0f;
  const int j_out = wave >> 2, rg = wave & 3;
  const int nb = 8 * rg + 4 * lk;            // first of the 4 output channels this lane stores (row h0 + j_out)
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (g.has_bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = nb + i < g.N ? bias[nb + i] : 0.0f;
  }

  wait_vm0();                                 // weights and bias are in: nothing of the prologue is pending inside the loop
  bool fresh = true;
  int h2 = g_begin % H2, d, b;
  {
    const int bd = g_begin / H2;
    b = bd / g.D; d = bd % g.D;
  }
  for (int G = g_begin; G < g_end; ++G) {
    const int h0 = 2 * h2;
    if (fresh) {
      stage_row(b, d, h0); stage_row(b, d, h0 + 1); stage_row(b, d, h0 + 2); stage_row(b, d, h0 + 3);
      wait_vm0();
      __syncthreads();
    }
    const bool same_plane = G + 1 < g_end && h2 + 1 < H2;
    if (same_plane) { stage_row(b, d, h0 + 4); stage_row(b, d, h0 + 5); }

    const int offA = ((h0 + ra) % kTwSlots) * kTapRowF, offB = ((h0 + rb) % kTwSlots) * kTapRowF;
    float* red = red2 + (G & 1) * kTwRedF;
    // y += result: the old values of this lane's output quad are requested here and have the whole tap walk to arrive
    float4 told = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.accumulate && (g.N & 3) == 0 && w0 + li < g.W && nb < g.N)
      told = *reinterpret_cast<const float4*>(Y + (((long)(b * g.D + d) * g.H + h0 + j_out) * g.W + w0 + li) * g.N + nb);
    f32x16 acc2[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[q][r] = 0.0f;
    float4 va[2], vb[2];
    auto fetch = [&](int c, float4 (&v)[2]) {
      const int u = li + c % 3;
      const float* colp = ring + (c / 3) * kTwPlaneF + u * 32;
      const int sw = (u >> 1) & 7;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int o = ((2 * (2 * kg + q) + lk) ^ sw) << 2;
        const float4 xa = *reinterpret_cast<const float4*>(colp + offA + o);
        const float4 xb = *reinterpret_cast<const float4*>(colp + offB + o);
        v[q] = make_float4(fmaf(sgn, xb.x, xa.x), fmaf(sgn, xb.y, xa.y), fmaf(sgn, xb.z, xa.z), fmaf(sgn, xb.w, xa.w));
      }
    };
    auto mm_head = [&](int c, const float4 (&v)[2]) { acc2[0] = mfma32(wr[c][0], v[0].x, acc2[0]); };
    auto mm_tail = [&](int c, const float4 (&v)[2]) {
      acc2[1] = mfma32(wr[c][4], v[1].x, acc2[1]);
      acc2[0] = mfma32(wr[c][1], v[0].y, acc2[0]);
      acc2[1] = mfma32(wr[c][5], v[1].y, acc2[1]);
      acc2[0] = mfma32(wr[c][2], v[0].z, acc2[0]);
      acc2[1] = mfma32(wr[c][6], v[1].z, acc2[1]);
      acc2[0] = mfma32(wr[c][3], v[0].w, acc2[0]);
      acc2[1] = mfma32(wr[c][7], v[1].w, acc2[1]);
    };
    // (r3: requesting the raw rows after the first MFMA and combining them only after the eighth was measured at the same
    // 0.557 ms per launch -- tools/taph_probe.py -- so the LDS latency of this step is not what holds the kernel at 63 %;
    // so was a phase-shifted schedule for the second wave of every SIMD (its non-MFMA window after the fourth MFMA of a
    // column instead of after the first, so that the two waves that leave each barrier together do not idle the pipe
    // together): 0.568 ms)
    fetch(0, va);
#pragma unroll
    for (int c = 0; c < 9; c += 2) {
      mm_head(c, va);
      __builtin_amdgcn_sched_barrier(0);
      if (c + 1 < 9) fetch(c + 1, vb);
      __builtin_amdgcn_sched_barrier(0);
      mm_tail(c, va);
      __builtin_amdgcn_sched_barrier(0);
      if (c + 1 < 9) {
        mm_head(c + 1, vb);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 2 < 9) fetch(c + 2, va);
        __builtin_amdgcn_sched_barrier(0);
        mm_tail(c + 1, vb);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc2[0][r] + acc2[1][r];
    wait_vm0();
    __syncthreads();
    {
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * rg + i;
        float m[4];
#pragma unroll
        for (int ff = 0; ff < 4; ++ff) m[ff] = red[(ff * 16 + r) * 64 + lane] + red[((ff + 4) * 16 + r) * 64 + lane];
        o[i] = (j_out == 0 ? (m[0] + m[1]) + m[2] : (m[1] - m[2]) - m[3]) + bv[i];
        if (g.relu) o[i] = fmaxf(o[i], 0.0f);
      }
      const int wv = w0 + li;
      if (wv < g.W) {
        float* dst = Y + (((long)(b * g.D + d) * g.H + h0 + j_out) * g.W + wv) * g.N + nb;
        if ((g.N & 3) == 0) {
          if (nb < g.N)
            *reinterpret_cast<float4*>(dst) = make_float4(o[0] + told.x, o[1] + told.y, o[2] + told.z, o[3] + told.w);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (nb + i < g.N) dst[i] = g.accumulate ? dst[i] + o[i] : o[i];
        }
      }
    }
    fresh = !same_plane;                     // no second barrier: the next pair publishes into the other half of red2
    if (++h2 == H2) {
      h2 = 0;
      if (++d == g.D) { d = 0; ++b; }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// conv_tapdh_kernel (round 4): the same <= 32-channel layers with Winograd F(2,3) along d AND h (fp32, even D and H).
// conv_taph_kernel sits at 0.61 of the fp32 matrix pipe whatever is done to its schedule (round 3); what is left is fewer
// multiply-adds.  A block of 2 x 2 output (plane, row) pairs needs the 4 x 4 input (plane, row)s d0-1 .. d0+2, h0-1 .. h0+2:
//     V[fd][fh] = B^T x B            (B^T of F(2,3): V0 = x0 - x2, V1 = x1 + x2, V2 = x2 - x1, V3 = x1 - x3, on both axes)
//     M[fd][fh] = sum over (kw, k) U[fd][fh][n][k][kw] * V[fd][fh]            U = G w G^T over (kd, kh)  (pack_tapdh_kernel)
//     y = A^T M A                    (y0 = M0 + M1 + M2, y1 = M1 - M2 - M3, on both axes)
// 16 x 3 weight matrices per 2 x 2 outputs instead of 4 x 27: 2.25x fewer MFMAs than the direct kernel (taph: 1.5x), +-1
// transforms only (no F(4,3) constants: the rounding behaviour of taph).  Sixteen waves (1024 threads, four per SIMD at
// <= 128 VGPRs, so a wave's LDS reads and its three add/subs per operand hide behind the other waves' MFMAs): wave =
// frequency (fd, fh) keeps its three 32 x 32 U slices (kw) in 48 VGPRs, reads FOUR ring rows per operand quad (2 planes x 2
// rows) and owns one 32 x 32 accumulator.  The ring holds 6 row slots x 4 planes: rows h0 .. h0+3 in use, h0+4 AND h0+5 in
// flight during the walk (round 6: with 5 slots the second new row could only be requested behind the walk and its HBM
// latency stood between the fold and the next walk).  The sixteen partial tiles meet in LDS, where the output transform is a
// signed sum of nine tiles per output: eight tiles in a 32 KB fold buffer, the other eight in the ring slots of rows h0 and
// h0+1 of the four planes, which are dead from the end of the walk until the next block's rows are requested into them
// (ring 102 KB + fold 32 KB + 16 KB for the old values of an accumulating launch = 150 KB; a separate 64 KB fold buffer
// beside six slots does not fit the CU's 160 KB).  A
// workgroup stages 4 planes for 2 output planes: every input plane is fetched twice instead of taph's three times.
constexpr int kDhSlots = 6, kDhPlaneF = kDhSlots * kTapRowF, kDhRingF = 4 * kDhPlaneF;
constexpr int kDhRedTiles = 8, kDhRedF = kDhRedTiles * 16 * 64;
static_assert(kTapRowF >= 16 * 64, "conv_tapdh_kernel: a partial tile must fit a dead ring row");
constexpr int kDhOldF = 16 * 64 * 4;         // `accumulate`: the old values of a block's outputs, one float4 per thread
constexpr size_t kDhLdsBytes = (size_t)(kDhRingF + kDhRedF + 32 + kDhOldF) * sizeof(float);       // + bias row + old values
constexpr int kDhPackedElems = 16 * 3 * 16 * 64;
static_assert(kDhLdsBytes <= 160 * 1024, "conv_tapdh_kernel: ring + fold buffer must fit the CU's LDS");

// F(2,3) weight transform G along one axis: (t0, t1, t2) -> frequency f
__device__ __forceinline__ float wino_g23(float t0, float t1, float t2, int f) {
  return f == 0 ? t0 : f == 1 ? 0.5f * ((t0 + t2) + t1) : f == 2 ? 0.5f * ((t0 + t2) - t1) : t2;
}

// w_packed[((wave * 3 + kw) * 16 + r) * 64 + lane] = U[fd = wave >> 2][fh = wave & 3][n = lane & 31][k][kw],
// k = 8 (r >> 2) + 4 (lane >> 5) + (r & 3); Weff as in pack_tap_kernel (mode 1: roles swapped, taps mirrored)
__global__ void __launch_bounds__(256)
pack_tapdh_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int mode) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= kDhPackedElems) return;
  const int lane = i & 63, r = (i >> 6) & 15, wk = i >> 10;
  const int kw = wk % 3, wave = wk / 3, fd = wave >> 2, fh = wave & 3;
  const int n = lane & 31, k = 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
  const int K = mode == 0 ? Cin : Cout, N = mode == 0 ? Cout : Cin;
  float v = 0.0f;
  if (n < N && k < K) {
    float th[3];
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) {
      float t[3];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int tap = kd * 9 + kh * 3 + kw;
        t[kh] = mode == 0 ? w[((size_t)n * Cin + k) * 27 + tap] : w[((size_t)k * Cin + n) * 27 + (26 - tap)];
      }
      th[kd] = wino_g23(t[0], t[1], t[2], fh);
    }
    v = wino_g23(th[0], th[1], th[2], fd);
  }
  wp[i] = v;
}

__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4)))
conv_tapdh_kernel(const float* __restrict__ X, const float* __restrict__ wp, const float* __restrict__ bias,
                  float* __restrict__ Y, ConvTapGeom g) {     // g.NG / g.gpc count 2 x 2 (plane, row) blocks
  extern __shared__ __align__(16) float tl[];
  float* ring = tl;                          // [4 planes][6 slots][34 voxels][32 channels], 16-byte swizzled
  // partial tile t ([32 voxels][8 channel quads][4], see the publish phase) of a block whose first row sits in slot s0: tiles 0-7 in the fold buffer behind the
  // ring, tiles 8-15 in the ring rows (plane (t - 8) & 3, row h0 + ((t - 8) >> 2)) that the finished walk has left dead
  auto tile_off = [](int t, int s0) {
    const int sl = s0 + ((t - kDhRedTiles) >> 2);             // s0 is even, < kDhSlots: no wrap
    return t < kDhRedTiles ? kDhRingF + t * 1024 : ((t - kDhRedTiles) & 3) * kDhPlaneF + sl * kTapRowF;
  };
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lk = lane >> 5;
  const int fd = wave >> 2, fh = wave & 3;

  float wr[3][16];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) wr[c][r] = wp[((wave * 3 + c) * 16 + r) * 64 + lane];

  unsigned chunk_id;
  {
    const unsigned n = gridDim.x, L = blockIdx.x;
    const unsigned xcd = L & 7, q = n >> 3, r = n & 7;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    chunk_id = base + (L >> 3);
  }
  const int seg = chunk_id % g.nseg, range = chunk_id / g.nseg;
  const int H2 = g.H >> 1, D2 = g.D >> 1;
  const int g_begin = range * g.gpc, g_end = min(g.NG, g_begin + g.gpc);
  const int w0 = seg * kTapWseg;

  constexpr int nxc = (kTapCols * 8 + 63) / 64;                 // 5 staging entries per (plane, row), see conv_tap_kernel
  int xoff[2], xmeta[2];                                        // 4 planes x 5 entries = 20 per row: waves 0-3 take two
  const int plane_g = g.H * g.W * g.K;                          // (the per-lane offsets stay in registers: recomputing them per
                                                                //  request measured 800 clocks of VALU per row and SIMD)
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int q = wave + n * 16;
    int off = -2, meta = -1;
    if (q < 4 * nxc) {
      const int pl = q / nxc, ch = q % nxc;
      const int j = ch * 64 + lane;
      if (j < kTapCols * 8) {
        const int u = j >> 3, c = (((j & 7) ^ ((u >> 1) & 7)) << 2), wsrc = w0 + u - 1;
        off = (wsrc >= 0 && wsrc < g.W && c < g.K) ? pl * plane_g + wsrc * g.K + c : -1;
      }
      meta = pl | ((pl * kDhPlaneF + ch * 256) << 4);
    }
    xoff[n] = off;
    xmeta[n] = __builtin_amdgcn_readfirstlane(meta);
  }
  // padded row hp (= input row hp - 1) of the planes 2 d2 - 1 .. 2 d2 + 2 -> slot hp % 6
  auto stage_row = [&](int b, int d2, int hp) {
    const float* base = X + ((long)(b * g.D + 2 * d2 - 1) * g.H + (hp - 1)) * (long)(g.W * g.K);
    const int h = hp - 1;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int meta = xmeta[n];
      if (meta < 0) break;
      const int pl = meta & 3, dp = 2 * d2 - 1 + pl;
      const bool rowok = h >= 0 && h < g.H && dp >= 0 && dp < g.D;
      float* dst = ring + (meta >> 4) + (hp % kDhSlots) * kTapRowF;
      const int off = xoff[n];
      const float* src = (rowok && off >= 0) ? base + off : kWgZeros;
      if (off != -2) glds16(src, dst);
    }
  };

  // B^T of F(2,3): frequency f combines inputs (a, b) with sign s:  0: x0 - x2   1: x1 + x2   2: x2 - x1   3: x1 - x3
  const int ra = fh == 0 ? 0 : (fh == 2 ? 2 : 1), rb = fh == 3 ? 3 : (fh == 2 ? 1 : 2);
  const float sh = fh == 1 ? 1.0f : -1.0f;
  const int pa = fd == 0 ? 0 : (fd == 2 ? 2 : 1), pb = fd == 3 ? 3 : (fd == 2 ? 1 : 2);
  const float sd = fd == 1 ? 1.0f : -1.0f;
  const int offPA = pa * kDhPlaneF, offPB = pb * kDhPlaneF;
  // store phase: wave = (output plane jd, output row jh, voxel octet vq): lane = (voxel 8 vq + (lane >> 3), channel quad
  // lane & 7), so that a wave's one 16-byte store per lane covers eight whole 128-byte voxels (round 6; a wave used to own
  // 8 channels of 32 voxels: sixty-four 32-byte pieces per store instruction, 1 k clocks of address processing per block)
  const int jd = wave >> 3, jh = (wave >> 2) & 1, vq = wave & 3;
  // Register discipline (round 6): the walk owns the file -- 48 weight registers, the accumulator, two operand quads and the
  // eight ring reads behind them -- and a spill reload anywhere between the row requests and the end of the walk is a
  // `s_waitcnt vmcnt(0)`, i.e. a wait for the rows (vmcnt counts the LDS-DMA too).  So nothing per-lane that only the fold needs
  // lives across the walk: it derives its lane constants from a lane id the compiler cannot see through (no hoisting out of
  // the block loop), the bias sits in LDS, the store address is formed behind the walk, old values travel by LDS-DMA.
  auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
  float* biasl = tl + kDhRingF + kDhRedF;
  if (tid < 32) biasl[tid] = (g.has_bias && tid < g.N) ? bias[tid] : 0.0f;      // read behind the first block's barriers
  // accumulate: the old output values travel global -> LDS beside the rows (no registers across the walk, and no load the
  // compiler would have to wait for in front of it)
  float* oldl = biasl + 32 + wave * 256;
  const bool acc4 = g.accumulate && (g.N & 3) == 0;
  // A^T of F(2,3): output j sums the frequencies j, j + 1, j + 2 with signs (+, +, +) for j = 0 and (+, -, -) for j = 1
  const float s1 = 1.0f - 2.0f * jh, t1 = 1.0f - 2.0f * jd;     // sign of the 2nd and 3rd term along h / d

  wait_vm0();                                 // weights and bias are in: nothing of the prologue is pending inside the loop
  unsigned long long tk[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long t_prev = __builtin_readcyclecounter();
  const unsigned long long t_start = t_prev;
#ifdef SSBEV_TAPDH_CLOCKS       // build-time tuning hook (-DSSBEV_TAPDH_CLOCKS + SSBEV_TAPDH_TIMES=1): per-phase shader clocks
#define TAPDH_TICK(i) if (g.dbg) { const unsigned long long t_now = __builtin_readcyclecounter(); tk[i] += t_now - t_prev; t_prev = t_now; }
#else
#define TAPDH_TICK(i)
#endif
  bool fresh = true;
  int h2 = g_begin % H2, d2, b;
  {
    const int bd = g_begin / H2;
    b = bd / D2; d2 = bd % D2;
  }
  // the four (plane, row) ring offsets of a block whose first row sits in slot s0, as opaque scalars: left to itself the compiler
  // adds the loop-invariant plane offsets into the twelve per-lane column bases ahead of the loop and keeps 24 address registers
  // across the walk
  int oAA = 0, oAB = 0, oBA = 0, oBB = 0;
  auto block_offsets = [&](int s0) {
    const int offA = ((s0 + ra) % kDhSlots) * kTapRowF, offB = ((s0 + rb) % kDhSlots) * kTapRowF;
    oAA = offPA + offA; oAB = offPA + offB; oBA = offPB + offA; oBB = offPB + offB;
    asm volatile("" : "+s"(oAA), "+s"(oAB), "+s"(oBA), "+s"(oBB));
  };
  // operand quad (kw = c, channel quad q): four ring reads, B^T of F(2,3) on both axes
  auto fetch = [&](int c, int q, float4& v) {
    const int u = li + c;
    const float* colp = ring + u * 32 + ((((2 * q + lk) ^ ((u >> 1) & 7))) << 2);
    const float4 aa = *reinterpret_cast<const float4*>(colp + oAA);
    const float4 ab = *reinterpret_cast<const float4*>(colp + oAB);
    const float4 ba = *reinterpret_cast<const float4*>(colp + oBA);
    const float4 bb = *reinterpret_cast<const float4*>(colp + oBB);
    const float4 pa4 = make_float4(fmaf(sh, ab.x, aa.x), fmaf(sh, ab.y, aa.y), fmaf(sh, ab.z, aa.z), fmaf(sh, ab.w, aa.w));
    const float4 pb4 = make_float4(fmaf(sh, bb.x, ba.x), fmaf(sh, bb.y, ba.y), fmaf(sh, bb.z, ba.z), fmaf(sh, bb.w, ba.w));
    v = make_float4(fmaf(sd, pb4.x, pa4.x), fmaf(sd, pb4.y, pa4.y), fmaf(sd, pb4.z, pa4.z), fmaf(sd, pb4.w, pa4.w));
  };
  float4 va, vb;
  for (int G = g_begin; G < g_end; ++G) {
    const int h0 = 2 * h2;
    if (fresh) {
      stage_row(b, d2, h0); stage_row(b, d2, h0 + 1); stage_row(b, d2, h0 + 2); stage_row(b, d2, h0 + 3);
      wait_vm0();
      __syncthreads();
    }
    TAPDH_TICK(0)
    const bool same_plane = G + 1 < g_end && h2 + 1 < H2;
    const int s0 = h0 % kDhSlots;
    if (acc4) {
      const int ln = opaque(lane), sl = 8 * vq + (ln >> 3), nb = 4 * (ln & 7);
      const float* src = (w0 + sl < g.W && nb < g.N)
          ? Y + (((long)(b * g.D + 2 * d2 + jd) * g.H + h0 + jh) * g.W + w0) * g.N + (sl * g.N + nb) : kWgZeros;
      glds16(src, oldl);
    }
    block_offsets(s0);
    fetch(0, 0, va);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 12; s += 2) {          // step s = (kw = s / 4, channel quad = s % 4): 4 MFMAs each
      // rows h0 + 4, h0 + 5 are requested INSIDE the walk, into the slots of rows h0 - 2, h0 - 1 (the previous block's fold has
      // read its tiles there): measured, requests at the head of the walk cost it 400 clocks per row, after the first third
      // they cost nothing (steps 2 .. 10 measure alike) and land long before the publish phase waits for them
      if (s == 4 && same_plane) {
        stage_row(b, d2, h0 + 4);
        stage_row(b, d2, h0 + 5);
      }
      // issue priority falls as a wave advances (2, 1, 0 over the twelve steps; 3 outside the walk): the arbiter serves the
      // OLDEST ready wave first, which left alone makes the four waves of a SIMD finish their walks 2 k clocks apart (measured
      // per wave slot: 8.5 / 10.3 / 12.6 / 16 k clocks after the block's start) -- the youngest one runs its last third alone,
      // at a single wave's latency-bound rate, with fifteen waves parked at the barrier.  With the priorities the four finish
      // within 1.2 k clocks of each other and the block takes 18.7 k clocks instead of 20.0 k
      if (s == 0) __builtin_amdgcn_s_setprio(2);
      if (s == 4) __builtin_amdgcn_s_setprio(1);
      if (s == 8) __builtin_amdgcn_s_setprio(0);
      fetch((s + 1) >> 2, (s + 1) & 3, vb);
      {
        const int c = s >> 2, q = s & 3;
        acc = mfma32(wr[c][4 * q + 0], va.x, acc);
        acc = mfma32(wr[c][4 * q + 1], va.y, acc);
        acc = mfma32(wr[c][4 * q + 2], va.z, acc);
        acc = mfma32(wr[c][4 * q + 3], va.w, acc);
      }
      if (s + 2 < 12) fetch((s + 2) >> 2, (s + 2) & 3, va);
      {
        const int c = (s + 1) >> 2, q = (s + 1) & 3;
        acc = mfma32(wr[c][4 * q + 0], vb.x, acc);
        acc = mfma32(wr[c][4 * q + 1], vb.y, acc);
        acc = mfma32(wr[c][4 * q + 2], vb.z, acc);
        acc = mfma32(wr[c][4 * q + 3], vb.w, acc);
      }
    }
    TAPDH_TICK(1)
    __builtin_amdgcn_s_setprio(3);                            // publish, fold and stores ahead of the other waves' walks
    // store phase: voxel sl of the segment, nb = first of the 4 output channels this lane stores
    const int sln = opaque(lane), sl = 8 * vq + (sln >> 3), nb = 4 * (sln & 7);
    float* dst = Y + (((long)(b * g.D + 2 * d2 + jd) * g.H + h0 + jh) * g.W + w0) * g.N + (sl * g.N + nb);
    {
      // tile = [voxel 32][channel quad 8, XOR-swizzled by (voxel >> 1) & 7][4]: the output's own layout.  The publishing lane
      // (voxel li, half lk) holds the channel quads 2 r + lk in its registers 4 r .. 4 r + 3
      const int pli = sln & 31, plk = sln >> 5, psw = (pli >> 1) & 7;
      float* pub = tl + tile_off(wave, s0) + pli * 32;
      if (wave < kDhRedTiles) {
#pragma unroll
        for (int r = 0; r < 16; r += 4)
          *reinterpret_cast<float4*>(pub + ((((r >> 1) + plk) ^ psw) << 2)) = make_float4(acc[r], acc[r + 1], acc[r + 2], acc[r + 3]);
      }
      barrier_lds();                                            // every wave is done with rows h0, h0 + 1: their slots take tiles
      if (wave >= kDhRedTiles) {
#pragma unroll
        for (int r = 0; r < 16; r += 4)
          *reinterpret_cast<float4*>(pub + ((((r >> 1) + plk) ^ psw) << 2)) = make_float4(acc[r], acc[r + 1], acc[r + 2], acc[r + 3]);
      }
    }
    wait_vm0();            // rows h0 + 4, h0 + 5 have landed (and the previous block's stores)
    __syncthreads();                                            // every wave has published
    TAPDH_TICK(2)
    {
      // tile (fd = jd + a, fh = jh + c), this lane's (voxel, channel quad): two per-lane bases (fold buffer / dead ring rows), everything else is
      // an immediate offset of the read once the branch on jd (wave-uniform) has fixed which fd lives where.  All nine reads are
      // requested before the first add (a wait per tile made this phase 3.4 k clocks of LDS latency).
      const int fo = sl * 32 + (((sln & 7) ^ ((sl >> 1) & 7)) << 2);
      const float* redp = tl + kDhRingF + jh * 1024 + fo;
      const float* deadp = tl + jh * kDhPlaneF + s0 * kTapRowF + fo;
      const float4 bv4 = *reinterpret_cast<const float4*>(biasl + nb);
      float4 told = make_float4(0.f, 0.f, 0.f, 0.f);
      if (acc4) told = *reinterpret_cast<const float4*>(oldl + 4 * sln);
      const float bv[4] = {bv4.x, bv4.y, bv4.z, bv4.w};
      // nine 16-byte reads by asm: with plain loads the compiler merges the two branches into one sequence of 36 four-byte reads
      // behind selected addresses
      typedef float f32x4_t __attribute__((ext_vector_type(4)));
      f32x4_t mv[3][3];
      const unsigned redb = (unsigned)(size_t)redp, deadb = (unsigned)(size_t)deadp;
#define DH_RD(dst, base, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(off))
#define DH_RD_RED(a, fd)  DH_RD(mv[a][0], redb, ((fd) * 4 + 0) * 4096); DH_RD(mv[a][1], redb, ((fd) * 4 + 1) * 4096); \
                          DH_RD(mv[a][2], redb, ((fd) * 4 + 2) * 4096)
#define DH_RD_DEAD(a, fd) DH_RD(mv[a][0], deadb, ((fd) - 2) * kTapRowF * 4); DH_RD(mv[a][1], deadb, (((fd) - 2) * kTapRowF + kDhPlaneF) * 4); \
                          DH_RD(mv[a][2], deadb, (((fd) - 2) * kTapRowF + 2 * kDhPlaneF) * 4)
      if (jd == 0) { DH_RD_RED(0, 0); DH_RD_RED(1, 1); DH_RD_DEAD(2, 2); }
      else         { DH_RD_RED(0, 1); DH_RD_DEAD(1, 2); DH_RD_DEAD(2, 3); }
#undef DH_RD_DEAD
#undef DH_RD_RED
#undef DH_RD
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // the tiles are read: their slots take the next rows (requested inside the next walk), the fold buffer the next tiles.
      // The output transform and the stores run behind this barrier, each wave at its own pace.  (Requesting the next block's
      // first operand quad here as well -- a software pipeline across blocks -- needs four more registers than the file has:
      // ten spill slots, with reloads inside the walk.)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      TAPDH_TICK(3)
      float m[3][3][4];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int i = 0; i < 4; ++i) m[a][c][i] = mv[a][c][i];
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float t[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) t[a] = (m[a][0][i] + s1 * m[a][1][i]) + s1 * m[a][2][i];
        o[i] = ((t[0] + t1 * t[1]) + t1 * t[2]) + bv[i];
        if (g.relu) o[i] = fmaxf(o[i], 0.0f);
      }
      if (w0 + sl < g.W) {
        if ((g.N & 3) == 0) {
          if (nb < g.N)
            *reinterpret_cast<float4*>(dst) = make_float4(o[0] + told.x, o[1] + told.y, o[2] + told.z, o[3] + told.w);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (nb + i < g.N) dst[i] = g.accumulate ? dst[i] + o[i] : o[i];
        }
      }
    }
    TAPDH_TICK(4)
    fresh = !same_plane;
    if (++h2 == H2) {
      h2 = 0;
      if (++d2 == D2) { d2 = 0; ++b; }
    }
  }
  if (g.dbg && lane == 0) {
    unsigned long long* o = g.dbg + ((size_t)blockIdx.x * 16 + wave) * 8;
#pragma unroll
    for (int i = 0; i < 5; ++i) o[i] = tk[i];
    o[5] = t_start; o[6] = __builtin_readcyclecounter(); o[7] = g_end - g_begin;
  }
#undef TAPDH_TICK
}

// ------------------------------------------------------------------------------------------------
// wgrad_tapdh_kernel (round 4): weight gradient of the same <= 32-channel stride-1 3x3x3 layers in the F(2,3) x F(2,3)
// domain over (d, h) -- the transpose of conv_tapdh_kernel:
//     gU[fd][fh][kw][k][n] = sum over 2 x 2 (plane, row) blocks and voxels v of  V[fd][fh](x)[v + kw][k] * Z[fd][fh](gy)[v][n]
//     V = B^T x B (as in the forward kernel)        Z = A gy A^T  (Z0 = g0, Z1 = g0 + g1, Z2 = g0 - g1, Z3 = -g1 per axis)
//     gw = G^T gU G over (kd, kh)                    (wgrad_dh_finish_kernel, after the fixed-order fold of the chunk partials)
// 16 x 3 products per 2 x 2 outputs instead of 4 x 27 (wgrad_lds_kernel<.., WINO>: 4 x 9 per row pair = 1.5x; here 2.25x).
// Sixteen waves = sixteen frequencies, three 32 x 32 accumulators (kw) each that live for the whole chunk: no fold, no
// publish -- one barrier per block.  MFMA roles: rows = input channel k, columns = output channel n, reduction = voxels (two
// per instruction, one per lane half); both operands are formed from four ds_read_b32 each (2 planes x 2 rows; conflict-free:
// the 32 lanes of a half read the 32 channels of one voxel line) and the column V(v + 2) of a step is V(v) of the next.
// LDS: x ring of 6 row slots x 4 planes (rows h0 .. h0+3 in use, h0+4 / h0+5 landing during the walk) + two buffers of the
// block's four gy rows = 134 KB, filled by glds16; partial tiles [chunk][48][k][n] -> launch_wgrad_reduce -> finish kernel.
constexpr int kWdSlots = 6, kWdPlaneF = kWdSlots * kTapRowF, kWdRingF = 4 * kWdPlaneF;
constexpr int kWdGRowF = 32 * 32, kWdGBufF = 4 * kWdGRowF;
constexpr size_t kWdLdsBytes = (size_t)(kWdRingF + 2 * kWdGBufF) * sizeof(float);
static_assert(kWdLdsBytes <= 160 * 1024, "wgrad_tapdh_kernel: rings must fit the CU's LDS");

struct WgradDhGeom {
  int B, D, H, W, Cq, Cp;         // Cq = channels of x (rows of a tile), Cp = channels of gy (columns)
  int nseg, NG, gpc;              // 32-voxel segments per row, 2 x 2 blocks in total, blocks per chunk
};

__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4)))
wgrad_tapdh_kernel(const float* __restrict__ X, const float* __restrict__ GY, float* __restrict__ ws, WgradDhGeom g) {
  extern __shared__ __align__(16) float tl[];
  float* xr = tl;                            // [4 planes][6 slots][34 voxels][32 channels], linear
  float* gr = tl + kWdRingF;                 // [2 buffers][2 planes][2 rows][32 voxels][32 channels]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lk = lane >> 5;
  const int fd = wave >> 2, fh = wave & 3;

  unsigned chunk_id;
  {
    const unsigned n = gridDim.x, L = blockIdx.x;
    const unsigned xcd = L & 7, q = n >> 3, r = n & 7;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    chunk_id = base + (L >> 3);
  }
  const int seg = chunk_id % g.nseg, range = chunk_id / g.nseg;
  const int H2 = g.H >> 1, D2 = g.D >> 1;
  const int g_begin = range * g.gpc, g_end = min(g.NG, g_begin + g.gpc);
  const int w0 = seg * kTapWseg;

  // staging lists.  x: one call brings TWO padded rows of the four planes = 2 x 4 x 5 entries of 64 sixteen-byte items;
  // gy: the block's 2 planes x 2 rows = 16 entries (one per wave)
  constexpr int nxc = (kTapCols * 8 + 63) / 64;
  int xoff[3], xmeta[3];
  const int rowq = g.W * g.Cq, planeq = g.H * rowq;
#pragma unroll
  for (int n = 0; n < 3; ++n) {
    const int q = wave + n * 16;
    int off = -2, meta = -1;
    if (q < 2 * 4 * nxc) {
      const int rr = q / (4 * nxc), pl = (q % (4 * nxc)) / nxc, ch = q % nxc;
      const int j = ch * 64 + lane;
      if (j < kTapCols * 8) {
        const int u = j >> 3, c = (j & 7) << 2, wsrc = w0 + u - 1;
        off = (wsrc >= 0 && wsrc < g.W && c < g.Cq) ? pl * planeq + rr * rowq + wsrc * g.Cq + c : -1;
      }
      meta = rr | (pl << 2) | ((pl * kWdPlaneF + ch * 256) << 8);
    }
    xoff[n] = off;
    xmeta[n] = __builtin_amdgcn_readfirstlane(meta);
  }
  int goff, gmeta;
  {
    const int rowp = g.W * g.Cp, planep = g.H * rowp;
    const int pl = wave >> 3, r = (wave >> 2) & 1, ch = wave & 3;
    const int j = ch * 64 + lane, v = j >> 3, c = (j & 7) << 2, wv = w0 + v;
    goff = (wv < g.W && c < g.Cp) ? pl * planep + r * rowp + wv * g.Cp + c : -1;
    gmeta = (pl * 2 + r) * kWdGRowF + ch * 256;
  }
  // padded rows hp0, hp0 + 1 (= input rows hp0 - 1, hp0) of the planes 2 d2 - 1 .. 2 d2 + 2 -> slots hp % 6
  auto stage_x = [&](int b, int d2, int hp0) {
    const float* base = X + ((long)(b * g.D + 2 * d2 - 1) * g.H + (hp0 - 1)) * (long)rowq;
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      const int meta = xmeta[n];
      if (meta < 0) break;
      const int rr = meta & 1, pl = (meta >> 2) & 3, dp = 2 * d2 - 1 + pl, h = hp0 - 1 + rr;
      const bool rowok = h >= 0 && h < g.H && dp >= 0 && dp < g.D;
      float* dst = xr + (meta >> 8) + ((hp0 + rr) % kWdSlots) * kTapRowF;
      const int off = xoff[n];
      const float* src = (rowok && off >= 0) ? base + off : kWgZeros;
      if (off != -2) glds16(src, dst);
    }
  };
  auto stage_g = [&](int b, int d2, int h2, int buf) {
    const float* base = GY + ((long)(b * g.D + 2 * d2) * g.H + 2 * h2) * (long)(g.W * g.Cp);
    glds16(goff >= 0 ? base + goff : kWgZeros, gr + buf * kWdGBufF + gmeta);
  };

  // B^T of F(2,3) on x (as conv_tapdh_kernel) and A of F(2,3) on gy: Z_f = z0[f] g0 + z1[f] g1
  const int ra = fh == 0 ? 0 : (fh == 2 ? 2 : 1), rb = fh == 3 ? 3 : (fh == 2 ? 1 : 2);
  const float sh = fh == 1 ? 1.0f : -1.0f;
  const int pa = fd == 0 ? 0 : (fd == 2 ? 2 : 1), pb = fd == 3 ? 3 : (fd == 2 ? 1 : 2);
  const float sd = fd == 1 ? 1.0f : -1.0f;
  const float zh0 = fh == 3 ? 0.0f : 1.0f, zh1 = fh == 0 ? 0.0f : (fh == 1 ? 1.0f : -1.0f);
  const float zd0 = fd == 3 ? 0.0f : 1.0f, zd1 = fd == 0 ? 0.0f : (fd == 1 ? 1.0f : -1.0f);
  const float c00 = zd0 * zh0, c01 = zd0 * zh1, c10 = zd1 * zh0, c11 = zd1 * zh1;

  f32x16 acc[3];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;

  bool fresh = true;
  int cur = 0;
  int h2 = g_begin % H2, d2, b;
  {
    const int bd = g_begin / H2;
    b = bd / D2; d2 = bd % D2;
  }
  for (int G = g_begin; G < g_end; ++G) {
    const int h0 = 2 * h2;
    if (fresh) {                              // first block of the chunk or of a plane pair: the ring is restaged
      if (G != g_begin) __syncthreads();      // ... once every wave has left the previous pair's last walk
      stage_x(b, d2, h0); stage_x(b, d2, h0 + 2);
      if (G == g_begin) stage_g(b, d2, h2, cur);
    }
    wait_vm0();                               // this block's rows (issued during the previous block, or just above)
    __syncthreads();                          // ... of every wave; and every wave has left the previous walk
    const bool has_next = G + 1 < g_end, same_plane = has_next && h2 + 1 < H2;
    auto stage_next = [&]() {
      if (same_plane) stage_x(b, d2, h0 + 4);   // the next block's two new rows -> the slots of h0 - 2, h0 - 1 (free since the barrier)
      if (has_next) {
        int nh2 = h2 + 1, nd2 = d2, nb = b;
        if (nh2 == H2) { nh2 = 0; if (++nd2 == D2) { nd2 = 0; ++nb; } }
        stage_g(nb, nd2, nh2, cur ^ 1);
      }
    };

    const float* xA = xr + pa * kWdPlaneF + ((h0 + ra) % kWdSlots) * kTapRowF + lk * 32 + li;
    const float* xB = xr + pa * kWdPlaneF + ((h0 + rb) % kWdSlots) * kTapRowF + lk * 32 + li;
    const float* xC = xr + pb * kWdPlaneF + ((h0 + ra) % kWdSlots) * kTapRowF + lk * 32 + li;
    const float* xD = xr + pb * kWdPlaneF + ((h0 + rb) % kWdSlots) * kTapRowF + lk * 32 + li;
    const float* gA = gr + cur * kWdGBufF + lk * 32 + li;
    auto vcol = [&](int u) {                  // V at voxel column u + lk of the staged row (column 0 = voxel w0 - 1)
      const float a = fmaf(sh, xB[u * 32], xA[u * 32]);
      const float c = fmaf(sh, xD[u * 32], xC[u * 32]);
      return fmaf(sd, c, a);
    };
    auto zcol = [&](int v) {
      float z = c00 * gA[v * 32];
      z = fmaf(c01, gA[kWdGRowF + v * 32], z);
      z = fmaf(c10, gA[2 * kWdGRowF + v * 32], z);
      return fmaf(c11, gA[3 * kWdGRowF + v * 32], z);
    };
    float vc = vcol(0);
#pragma unroll
    for (int s = 0; s < 16; ++s) {            // voxels 2 s + lk of the segment
      // requests inside the walk and issue priority falling along it, as in conv_tapdh_kernel (round 6: 406 -> 385 us with the
      // finish kernels, same bits)
      if (s == 5) stage_next();
      if (s == 0) __builtin_amdgcn_s_setprio(2);
      if (s == 6) __builtin_amdgcn_s_setprio(1);
      if (s == 11) __builtin_amdgcn_s_setprio(0);
      const float v1 = vcol(2 * s + 1), v2 = vcol(2 * s + 2), z = zcol(2 * s);
      acc[0] = mfma32(vc, z, acc[0]);
      acc[1] = mfma32(v1, z, acc[1]);
      acc[2] = mfma32(v2, z, acc[2]);
      vc = v2;
    }
    __builtin_amdgcn_s_setprio(3);
    cur ^= 1;
    fresh = !same_plane;
    if (++h2 == H2) {
      h2 = 0;
      if (++d2 == D2) { d2 = 0; ++b; }
    }
  }
  // partial tiles of this chunk: ws[chunk][T = wave * 3 + kw][k][n]; accumulator row = (r & 3) + 8 (r >> 2) + 4 lk, column li
  float* wo = ws + ((size_t)chunk_id * 48 + wave * 3) * g.Cq * g.Cp;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = (r & 3) + 8 * (r >> 2) + 4 * lk;
      if (k < g.Cq && li < g.Cp) wo[((size_t)c * g.Cq + k) * g.Cp + li] = acc[c][r];
    }
}

// gw[p][q][kd][kh][kw] = sum over (fd, fh) of G[fd][kd] G[fh][kh] gU[p][q][(fd * 4 + fh) * 3 + kw]   (G^T gU G, G of F(2,3))
__global__ void __launch_bounds__(256)
wgrad_dh_finish_kernel(const float* __restrict__ gU, float* __restrict__ gw, int npq) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= npq * 3) return;
  const int kw = i % 3, pq = i / 3;
  float t[4][3];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    float u[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) u[c] = gU[(size_t)pq * 48 + (a * 4 + c) * 3 + kw];
    t[a][0] = u[0] + 0.5f * (u[1] + u[2]);
    t[a][1] = 0.5f * (u[1] - u[2]);
    t[a][2] = 0.5f * (u[1] + u[2]) + u[3];
  }
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    gw[(size_t)pq * 27 + 0 * 9 + kh * 3 + kw] = t[0][kh] + 0.5f * (t[1][kh] + t[2][kh]);
    gw[(size_t)pq * 27 + 1 * 9 + kh * 3 + kw] = 0.5f * (t[1][kh] - t[2][kh]);
    gw[(size_t)pq * 27 + 2 * 9 + kh * 3 + kw] = 0.5f * (t[1][kh] + t[2][kh]) + t[3][kh];
  }
}

// ------------------------------------------------------------------------------------------------
// conv_pw32_kernel (round 4): 1x1x1 stride-1 convolution (forward and data gradient) with <= 32 channels on both sides -- the
// redirect convolutions of the hourglasses on the 189 MB cost-volume tensors (VT:83-96).  2 kFLOP per 256 bytes moved: a pure
// HBM stream, which conv_gather_kernel runs at 3.2 TB/s because every lane fetches 16-byte pieces of 128-byte voxel lines
// (eight L1 accesses per line).  Here a wave owns a run of 32-voxel tiles; a tile travels global -> LDS as four whole
// kilobytes (glds16, two tiles in flight per wave, no workgroup barrier: the LDS tiles are private to the wave), is read back
// with the conflict-free swizzled ds_read_b128 of the tap kernels, multiplied by the 32 x 32 weights held in 16 VGPRs and
// leaves as four 16-byte stores per lane.
constexpr int kPwPackedElems = 16 * 64;

// w_packed[r * 64 + lane] = Weff[n = lane & 31][k = 8 (r >> 2) + 4 (lane >> 5) + (r & 3)];  mode 0: Weff[n][k] = w[n][k]
// (torch [Cout][Cin]), mode 1 (data gradient): Weff[n][k] = w[k][n]
__global__ void __launch_bounds__(256)
pack_pw32_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int mode) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= kPwPackedElems) return;
  const int lane = i & 63, r = i >> 6;
  const int n = lane & 31, k = 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
  const int K = mode == 0 ? Cin : Cout, N = mode == 0 ? Cout : Cin;
  float v = 0.0f;
  if (n < N && k < K) v = mode == 0 ? w[(size_t)n * Cin + k] : w[(size_t)k * Cin + n];
  wp[i] = v;
}

struct PwGeom {
  long M;                         // voxels
  int K, N;                       // input / output channels of this pass (multiples of 4, <= 32)
  int relu, has_bias, accumulate;
  long tpw;                       // tiles per wave
};

__global__ void __launch_bounds__(256)
conv_pw32_kernel(const float* __restrict__ X, const float* __restrict__ wp, const float* __restrict__ bias,
                 float* __restrict__ Y, PwGeom g) {
  __shared__ __align__(16) float tl[4 * 2 * 1024];          // [4 waves][2 tiles][32 voxels][32 channels], 16-byte swizzled
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lk = lane >> 5;
  float* mine = tl + wave * 2048;
  float wr[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) wr[r] = wp[r * 64 + lane];
  const int cq = lane & 7, nq = 4 * cq;                   // store phase: lane = (voxel lane >> 3 of an octet, channel quad cq)
  float bv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bv[i] = (g.has_bias && nq + i < g.N) ? bias[nq + i] : 0.0f;
  const long ntiles = (g.M + 31) >> 5;
  const long t0 = ((long)blockIdx.x * 4 + wave) * g.tpw, t1 = min(ntiles, t0 + g.tpw);
  // staging: instruction e copies the voxels 8 e .. 8 e + 7 of the tile; lane = (voxel u, physical quad p), source quad p ^ key(u)
  int soff[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int j = e * 64 + lane, u = j >> 3, c = ((j & 7) ^ ((u >> 1) & 7)) << 2;
    soff[e] = c < g.K ? u * g.K + c : -1;
  }
  auto stage = [&](long t, int buf) {
    const float* base = X + t * 32 * (long)g.K;
    const long left = g.M - t * 32;                        // voxels of this tile that exist
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int u = (e * 64 + lane) >> 3;
      const float* src = (soff[e] >= 0 && u < left) ? base + soff[e] : kWgZeros;
      glds16(src, mine + buf * 1024 + e * 256);
    }
  };
  wait_vm0();                                               // weights and bias are in
  if (t0 < t1) stage(t0, 0);
  for (long t = t0; t < t1; ++t) {
    const int buf = (int)(t - t0) & 1;
    if (t + 1 < t1) {
      stage(t + 1, buf ^ 1);
      __builtin_amdgcn_s_waitcnt(0x0F74);                   // vmcnt(4): everything but the four copies just issued has landed
    } else {
      wait_vm0();
    }
    asm volatile("" ::: "memory");
    const float* tp = mine + buf * 1024 + li * 32;
    const int sw = (li >> 1) & 7;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 xv = *reinterpret_cast<const float4*>(tp + (((2 * q + lk) ^ sw) << 2));
      acc = mfma32(wr[4 * q + 0], xv.x, acc);
      acc = mfma32(wr[4 * q + 1], xv.y, acc);
      acc = mfma32(wr[4 * q + 2], xv.z, acc);
      acc = mfma32(wr[4 * q + 3], xv.w, acc);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the tile has been read: its buffer is restaged next iteration
    // round 6: the 32 x 32 output tile goes back through the wave's (consumed) input buffer and leaves as four stores of 1 KiB
    // of whole voxels each (the direct form stored sixty-four 32-byte pieces per instruction); bias, ReLU and the old values
    // of an accumulating launch are applied on the way out, in the order of the direct form
    float* ob = mine + buf * 1024;
    {
      const int sw = (li >> 1) & 7;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq)
        *reinterpret_cast<float4*>(ob + li * 32 + (((2 * rq + lk) ^ sw) << 2)) =
            make_float4(acc[4 * rq], acc[4 * rq + 1], acc[4 * rq + 2], acc[4 * rq + 3]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const bool n4 = (g.N & 3) == 0;
    float4 ov[4];
    if (g.accumulate && n4) {                                // every old value is requested before the first store
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const long v = t * 32 + it * 8 + (lane >> 3);
        ov[it] = (v < g.M && nq < g.N) ? *reinterpret_cast<const float4*>(Y + v * g.N + nq) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int u = it * 8 + (lane >> 3);
      const long v = t * 32 + u;
      const float4 a4 = *reinterpret_cast<const float4*>(ob + u * 32 + ((cq ^ ((u >> 1) & 7)) << 2));
      float o[4] = {a4.x + bv[0], a4.y + bv[1], a4.z + bv[2], a4.w + bv[3]};
      if (g.relu) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = fmaxf(o[i], 0.0f);
      }
      float* dst = Y + v * g.N + nq;
      if (n4) {
        if (g.accumulate) { o[0] += ov[it].x; o[1] += ov[it].y; o[2] += ov[it].z; o[3] += ov[it].w; }
        if (v < g.M && nq < g.N) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
      } else if (v < g.M) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (nq + i < g.N) dst[i] = g.accumulate ? dst[i] + o[i] : o[i];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the buffer is read back before the next iteration restages it
  }
}

// ------------------------------------------------------------------------------------------------
// conv_tap2_kernel (round 3): the stride-2 3x3x3 "down" gather with <= 32 input and 33..64 output channels on the LDS-ring /
// register-weights design of conv_tap_kernel -- the first convolution of every hourglass (VT:73-76, 32 -> 64 on the
// 192 x 48 x 160 cost volume) and the data gradient of its last transposed convolution (VT:86-88, 64 -> 32), 8 launches per
// step that conv_gather_kernel runs at 66-69 TF/s (parked on its 27 x 128-byte operand lines: SQ_WAIT_ANY 37 %).
//   * 8 waves = 4 tap groups x 2 output-channel halves; a wave keeps its 7 taps x (32 k x 32 n) weights in 112 VGPRs.
//   * MFMA columns = 2 output rows x 16 output voxels (the 48 x 160 -> 24 x 80 level has 80 = 5 x 16 voxels per row: 32-voxel
//     segments would leave a sixth of the columns empty), rows = output channels, k = input channels.
//   * the ring holds the five source rows 2 h0 - 1 .. 2 h0 + 3 of three source planes (34 voxels each, filled by
//     global_load_lds with the swizzle of conv_tap_kernel); the four NEW rows of the next row pair land while this pair
//     computes (9 slots).  The B operand of a tap is one ds_read_b128 at voxel 2 v + kw of row 2 (h0 + r) + kh.
//   * fold of the four tap groups through LDS per channel half, float4 stores; `accumulate` requests the old values first.
constexpr int kT2Slots = 9, kT2PlaneF = kT2Slots * kTapRowF, kT2RingF = 3 * kT2PlaneF;
constexpr int kT2RedF = 8 * 16 * 64;
constexpr size_t kT2LdsBytes = (size_t)(kT2RingF + kT2RedF) * sizeof(float);
constexpr int kT2PackedElems = 2 * 4 * 7 * 16 * 64;

// w_packed[(((nh * 4 + tg) * 7 + tt) * 16 + q * 4 + c) * 64 + lane] = Weff[n = 32 nh + (lane & 31)][k = 8 q + 4 (lane >> 5) + c][tap = tg + 4 tt]
// with Weff[n][k][tap] = w[(n * K + k) * 27 + tap]: conv forward (torch [Cout][Cin][27]) and transposed-conv data gradient
// (torch [Cin][Cout][27], N = Cin, K = Cout; taps NOT mirrored: gx[i] = sum_k gy[2 i - 1 + k] w[k]) share the formula
__global__ void __launch_bounds__(256)
pack_tap2_kernel(const float* __restrict__ w, float* __restrict__ wp, int N, int K) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= kT2PackedElems) return;
  const int lane = i & 63, r = (i >> 6) & 15, wt = i >> 10;
  const int tt = wt % 7, wv = wt / 7, tg = wv & 3, nh = wv >> 2;
  const int tap = tg + 4 * tt, n = 32 * nh + (lane & 31), k = 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
  wp[i] = (tap < 27 && n < N && k < K) ? w[((size_t)n * K + k) * 27 + tap] : 0.0f;
}

__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
conv_tap2_kernel(const float* __restrict__ X, const float* __restrict__ wp, const float* __restrict__ bias,
                 float* __restrict__ Y, ConvTapGeom g) {      // g.NG / g.gpc count output row PAIRS
  extern __shared__ __align__(16) float tl[];
  float* ring = tl;                          // [3 planes][9 slots][34 voxels][32 channels], 16-byte swizzled
  float* red = tl + kT2RingF;                // [8 waves][16 rows][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lk = lane >> 5;
  const int tg = wave & 3, nh = wave >> 2;
  const int vx = li & 15, rw = li >> 4;      // output voxel within the segment, output row within the pair

  float wr[7][16];
#pragma unroll
  for (int tt = 0; tt < 7; ++tt)
#pragma unroll
    for (int r = 0; r < 16; ++r) wr[tt][r] = wp[((wave * 7 + tt) * 16 + r) * 64 + lane];
  const int ntap = tg < 3 ? 7 : 6;

  unsigned chunk_id;
  {
    const unsigned n = gridDim.x, L = blockIdx.x;
    const unsigned xcd = L & 7, q = n >> 3, r = n & 7;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    chunk_id = base + (L >> 3);
  }
  const int seg = chunk_id % g.nseg, range = chunk_id / g.nseg;
  const int w0 = seg * 16;                   // first output voxel of the segment; its first source voxel is 2 w0 - 1
  const int g_begin = range * g.gpc, g_end = min(g.NG, g_begin + g.gpc);
  const int H2 = (g.H + 1) >> 1;

  constexpr int nxc = (kTapCols * 8 + 63) / 64;                 // 5 staging entries per (plane, row)
  int xoff[2], xmeta[2];
  const int plane_g = g.Hs * g.Ws * g.K;
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int q = wave + n * 8;
    int off = -2, meta = -1;
    if (q < 3 * nxc) {
      const int pl = q / nxc, ch = q % nxc;
      const int j = ch * 64 + lane;
      if (j < kTapCols * 8) {
        // LDS slot j >> 3 holds voxel column u = slot ^ ((slot >> 4) & 1): the walk reads columns of ONE parity (2 v + kw), whose
        // 128-byte lines all start in the same half of the 64 banks -- with the columns 16..31 swapped in pairs the sixteen
        // lanes of a ds_read_b128 group use both halves (r4 SQ counters: 38.6 % of this kernel's LDS cycles were conflicts)
        const int sl = j >> 3, u = sl ^ ((sl >> 4) & 1), c = (((j & 7) ^ ((u >> 1) & 7)) << 2), wsrc = 2 * w0 - 1 + u;
        off = (wsrc >= 0 && wsrc < g.Ws && c < g.K) ? pl * plane_g + wsrc * g.K + c : -1;
      }
      meta = pl | ((pl * kT2PlaneF + ch * 256) << 4);
    }
    xoff[n] = off;
    xmeta[n] = __builtin_amdgcn_readfirstlane(meta);
  }
  // padded source row hp (= source row hp - 1) of source planes 2 d - 1 .. 2 d + 1 -> ring slot hp % 9
  auto stage_row = [&](int b, int d, int hp) {
    const float* base = X + ((long)(b * g.Ds + 2 * d - 1) * g.Hs + (hp - 1)) * (long)(g.Ws * g.K);
    const int h = hp - 1;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int meta = xmeta[n];
      if (meta < 0) break;
      const int pl = meta & 3, dp = 2 * d - 1 + pl;
      const bool rowok = h >= 0 && h < g.Hs && dp >= 0 && dp < g.Ds;
      float* dst = ring + (meta >> 4) + (hp % kT2Slots) * kTapRowF;
      const int off = xoff[n];
      const float* src = (rowok && off >= 0) ? base + off : kWgZeros;
      if (off != -2) glds16(src, dst);
    }
  };

  int tp_plane[7], tp_kh[7], tp_kw[7];
#pragma unroll
  for (int tt = 0; tt < 7; ++tt) {
    const int t = min(tg + 4 * tt, 26);
    tp_plane[tt] = (t / 9) * kT2PlaneF; tp_kh[tt] = (t / 3) % 3; tp_kw[tt] = t % 3;
  }
  const int nb = 32 * nh + 8 * tg + 4 * lk;  // first of the 4 output channels this lane stores
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (g.has_bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = nb + i < g.N ? bias[nb + i] : 0.0f;
  }

  wait_vm0();                                 // weights and bias are in: nothing of the prologue is pending inside the loop
  bool fresh = true;
  int h2 = g_begin % H2, d, b;
  {
    const int bd = g_begin / H2;
    b = bd / g.D; d = bd % g.D;
  }
  for (int G = g_begin; G < g_end; ++G) {
    const int h0 = 2 * h2;
    if (fresh) {
#pragma unroll
      for (int r = 0; r < 5; ++r) stage_row(b, d, 2 * h0 + r);
      wait_vm0();
      __syncthreads();
    }
    const bool same_plane = G + 1 < g_end && h2 + 1 < H2;
    const int wv = w0 + vx, hrow = h0 + rw;
    const bool ok = wv < g.W && hrow < g.H && nb < g.N;
    float* dst = Y + (((long)(b * g.D + d) * g.H + hrow) * g.W + wv) * g.N + nb;
    float4 told = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.accumulate && ok && (g.N & 3) == 0) told = *reinterpret_cast<const float4*>(dst);

    // slot offsets of the padded rows 2 (h0 + rw) + kh, kh = 0..2 (per lane: rw differs between the half rows of a wave)
    int srow[3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) srow[kh] = ((2 * h0 + 2 * rw + kh) % kT2Slots) * kTapRowF;

    f32x16 acc2[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[q][r] = 0.0f;
    float4 xa[4], xb4[4];
    auto fetch = [&](int tt, float4 (&xv)[4]) {
      const int u = 2 * vx + tp_kw[tt];
      const int sr = tp_kh[tt] == 0 ? srow[0] : (tp_kh[tt] == 1 ? srow[1] : srow[2]);   // (a select: srow[tp_kh[tt]] is a scratch array)
      const float* rowp = ring + tp_plane[tt] + sr + (u ^ ((u >> 4) & 1)) * 32;
      const int sw = (u >> 1) & 7;
#pragma unroll
      for (int q = 0; q < 4; ++q) xv[q] = *reinterpret_cast<const float4*>(rowp + (((2 * q + lk) ^ sw) << 2));
    };
    auto mm_head = [&](int tt, const float4 (&xv)[4]) { acc2[0] = mfma32(wr[tt][0], xv[0].x, acc2[0]); };
    auto mm_tail = [&](int tt, const float4 (&xv)[4]) {
#pragma unroll
      for (int q = 1; q < 4; ++q) acc2[q & 1] = mfma32(wr[tt][4 * q + 0], xv[q].x, acc2[q & 1]);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc2[q & 1] = mfma32(wr[tt][4 * q + 1], xv[q].y, acc2[q & 1]);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc2[q & 1] = mfma32(wr[tt][4 * q + 2], xv[q].z, acc2[q & 1]);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc2[q & 1] = mfma32(wr[tt][4 * q + 3], xv[q].w, acc2[q & 1]);
    };
    fetch(0, xa);
#pragma unroll
    for (int tt = 0; tt < 7; tt += 2) {
      // row requests inside the walk and issue priority falling along it: see conv_tapdh_kernel (round 6: -2 %, same bits)
      if (tt == 2 && same_plane) {
#pragma unroll
        for (int r = 5; r < 9; ++r) stage_row(b, d, 2 * h0 + r);
      }
      if (tt == 0) __builtin_amdgcn_s_setprio(2);
      if (tt == 2) __builtin_amdgcn_s_setprio(1);
      if (tt == 4) __builtin_amdgcn_s_setprio(0);
      if (tt < ntap) {
        mm_head(tt, xa);
        __builtin_amdgcn_sched_barrier(0);
        if (tt + 1 < 7 && tt + 1 < ntap) fetch(tt + 1, xb4);
        __builtin_amdgcn_sched_barrier(0);
        mm_tail(tt, xa);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (tt + 1 < 7 && tt + 1 < ntap) {
        mm_head(tt + 1, xb4);
        __builtin_amdgcn_sched_barrier(0);
        if (tt + 2 < 7 && tt + 2 < ntap) fetch(tt + 2, xa);
        __builtin_amdgcn_sched_barrier(0);
        mm_tail(tt + 1, xb4);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_setprio(3);
    // fold the four tap groups of each channel half: every wave publishes its partial tile, then sums rows 4 tg .. 4 tg + 3
#pragma unroll
    for (int r = 0; r < 16; r += 4)           // tile = [register quad][lane][4]: 16-byte LDS accesses (round 6)
      *reinterpret_cast<float4*>(red + ((wave * 4 + (r >> 2)) * 64 + lane) * 4) =
          make_float4(acc2[0][r] + acc2[1][r], acc2[0][r + 1] + acc2[1][r + 1], acc2[0][r + 2] + acc2[1][r + 2], acc2[0][r + 3] + acc2[1][r + 3]);
    wait_vm0();
    __syncthreads();
    {
      float o[4];
      {
        const float* rp = red + (((nh * 4) * 4 + tg) * 64 + lane) * 4;
        const float4 p0 = *reinterpret_cast<const float4*>(rp), p1 = *reinterpret_cast<const float4*>(rp + 1024);
        const float4 p2 = *reinterpret_cast<const float4*>(rp + 2048), p3 = *reinterpret_cast<const float4*>(rp + 3072);
        o[0] = ((p0.x + p1.x) + p2.x) + p3.x + bv[0];
        o[1] = ((p0.y + p1.y) + p2.y) + p3.y + bv[1];
        o[2] = ((p0.z + p1.z) + p2.z) + p3.z + bv[2];
        o[3] = ((p0.w + p1.w) + p2.w) + p3.w + bv[3];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (g.relu) o[i] = fmaxf(o[i], 0.0f);
      if (ok) {
        if ((g.N & 3) == 0) {
          *reinterpret_cast<float4*>(dst) = make_float4(o[0] + told.x, o[1] + told.y, o[2] + told.z, o[3] + told.w);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (nb + i < g.N) dst[i] = g.accumulate ? dst[i] + o[i] : o[i];
        }
      }
    }
    barrier_lds();                         // the partial buffer is reused by the next row pair; the stores stay in flight
    fresh = !same_plane;
    if (++h2 == H2) {
      h2 = 0;
      if (++d == g.D) { d = 0; ++b; }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// conv_tap2up_kernel (round 3): the stride-2 3x3x3 "up" gather -- transposed-conv forward 64 -> 32 (VT:86-88, the last layer of
// every hourglass) and the data gradient of the stride-2 conv 32 -> 64 (VT:73-76) -- with 33..64 source channels on the coarse
// grid and <= 32 destination channels on the fine grid (fine = 2 x coarse).  Along one axis a fine index 2 c + par receives
//     par = 0:  w[k = 1] * x[c]                    par = 1:  w[k = 2] * x[c] + w[k = 0] * x[c + 1]
// so the 8 fine voxels of a coarse cell are 8 parity CLASSES with 1, 2, 2, 2, 4, 4, 4, 8 taps (27 in all), each a GEMM over
// the 64 source channels of the cell or of its +1 neighbours.  MFMA columns = 2 coarse rows x 16 coarse cells, rows = the 32
// destination channels, k = source channels; the taps are dealt in pairs to fourteen waves by class (four waves share the 8-tap
// class), every wave keeps its 2 taps x (64 k x 32 n) weights in 64 VGPRs; the ring
// holds three coarse rows of two coarse planes (17 voxels x 64 channels, filled by global_load_lds, 16-byte quads XOR-swizzled
// by the voxel index: the 16 lanes of a ds_read_b128 group read 16 different voxels); the class tiles meet in LDS and leave
// as 16-byte stores (four per lane and class: 32 contiguous bytes per voxel and instruction).
constexpr int kUpCols = 17, kUpRowF = kUpCols * 64, kUpSlots = 5, kUpPlaneF = kUpSlots * kUpRowF, kUpRingF = 2 * kUpPlaneF;
constexpr int kUpTiles = 14, kUpRedF = kUpTiles * 16 * 64;
constexpr size_t kUpLdsBytes = (size_t)(kUpRingF + kUpRedF) * sizeof(float);
constexpr int kUpPackedElems = 16 * 2 * 32 * 64;
static_assert(kUpRowF == kTapRowF, "17 voxels x 64 channels = 34 voxels x 32 channels");

// SIXTEEN waves (1024 threads, four per SIMD at <= 128 VGPRs: a first version with eight waves x four taps needed 256 VGPRs
// and spilled): wave w < 14 owns two taps of ONE parity class (kd * 9 + kh * 3 + kw, -1 = none) and one partial tile.
// Per axis k = 1: parity 0; k = 2: parity 1, same cell; k = 0: parity 1, cell + 1.
//   waves 0-3: class (1,1,1)   4, 5: (0,1,1)   6, 7: (1,0,1)   8, 9: (1,1,0)   10: (0,0,1)   11: (0,1,0)   12: (1,0,0)   13: (0,0,0)
__device__ __constant__ int kUpTaps[16][2] = {{0, 2}, {6, 8}, {18, 20}, {24, 26}, {9, 11}, {15, 17}, {3, 5}, {21, 23},
                                              {1, 7}, {19, 25}, {12, 14}, {10, 16}, {4, 22}, {13, -1}, {-1, -1}, {-1, -1}};
// store phase: waves 2 c, 2 c + 1 write class kUpStoreClass[c] (pd * 4 + ph * 2 + pw) = the sum of its partial tiles
__device__ __constant__ int kUpStoreClass[8] = {7, 3, 5, 6, 1, 2, 4, 0};
__device__ __constant__ int kUpStoreTile[8] = {0, 4, 6, 8, 10, 11, 12, 13};
__device__ __constant__ int kUpStoreNTiles[8] = {4, 2, 2, 2, 1, 1, 1, 1};

// w_packed[((wave * 2 + tt) * 32 + q * 4 + c) * 64 + lane] = Weff[n = lane & 31][k = 8 q + 4 (lane >> 5) + c][tap kUpTaps[wave][tt]],
// Weff[n][k][tap] = w[(k * N + n) * 27 + tap]: transposed-conv forward (torch [Cin = K][Cout = N][27]) and conv data gradient
// (torch [Cout = K][Cin = N][27]) share the formula, taps NOT mirrored
__global__ void __launch_bounds__(256)
pack_tap2up_kernel(const float* __restrict__ w, float* __restrict__ wp, int N, int K) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= kUpPackedElems) return;
  const int lane = i & 63, r = (i >> 6) & 31, tt = (i >> 11) & 1, wave = i >> 12;
  const int tap = kUpTaps[wave][tt], n = lane & 31, k = 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
  wp[i] = (tap >= 0 && n < N && k < K) ? w[((size_t)k * N + n) * 27 + tap] : 0.0f;
}

__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4)))
conv_tap2up_kernel(const float* __restrict__ X, const float* __restrict__ wp, const float* __restrict__ bias,
                   float* __restrict__ Y, ConvTapGeom g) {    // g.Ds/Hs/Ws: coarse source grid; g.D/H/W: fine grid = 2 x coarse
  extern __shared__ __align__(16) float tl[];
  float* ring = tl;                          // [2 planes][5 slots][17 voxels][64 channels], 16-byte swizzled
  float* red = tl + kUpRingF;                // [14 tiles][16 rows][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lk = lane >> 5;
  const int vx = li & 15, rw = li >> 4;      // coarse cell within the segment, coarse row within the pair

  // this wave's two taps: weights (A operands) in registers, per-tap cell shifts as scalars
  float wr[2][32];
  int t_pl[2], t_dh[2], t_dw[2];
  int ntap = 0;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int tap = kUpTaps[wave][tt];
#pragma unroll
    for (int r = 0; r < 32; ++r) wr[tt][r] = wp[((wave * 2 + tt) * 32 + r) * 64 + lane];
    const int t = max(tap, 0), kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
    t_pl[tt] = (kd == 0 ? 1 : 0) * kUpPlaneF; t_dh[tt] = kh == 0 ? 1 : 0; t_dw[tt] = kw == 0 ? 1 : 0;
    ntap += tap >= 0 ? 1 : 0;
  }
  ntap = __builtin_amdgcn_readfirstlane(ntap);

  unsigned chunk_id;
  {
    const unsigned n = gridDim.x, L = blockIdx.x;
    const unsigned xcd = L & 7, q = n >> 3, r = n & 7;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    chunk_id = base + (L >> 3);
  }
  const int seg = chunk_id % g.nseg, range = chunk_id / g.nseg;
  const int c0 = seg * 16;                   // first coarse cell of the segment
  const int g_begin = range * g.gpc, g_end = min(g.NG, g_begin + g.gpc);
  const int H2 = (g.Hs + 1) >> 1;

  constexpr int nxc = (kUpCols * 16 + 63) / 64;                 // 5 staging entries per (plane, row): waves 0..9 take one each
  int xoff = -2, xmeta = -1;
  const int plane_g = g.Hs * g.Ws * g.K;
  if (wave < 2 * nxc) {
    const int pl = wave / nxc, ch = wave % nxc;
    const int j = ch * 64 + lane;
    if (j < kUpCols * 16) {
      const int u = j >> 4, c = (((j & 15) ^ (u & 15)) << 2), wsrc = c0 + u;
      xoff = (wsrc < g.Ws && c < g.K) ? pl * plane_g + wsrc * g.K + c : -1;
    }
    xmeta = pl | ((pl * kUpPlaneF + ch * 256) << 4);
  }
  auto stage_row = [&](int b, int d, int hr) {      // coarse row hr of coarse planes d, d + 1 -> ring slot hr % 5
    if (xmeta < 0) return;
    const float* base = X + ((long)(b * g.Ds + d) * g.Hs + hr) * (long)(g.Ws * g.K);
    const int pl = xmeta & 3;
    const bool rowok = hr < g.Hs && d + pl < g.Ds;
    float* dst = ring + (xmeta >> 4) + (hr % kUpSlots) * kUpRowF;
    const float* src = (rowok && xoff >= 0) ? base + xoff : kWgZeros;
    if (xoff != -2) glds16(src, dst);
  };

  const int sc = wave >> 1, jp = wave & 1;   // store phase: class slot and channel-group pair of this wave
  const int scls = kUpStoreClass[sc], stile = kUpStoreTile[sc], sntile = kUpStoreNTiles[sc];
  const int spd = scls >> 2, sph = (scls >> 1) & 1, spw = scls & 1;

  wait_vm0();                                 // weights and bias are in: nothing of the prologue is pending inside the loop
  bool fresh = true;
  int h2 = g_begin % H2, d, b;
  {
    const int bd = g_begin / H2;
    b = bd / g.Ds; d = bd % g.Ds;
  }
  for (int G = g_begin; G < g_end; ++G) {
    const int h0 = 2 * h2;
    if (fresh) {
      stage_row(b, d, h0); stage_row(b, d, h0 + 1); stage_row(b, d, h0 + 2);
      wait_vm0();
      __syncthreads();
    }
    const bool same_plane = G + 1 < g_end && h2 + 1 < H2;

    // ring offsets of the coarse rows h0 + rw (+ 1)
    const int srow0 = ((h0 + rw) % kUpSlots) * kUpRowF, srow1 = ((h0 + rw + 1) % kUpSlots) * kUpRowF;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    // group gi = (tap tt = gi >> 1, channel half = gi & 1): four 16-byte quads = 32 source channels, 16 MFMAs; the other
    // three waves of the SIMD cover the LDS latency of a group, so the operand buffer is single
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      const int tt = gi >> 1, hq = gi & 1;
      // row requests inside the walk and issue priority falling along it: see conv_tapdh_kernel (round 6: -2 %, same bits)
      if (gi == 1 && same_plane) { stage_row(b, d, h0 + 3); stage_row(b, d, h0 + 4); }
      if (gi == 0) __builtin_amdgcn_s_setprio(2);
      if (gi == 2) __builtin_amdgcn_s_setprio(1);
      if (gi == 3) __builtin_amdgcn_s_setprio(0);
      if (tt < ntap) {
        const int u = vx + t_dw[tt];
        const float* rowp = ring + t_pl[tt] + (t_dh[tt] ? srow1 : srow0) + u * 64;
        const int sw = u & 15;
        float4 xc = *reinterpret_cast<const float4*>(rowp + (((2 * (4 * hq) + lk) ^ sw) << 2));
#pragma unroll
        for (int q = 0; q < 4; ++q) {          // one quad ahead (four in flight left no room for 13 of the weight registers)
          float4 xn = xc;
          if (q + 1 < 4) xn = *reinterpret_cast<const float4*>(rowp + (((2 * (4 * hq + q + 1) + lk) ^ sw) << 2));
          acc = mfma32(wr[tt][16 * hq + 4 * q + 0], xc.x, acc);
          acc = mfma32(wr[tt][16 * hq + 4 * q + 1], xc.y, acc);
          acc = mfma32(wr[tt][16 * hq + 4 * q + 2], xc.z, acc);
          acc = mfma32(wr[tt][16 * hq + 4 * q + 3], xc.w, acc);
          xc = xn;
        }
      }
    }
    __builtin_amdgcn_s_setprio(3);
    if (wave < kUpTiles) {
#pragma unroll
      for (int r = 0; r < 16; r += 4)         // tile = [register quad][lane][4]: 16-byte LDS accesses (round 6)
        *reinterpret_cast<float4*>(red + ((wave * 4 + (r >> 2)) * 64 + lane) * 4) = make_float4(acc[r], acc[r + 1], acc[r + 2], acc[r + 3]);
    }
    wait_vm0();
    __syncthreads();
    {
      // waves 2 c, 2 c + 1 store class kUpStoreClass[c]: lane (cell li, lk) holds channels 8 j + 4 lk .. + 3; this wave j = 2 jp, 2 jp + 1.
      // Round 4: every global load of the phase (old values, bias) is issued first, then ALL partial values of the lane's
      // eight outputs (one uniform branch per tile, reads batched inside), then the sums in the fixed tile order -- the r3
      // form (a run-time loop per output with a wait per tile, a bias load per element) serialised ~40 LDS / L2 latencies.
      const int cw = c0 + vx, chr = h0 + rw;
      const bool ok = cw < g.Ws && chr < g.Hs;
      float* dst = Y + (((long)(b * g.D + 2 * d + spd) * g.H + 2 * chr + sph) * g.W + 2 * cw + spw) * g.N + 4 * lk;
      const int n0 = 16 * jp + 4 * lk;           // first channel of jj = 0; jj = 1 is 8 further
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {           // (one output quad at a time: both at once spilled 19 registers)
        const bool chok = n0 + 8 * jj < g.N;
        float4 ov = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.accumulate && ok && chok) ov = *reinterpret_cast<const float4*>(dst + 16 * jp + 8 * jj);
        float bb[4] = {0.f, 0.f, 0.f, 0.f};
        if (g.has_bias) {
#pragma unroll
          for (int i = 0; i < 4; ++i) bb[i] = n0 + 8 * jj + i < g.N ? bias[n0 + 8 * jj + i] : 0.0f;
        }
        float pv[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (t < sntile) {
            const float4 v = *reinterpret_cast<const float4*>(red + (((stile + t) * 4 + 2 * jp + jj) * 64 + lane) * 4);
            pv[t][0] = v.x; pv[t][1] = v.y; pv[t][2] = v.z; pv[t][3] = v.w;
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) pv[t][i] = 0.0f;
          }
        }
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = pv[0][i];
          if (sntile > 1) v += pv[1][i];
          if (sntile > 2) { v += pv[2][i]; v += pv[3][i]; }
          v += bb[i];
          o[i] = g.relu ? fmaxf(v, 0.0f) : v;
        }
        if (ok && chok) *reinterpret_cast<float4*>(dst + 16 * jp + 8 * jj) = make_float4(o[0] + ov.x, o[1] + ov.y, o[2] + ov.z, o[3] + ov.w);
      }
    }
    barrier_lds();                         // the fold buffer is reused by the next row pair; the stores stay in flight
    fresh = !same_plane;
    if (++h2 == H2) {
      h2 = 0;
      if (++d == g.Ds) { d = 0; ++b; }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the 3x3x3 stride-1 "heads" with 32 input channels and <= 4 output channels (classif3_2 / redir2:
// 32 -> 1).  On the MFMA kernel such a layer costs as much as a full 32 -> 32 one (31 of 32 tile columns are padding).
// Here it is a VALU reduction over the LDS ring of conv_tap_kernel:
//     gw[tap][ci][n] = sum_v x[v + tap][ci] * gy[v][n]
// thread (tap = tid / 8, channel quad = tid % 8) keeps 4 x NP accumulators; per voxel it reads its x quad from the ring
// (one ds_read_b128) and gy[v][0..NP) as an LDS broadcast.  Partial tiles per workgroup, folded by wgrad_reduce_kernel.
constexpr int kThinNP = 4;

__global__ void __launch_bounds__(256)
wgrad_thin_kernel(const float* __restrict__ X, const float* __restrict__ GY, float* __restrict__ ws, ConvTapGeom g) {
  extern __shared__ __align__(16) float tl[];
  float* ring = tl;                          // [3 planes][4 slots][34 voxels][32 channels], 16-byte swizzled
  float* gyl = tl + kTapRingF;               // [2 buffers][32 voxels][NP]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned chunk_id;
  {
    const unsigned n = gridDim.x, L = blockIdx.x;
    const unsigned xcd = L & 7, q = n >> 3, r = n & 7;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    chunk_id = base + (L >> 3);
  }
  const int seg = chunk_id % g.nseg, range = chunk_id / g.nseg;
  const int w0 = seg * kTapWseg;
  const int g_begin = range * g.gpc, g_end = min(g.NG, g_begin + g.gpc);

  // x staging exactly as in conv_tap_kernel (K = 32 channels of x)
  constexpr int nxc = (kTapCols * 8 + 63) / 64;
  int xoff[4], xmeta[4];
  const int plane_g = g.H * g.W * g.K;
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int q = wave + n * 4;
    int off = -2, meta = -1;
    if (q < 3 * nxc) {
      const int pl = q / nxc, ch = q % nxc;
      const int j = ch * 64 + lane;
      if (j < kTapCols * 8) {
        const int u = j >> 3, c = (((j & 7) ^ (u & 7)) << 2), wsrc = w0 + u - 1;
        off = (wsrc >= 0 && wsrc < g.W && c < g.K) ? pl * plane_g + wsrc * g.K + c : -1;
      }
      meta = pl | ((pl * kTapPlaneF + ch * 256) << 4);
    }
    xoff[n] = off;
    xmeta[n] = __builtin_amdgcn_readfirstlane(meta);
  }
  auto stage_row = [&](int b, int d, int hp) {
    const float* base = X + ((long)(b * g.D + d - 1) * g.H + (hp - 1)) * (long)(g.W * g.K);
    const int h = hp - 1;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int meta = xmeta[n];
      if (meta < 0) break;
      const int pl = meta & 3, dp = d - 1 + pl;
      const bool rowok = h >= 0 && h < g.H && dp >= 0 && dp < g.D;
      float* dst = ring + (meta >> 4) + (hp & 3) * kTapRowF;
      const int off = xoff[n];
      const float* src = (rowok && off >= 0) ? base + off : kWgZeros;
      if (off != -2) __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
    }
  };
  // gy row of the step: 32 voxels x N floats (N <= 4), zero-padded to NP, through registers (tiny)
  auto load_gy = [&](int b, int d, int h, int buf) {
    if (tid < kTapWseg * kThinNP) {
      const int v = tid / kThinNP, n = tid % kThinNP, wv = w0 + v;
      float val = 0.0f;
      if (wv < g.W && n < g.N) val = GY[(((long)(b * g.D + d) * g.H + h) * g.W + wv) * g.N + n];
      gyl[buf * kTapWseg * kThinNP + tid] = val;
    }
  };

  const int tap = tid >> 3, quad = tid & 7;          // 32 tap slots (27 valid) x 8 channel quads
  const int tpl = min(tap, 26);
  const int kd = tpl / 9, kh = (tpl / 3) % 3, kw = tpl % 3;
  float acc[kThinNP][4];
#pragma unroll
  for (int n = 0; n < kThinNP; ++n)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[n][c] = 0.0f;

  bool fresh = true;
  int h = g_begin % g.H, d, b;
  {
    const int bd = g_begin / g.H;
    b = bd / g.D; d = bd % g.D;
  }
  for (int G = g_begin; G < g_end; ++G) {
    const int buf = G & 1;
    if (fresh) {
      stage_row(b, d, h); stage_row(b, d, h + 1); stage_row(b, d, h + 2);
      load_gy(b, d, h, buf);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    const bool has_next = G + 1 < g_end;
    const bool same_plane = has_next && h + 1 < g.H;
    if (same_plane) stage_row(b, d, h + 3);
    if (has_next) {                                   // next row's gy into the other buffer
      int hn = h + 1, dn = d, bn = b;
      if (hn == g.H) { hn = 0; if (++dn == g.D) { dn = 0; ++bn; } }
      load_gy(bn, dn, hn, buf ^ 1);
    }
    if (tap < 27) {
      const float* rowp = ring + kd * kTapPlaneF + ((h + kh) & 3) * kTapRowF;
      const float* gp = gyl + buf * kTapWseg * kThinNP;
#pragma unroll 4
      for (int v = 0; v < kTapWseg; ++v) {
        const int u = v + kw;
        const float4 xq = *reinterpret_cast<const float4*>(rowp + u * 32 + ((quad ^ (u & 7)) << 2));
        const float4 gv = *reinterpret_cast<const float4*>(gp + v * kThinNP);
        const float gn[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int n = 0; n < kThinNP; ++n) {
          acc[n][0] += gn[n] * xq.x; acc[n][1] += gn[n] * xq.y; acc[n][2] += gn[n] * xq.z; acc[n][3] += gn[n] * xq.w;
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    fresh = !same_plane;
    if (++h == g.H) {
      h = 0;
      if (++d == g.D) { d = 0; ++b; }
    }
  }
  // partial tile of this workgroup: ws[chunk][tap][ci][n]  (Cq = K, Cp = N)
  if (tap < 27) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int ci = quad * 4 + c;
      if (ci >= g.K) continue;
#pragma unroll
      for (int n = 0; n < kThinNP; ++n)
        if (n < g.N) ws[(((size_t)chunk_id * 27 + tap) * g.K + ci) * g.N + n] = acc[n][c];
    }
  }
}

struct WgradThinPlan { bool ok; ConvTapGeom g; int nchunks; };

WgradThinPlan plan_wgrad_thin(const ssbev_conv_dims* d) {
  WgradThinPlan p;
  p.ok = false;
  if (d->transposed || d->kd != 3 || d->kh != 3 || d->kw != 3 || d->sd != 1 || d->sh != 1 || d->sw != 1) return p;
  if (d->pd != 1 || d->ph != 1 || d->pw != 1 || d->dd != 1 || d->dh != 1 || d->dw != 1) return p;
  if (d->Di != d->Do || d->Hi != d->Ho || d->Wi != d->Wo || d->tile_hint == 7) return p;
  if (d->Cin > 32 || d->Cin < 16 || d->Cin % 4 != 0 || d->Cout > kThinNP) return p;
  ConvTapGeom& g = p.g;
  g.B = d->B; g.D = d->Do; g.H = d->Ho; g.W = d->Wo; g.K = d->Cin; g.N = d->Cout;
  g.nseg = (g.W + kTapWseg - 1) / kTapWseg;
  g.NG = g.B * g.D * g.H;
  g.relu = 0; g.has_bias = 0;
  if ((long)g.NG * g.nseg < 1024L * 16 && d->tile_hint != 9) return p;
  long nranges = std::max(1L, 1024L / g.nseg);              // ~4 workgroups per CU (LDS 53 KB, light registers)
  if (nranges > g.NG) nranges = g.NG;
  g.gpc = (int)((g.NG + nranges - 1) / nranges);
  nranges = (g.NG + g.gpc - 1) / g.gpc;
  p.nchunks = (int)(nranges * g.nseg);
  p.ok = true;
  return p;
}

// ------------------------------------------------------------------------------------------------
// Forward / data gradient of the same heads (32 input channels, <= 4 output channels; also the data gradient of the
// 2 -> 32 layer, whose output side is the thin one): out[v][n] = sum_tap sum_k x[v + tap][k] * Weff[n][k][tap] as a VALU
// dot product over the LDS ring.  Thread (voxel = tid / 8, channel quad = tid % 8): per tap one ds_read_b128 of x and one
// of the weights per output channel (LDS-resident, 13.8 KB, broadcast across the voxels), then an xor-shuffle fold over the
// 8 quads.  wt[(tap * NP + n) * 32 + k] = Weff[n][k][tap] (pack_thin_kernel; mode 0 forward, mode 1 data gradient).
__global__ void __launch_bounds__(256)
pack_thin_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int Cin, int mode) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 27 * kThinNP * 32) return;
  const int k = i & 31, n = (i >> 5) % kThinNP, tap = i / (32 * kThinNP);
  const int K = mode == 0 ? Cin : Cout, N = mode == 0 ? Cout : Cin;
  float v = 0.0f;
  if (n < N && k < K) v = mode == 0 ? w[((size_t)n * Cin + k) * 27 + tap] : w[((size_t)k * Cin + n) * 27 + (26 - tap)];
  wt[i] = v;
}

__global__ void __launch_bounds__(256)
conv_thin_kernel(const float* __restrict__ X, const float* __restrict__ wt, const float* __restrict__ bias,
                 float* __restrict__ Y, ConvTapGeom g) {
  extern __shared__ __align__(16) float tl[];
  float* ring = tl;                          // [3 planes][4 slots][34 voxels][32 channels], 16-byte swizzled
  float* wl = tl + kTapRingF;                // [27 taps][NP][32 k]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 27 * kThinNP * 32; i += 256) wl[i] = wt[i];
  unsigned chunk_id;
  {
    const unsigned n = gridDim.x, L = blockIdx.x;
    const unsigned xcd = L & 7, q = n >> 3, r = n & 7;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    chunk_id = base + (L >> 3);
  }
  const int seg = chunk_id % g.nseg, range = chunk_id / g.nseg;
  const int w0 = seg * kTapWseg;
  const int g_begin = range * g.gpc, g_end = min(g.NG, g_begin + g.gpc);
  constexpr int nxc = (kTapCols * 8 + 63) / 64;
  int xoff[4], xmeta[4];
  const int plane_g = g.H * g.W * g.K;
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int q = wave + n * 4;
    int off = -2, meta = -1;
    if (q < 3 * nxc) {
      const int pl = q / nxc, ch = q % nxc;
      const int j = ch * 64 + lane;
      if (j < kTapCols * 8) {
        const int u = j >> 3, c = (((j & 7) ^ (u & 7)) << 2), wsrc = w0 + u - 1;
        off = (wsrc >= 0 && wsrc < g.W && c < g.K) ? pl * plane_g + wsrc * g.K + c : -1;
      }
      meta = pl | ((pl * kTapPlaneF + ch * 256) << 4);
    }
    xoff[n] = off;
    xmeta[n] = __builtin_amdgcn_readfirstlane(meta);
  }
  auto stage_row = [&](int b, int d, int hp) {
    const float* base = X + ((long)(b * g.D + d - 1) * g.H + (hp - 1)) * (long)(g.W * g.K);
    const int h = hp - 1;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int meta = xmeta[n];
      if (meta < 0) break;
      const int pl = meta & 3, dp = d - 1 + pl;
      const bool rowok = h >= 0 && h < g.H && dp >= 0 && dp < g.D;
      float* dst = ring + (meta >> 4) + (hp & 3) * kTapRowF;
      const int off = xoff[n];
      const float* src = (rowok && off >= 0) ? base + off : kWgZeros;
      if (off != -2) __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
    }
  };
  const int v = tid >> 3, quad = tid & 7;
  float bv[kThinNP];
#pragma unroll
  for (int n = 0; n < kThinNP; ++n) bv[n] = (g.has_bias && n < g.N) ? bias[n] : 0.0f;

  bool fresh = true;
  int h = g_begin % g.H, d, b;
  {
    const int bd = g_begin / g.H;
    b = bd / g.D; d = bd % g.D;
  }
  for (int G = g_begin; G < g_end; ++G) {
    if (fresh) {
      stage_row(b, d, h); stage_row(b, d, h + 1); stage_row(b, d, h + 2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    const bool same_plane = G + 1 < g_end && h + 1 < g.H;
    if (same_plane) stage_row(b, d, h + 3);
    float acc[kThinNP];
#pragma unroll
    for (int n = 0; n < kThinNP; ++n) acc[n] = 0.0f;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const float* rowp = ring + kd * kTapPlaneF + ((h + kh) & 3) * kTapRowF;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int u = v + kw, tap = (kd * 3 + kh) * 3 + kw;
          const float4 xq = *reinterpret_cast<const float4*>(rowp + u * 32 + ((quad ^ (u & 7)) << 2));
#pragma unroll
          for (int n = 0; n < kThinNP; ++n) {
            if (n < g.N) {
              const float4 wq = *reinterpret_cast<const float4*>(wl + (tap * kThinNP + n) * 32 + quad * 4);
              acc[n] += xq.x * wq.x + xq.y * wq.y + xq.z * wq.z + xq.w * wq.w;
            }
          }
        }
      }
    // fold the 8 channel quads of a voxel (adjacent lanes)
#pragma unroll
    for (int n = 0; n < kThinNP; ++n) {
      acc[n] += __shfl_xor(acc[n], 1, 64);
      acc[n] += __shfl_xor(acc[n], 2, 64);
      acc[n] += __shfl_xor(acc[n], 4, 64);
    }
    const int wv = w0 + v;
    if (quad == 0 && wv < g.W) {
      float* dst = Y + (((long)(b * g.D + d) * g.H + h) * g.W + wv) * g.N;
#pragma unroll
      for (int n = 0; n < kThinNP; ++n)
        if (n < g.N) {
          float o = acc[n] + bv[n];
          dst[n] = g.relu ? fmaxf(o, 0.0f) : o;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    fresh = !same_plane;
    if (++h == g.H) {
      h = 0;
      if (++d == g.D) { d = 0; ++b; }
    }
  }
}

bool conv_thin_applicable(const ssbev_conv_dims* d, int mode) {
  if (d->transposed || d->kd != 3 || d->kh != 3 || d->kw != 3 || d->sd != 1 || d->sh != 1 || d->sw != 1) return false;
  if (d->pd != 1 || d->ph != 1 || d->pw != 1 || d->dd != 1 || d->dh != 1 || d->dw != 1) return false;
  if (d->Di != d->Do || d->Hi != d->Ho || d->Wi != d->Wo || d->accumulate || d->tile_hint == 8) return false;
  const int K = mode == 0 ? d->Cin : d->Cout, N = mode == 0 ? d->Cout : d->Cin;
  if (K > 32 || K < 16 || K % 4 != 0 || N > kThinNP) return false;
  return d->tile_hint == 9 || (long)d->B * d->Do * d->Ho * ((d->Wo + kTapWseg - 1) / kTapWseg) >= 1024L * 16;
}

int launch_conv_thin(const float* x, const float* wt, const float* bias, float* y, const ssbev_conv_dims* d, int mode,
                     hipStream_t st) {
  ConvTapGeom g;
  g.B = d->B; g.D = d->Do; g.H = d->Ho; g.W = d->Wo;
  g.K = mode == 0 ? d->Cin : d->Cout;
  g.N = mode == 0 ? d->Cout : d->Cin;
  g.nseg = (g.W + kTapWseg - 1) / kTapWseg;
  g.NG = g.B * g.D * g.H;
  g.relu = mode == 0 ? d->relu : 0;
  g.has_bias = (mode == 0 && bias) ? 1 : 0;
  long nranges = std::max(1L, 1024L / g.nseg);
  if (nranges > g.NG) nranges = g.NG;
  g.gpc = (int)((g.NG + nranges - 1) / nranges);
  nranges = (g.NG + g.gpc - 1) / g.gpc;
  const size_t lds = (size_t)(kTapRingF + 27 * kThinNP * 32) * sizeof(float);
  auto kern = conv_thin_kernel;
  if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)lds) != hipSuccess)
    return SSBEV_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3((unsigned)(nranges * g.nseg)), dim3(256), lds, st, x, wt, bias, y, g);
  return ssbev_launch_status();
}

// Stride-2 "down" gather on conv_tap2_kernel: conv forward (mode 0, !transposed) or transposed-conv data gradient (mode 1,
// transposed) with k3 s2 p1 (output_padding 1: source = 2 x destination), K <= 32 source channels, 33..64 destination channels.
// tile_hint 8 keeps the generic gather kernel, 5 forces this one on small problems (tests).
bool conv_tap2_applicable(const ssbev_conv_dims* d, int mode) {
  static const bool enabled = !(ssbev_tune("SSBEV_TAP2") && atoi(ssbev_tune("SSBEV_TAP2")) == 0);          // A/B hook
  if (!enabled && d->tile_hint != 5) return false;
  if (!((mode == 0 && !d->transposed) || (mode == 1 && d->transposed))) return false;
  if (d->kd != 3 || d->kh != 3 || d->kw != 3 || d->sd != 2 || d->sh != 2 || d->sw != 2) return false;
  if (d->pd != 1 || d->ph != 1 || d->pw != 1 || d->dd != 1 || d->dh != 1 || d->dw != 1) return false;
  if (d->precision != 0 || d->tile_hint == 8 || (d->tile_hint != 0 && d->tile_hint != 5)) return false;
  const int K = mode == 0 ? d->Cin : d->Cout, N = mode == 0 ? d->Cout : d->Cin;
  if (K > 32 || K < 16 || K % 4 != 0 || N <= 32 || N > 64 || N % 4 != 0) return false;
  // source / destination grids of the gather and their stride-2 relation
  const int Ds = mode == 0 ? d->Di : d->Do, Hs = mode == 0 ? d->Hi : d->Ho, Ws = mode == 0 ? d->Wi : d->Wo;
  const int Dd = mode == 0 ? d->Do : d->Di, Hd = mode == 0 ? d->Ho : d->Hi, Wd = mode == 0 ? d->Wo : d->Wi;
  if (Dd != (Ds - 1) / 2 + 1 || Hd != (Hs - 1) / 2 + 1 || Wd != (Ws - 1) / 2 + 1) return false;
  if ((long)Hs * Ws * K >= (1L << 30)) return false;
  // worth it when the row-pair walks fill the chip: >= 256 workgroups of >= 8 row pairs
  return d->tile_hint == 5 || (long)d->B * Dd * ((Hd + 1) / 2) * ((Wd + 15) / 16) >= 256L * 8;
}

int launch_conv_tap2(const float* x, const float* wp, const float* bias, float* y, const ssbev_conv_dims* d, int mode,
                     hipStream_t st) {
  ConvTapGeom g;
  g.B = d->B;
  g.Ds = mode == 0 ? d->Di : d->Do; g.Hs = mode == 0 ? d->Hi : d->Ho; g.Ws = mode == 0 ? d->Wi : d->Wo;
  g.D = mode == 0 ? d->Do : d->Di; g.H = mode == 0 ? d->Ho : d->Hi; g.W = mode == 0 ? d->Wo : d->Wi;
  g.K = mode == 0 ? d->Cin : d->Cout;
  g.N = mode == 0 ? d->Cout : d->Cin;
  g.nseg = (g.W + 15) / 16;
  const int H2 = (g.H + 1) / 2;
  g.NG = g.B * g.D * H2;                     // output row pairs
  g.relu = mode == 0 ? d->relu : 0;
  g.has_bias = (mode == 0 && bias) ? 1 : 0;
  g.accumulate = d->accumulate;
  // one 512-thread workgroup per CU (150 KB of LDS): whole rounds of 256 workgroups, chunk start-up ~1 pair, plane crossing ~0.5
  double best = 1e30;
  g.gpc = 1;
  for (int c = 1; c <= g.NG && c <= 96; ++c) {
    const long blocks = (long)((g.NG + c - 1) / c) * g.nseg;
    const long rounds = (blocks + 255) / 256;
    const double crossings = H2 % c == 0 ? 0.0 : (c % H2 == 0 ? c / H2 - 1 : (double)c / H2);
    const double cost = rounds * (c + 1.0 + 0.5 * crossings);
    if (cost < best) { best = cost; g.gpc = c; }
  }
  if (const char* e = ssbev_tune("SSBEV_TAP2_GPC")) { const int v = atoi(e); if (v > 0) g.gpc = v; }   // tuning hook
  const long nranges = (g.NG + g.gpc - 1) / g.gpc;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tap2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)kT2LdsBytes) != hipSuccess)
    return SSBEV_ELAUNCH;
  hipLaunchKernelGGL(conv_tap2_kernel, dim3((unsigned)(nranges * g.nseg)), dim3(512), kT2LdsBytes, st, x, wp, bias, y, g);
  return ssbev_launch_status();
}

// Stride-2 "up" gather on conv_tap2up_kernel: transposed-conv forward (mode 0, transposed) or conv data gradient (mode 1,
// !transposed) with k3 s2 p1, fine grid = 2 x coarse grid, 33..64 source channels, <= 32 destination channels (multiple of 8).
bool conv_tap2up_applicable(const ssbev_conv_dims* d, int mode) {
  static const bool enabled = !(ssbev_tune("SSBEV_TAP2UP") && atoi(ssbev_tune("SSBEV_TAP2UP")) == 0);      // A/B hook
  if (!enabled && d->tile_hint != 5) return false;
  if (!((mode == 0 && d->transposed) || (mode == 1 && !d->transposed))) return false;
  if (d->kd != 3 || d->kh != 3 || d->kw != 3 || d->sd != 2 || d->sh != 2 || d->sw != 2) return false;
  if (d->pd != 1 || d->ph != 1 || d->pw != 1 || d->dd != 1 || d->dh != 1 || d->dw != 1) return false;
  if (d->precision != 0 || d->tile_hint == 8 || (d->tile_hint != 0 && d->tile_hint != 5)) return false;
  const int K = mode == 0 ? d->Cin : d->Cout, N = mode == 0 ? d->Cout : d->Cin;
  if (K <= 32 || K > 64 || K % 4 != 0 || N > 32 || N < 16 || N % 8 != 0) return false;
  // coarse (source) and fine (destination) grids
  const int Ds = mode == 0 ? d->Di : d->Do, Hs = mode == 0 ? d->Hi : d->Ho, Ws = mode == 0 ? d->Wi : d->Wo;
  const int Dd = mode == 0 ? d->Do : d->Di, Hd = mode == 0 ? d->Ho : d->Hi, Wd = mode == 0 ? d->Wo : d->Wi;
  if (Dd != 2 * Ds || Hd != 2 * Hs || Wd != 2 * Ws) return false;
  if ((long)Hs * Ws * K >= (1L << 30)) return false;
  return d->tile_hint == 5 || (long)d->B * Ds * ((Hs + 1) / 2) * ((Ws + 15) / 16) >= 256L * 8;
}

int launch_conv_tap2up(const float* x, const float* wp, const float* bias, float* y, const ssbev_conv_dims* d, int mode,
                       hipStream_t st) {
  ConvTapGeom g;
  g.B = d->B;
  g.Ds = mode == 0 ? d->Di : d->Do; g.Hs = mode == 0 ? d->Hi : d->Ho; g.Ws = mode == 0 ? d->Wi : d->Wo;
  g.D = 2 * g.Ds; g.H = 2 * g.Hs; g.W = 2 * g.Ws;
  g.K = mode == 0 ? d->Cin : d->Cout;
  g.N = mode == 0 ? d->Cout : d->Cin;
  g.nseg = (g.Ws + 15) / 16;
  const int H2 = (g.Hs + 1) / 2;
  g.NG = g.B * g.Ds * H2;                    // coarse row pairs
  g.relu = mode == 0 ? d->relu : 0;
  g.has_bias = (mode == 0 && bias) ? 1 : 0;
  g.accumulate = d->accumulate;
  double best = 1e30;
  g.gpc = 1;
  for (int c = 1; c <= g.NG && c <= 96; ++c) {
    const long blocks = (long)((g.NG + c - 1) / c) * g.nseg;
    const long rounds = (blocks + 255) / 256;
    const double crossings = H2 % c == 0 ? 0.0 : (c % H2 == 0 ? c / H2 - 1 : (double)c / H2);
    const double cost = rounds * (c + 1.0 + 0.5 * crossings);
    if (cost < best) { best = cost; g.gpc = c; }
  }
  if (const char* e = ssbev_tune("SSBEV_TAP2UP_GPC")) { const int v = atoi(e); if (v > 0) g.gpc = v; }   // tuning hook
  const long nranges = (g.NG + g.gpc - 1) / g.gpc;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tap2up_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)kUpLdsBytes) != hipSuccess)
    return SSBEV_ELAUNCH;
  hipLaunchKernelGGL(conv_tap2up_kernel, dim3((unsigned)(nranges * g.nseg)), dim3(1024), kUpLdsBytes, st, x, wp, bias, y, g);
  return ssbev_launch_status();
}

// tile_hint 8 forces the generic gather kernels (A/B timing), 9 forces this kernel on small problems (tests)
bool conv_tap_applicable(const ssbev_conv_dims* d, int mode) {
  if (d->transposed || d->kd != 3 || d->kh != 3 || d->kw != 3 || d->sd != 1 || d->sh != 1 || d->sw != 1) return false;
  if (d->pd != 1 || d->ph != 1 || d->pw != 1 || d->dd != 1 || d->dh != 1 || d->dw != 1) return false;
  if (d->Di != d->Do || d->Hi != d->Ho || d->Wi != d->Wo || d->tile_hint == 8) return false;
  if (d->Cin > 32 || d->Cout > 32) return false;
  const int K = mode == 0 ? d->Cin : d->Cout;
  if (K % 4 != 0) return false;
  if (K < 16 && d->tile_hint != 9 && d->tile_hint != 6) return false;     // few input channels: the generic kernel has little to fetch
  // worth it only when the volume fills the chip with row walks: >= 1024 (segment, row-range) workgroups of >= 16 rows
  // (tile_hint 9 forces it on any size: tests)
  return d->tile_hint == 9 || d->tile_hint == 6 ||
         (long)d->B * d->Do * d->Ho * ((d->Wo + kTapWseg - 1) / kTapWseg) >= 1024L * 16;
}

// F(2,3)-along-h variant: fp32, even H.  tile_hint 6 keeps the plain tap kernel (tests / A-B timing)
bool conv_taph_applicable(const ssbev_conv_dims* d, int mode) {
  return conv_tap_applicable(d, mode) && d->Ho % 2 == 0 && d->precision == 0 && d->tile_hint != 6;
}

// 1x1x1 stride-1 layers with <= 32 channels on both sides on conv_pw32_kernel (tile_hint 8 keeps the generic gather kernel)
bool conv_pw32_applicable(const ssbev_conv_dims* d, int mode) {
  static const int off = ssbev_tune("SSBEV_PW32") ? atoi(ssbev_tune("SSBEV_PW32")) == 0 : 0;
  if (off || d->transposed || d->precision != 0 || d->tile_hint == 8) return false;
  if (d->kd != 1 || d->kh != 1 || d->kw != 1 || d->sd != 1 || d->sh != 1 || d->sw != 1) return false;
  if (d->pd != 0 || d->ph != 0 || d->pw != 0) return false;
  if (d->Di != d->Do || d->Hi != d->Ho || d->Wi != d->Wo) return false;
  if (d->Cin > 32 || d->Cout > 32 || d->Cin % 4 || d->Cout % 4 || d->Cin < 8 || d->Cout < 8) return false;
  (void)mode;
  return (long)d->B * d->Do * d->Ho * d->Wo >= 32768;
}

int launch_conv_pw32(const float* x, const float* wp, const float* bias, float* y, const ssbev_conv_dims* d, int mode,
                     hipStream_t st) {
  PwGeom g;
  g.M = (long)d->B * d->Do * d->Ho * d->Wo;
  g.K = mode == 0 ? d->Cin : d->Cout;
  g.N = mode == 0 ? d->Cout : d->Cin;
  g.relu = mode == 0 ? d->relu : 0;
  g.has_bias = (mode == 0 && bias) ? 1 : 0;
  g.accumulate = d->accumulate;
  const long ntiles = (g.M + 31) / 32;
  // ~16 waves per CU, each with a run of at least 4 tiles
  long waves = 256L * 16;
  if (waves * 4 > ntiles) waves = (ntiles + 3) / 4;
  g.tpw = (ntiles + waves - 1) / waves;
  const long nblocks = ((ntiles + g.tpw - 1) / g.tpw + 3) / 4;
  hipLaunchKernelGGL(conv_pw32_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, x, wp, bias, y, g);
  return ssbev_launch_status();
}

// F(2,3) along d and h: even D as well.  tile_hint 4 keeps the h-only kernel (A/B timing, tests)
bool conv_tapdh_applicable(const ssbev_conv_dims* d, int mode) {
  static const int off = ssbev_tune("SSBEV_TAPDH") ? atoi(ssbev_tune("SSBEV_TAPDH")) == 0 : 0;
  return !off && conv_taph_applicable(d, mode) && d->Do % 2 == 0 && d->tile_hint != 4;
}

int launch_conv_tapdh(const float* x, const float* wp, const float* bias, float* y, const ssbev_conv_dims* d, int mode,
                      hipStream_t st) {
  ConvTapGeom g;
  g.B = d->B; g.D = d->Do; g.H = d->Ho; g.W = d->Wo;
  g.K = mode == 0 ? d->Cin : d->Cout;
  g.N = mode == 0 ? d->Cout : d->Cin;
  g.nseg = (g.W + kTapWseg - 1) / kTapWseg;
  g.NG = g.B * (g.D / 2) * (g.H / 2);        // 2 x 2 (plane, row) blocks
  g.relu = mode == 0 ? d->relu : 0;
  g.has_bias = (mode == 0 && bias) ? 1 : 0;
  g.accumulate = d->accumulate;
  // one 1024-thread workgroup per CU (149 KB of LDS); chunk length as in launch_conv_taph: whole rounds of 256 workgroups,
  // then the start-up of a chunk (weights + first four rows of four planes) and the restage at every plane-pair crossing
  const int H2 = g.H / 2;
  double best = 1e30;
  g.gpc = 1;
  for (int c = 1; c <= g.NG && c <= 96; ++c) {
    const long blocks = (long)((g.NG + c - 1) / c) * g.nseg;
    const long rounds = (blocks + 255) / 256;
    const double crossings = H2 % c == 0 ? 0.0 : (c % H2 == 0 ? c / H2 - 1 : (double)c / H2);
    const double cost = rounds * (c + 0.5 + 0.3 * crossings);
    if (cost < best) { best = cost; g.gpc = c; }
  }
  if (const char* e = ssbev_tune("SSBEV_TAPDH_GPC")) { const int v = atoi(e); if (v > 0) g.gpc = v; }   // tuning hook
  const long nranges = (g.NG + g.gpc - 1) / g.gpc;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tapdh_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)kDhLdsBytes) != hipSuccess)
    return SSBEV_ELAUNCH;
  static const int times = ssbev_tune("SSBEV_TAPDH_TIMES") ? atoi(ssbev_tune("SSBEV_TAPDH_TIMES")) : 0;
  if (times) {      // tuning hook: per-phase shader clocks of every wave (fresh stage / walk / publish + barrier / fold / tail barrier)
    const size_t nwg = (size_t)(nranges * g.nseg), n = nwg * 16 * 8;
    unsigned long long* dev = nullptr;
    if (hipMalloc(&dev, n * 8) != hipSuccess) return SSBEV_ELAUNCH;
    g.dbg = dev;
    hipLaunchKernelGGL(conv_tapdh_kernel, dim3((unsigned)nwg), dim3(1024), kDhLdsBytes, st, x, wp, bias, y, g);
    std::vector<unsigned long long> h(n);
    hipStreamSynchronize(st);
    hipMemcpy(h.data(), dev, n * 8, hipMemcpyDeviceToHost);
    hipFree(dev);
    double ph[5] = {0, 0, 0, 0, 0}, life = 0, blocks = 0;
    unsigned long long t0 = ~0ull, t1 = 0;
    for (size_t w = 0; w < nwg * 16; ++w) {
      for (int i = 0; i < 5; ++i) ph[i] += (double)h[w * 8 + i];
      life += (double)(h[w * 8 + 6] - h[w * 8 + 5]);
      blocks += (double)h[w * 8 + 7];
      t0 = std::min(t0, h[w * 8 + 5]); t1 = std::max(t1, h[w * 8 + 6]);
    }
    if (times > 1) {        // per wave slot: where the sixteen waves stand when their walk ends / their publish phase ends
      for (int wv = 0; wv < 16; ++wv) {
        double a[5] = {0, 0, 0, 0, 0}, nb = 0;
        for (size_t w = wv; w < nwg * 16; w += 16) { for (int i = 0; i < 5; ++i) a[i] += (double)h[w * 8 + i]; nb += (double)h[w * 8 + 7]; }
        fprintf(stderr, "  wave %2d: stage %.0f walk %.0f publish %.0f fold %.0f tail %.0f\n", wv, a[0] / nb, a[1] / nb, a[2] / nb, a[3] / nb, a[4] / nb);
      }
    }
    fprintf(stderr, "tapdh clocks per 2x2 block and wave (s_memtime ticks): stage %.0f walk %.0f publish+barrier %.0f fold %.0f tail barrier %.0f"
            " | per chunk: life %.0f for %.1f blocks | kernel span %.0f ticks, %zu workgroups\n",
            ph[0] / blocks, ph[1] / blocks, ph[2] / blocks, ph[3] / blocks, ph[4] / blocks, life / (nwg * 16), blocks / (nwg * 16),
            (double)(t1 - t0), nwg);
    return ssbev_launch_status();
  }
  hipLaunchKernelGGL(conv_tapdh_kernel, dim3((unsigned)(nranges * g.nseg)), dim3(1024), kDhLdsBytes, st, x, wp, bias, y, g);
  return ssbev_launch_status();
}

// weight gradient of the conv_tapdh layers in the F(2,3) x F(2,3) domain (wgrad_tapdh_kernel).  tile_hint 9 forces it on
// small problems (tests); 4 / 5 / 6 / 7 keep the older kernels (A/B timing)
size_t align256b(size_t x);
struct WgradDhPlan { bool ok; WgradDhGeom g; int nchunks; };
WgradDhPlan plan_wgrad_dh(const ssbev_conv_dims* d) {
  static const int off = ssbev_tune("SSBEV_WGRAD_DH") ? atoi(ssbev_tune("SSBEV_WGRAD_DH")) == 0 : 0;
  WgradDhPlan p;
  p.ok = false; p.nchunks = 0;
  if (off || d->transposed || d->precision != 0) return p;
  if (d->kd != 3 || d->kh != 3 || d->kw != 3 || d->sd != 1 || d->sh != 1 || d->sw != 1) return p;
  if (d->pd != 1 || d->ph != 1 || d->pw != 1 || d->dd != 1 || d->dh != 1 || d->dw != 1) return p;
  if (d->Di != d->Do || d->Hi != d->Ho || d->Wi != d->Wo || d->Do % 2 || d->Ho % 2) return p;
  if (d->Cin > 32 || d->Cout > 32 || d->Cin % 4 || d->Cout % 4 || d->Cin < 16 || d->Cout < 8) return p;
  if (d->tile_hint == 4 || d->tile_hint == 5 || d->tile_hint == 6 || d->tile_hint == 7 || d->tile_hint == 8) return p;
  const int nseg = (d->Wo + kTapWseg - 1) / kTapWseg;
  if (d->tile_hint != 9 && (long)d->B * d->Do * d->Ho * nseg < 1024L * 16) return p;
  if ((long)d->Ho * d->Wo * 32 >= (1L << 30)) return p;
  WgradDhGeom& g = p.g;
  g.B = d->B; g.D = d->Do; g.H = d->Ho; g.W = d->Wo; g.Cq = d->Cin; g.Cp = d->Cout;
  g.nseg = nseg;
  g.NG = g.B * (g.D / 2) * (g.H / 2);
  // one 1024-thread workgroup per CU; the chunk partials (48 tiles per chunk) are folded afterwards, so ONE round of long
  // chunks: whole rounds of 256 workgroups, start-up + plane-pair crossings as in launch_conv_tapdh
  const int H2 = g.H / 2;
  double best = 1e30;
  g.gpc = 1;
  for (int c = 1; c <= g.NG && c <= 192; ++c) {
    const long blocks = (long)((g.NG + c - 1) / c) * g.nseg;
    const long rounds = (blocks + 255) / 256;
    const double crossings = H2 % c == 0 ? 0.0 : (c % H2 == 0 ? c / H2 - 1 : (double)c / H2);
    const double cost = rounds * (c + 1.0 + 0.3 * crossings);
    if (cost < best) { best = cost; g.gpc = c; }
  }
  if (const char* e = ssbev_tune("SSBEV_WGRAD_DH_GPC")) { const int v = atoi(e); if (v > 0) g.gpc = v; }   // tuning hook
  p.nchunks = ((g.NG + g.gpc - 1) / g.gpc) * g.nseg;
  p.ok = true;
  return p;
}

size_t wgrad_dh_workspace(const WgradDhPlan& p) {
  const size_t tile = (size_t)48 * p.g.Cq * p.g.Cp * sizeof(float);
  return align256b((size_t)p.nchunks * tile) + align256b(tile);
}

int run_wgrad_dh(const float* x, const float* gy, float* gw, const WgradDhPlan& p, void* ws, hipStream_t st) {
  float* partial = static_cast<float*>(ws);
  float* gU = reinterpret_cast<float*>(static_cast<char*>(ws) + align256b((size_t)p.nchunks * 48 * p.g.Cq * p.g.Cp * sizeof(float)));
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_tapdh_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)kWdLdsBytes) != hipSuccess)
    return SSBEV_ELAUNCH;
  hipLaunchKernelGGL(wgrad_tapdh_kernel, dim3((unsigned)p.nchunks), dim3(1024), kWdLdsBytes, st, x, gy, partial, p.g);
  launch_wgrad_reduce(partial, gU, p.nchunks, 48, p.g.Cq, p.g.Cp, st);          // -> gU[n][k][48]
  const int npq = p.g.Cq * p.g.Cp;
  hipLaunchKernelGGL(wgrad_dh_finish_kernel, dim3(cdiv((size_t)npq * 3, 256)), dim3(256), 0, st, gU, gw, npq);
  return ssbev_launch_status();
}

int launch_conv_taph(const float* x, const float* wp, const float* bias, float* y, const ssbev_conv_dims* d, int mode,
                     hipStream_t st) {
  ConvTapGeom g;
  g.B = d->B; g.D = d->Do; g.H = d->Ho; g.W = d->Wo;
  g.K = mode == 0 ? d->Cin : d->Cout;
  g.N = mode == 0 ? d->Cout : d->Cin;
  g.nseg = (g.W + kTapWseg - 1) / kTapWseg;
  g.NG = g.B * g.D * (g.H / 2);              // row pairs
  g.relu = mode == 0 ? d->relu : 0;
  g.has_bias = (mode == 0 && bias) ? 1 : 0;
  g.accumulate = d->accumulate;
  // One 512-thread workgroup per CU (141 KB of LDS).  Chunk length (row pairs per workgroup) by a small cost model fitted on
  // the 192 x 48 x 160 layer (tools/taph_gpc_probe.py): whole rounds of 256 workgroups matter most (the last round's idle
  // CUs: 12 pairs -> 7.5 rounds 0.580 ms, 18 pairs -> 5.0 rounds 0.537 ms), then the start-up of a chunk (weights + first
  // four rows, ~0.6 pair) and the ring restage at every depth-plane crossing (~0.3 pair)
  const int H2 = g.H / 2;
  double best = 1e30;
  g.gpc = 1;
  for (int c = 1; c <= g.NG && c <= 96; ++c) {
    const long blocks = (long)((g.NG + c - 1) / c) * g.nseg;
    const long rounds = (blocks + 255) / 256;
    const double crossings = H2 % c == 0 ? 0.0 : (c % H2 == 0 ? c / H2 - 1 : (double)c / H2);
    const double cost = rounds * (c + 0.6 + 0.3 * crossings);
    if (cost < best) { best = cost; g.gpc = c; }
  }
  if (const char* e = ssbev_tune("SSBEV_TAPH_GPC")) { const int v = atoi(e); if (v > 0) g.gpc = v; }   // tuning hook
  // (Round 3 built a plane-aligned chunk order -- XCD x owns 24 consecutive planes, plane index fastest, so that the chunks of
  // d - 1, d, d + 1 meet in one L2: 572 -> 228 MB of HBM reads per launch, but the workgroups moving in step cost the kernel
  // 8 % (0.533 -> 0.579 ms, profiles/r3y_taph_plane_aligned.txt).  The kernel is not HBM-bound; removed in round 6.)
  const long nranges = (g.NG + g.gpc - 1) / g.gpc;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_taph_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)kTwLdsBytes) != hipSuccess)
    return SSBEV_ELAUNCH;
  hipLaunchKernelGGL(conv_taph_kernel, dim3((unsigned)(nranges * g.nseg)), dim3(512), kTwLdsBytes, st, x, wp,
                     bias, y, g);
  return ssbev_launch_status();
}

int launch_conv_tap(const float* x, const float* wp, const float* bias, float* y, const ssbev_conv_dims* d, int mode,
                    hipStream_t st) {
  ConvTapGeom g;
  g.B = d->B; g.D = d->Do; g.H = d->Ho; g.W = d->Wo;
  g.K = mode == 0 ? d->Cin : d->Cout;
  g.N = mode == 0 ? d->Cout : d->Cin;
  g.nseg = (g.W + kTapWseg - 1) / kTapWseg;
  g.NG = g.B * g.D * g.H;
  g.relu = mode == 0 ? d->relu : 0;
  g.has_bias = (mode == 0 && bias) ? 1 : 0;
  g.accumulate = d->accumulate;
  // two workgroups per CU; whole rounds of 512 workgroups, >= 16 rows each
  long nranges = 512 / g.nseg;
  for (long rounds = 8; rounds >= 1; --rounds) {
    const long nr = (512 * rounds) / g.nseg;
    if ((g.NG + nr - 1) / nr >= 24) { nranges = nr; break; }
  }
  if (nranges < 1) nranges = 1;
  g.gpc = (int)((g.NG + nranges - 1) / nranges);
  nranges = (g.NG + g.gpc - 1) / g.gpc;
  auto kern = d->precision == 1 ? conv_tap_kernel<true> : conv_tap_kernel<false>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)kTapLdsBytes) != hipSuccess)
    return SSBEV_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3((unsigned)(nranges * g.nseg)), dim3(256), kTapLdsBytes, st, x, wp, bias, y, g);
  return ssbev_launch_status();
}

bool wgrad_cf_applicable(const ssbev_conv_dims* d) {
  if (d->transposed || d->sd != 1 || d->sh != 1 || d->sw != 1) return false;
  if (d->Di != d->Do || d->Hi != d->Ho || d->Wi != d->Wo) return false;
  if (d->Wo % 4 != 0 || ((long)d->B * d->Do * d->Ho * d->Wo) % 8 != 0) return false;
  if (2 * d->pd != d->dd * (d->kd - 1) || 2 * d->ph != d->dh * (d->kh - 1) || 2 * d->pw != d->dw * (d->kw - 1)) return false;
  return true;
}

WgradCfGeom make_wgrad_cf_geom(const ssbev_conv_dims* d, WgradCfg* cfg_out) {
  WgradCfGeom g;
  g.B = d->B; g.Cp = d->Cout; g.Cq = d->Cin; g.D = d->Do; g.H = d->Ho; g.W = d->Wo;
  g.kd = d->kd; g.kh = d->kh; g.kw = d->kw; g.dd = d->dd; g.dh = d->dh; g.dw = d->dw;
  g.Dp = d->Di + 2 * d->pd; g.Hp = d->Hi + 2 * d->ph;
  g.Wp = ((d->Wi + 2 * d->pw + 3) & ~3) + 4;          // +4: slack for the last row's unaligned float4
  WgradCfg c;
  if (d->kh * d->kw == 1) c = {2, 2, 1, 1};
  else if (g.Cp > 32 && g.Cq > 32) c = {2, 2, 1, 3};
  else if (g.Cp <= 32 && g.Cq <= 32) c = {1, 1, 3, 3};
  else c = {1, 1, 1, 3};
  *cfg_out = c;
  const long N = (long)g.B * g.D * g.H * g.W;
  const long tiles = (long)((g.Cp + 32 * c.MP - 1) / (32 * c.MP)) * ((g.Cq + 32 * c.MQ - 1) / (32 * c.MQ)) * g.kd *
                     ((g.kh + c.TH - 1) / c.TH) * ((g.kw + c.TW - 1) / c.TW);
  // Workgroups (= 4 chunks x one tile) are sized so that their count is just BELOW a multiple of the
  // number of workgroups the chip holds at once (PMC: with 3.09 "rounds" the SIMDs idled 27 % of the time).
  const int acc_regs = 16 * c.MQ * c.MP * c.TH * c.TW;
  const long resident = 256L * (acc_regs > 128 ? 1 : (acc_regs > 64 ? 2 : 3));
  long best_cb = N / (4 * 512L) / (tiles > 0 ? 1 : 1);     // fallback: as many 512-voxel chunks as exist
  if (best_cb < 1) best_cb = 1;
  if (best_cb > resident) best_cb = resident;               // (one tile, huge volume: a single round is enough)
  for (long rounds = 8; rounds >= 1; --rounds) {
    const long cb = (resident * rounds) / (tiles > 0 ? tiles : 1);       // chunk-blocks per tile
    if (cb >= 1 && (N + 4 * cb - 1) / (4 * cb) >= 512) { best_cb = cb; break; }
  }
  long chunk = (N + 4 * best_cb - 1) / (4 * best_cb);
  if (chunk < 512) chunk = 512;
  chunk = (chunk + 31) & ~31L;
  g.chunk = (int)chunk;
  g.nchunks = (int)((N + chunk - 1) / chunk);
  return g;
}

size_t align256b(size_t x) { return (x + 255) & ~(size_t)255; }

template <int MQ, int MP, int TH, int TW>
void launch_wgrad_cf(const float* Pt, const float* Qp, float* ws, const WgradCfGeom& g, hipStream_t st) {
  const int ytiles = ((g.Cq + 32 * MQ - 1) / (32 * MQ)) * ((g.Cp + 32 * MP - 1) / (32 * MP));
  dim3 grid(cdiv(g.nchunks, 4), ytiles, g.kd * ((g.kh + TH - 1) / TH) * ((g.kw + TW - 1) / TW));
  hipLaunchKernelGGL((wgrad_cf_kernel<MQ, MP, TH, TW>), grid, dim3(256), 0, st, Pt, Qp, ws, g);
}

// tile configuration of the weight-gradient kernel for a problem
WgradCfg wgrad_cfg(int Cp, int Cq, int kh, int kw) {
  if (kh * kw == 1) return {2, 2, 1, 1};
  if (Cp > 32 && Cq > 32) return {2, 2, 1, 3};
  if (Cp <= 32 && Cq <= 32) return {1, 1, 3, 3};
  return {1, 1, 1, 3};
}

WgradGeom make_wgrad_geom(const ssbev_conv_dims* d) {
  WgradGeom g;
  g.B = d->B;
  if (!d->transposed) {
    g.Cp = d->Cout; g.Cq = d->Cin;
    g.Ds = d->Do; g.Hs = d->Ho; g.Ws = d->Wo; g.Dq = d->Di; g.Hq = d->Hi; g.Wq = d->Wi;
  } else {
    g.Cp = d->Cin; g.Cq = d->Cout;
    g.Ds = d->Di; g.Hs = d->Hi; g.Ws = d->Wi; g.Dq = d->Do; g.Hq = d->Ho; g.Wq = d->Wo;
  }
  g.kd = d->kd; g.kh = d->kh; g.kw = d->kw; g.sd = d->sd; g.sh = d->sh; g.sw = d->sw;
  g.pd = d->pd; g.ph = d->ph; g.pw = d->pw; g.dd = d->dd; g.dh = d->dh; g.dw = d->dw;
  const long Mtot = (long)g.B * g.Ds * g.Hs * g.Ws;
  const WgradCfg c = wgrad_cfg(g.Cp, g.Cq, g.kh, g.kw);
  // aim for ~3000 wave-tasks in total, chunks of at least 256 voxels
  const long tiles = (long)((g.Cp + 32 * c.MP - 1) / (32 * c.MP)) * ((g.Cq + 32 * c.MQ - 1) / (32 * c.MQ)) * g.kd *
                     ((g.kh + c.TH - 1) / c.TH) * ((g.kw + c.TW - 1) / c.TW);
  long want = 3072 / (tiles > 0 ? tiles : 1);
  if (want < 1) want = 1;
  long chunk = (Mtot + want - 1) / want;
  if (chunk < 256) chunk = 256;
  chunk = (chunk + 1) & ~1L;
  g.chunk = (int)chunk;
  g.nchunks = (int)((Mtot + chunk - 1) / chunk);
  return g;
}

template <int MQ, int MP, int TH, int TW>
void launch_wgrad(const float* P, const float* Qt, float* ws, const WgradGeom& g, hipStream_t st) {
  const int ytiles = ((g.Cq + 32 * MQ - 1) / (32 * MQ)) * ((g.Cp + 32 * MP - 1) / (32 * MP));
  dim3 grid(cdiv(g.nchunks, 4), ytiles, g.kd * ((g.kh + TH - 1) / TH) * ((g.kw + TW - 1) / TW));
  hipLaunchKernelGGL((wgrad_kernel<MQ, MP, TH, TW>), grid, dim3(256), 0, st, P, Qt, ws, g);
}

}  // namespace

namespace ssbev_detail {
// fixed-order fold of per-chunk partial weight gradients [chunk][tap][Cq][Cp] into the torch layout (shared with conv_bf16.hip)
void wgrad_reduce(float* partial, float* gw, int nchunks, int taps, int Cq, int Cp, hipStream_t st) {
  launch_wgrad_reduce(partial, gw, nchunks, taps, Cq, Cp, st);
}
}  // namespace ssbev_detail

extern "C" {

int ssbev_conv_kernel_class(const ssbev_conv_dims* d, int mode) {
  if (!conv_dims_ok(d) || mode < 0 || mode > 2) return SSBEV_EINVAL;
  if (ssbev_bf16::storage_mode(d)) return ssbev_bf16::dims_ok(d, mode) ? ssbev_bf16::kernel_class(d, mode) : SSBEV_EINVAL;
  if (mode == 2) return ssbev_thin::wgrad_applicable(d) ? 6 : 0;      // weight gradient: 6 = wgrad_thinside_kernel (unpadded thin side)
  if (ssbev_thin::thinin_applicable(d, mode)) return 4;
  if (ssbev_thin::thinout_applicable(d, mode)) return 5;     // ssbev_conv_thin_* (caller-owned workspace); ssbev_conv_fwd falls back to class 3 / 0
  if (conv_thin_applicable(d, mode)) return 3;
  if (conv_tap2_applicable(d, mode)) return 7;                // stride-2 "down" gather on conv_tap2_kernel
  if (conv_tap2up_applicable(d, mode)) return 8;              // stride-2 "up" gather on conv_tap2up_kernel
  if (conv_pw32_applicable(d, mode)) return 10;               // 1x1x1, <= 32 channels: streaming kernel
  if (conv_tapdh_applicable(d, mode)) return 9;               // F(2,3) along d and h inside the tap walk
  if (conv_taph_applicable(d, mode)) return 2;
  if (conv_tap_applicable(d, mode)) return 1;
  {                                                           // the generic gather: conv_igemm_kernel (11) or conv_gather_kernel (0)
    ConvGeom g;
    const bool fwd = mode == 0;
    g.B = d->B; g.Cin = fwd ? d->Cin : d->Cout; g.Cout = fwd ? d->Cout : d->Cin; g.CinPad = pad8(g.Cin); g.CoutPad = pad32(g.Cout);
    g.Di = fwd ? d->Di : d->Do; g.Hi = fwd ? d->Hi : d->Ho; g.Wi = fwd ? d->Wi : d->Wo;
    g.Do = fwd ? d->Do : d->Di; g.Ho = fwd ? d->Ho : d->Hi; g.Wo = fwd ? d->Wo : d->Wi;
    g.kd = d->kd; g.kh = d->kh; g.kw = d->kw; g.sd = d->sd; g.sh = d->sh; g.sw = d->sw;
    g.pd = d->pd; g.ph = d->ph; g.pw = d->pw; g.dd = d->dd; g.dh = d->dh; g.dw = d->dw;
    g.form = (d->transposed != 0) == fwd ? 1 : 0; g.relu = 0; g.accumulate = 0;
    g.hint = d->tile_hint >= 10 ? d->tile_hint : 0; g.chunk_taps = 0; g.bf16 = d->precision == 1;
    if (g.Cin % 4 == 0 && conv_igemm_applicable(g)) return 11;
  }
  return 0;
}

size_t ssbev_conv_packed_weight_elems(const ssbev_conv_dims* d) {
  if (!conv_dims_ok(d)) return 0;
  if (ssbev_bf16::storage_mode(d)) return ssbev_bf16::packed_elems(d);
  // big enough for either role assignment (forward or data-gradient operand)
  const size_t taps = (size_t)d->kd * d->kh * d->kw;
  const size_t a = (size_t)pad8(d->Cin) * pad32(d->Cout), b = (size_t)pad8(d->Cout) * pad32(d->Cin);
  const size_t generic = taps * (a > b ? a : b);
  const size_t special = (size_t)std::max(std::max(kTwPackedElems, kDhPackedElems), std::max(kT2PackedElems, kUpPackedElems));
  return generic > special ? generic : special;
}

int ssbev_conv_pack_weight(const float* w_src, float* w_packed, const ssbev_conv_dims* d, int mode,
                           ssbev_stream_t stream) {
  if (!conv_dims_ok(d) || !w_src || !w_packed || (mode != 0 && mode != 1)) return SSBEV_EINVAL;
  if (ssbev_bf16::storage_mode(d)) return ssbev_bf16::pack(w_src, w_packed, d, mode, as_stream(stream));
  if (ssbev_thin::thinin_applicable(d, mode)) return ssbev_thin::thinin_pack(w_src, w_packed, d, mode, as_stream(stream));
  if (conv_thin_applicable(d, mode)) {       // <= 4 output channels: LDS-resident [tap][n][k] table (conv_thin_kernel)
    hipLaunchKernelGGL(pack_thin_kernel, dim3(cdiv(27 * kThinNP * 32, 256)), dim3(256), 0, as_stream(stream), w_src,
                       w_packed, d->Cout, d->Cin, mode);
    return ssbev_launch_status();
  }
  if (conv_tap2_applicable(d, mode)) {       // stride-2 "down" gather: [N][K][27] in both roles, see pack_tap2_kernel
    const int K = mode == 0 ? d->Cin : d->Cout, N = mode == 0 ? d->Cout : d->Cin;
    hipLaunchKernelGGL(pack_tap2_kernel, dim3(cdiv(kT2PackedElems, 256)), dim3(256), 0, as_stream(stream), w_src, w_packed, N, K);
    return ssbev_launch_status();
  }
  if (conv_tap2up_applicable(d, mode)) {     // stride-2 "up" gather: [K][N][27] in both roles, see pack_tap2up_kernel
    const int K = mode == 0 ? d->Cin : d->Cout, N = mode == 0 ? d->Cout : d->Cin;
    hipLaunchKernelGGL(pack_tap2up_kernel, dim3(cdiv(kUpPackedElems, 256)), dim3(256), 0, as_stream(stream), w_src, w_packed, N, K);
    return ssbev_launch_status();
  }
  if (conv_pw32_applicable(d, mode)) {
    hipLaunchKernelGGL(pack_pw32_kernel, dim3(cdiv(kPwPackedElems, 256)), dim3(256), 0, as_stream(stream), w_src, w_packed,
                       d->Cout, d->Cin, mode);
    return ssbev_launch_status();
  }
  if (conv_tapdh_applicable(d, mode)) {      // Winograd along d and h: U = G w G^T per kw
    hipLaunchKernelGGL(pack_tapdh_kernel, dim3(cdiv(kDhPackedElems, 256)), dim3(256), 0, as_stream(stream), w_src, w_packed,
                       d->Cout, d->Cin, mode);
    return ssbev_launch_status();
  }
  if (conv_taph_applicable(d, mode)) {       // Winograd-along-h variant of the tap kernel: U = G w per (kd, kw)
    hipLaunchKernelGGL(pack_taph_kernel, dim3(cdiv(kTwPackedElems, 256)), dim3(256), 0, as_stream(stream), w_src, w_packed,
                       d->Cout, d->Cin, mode);
    return ssbev_launch_status();
  }
  if (conv_tap_applicable(d, mode)) {        // register-resident tap-split layout (see conv_tap_kernel)
    hipLaunchKernelGGL(pack_tap_kernel, dim3(cdiv(kTapPackedElems, 256)), dim3(256), 0, as_stream(stream), w_src, w_packed,
                       d->Cout, d->Cin, mode);
    return ssbev_launch_status();
  }
  const int taps = d->kd * d->kh * d->kw;
  // forward: K = Cin, N = Cout; data gradient: K = Cout, N = Cin
  const int K = mode == 0 ? d->Cin : d->Cout, N = mode == 0 ? d->Cout : d->Cin;
  // torch layout: conv [Cout,Cin,k], deconv [Cin,Cout,k]
  //   fwd conv  : K=A1 (layout 0)   fwd deconv: K=A0 (layout 1)
  //   bwd conv  : K=A0 (layout 1)   bwd deconv: K=A1 (layout 0)
  const int layout = (mode == 0) == (d->transposed == 0) ? 0 : 1;
  const int KPad = pad8(K), NPad = pad32(N);
  const long total = (long)taps * KPad * NPad;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), w_src, w_packed, K,
                     N, KPad, NPad, taps, layout, total);
  return ssbev_launch_status();
}

// bf16-STORAGE twins (ssbev_conv_dims.precision = 2 / 3, csrc/conv_bf16.hip): the activation tensors are bf16 bit patterns and the
// signatures say so; a call whose `precision` does not match its entry point is SSBEV_EINVAL on either side (round 4 passed the
// bf16 tensors through the float* entry points: a wrong `precision` value was silent garbage).
int ssbev_conv_fwd_bf16(const uint16_t* x, const float* w_packed, const float* bias, void* y, const ssbev_conv_dims* d,
                        ssbev_stream_t stream) {
  if (!conv_dims_ok(d) || !ssbev_bf16::storage_mode(d) || !x || !w_packed || !y) return SSBEV_EINVAL;
  return ssbev_bf16::forward(x, w_packed, bias, y, d, as_stream(stream));
}

int ssbev_conv_bwd_data_bf16(const uint16_t* gy, const float* w_packed_t, void* gx, const ssbev_conv_dims* d,
                             ssbev_stream_t stream) {
  if (!conv_dims_ok(d) || !ssbev_bf16::storage_mode(d) || !gy || !w_packed_t || !gx) return SSBEV_EINVAL;
  return ssbev_bf16::backward_data(gy, w_packed_t, gx, d, as_stream(stream));
}

int ssbev_conv_bwd_weight_bf16(const uint16_t* x, const uint16_t* gy, float* gw, const ssbev_conv_dims* d, void* ws,
                               size_t ws_bytes, ssbev_stream_t stream) {
  if (!conv_dims_ok(d) || d->precision != 2 || !x || !gy || !gw || !ws) return SSBEV_EINVAL;
  return ssbev_bf16::backward_weight(x, gy, gw, d, ws, ws_bytes, as_stream(stream));
}

int ssbev_conv_fwd(const float* x, const float* w_packed, const float* bias, float* y,
                   const ssbev_conv_dims* d, ssbev_stream_t stream) {
  if (!conv_dims_ok(d) || !x || !w_packed || !y) return SSBEV_EINVAL;
  if (ssbev_bf16::storage_mode(d)) return SSBEV_EINVAL;      // bf16 tensors go through ssbev_conv_fwd_bf16 (typed pointers)
  if (ssbev_thin::thinin_applicable(d, 0)) return ssbev_thin::thinin_launch(x, w_packed, bias, y, d, 0, as_stream(stream));
  if (d->Cin % 4 != 0) return SSBEV_EINVAL;   // float4 channel loads: caller pads K channels to 4
  if (conv_thin_applicable(d, 0)) return launch_conv_thin(x, w_packed, bias, y, d, 0, as_stream(stream));
  if (conv_tap2_applicable(d, 0)) return launch_conv_tap2(x, w_packed, bias, y, d, 0, as_stream(stream));
  if (conv_tap2up_applicable(d, 0)) return launch_conv_tap2up(x, w_packed, bias, y, d, 0, as_stream(stream));
  if (conv_pw32_applicable(d, 0)) return launch_conv_pw32(x, w_packed, bias, y, d, 0, as_stream(stream));
  if (conv_tapdh_applicable(d, 0)) return launch_conv_tapdh(x, w_packed, bias, y, d, 0, as_stream(stream));
  if (conv_taph_applicable(d, 0)) return launch_conv_taph(x, w_packed, bias, y, d, 0, as_stream(stream));
  if (conv_tap_applicable(d, 0)) return launch_conv_tap(x, w_packed, bias, y, d, 0, as_stream(stream));
  ConvGeom g;
  g.B = d->B; g.Cin = d->Cin; g.Cout = d->Cout; g.CinPad = pad8(d->Cin); g.CoutPad = pad32(d->Cout);
  g.Di = d->Di; g.Hi = d->Hi; g.Wi = d->Wi; g.Do = d->Do; g.Ho = d->Ho; g.Wo = d->Wo;
  g.kd = d->kd; g.kh = d->kh; g.kw = d->kw; g.sd = d->sd; g.sh = d->sh; g.sw = d->sw;
  g.pd = d->pd; g.ph = d->ph; g.pw = d->pw; g.dd = d->dd; g.dh = d->dh; g.dw = d->dw;
  g.form = d->transposed ? 1 : 0; g.relu = d->relu; g.accumulate = d->accumulate;
  g.hint = d->tile_hint >= 10 ? d->tile_hint : 0;       // hints below 10 select other kernel families
  g.chunk_taps = 0; g.bf16 = d->precision == 1;
  return dispatch_gather(x, w_packed, bias, y, g, as_stream(stream));
}

int ssbev_conv_bwd_data(const float* gy, const float* w_packed_t, float* gx,
                        const ssbev_conv_dims* d, ssbev_stream_t stream) {
  if (!conv_dims_ok(d) || !gy || !w_packed_t || !gx) return SSBEV_EINVAL;
  if (ssbev_bf16::storage_mode(d)) return SSBEV_EINVAL;      // -> ssbev_conv_bwd_data_bf16
  if (ssbev_thin::thinin_applicable(d, 1)) return ssbev_thin::thinin_launch(gy, w_packed_t, nullptr, gx, d, 1, as_stream(stream));
  if (conv_thin_applicable(d, 1)) return launch_conv_thin(gy, w_packed_t, nullptr, gx, d, 1, as_stream(stream));
  if (conv_tap2_applicable(d, 1)) return launch_conv_tap2(gy, w_packed_t, nullptr, gx, d, 1, as_stream(stream));
  if (conv_tap2up_applicable(d, 1)) return launch_conv_tap2up(gy, w_packed_t, nullptr, gx, d, 1, as_stream(stream));
  if (conv_pw32_applicable(d, 1)) return launch_conv_pw32(gy, w_packed_t, nullptr, gx, d, 1, as_stream(stream));
  if (conv_tapdh_applicable(d, 1)) return launch_conv_tapdh(gy, w_packed_t, nullptr, gx, d, 1, as_stream(stream));
  if (conv_taph_applicable(d, 1)) return launch_conv_taph(gy, w_packed_t, nullptr, gx, d, 1, as_stream(stream));
  if (conv_tap_applicable(d, 1)) return launch_conv_tap(gy, w_packed_t, nullptr, gx, d, 1, as_stream(stream));
  ConvGeom g;   // roles swapped: source grid = forward output grid, K = Cout, N = Cin
  g.B = d->B; g.Cin = d->Cout; g.Cout = d->Cin; g.CinPad = pad8(d->Cout); g.CoutPad = pad32(d->Cin);
  g.Di = d->Do; g.Hi = d->Ho; g.Wi = d->Wo; g.Do = d->Di; g.Ho = d->Hi; g.Wo = d->Wi;
  g.kd = d->kd; g.kh = d->kh; g.kw = d->kw; g.sd = d->sd; g.sh = d->sh; g.sw = d->sw;
  g.pd = d->pd; g.ph = d->ph; g.pw = d->pw; g.dd = d->dd; g.dh = d->dh; g.dw = d->dw;
  g.form = d->transposed ? 0 : 1;   // grad of a conv gathers like a deconv and vice versa
  g.relu = 0; g.accumulate = d->accumulate; g.hint = d->tile_hint >= 10 ? d->tile_hint : 0; g.chunk_taps = 0;
  g.bf16 = d->precision == 1;
  if (g.Cin % 4 != 0) return SSBEV_EINVAL;
  return dispatch_gather(gy, w_packed_t, nullptr, gx, g, as_stream(stream));
}

size_t ssbev_conv_bwd_weight_workspace(const ssbev_conv_dims* d) {
  if (!conv_dims_ok(d)) return 0;
  if (ssbev_bf16::storage_mode(d)) return ssbev_bf16::wgrad_workspace(d);
  if (ssbev_thin::wgrad_applicable(d)) return ssbev_thin::wgrad_workspace(d);
  {
    const WgradThinPlan tp = plan_wgrad_thin(d);
    if (tp.ok) return align256b((size_t)tp.nchunks * 27 * d->Cin * d->Cout * sizeof(float));
  }
  {
    const Wgrad1x1Plan p1 = plan_wgrad_1x1(d);
    if (p1.ok && d->tile_hint != 7) return align256b((size_t)p1.nchunks * p1.Cq * p1.Cp * sizeof(float));
  }
  {
    const WgradDhPlan hp = plan_wgrad_dh(d);
    if (hp.ok) return wgrad_dh_workspace(hp);
  }
  {
    const WgradLdsPlan lp = plan_wgrad_lds(d);
    if (lp.ok && d->tile_hint != 7)
      return align256b((size_t)lp.nchunks * lp.ksplit * d->kd * 9 * d->Cout * d->Cin * sizeof(float));
  }
  if (wgrad_cf_applicable(d)) {
    WgradCfg c;
    const WgradCfGeom g = make_wgrad_cf_geom(d, &c);
    const size_t N = (size_t)g.B * g.D * g.H * g.W;
    return align256b((size_t)g.nchunks * d->kd * d->kh * d->kw * g.Cp * g.Cq * sizeof(float)) +
           align256b((size_t)g.Cp * N * sizeof(float)) +
           align256b(((size_t)g.Cq * g.B * g.Dp * g.Hp * g.Wp + 16) * sizeof(float));
  }
  const WgradGeom g = make_wgrad_geom(d);
  return (size_t)g.nchunks * d->kd * d->kh * d->kw * g.Cp * g.Cq * sizeof(float);
}

int ssbev_conv_bwd_weight(const float* x, const float* gy, float* gw, const ssbev_conv_dims* d,
                          void* ws, size_t ws_bytes, ssbev_stream_t stream) {
  if (!conv_dims_ok(d) || !x || !gy || !gw || !ws) return SSBEV_EINVAL;
  if (ssbev_bf16::storage_mode(d)) return SSBEV_EINVAL;      // -> ssbev_conv_bwd_weight_bf16
  if (ws_bytes < ssbev_conv_bwd_weight_workspace(d)) return SSBEV_EWORKSPACE;
  if (ssbev_thin::wgrad_applicable(d)) return ssbev_thin::wgrad_launch(x, gy, gw, d, ws, ws_bytes, as_stream(stream));
  {
    const WgradThinPlan tp = plan_wgrad_thin(d);
    if (tp.ok) {                                 // 32 -> (<= 4) heads: VALU reduction over the LDS ring
      hipStream_t st = as_stream(stream);
      float* partial = static_cast<float*>(ws);
      const size_t lds = (size_t)(kTapRingF + 2 * kTapWseg * kThinNP) * sizeof(float);
      hipLaunchKernelGGL(wgrad_thin_kernel, dim3(tp.nchunks), dim3(256), lds, st, x, gy, partial, tp.g);
      launch_wgrad_reduce(partial, gw, tp.nchunks, 27, d->Cin, d->Cout, st);
      return ssbev_launch_status();
    }
  }
  {
    const Wgrad1x1Plan p1 = plan_wgrad_1x1(d);
    if (p1.ok && d->tile_hint != 7) {
      hipStream_t st = as_stream(stream);
      const float* P = d->transposed ? x : gy;
      const float* Q = d->transposed ? gy : x;
      float* partial = static_cast<float*>(ws);
      dim3 grid(p1.nchunks, cdiv(p1.Cq, 32 * p1.MQ) * cdiv(p1.Cp, 32 * p1.MP)), block(1024);
      if (p1.MQ == 2 && p1.MP == 2)
        hipLaunchKernelGGL((wgrad_1x1_kernel<2, 2>), grid, block, 0, st, P, Q, partial, p1.N, p1.Cq, p1.Cp, p1.chunk, p1.nchunks);
      else if (p1.MQ == 2)
        hipLaunchKernelGGL((wgrad_1x1_kernel<2, 1>), grid, block, 0, st, P, Q, partial, p1.N, p1.Cq, p1.Cp, p1.chunk, p1.nchunks);
      else if (p1.MP == 2)
        hipLaunchKernelGGL((wgrad_1x1_kernel<1, 2>), grid, block, 0, st, P, Q, partial, p1.N, p1.Cq, p1.Cp, p1.chunk, p1.nchunks);
      else
        hipLaunchKernelGGL((wgrad_1x1_kernel<1, 1>), grid, block, 0, st, P, Q, partial, p1.N, p1.Cq, p1.Cp, p1.chunk, p1.nchunks);
      launch_wgrad_reduce(partial, gw, p1.nchunks, 1, p1.Cq, p1.Cp, st);
      return ssbev_launch_status();
    }
  }
  {
    const WgradDhPlan hp = plan_wgrad_dh(d);   // <= 32 x 32 channels, even D and H: F(2,3) along d and h (round 4)
    if (hp.ok) return run_wgrad_dh(x, gy, gw, hp, ws, as_stream(stream));
  }
  {
    const WgradLdsPlan lp = plan_wgrad_lds(d);
    if (lp.ok && d->tile_hint != 7) {          // tile_hint 7: force the channel-major path (tests / A-B timing)
      hipStream_t st = as_stream(stream);
      float* partial = static_cast<float*>(ws);
      const int rc = run_wgrad_lds(x, gy, partial, d, lp, st);
      if (rc != SSBEV_OK) return rc;
      launch_wgrad_reduce(partial, gw, lp.nchunks * lp.ksplit, d->kd * 9, lp.g.Cq, lp.g.Cp, st);
      return ssbev_launch_status();
    }
  }
  if (wgrad_cf_applicable(d)) {
    hipStream_t st = as_stream(stream);
    WgradCfg c;
    const WgradCfGeom g = make_wgrad_cf_geom(d, &c);
    const size_t N = (size_t)g.B * g.D * g.H * g.W;
    const int taps = g.kd * g.kh * g.kw;
    char* base = static_cast<char*>(ws);
    float* partial = reinterpret_cast<float*>(base);
    float* Pt = reinterpret_cast<float*>(base + align256b((size_t)g.nchunks * taps * g.Cp * g.Cq * sizeof(float)));
    float* Qp = reinterpret_cast<float*>(reinterpret_cast<char*>(Pt) + align256b((size_t)g.Cp * N * sizeof(float)));
    const size_t qbytes = ((size_t)g.Cq * g.B * g.Dp * g.Hp * g.Wp + 16) * sizeof(float);
    if (hipMemsetAsync(Qp, 0, qbytes, st) != hipSuccess) return SSBEV_ELAUNCH;
    hipLaunchKernelGGL(transpose_pad_kernel, dim3(cdiv(N, 64), cdiv(g.Cp, 32)), dim3(256), 0, st, gy, Pt, g.Cp, g.B,
                       g.D, g.H, g.W, 0, 0, 0, g.D, g.H, g.W);
    hipLaunchKernelGGL(transpose_pad_kernel, dim3(cdiv(N, 64), cdiv(g.Cq, 32)), dim3(256), 0, st, x, Qp, g.Cq, g.B,
                       g.D, g.H, g.W, d->pd, d->ph, d->pw, g.Dp, g.Hp, g.Wp);
    if (c.TH == 1 && c.TW == 1) launch_wgrad_cf<2, 2, 1, 1>(Pt, Qp, partial, g, st);
    else if (c.MQ == 2) launch_wgrad_cf<2, 2, 1, 3>(Pt, Qp, partial, g, st);
    else if (c.TH == 3) launch_wgrad_cf<1, 1, 3, 3>(Pt, Qp, partial, g, st);
    else launch_wgrad_cf<1, 1, 1, 3>(Pt, Qp, partial, g, st);
    launch_wgrad_reduce(partial, gw, g.nchunks, taps, g.Cq, g.Cp, st);
    return ssbev_launch_status();
  }
  const WgradGeom g = make_wgrad_geom(d);
  const float* P = d->transposed ? x : gy;
  const float* Qt = d->transposed ? gy : x;
  hipStream_t st = as_stream(stream);
  const int taps = g.kd * g.kh * g.kw;
  if ((long)g.B * g.Ds * g.Hs * g.Ws >= (1L << 31)) return SSBEV_EINVAL;
  const WgradCfg c = wgrad_cfg(g.Cp, g.Cq, g.kh, g.kw);
  float* wsf = static_cast<float*>(ws);
  if (c.TH == 1 && c.TW == 1) launch_wgrad<2, 2, 1, 1>(P, Qt, wsf, g, st);
  else if (c.MQ == 2) launch_wgrad<2, 2, 1, 3>(P, Qt, wsf, g, st);
  else if (c.TH == 3) launch_wgrad<1, 1, 3, 3>(P, Qt, wsf, g, st);
  else launch_wgrad<1, 1, 1, 3>(P, Qt, wsf, g, st);
  launch_wgrad_reduce(wsf, gw, g.nchunks, taps, g.Cq, g.Cp, st);
  return ssbev_launch_status();
}

}  // extern "C"
