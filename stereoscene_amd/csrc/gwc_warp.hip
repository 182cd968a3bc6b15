// Stereo plane-sweep cost volume (SURVEY a2+a3), gfx950: group-wise correlation fused with the
// disparity -> metric-depth bilinear resample of the reference's `warp` (VT:104-156).
//
//   cost[g', d](w) = 1/cpg * sum_{c in g'} L[w, c] * R[w - d, c]          (0 for w < d)
//   vol[k, g](w)   = sum_{ty in {y0,y0+1}} sum_{tx in {x0,x0+1}} wy*wx * cost[ty, tx](w)
// with (x0, wx) from ix(k) = unnormalise(2*(calib/(4*down)/(k+1))/(D-1) - 1) and (y0, wy) from
// iy(g) = unnormalise(2*g/(G-1) - 1): exactly grid_sample's arithmetic, both conventions.
//
// MI355X mapping: one workgroup per (batch, image row, depth-plane chunk).  The right-view row
// (W x C fp32, 40 KB at 160x64) is staged ONCE into LDS with a +4-float row pad (conflict-free
// 8-byte reads); the left-view operands live in registers for the whole chunk; every plane is
// written as W*G contiguous floats (channels-last volume) => the kernel is a pure streaming
// write of the 189 MB volume (D=192), its only HBM-sized traffic.  The 2-channel group
// reduction happens inside one lane (no cross-lane traffic needed for cpg=2..8).
//
// What makes it a streaming kernel (r2): the disparity tap x0(k) = floor(ix(k)) ~ calib/4/(k+1) takes only ~25 distinct
// values over the 192 metric depths (x0 = 0 for every k >= 98, x0 = 1 for 49..97, ...), so the raw correlations
// cost[.][x0], cost[.][x0+1] of a (pixel, group) are computed ONCE per run of planes with the same x0 and every plane of
// the run is just out = wx0(k) * m0 + wx1(k) * m1: two VALU ops and one 16-byte store per four outputs (a lane owns four
// consecutive groups, a wave stores 1 KB per instruction).  The per-plane LDS reads and the 2 x 2 x CPG multiply-adds of
// the first version (113 us, 1.7 TB/s) only happen at the ~40 run starts.  Plane chunks are short where x0 changes at
// every plane (k < 16) and long behind, so that the workgroups carry equal work.
#include "common.h"

#include <algorithm>
#include <cstdlib>

namespace {

constexpr int KCHUNK = 16;

struct XTap { int x0; float w0, w1; };

// grid_sample coordinate un-normalisation (ATen grid_sampler_unnormalize)
__device__ __forceinline__ float unnormalise(float coord, int size, int align_corners) {
  return align_corners ? __fmul_rn(__fdiv_rn(__fadd_rn(coord, 1.0f), 2.0f), (float)(size - 1))
                       : __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(coord, 1.0f), (float)size), 1.0f), 2.0f);
}

__device__ __forceinline__ XTap depth_tap(float calib, int k, int D, float down, int align_corners) {
  const float xx = __fdiv_rn(__fdiv_rn(calib, __fmul_rn(down, 4.0f)), (float)(k + 1));
  const float gx = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, xx), (float)max(D - 1, 1)), 1.0f);
  const float ix = unnormalise(gx, D, align_corners);
  const float fl = floorf(ix);
  XTap t;
  // clamp before the int cast: only taps inside [-1, D] can ever contribute
  t.x0 = (int)fminf(fmaxf(fl, -2.0f), (float)D + 1.0f);
  t.w1 = __fsub_rn(ix, fl);
  t.w0 = __fsub_rn(1.0f, t.w1);
  if (!(ix == ix)) { t.x0 = -2; t.w0 = t.w1 = 0.0f; }
  return t;
}

__device__ __forceinline__ void group_tap(int g, int G, int align_corners, int* y0, float* w0, float* w1) {
  const float gy = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, (float)g), (float)max(G - 1, 1)), 1.0f);
  const float iy = unnormalise(gy, G, align_corners);
  const float fl = floorf(iy);
  *y0 = (int)fl;
  *w1 = __fsub_rn(iy, fl);
  *w0 = __fsub_rn(1.0f, *w1);
}

// stage one [W, C] row into LDS with padded stride
__device__ __forceinline__ void stage_row(const float* __restrict__ src, float* lds, int W, int C, int stride) {
  const int vec_per_row = C >> 2;
  for (int i = threadIdx.x; i < W * vec_per_row; i += blockDim.x) {
    const int w = i / vec_per_row, v = i - w * vec_per_row;
    const float4 t = reinterpret_cast<const float4*>(src)[i];
    *reinterpret_cast<float4*>(lds + w * stride + 4 * v) = t;
  }
}

template <int CPG>
__global__ void __launch_bounds__(256)
gwc_warp_fwd_kernel(const float* __restrict__ left, const float* __restrict__ right,
                    const float* __restrict__ calib, float* __restrict__ vol, int B, int C, int G, int D, int H,
                    int W, float down, int align_corners) {
  extern __shared__ __align__(16) float lds[];
  const int stride = C + 4;
  float* Rrow = lds;                                   // [W][stride]
  XTap* taps = reinterpret_cast<XTap*>(lds + W * stride);  // [KCHUNK]
  const int bh = blockIdx.x;
  const int b = bh / H, h = bh - b * H;
  const int k_begin = blockIdx.y * KCHUNK;
  const int k_end = min(D, k_begin + KCHUNK);
  const float* Lrow_g = left + ((size_t)b * H + h) * W * C;
  stage_row(right + ((size_t)b * H + h) * W * C, Rrow, W, C, stride);
  if (threadIdx.x < KCHUNK && k_begin + threadIdx.x < D)
    taps[threadIdx.x] = depth_tap(calib[b], k_begin + threadIdx.x, D, down, align_corners);
  __syncthreads();

  const float inv_cpg = 1.0f / (float)CPG;
  for (int item = threadIdx.x; item < W * G; item += blockDim.x) {
    const int w = item / G, g = item - w * G;
    int y0;
    float wy0, wy1;
    group_tap(g, G, align_corners, &y0, &wy0, &wy1);
    const bool ok0 = y0 >= 0 && y0 < G, ok1 = y0 + 1 >= 0 && y0 + 1 < G;
    float l0[CPG], l1[CPG];
#pragma unroll
    for (int c = 0; c < CPG; ++c) {
      l0[c] = ok0 ? Lrow_g[w * C + y0 * CPG + c] : 0.0f;
      l1[c] = ok1 ? Lrow_g[w * C + (y0 + 1) * CPG + c] : 0.0f;
    }
    const int cy0 = ok0 ? y0 * CPG : 0, cy1 = ok1 ? (y0 + 1) * CPG : 0;
    float* dst = vol + ((((size_t)b * D + k_begin) * H + h) * W + w) * G + g;
    const size_t plane = (size_t)H * W * G;
    for (int k = k_begin; k < k_end; ++k) {
      const XTap t = taps[k - k_begin];
      float acc = 0.0f;
#pragma unroll
      for (int tx = 0; tx < 2; ++tx) {
        const int d = t.x0 + tx;
        const float wx = tx ? t.w1 : t.w0;
        if (d >= 0 && d < D && w >= d) {   // w >= d: the reference volume is zero left of the disparity
          const float* r = Rrow + (w - d) * stride;
          float c0 = 0.0f, c1 = 0.0f;
#pragma unroll
          for (int c = 0; c < CPG; ++c) {
            c0 += l0[c] * r[cy0 + c];
            c1 += l1[c] * r[cy1 + c];
          }
          acc += (wy0 * wx) * (c0 * inv_cpg) + (wy1 * wx) * (c1 * inv_cpg);
        }
      }
      *dst = acc;
      dst += plane;
    }
  }
}

// ---- forward, four groups per lane --------------------------------------------------------------------------------
constexpr int MAX_CHUNKS = 48;
struct PlaneChunks { int n; int start[MAX_CHUNKS + 1]; };   // chunk c covers planes [start[c], start[c+1])
constexpr int FWD_MAXLEN = 32;                              // longest chunk (taps staged in LDS)

typedef float v4f __attribute__((ext_vector_type(4)));

template <int CPG, bool NT, bool STAGE>
__global__ void __launch_bounds__(512)
gwc_warp_fwd4_kernel(const float* __restrict__ left, const float* __restrict__ right,
                     const float* __restrict__ calib, float* __restrict__ vol, PlaneChunks chunks, int B, int C, int G,
                     int D, int H, int W, float down, int align_corners, int rows_per_wg) {
  extern __shared__ __align__(16) float lds[];
  // STAGE: one row per workgroup, the right-view row staged in LDS (+4-float pad).  !STAGE: rows_per_wg consecutive rows
  // per workgroup, the right view read straight from L2 at the (rare) run starts -- no staging, no LDS footprint, and
  // rows_per_wg * W * G contiguous floats written per plane.
  const int stride = STAGE ? C + 4 : C;
  XTap* taps = reinterpret_cast<XTap*>(STAGE ? lds + W * stride : lds);  // [FWD_MAXLEN]
  const int row0 = blockIdx.x * rows_per_wg;                             // first (b, h) row of this workgroup
  const int nrows = min(rows_per_wg, B * H - row0);
  const int k_begin = chunks.start[blockIdx.y], k_end = chunks.start[blockIdx.y + 1];
  if (STAGE) stage_row(right + (size_t)row0 * W * C, lds, W, C, stride);
  // calib is per batch element: a workgroup's rows must share it (rows_per_wg divides H, checked by the launcher)
  if ((int)threadIdx.x < k_end - k_begin)
    taps[threadIdx.x] = depth_tap(calib[row0 / H], k_begin + threadIdx.x, D, down, align_corners);
  __syncthreads();

  const float inv_cpg = 1.0f / (float)CPG;
  const int G4 = G >> 2;
  const size_t plane = (size_t)H * W * G;
  // blockDim.x is a multiple of G4 (launcher): a thread keeps its four groups for all of its pixels, so the group-axis
  // taps (IEEE divisions) are computed once per thread instead of once per item
  const int g4 = (threadIdx.x % G4) << 2, ppp = blockDim.x / G4;      // pixels per pass of the workgroup
  float wy[4][2];
  int cy[4][2];
  bool okg[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int y0;
    float w0, w1;
    group_tap(g4 + j, G, align_corners, &y0, &w0, &w1);
    okg[j][0] = y0 >= 0 && y0 < G;
    okg[j][1] = y0 + 1 >= 0 && y0 + 1 < G;
    cy[j][0] = okg[j][0] ? y0 * CPG : 0;
    cy[j][1] = okg[j][1] ? (y0 + 1) * CPG : 0;
    wy[j][0] = okg[j][0] ? w0 : 0.0f;
    wy[j][1] = okg[j][1] ? w1 : 0.0f;
  }
  for (int pix = threadIdx.x / G4; pix < nrows * W; pix += ppp) {
    const int rr = pix / W, w = pix - rr * W;
    const int row = row0 + rr, b = row / H, h = row - b * H;
    const float* Lrow_g = left + (size_t)row * W * C;
    const float* Rrow = STAGE ? lds : right + (size_t)row * W * C;
    float l[4][2][CPG];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int c = 0; c < CPG; ++c) {
        l[j][0][c] = okg[j][0] ? Lrow_g[w * C + cy[j][0] + c] : 0.0f;
        l[j][1][c] = okg[j][1] ? Lrow_g[w * C + cy[j][1] + c] : 0.0f;
      }
    }
    float m[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j][0] = m[j][1] = 0.0f;
    int cur = INT32_MIN;
    float* dst = vol + ((((size_t)b * D + k_begin) * H + h) * W + w) * G + g4;
    for (int k = k_begin; k < k_end; ++k) {
      const XTap t = taps[k - k_begin];
      if (t.x0 != cur) {          // wave-uniform: a new run of planes that share their two disparity taps
        cur = t.x0;
#pragma unroll
        for (int tx = 0; tx < 2; ++tx) {
          const int d = cur + tx;
          const bool valid = d >= 0 && d < D && w >= d;   // w >= d: the reference volume is zero left of the disparity
          const float* r = Rrow + (valid ? (w - d) : 0) * stride;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float c0 = 0.0f, c1 = 0.0f;
#pragma unroll
            for (int c = 0; c < CPG; ++c) {
              c0 += l[j][0][c] * r[cy[j][0] + c];
              c1 += l[j][1][c] * r[cy[j][1] + c];
            }
            m[j][tx] = valid ? wy[j][0] * (c0 * inv_cpg) + wy[j][1] * (c1 * inv_cpg) : 0.0f;
          }
        }
      }
      v4f o;
      o.x = t.w0 * m[0][0] + t.w1 * m[0][1];
      o.y = t.w0 * m[1][0] + t.w1 * m[1][1];
      o.z = t.w0 * m[2][0] + t.w1 * m[2][1];
      o.w = t.w0 * m[3][0] + t.w1 * m[3][1];
      if (NT) __builtin_nontemporal_store(o, reinterpret_cast<v4f*>(dst));   // streaming write: keep it out of L2 / MALL
      else *reinterpret_cast<v4f*>(dst) = o;
      dst += plane;
    }
  }
}

// ---- forward, one pixel per thread (round 3) ---------------------------------------------------------------------------
// fwd4 walks its row in passes of blockDim / (G/4) pixels and loads the left-view operands of every pass from global
// memory in front of that pass's stores: five load latencies per workgroup, each queued BEHIND the store stream the same
// workgroups keep full (17 of its 47 us; the store pattern alone runs at 30 us, tools/micro/store_pattern.hip).  Here a
// workgroup owns (row, tile of blockDim / (G/4) pixels, plane chunk) and a thread ONE (pixel, group quad): the left-view
// operands are loaded once, before anything is stored, next to the staging of the right-view pixels the chunk can reach
// ([w0 - dmax(chunk), w0 + PX): 10 KB for the far chunks instead of the 40 KB row), and the plane loop is stores and
// multiply-adds only, with the next plane's tap read one iteration ahead.
template <int CPG, bool NT>
__global__ void __launch_bounds__(512)
gwc_warp_fwd5_kernel(const float* __restrict__ left, const float* __restrict__ right,
                     const float* __restrict__ calib, float* __restrict__ vol, PlaneChunks chunks, int B, int C, int G,
                     int D, int H, int W, float down, int align_corners, int ntiles) {
  extern __shared__ __align__(16) float lds[];
  const int stride = C + 4;
  XTap* taps = reinterpret_cast<XTap*>(lds + W * stride);               // [chunk length <= D]
  const int G4 = G >> 2, PX = blockDim.x / G4;
  const int row = blockIdx.x / ntiles, tile = blockIdx.x - row * ntiles;
  const int b = row / H, h = row - b * H;
  const int w0 = tile * PX;
  const int k_begin = chunks.start[blockIdx.y], k_end = chunks.start[blockIdx.y + 1];
  const int g4 = (threadIdx.x % G4) << 2, w = w0 + threadIdx.x / G4;
  const bool active = w < W;
  const float inv_cpg = 1.0f / (float)CPG;

  float wy[4][2];
  int cy[4][2];
  bool okg[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int y0;
    float a0, a1;
    group_tap(g4 + j, G, align_corners, &y0, &a0, &a1);
    okg[j][0] = y0 >= 0 && y0 < G;
    okg[j][1] = y0 + 1 >= 0 && y0 + 1 < G;
    cy[j][0] = okg[j][0] ? y0 * CPG : 0;
    cy[j][1] = okg[j][1] ? (y0 + 1) * CPG : 0;
    wy[j][0] = okg[j][0] ? a0 : 0.0f;
    wy[j][1] = okg[j][1] ? a1 : 0.0f;
  }
  // left-view operands: issued first, they land while the right-view pixels are staged
  const float* Lpix = left + ((size_t)row * W + (active ? w : 0)) * C;
  float l[4][2][CPG];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int c = 0; c < CPG; ++c) {
      l[j][0][c] = Lpix[cy[j][0] + c];
      l[j][1][c] = Lpix[cy[j][1] + c];
    }
  // right-view pixels this chunk can reach: x0(k) is monotone in k, so its extremes sit at the chunk's ends
  const float cal = calib[b];
  const XTap ta = depth_tap(cal, k_begin, D, down, align_corners), tb = depth_tap(cal, k_end - 1, D, down, align_corners);
  const int d_hi = min(max(max(ta.x0, tb.x0) + 1, 0), W);
  const int lo = max(0, w0 - d_hi), hi = min(W, w0 + PX);
  {
    const int vec_per_px = C >> 2;
    const float4* src = reinterpret_cast<const float4*>(right + ((size_t)row * W + lo) * C);
    for (int i = threadIdx.x; i < (hi - lo) * vec_per_px; i += blockDim.x) {
      const int p = i / vec_per_px, v = i - p * vec_per_px;
      *reinterpret_cast<float4*>(lds + p * stride + 4 * v) = src[i];
    }
  }
  for (int i = threadIdx.x; i < k_end - k_begin; i += blockDim.x)
    taps[i] = depth_tap(cal, k_begin + i, D, down, align_corners);
  __syncthreads();
  if (!active) return;

  float m[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j) m[j][0] = m[j][1] = 0.0f;
  int cur = INT32_MIN;
  const size_t plane = (size_t)H * W * G;
  float* dst = vol + ((((size_t)b * D + k_begin) * H + h) * W + w) * G + g4;
  XTap t = taps[0];
  for (int k = k_begin; k < k_end; ++k) {
    const XTap tn = taps[min(k + 1, k_end - 1) - k_begin];   // next plane's tap: its LDS latency hides behind this plane
    if (t.x0 != cur) {            // wave-uniform: a new run of planes that share their two disparity taps
      cur = t.x0;
#pragma unroll
      for (int tx = 0; tx < 2; ++tx) {
        const int d = cur + tx;
        const bool valid = d >= 0 && d < D && w >= d;   // w >= d: the reference volume is zero left of the disparity
        const float* r = lds + (valid ? (w - d - lo) : 0) * stride;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float c0 = 0.0f, c1 = 0.0f;
#pragma unroll
          for (int c = 0; c < CPG; ++c) {
            c0 += l[j][0][c] * r[cy[j][0] + c];
            c1 += l[j][1][c] * r[cy[j][1] + c];
          }
          m[j][tx] = valid ? wy[j][0] * (c0 * inv_cpg) + wy[j][1] * (c1 * inv_cpg) : 0.0f;
        }
      }
    }
    v4f o;
    o.x = t.w0 * m[0][0] + t.w1 * m[0][1];
    o.y = t.w0 * m[1][0] + t.w1 * m[1][1];
    o.z = t.w0 * m[2][0] + t.w1 * m[2][1];
    o.w = t.w0 * m[3][0] + t.w1 * m[3][1];
    if (NT) __builtin_nontemporal_store(o, reinterpret_cast<v4f*>(dst));
    else *reinterpret_cast<v4f*>(dst) = o;
    dst += plane;
    t = tn;
  }
}

// Backward.  For a source group g' the output groups that read it are those g with y0(g) == g'
// (weight wy0(g)) or y0(g)+1 == g' (weight wy1(g)); y0 is monotone in g so there are at most a few.
//   S(k, w)      = sum_{(g, wy)} wy * gvol[k, w, g]
//   gL[w , c]    = 1/cpg * sum_k sum_tx wx * S(k, w)       * R[w - d, c]      d = x0(k)+tx, w >= d
//   gR[w', c]    = 1/cpg * sum_k sum_tx wx * S(k, w' + d)  * L[w' + d, c]     w' + d < W
// One workgroup per (batch, row, 32-pixel tile); each thread owns (pixel, source group) and walks
// all D planes, so every gradient element is produced by exactly one thread (no atomics).
constexpr int MAX_SRC = 4;
constexpr int WTILE = 32;

template <int CPG, bool FOR_LEFT>
__global__ void __launch_bounds__(1024)
gwc_warp_bwd_kernel(const float* __restrict__ gvol, const float* __restrict__ other,
                    const float* __restrict__ calib, float* __restrict__ gout, int B, int C, int G, int D,
                    int H, int W, float down, int align_corners) {
  extern __shared__ __align__(16) float lds[];
  const int stride = C + 4;
  float* Orow = lds;                                         // the OTHER view's row [W][stride]
  XTap* taps = reinterpret_cast<XTap*>(lds + W * stride);    // [D]
  int* src_g = reinterpret_cast<int*>(taps + D);             // [G][MAX_SRC]
  float* src_w = reinterpret_cast<float*>(src_g + G * MAX_SRC);
  int* src_n = reinterpret_cast<int*>(src_w + G * MAX_SRC);  // [G]
  const int bh = blockIdx.x;
  const int b = bh / H, h = bh - b * H;
  const int w_begin = blockIdx.y * WTILE;
  stage_row(other + ((size_t)b * H + h) * W * C, Orow, W, C, stride);
  for (int k = threadIdx.x; k < D; k += blockDim.x) taps[k] = depth_tap(calib[b], k, D, down, align_corners);
  for (int gs = threadIdx.x; gs < G; gs += blockDim.x) {
    int n = 0;
    for (int g = 0; g < G; ++g) {
      int y0;
      float w0, w1;
      group_tap(g, G, align_corners, &y0, &w0, &w1);
      if (y0 == gs && n < MAX_SRC) { src_g[gs * MAX_SRC + n] = g; src_w[gs * MAX_SRC + n] = w0; ++n; }
      if (y0 + 1 == gs && n < MAX_SRC) { src_g[gs * MAX_SRC + n] = g; src_w[gs * MAX_SRC + n] = w1; ++n; }
    }
    src_n[gs] = n;
  }
  __syncthreads();
  const float inv_cpg = 1.0f / (float)CPG;
  const size_t plane = (size_t)H * W * G;
  const float* grow = gvol + (((size_t)b * D) * H + h) * W * G;   // plane 0 of this row
  for (int item = threadIdx.x; item < WTILE * G; item += blockDim.x) {
    const int w = w_begin + item / G, gs = item % G;
    if (w >= W) continue;
    const int n = src_n[gs];
    int sg[MAX_SRC];
    float sw[MAX_SRC];
#pragma unroll
    for (int i = 0; i < MAX_SRC; ++i) {
      sg[i] = i < n ? src_g[gs * MAX_SRC + i] : 0;
      sw[i] = i < n ? src_w[gs * MAX_SRC + i] : 0.0f;
    }
    float acc[CPG];
#pragma unroll
    for (int c = 0; c < CPG; ++c) acc[c] = 0.0f;
#pragma unroll 4
    for (int k = 0; k < D; ++k) {
      const XTap t = taps[k];
      const float* gk = grow + (size_t)k * plane;
      float s_here = 0.0f;
      if (FOR_LEFT) {
#pragma unroll
        for (int i = 0; i < MAX_SRC; ++i)
          if (i < n) s_here += sw[i] * gk[(size_t)w * G + sg[i]];
      }
#pragma unroll
      for (int tx = 0; tx < 2; ++tx) {
        const int d = t.x0 + tx;
        const float wx = tx ? t.w1 : t.w0;
        if (d < 0 || d >= D) continue;
        if (FOR_LEFT) {
          if (w < d) continue;
          const float* r = Orow + (w - d) * stride + gs * CPG;
          const float coef = wx * s_here * inv_cpg;
#pragma unroll
          for (int c = 0; c < CPG; ++c) acc[c] += coef * r[c];
        } else {
          const int ws = w + d;
          if (ws >= W) continue;
          float s = 0.0f;
#pragma unroll
          for (int i = 0; i < MAX_SRC; ++i)
            if (i < n) s += sw[i] * gk[(size_t)ws * G + sg[i]];
          const float* l = Orow + ws * stride + gs * CPG;
          const float coef = wx * s * inv_cpg;
#pragma unroll
          for (int c = 0; c < CPG; ++c) acc[c] += coef * l[c];
        }
      }
    }
    float* dst = gout + (((size_t)b * H + h) * W + w) * C + gs * CPG;
#pragma unroll
    for (int c = 0; c < CPG; ++c) dst[c] = acc[c];
  }
}

// ---- backward, ONE launch for both views: the gradient volume is read once ---------------------------------------
// Same run structure as the forward.  A workgroup owns (batch, row, plane chunk) and alternates two phases:
//   A (streaming): thread = (pixel, quad of output groups) reads gvol[k, w, g..g+3] as one 16-byte load per plane and
//     folds the run's planes into T[tx] = sum_k wx_tx(k) * gvol[k] -- two FMAs per value, nothing else;
//   B (run end):   the quads go to LDS (S[tx][w][g]); thread = (pixel, SOURCE group s) rebuilds
//     gcost[s][d](w) = sum_{g: y-tap of g hits s} wy * S[tx][w][g] and accumulates, in registers,
//       gL[w , s, :] += gcost[s][d](w)      * R[w - d, s, :]          (w >= d)
//       gR[w , s, :] += gcost[s][d](w + d)  * L[w + d, s, :]          (w + d < W)
//     with both feature rows resident in LDS.  Every output element has one owner: no atomics, fixed order.
// Chunks write partial rows; gwc_partial_reduce_kernel sums them in chunk order (deterministic).
template <int CPG> struct BwdCfg { static constexpr int MAXI = CPG >= 8 ? 2 : (CPG == 4 ? 4 : 8); };
constexpr int BWD_MAXQ = 2;
constexpr int BWD_MAXTHREADS = 768;     // 12 waves = 3 per SIMD: 170 VGPRs each

template <int CPG, int UNR, bool NT>
__global__ void __launch_bounds__(BWD_MAXTHREADS)
gwc_warp_bwd2_kernel(const float* __restrict__ gvol, const float* __restrict__ left, const float* __restrict__ right,
                     const float* __restrict__ calib, float* __restrict__ part_l, float* __restrict__ part_r,
                     PlaneChunks chunks, int B, int C, int G, int D, int H, int W, float down, int align_corners) {
  constexpr int MAXI = BwdCfg<CPG>::MAXI;
  extern __shared__ __align__(16) float lds[];
  const int stride = C + 4;
  float* Lrow = lds;                                         // [W][stride]
  float* Rrow = Lrow + W * stride;                           // [W][stride]
  float* S = Rrow + W * stride;                              // [2][W][G]
  XTap* taps = reinterpret_cast<XTap*>(S + 2 * W * G);       // [chunk length <= D]
  int* src_g = reinterpret_cast<int*>(taps + D);             // [G][MAX_SRC]
  float* src_w = reinterpret_cast<float*>(src_g + G * MAX_SRC);
  int* src_n = reinterpret_cast<int*>(src_w + G * MAX_SRC);  // [G]
  const int bh = blockIdx.x;
  const int b = bh / H, h = bh - b * H;
  const int k_begin = chunks.start[blockIdx.y], k_end = chunks.start[blockIdx.y + 1];
  const int tid = threadIdx.x, nthr = blockDim.x;
  stage_row(left + ((size_t)b * H + h) * W * C, Lrow, W, C, stride);
  stage_row(right + ((size_t)b * H + h) * W * C, Rrow, W, C, stride);
  for (int k = tid; k < k_end - k_begin; k += nthr) taps[k] = depth_tap(calib[b], k_begin + k, D, down, align_corners);
  for (int gs = tid; gs < G; gs += nthr) {
    int n = 0;
    for (int g = 0; g < G; ++g) {
      int y0;
      float w0, w1;
      group_tap(g, G, align_corners, &y0, &w0, &w1);
      if (y0 == gs && n < MAX_SRC) { src_g[gs * MAX_SRC + n] = g; src_w[gs * MAX_SRC + n] = w0; ++n; }
      if (y0 + 1 == gs && n < MAX_SRC) { src_g[gs * MAX_SRC + n] = g; src_w[gs * MAX_SRC + n] = w1; ++n; }
    }
    src_n[gs] = n;
  }
  __syncthreads();

  // phase-B identity of this thread: source group s is the same for all of its items (nthr % G == 0)
  const int s = tid % G, w_first = tid / G, w_step = nthr / G;
  int sg[MAX_SRC];
  float sw[MAX_SRC];
  {
    const int n = src_n[s];
#pragma unroll
    for (int i = 0; i < MAX_SRC; ++i) {
      sg[i] = i < n ? src_g[s * MAX_SRC + i] : 0;
      sw[i] = i < n ? src_w[s * MAX_SRC + i] : 0.0f;
    }
  }
  const int nsrc = src_n[s];
  float accL[MAXI][CPG], accR[MAXI][CPG];
#pragma unroll
  for (int it = 0; it < MAXI; ++it)
#pragma unroll
    for (int c = 0; c < CPG; ++c) accL[it][c] = accR[it][c] = 0.0f;

  const int nq = W * (G >> 2);                               // quads per plane row
  const size_t plane = (size_t)H * W * G;
  const float* grow = gvol + (((size_t)b * D) * H + h) * W * G;
  int k = k_begin;
  while (k < k_end) {
    const int x0 = taps[k - k_begin].x0;
    int ke = k + 1;
    while (ke < k_end && taps[ke - k_begin].x0 == x0) ++ke;
    // ---- phase A: stream the planes [k, ke) of this run
    v4f T0[BWD_MAXQ], T1[BWD_MAXQ];
#pragma unroll
    for (int it = 0; it < BWD_MAXQ; ++it) T0[it] = T1[it] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
    for (int kk = k; kk < ke; kk += UNR) {
      v4f gv[UNR][BWD_MAXQ];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const v4f* gk = reinterpret_cast<const v4f*>(grow + (size_t)min(kk + u, ke - 1) * plane);
#pragma unroll
        for (int it = 0; it < BWD_MAXQ; ++it) {
          const int q = min(tid + it * nthr, nq - 1);
          gv[u][it] = NT ? __builtin_nontemporal_load(gk + q) : gk[q];
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (kk + u < ke) {
          const XTap t = taps[kk + u - k_begin];
#pragma unroll
          for (int it = 0; it < BWD_MAXQ; ++it) {
            T0[it] += t.w0 * gv[u][it];
            T1[it] += t.w1 * gv[u][it];
          }
        }
      }
    }
    // ---- phase B: contract the run against the feature rows
#pragma unroll
    for (int it = 0; it < BWD_MAXQ; ++it) {
      const int q = tid + it * nthr;
      if (q < nq) {
        reinterpret_cast<v4f*>(S)[q] = T0[it];
        reinterpret_cast<v4f*>(S + W * G)[q] = T1[it];
      }
    }
    __syncthreads();
#pragma unroll
    for (int tx = 0; tx < 2; ++tx) {
      const int d = x0 + tx;
      if (d < 0 || d >= D) continue;
      const float* Sx = S + tx * W * G;
#pragma unroll
      for (int it = 0; it < MAXI; ++it) {
        const int w = w_first + it * w_step;
        if (w >= W) break;
        if (w >= d) {
          float gc = 0.0f;
#pragma unroll
          for (int i = 0; i < MAX_SRC; ++i)
            if (i < nsrc) gc += sw[i] * Sx[w * G + sg[i]];
          const float* r = Rrow + (w - d) * stride + s * CPG;
#pragma unroll
          for (int c = 0; c < CPG; ++c) accL[it][c] += gc * r[c];
        }
        if (w + d < W) {
          float gc = 0.0f;
#pragma unroll
          for (int i = 0; i < MAX_SRC; ++i)
            if (i < nsrc) gc += sw[i] * Sx[(w + d) * G + sg[i]];
          const float* l = Lrow + (w + d) * stride + s * CPG;
#pragma unroll
          for (int c = 0; c < CPG; ++c) accR[it][c] += gc * l[c];
        }
      }
    }
    __syncthreads();
    k = ke;
  }
  const float inv_cpg = 1.0f / (float)CPG;
  const size_t slab = (size_t)B * H * W * C;
  float* pl = part_l + blockIdx.y * slab + ((size_t)b * H + h) * W * C;
  float* pr = part_r + blockIdx.y * slab + ((size_t)b * H + h) * W * C;
#pragma unroll
  for (int it = 0; it < MAXI; ++it) {
    const int w = w_first + it * w_step;
    if (w >= W) break;
#pragma unroll
    for (int c = 0; c < CPG; ++c) {
      pl[(size_t)w * C + s * CPG + c] = accL[it][c] * inv_cpg;
      pr[(size_t)w * C + s * CPG + c] = accR[it][c] * inv_cpg;
    }
  }
}

// ---- backward, round 3: the plane stream never stops ---------------------------------------------------------------
// bwd2 alternates "stream the planes of a run" and "contract the run" with two barriers per run and nothing in flight
// while it contracts; per batch of planes it pays one full load latency (load, wait, use).  Here
//   * the planes are read in batches of <= UNR planes of ONE run, two register buffers deep (ping-pong): while a batch is
//     folded into the tap sums the next one is in flight, and it STAYS in flight across a contraction (the barrier is a
//     raw s_barrier behind s_waitcnt lgkmcnt(0) -- __syncthreads() would drain vmcnt).  The batch sequence follows from
//     runend[] (end of the run a plane belongs to: a binary search per plane in the prologue, x0(k) is monotone);
//   * the tap sums roll: a run with taps (x0, x0 + 1) is followed by one with (x0 - 1, x0) almost everywhere, so the sum
//     for d = x0 keeps accumulating (as the upper tap) and only the finished one (d = x0 + 1) is contracted -- one
//     contraction per run instead of two, one barrier each (the exchange buffer S is double-buffered);
//   * y-taps of weight zero (align_corners: every second one) are dropped from the source lists.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct __align__(16) BTap { int x0; float w0, w1; int runend; };   // runend: first plane behind the run of this plane

template <int CPG, int UNR>
__global__ void __launch_bounds__(BWD_MAXTHREADS)
gwc_warp_bwd3_kernel(const float* __restrict__ gvol, const float* __restrict__ left, const float* __restrict__ right,
                     const float* __restrict__ calib, float* __restrict__ part_l, float* __restrict__ part_r,
                     PlaneChunks chunks, int B, int C, int G, int D, int H, int W, float down, int align_corners) {
  constexpr int MAXI = BwdCfg<CPG>::MAXI;
  extern __shared__ __align__(16) float lds[];
  const int stride = C;                                      // the contraction reads runs of consecutive s: no pad needed
  // order matters: the contraction reads S[(w + d) * G ..], L[w + d], R[w - d] WITHOUT range checks (a selected 0 multiplies
  // what comes back), so an overshoot must land in initialised floats: S0 -> S1 -> Lrow <- Rrow (negative) / Lrow -> Rrow
  float* S = lds;                                            // [2][W][G]
  float* Lrow = S + 2 * W * G;                               // [W][C]
  float* Rrow = Lrow + W * stride;                           // [W][C]
  BTap* taps = reinterpret_cast<BTap*>(Rrow + W * stride);   // [chunk length <= D]
  int* src_g = reinterpret_cast<int*>(taps + D);             // [G][MAX_SRC]
  float* src_w = reinterpret_cast<float*>(src_g + G * MAX_SRC);
  int* src_n = reinterpret_cast<int*>(src_w + G * MAX_SRC);  // [G]
  int* ytap_y = src_n + G;                                   // [G]     y-tap of output group g: first source group ...
  float* ytap_w = reinterpret_cast<float*>(ytap_y + G);      // [G][2]  ... and the two weights
  int* nsrc_max = reinterpret_cast<int*>(ytap_w + 2 * G);    // [1]
  const int bh = blockIdx.x;
  const int b = bh / H, h = bh - b * H;
  const int k_begin = chunks.start[blockIdx.y], k_end = chunks.start[blockIdx.y + 1], len = k_end - k_begin;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const size_t plane4 = (size_t)H * W * G / 4;
  const v4f* grow = reinterpret_cast<const v4f*>(gvol + (((size_t)b * D) * H + h) * W * G);

  stage_row(left + ((size_t)b * H + h) * W * C, Lrow, W, C, stride);
  stage_row(right + ((size_t)b * H + h) * W * C, Rrow, W, C, stride);
  for (int k = tid; k < len; k += nthr) {
    const XTap t = depth_tap(calib[b], k_begin + k, D, down, align_corners);
    taps[k].x0 = t.x0; taps[k].w0 = t.w0; taps[k].w1 = t.w1;
  }
  for (int g = tid; g < G; g += nthr) {                       // the y-tap of every output group, once (IEEE divisions)
    int y0;
    float w0, w1;
    group_tap(g, G, align_corners, &y0, &w0, &w1);
    ytap_y[g] = y0; ytap_w[2 * g] = w0; ytap_w[2 * g + 1] = w1;
  }
  if (tid == 0) *nsrc_max = 0;
  __syncthreads();
  for (int gs = tid; gs < G; gs += nthr) {                    // source lists: who reads source group gs, with which weight
    int n = 0;                                                // (y0(g) is g - 1 or g in both conventions: +-2 is generous)
    for (int g = max(gs - 2, 0); g <= min(gs + 2, G - 1); ++g) {
      const int y0 = ytap_y[g];
      const float w0 = ytap_w[2 * g], w1 = ytap_w[2 * g + 1];
      if (y0 == gs && w0 != 0.0f && n < MAX_SRC) { src_g[gs * MAX_SRC + n] = g; src_w[gs * MAX_SRC + n] = w0; ++n; }
      if (y0 + 1 == gs && w1 != 0.0f && n < MAX_SRC) { src_g[gs * MAX_SRC + n] = g; src_w[gs * MAX_SRC + n] = w1; ++n; }
    }
    src_n[gs] = n;
    atomicMax(nsrc_max, n);
  }
  for (int i = tid; i < len; i += nthr) {                    // equal taps are contiguous (monotone): binary search for the run end
    const int xi = taps[i].x0;
    int lo = i + 1, hi = len;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (taps[mid].x0 == xi) lo = mid + 1; else hi = mid;
    }
    taps[i].runend = k_begin + lo;
  }
  __syncthreads();

  // contraction identity of this thread: source group s is the same for all of its items (nthr % G == 0)
  const int s = tid % G, w_first = tid / G, w_step = nthr / G;
  int sg[MAX_SRC];
  float sw[MAX_SRC];
  const int nsrc = *nsrc_max;                                // the longest source list, workgroup-uniform: shorter lists are
  {                                                          // padded with weight 0 on entry 0 (scalar branches below)
    const int n = src_n[s];
#pragma unroll
    for (int i = 0; i < MAX_SRC; ++i) {
      sg[i] = i < n ? src_g[s * MAX_SRC + i] : 0;
      sw[i] = i < n ? src_w[s * MAX_SRC + i] : 0.0f;
    }
  }
  float accL[MAXI][CPG], accR[MAXI][CPG];
#pragma unroll
  for (int it = 0; it < MAXI; ++it)
#pragma unroll
    for (int c = 0; c < CPG; ++c) accL[it][c] = accR[it][c] = 0.0f;

  v4f Tlo[BWD_MAXQ], Thi[BWD_MAXQ];                          // tap sums for d = cur and d = cur + 1
#pragma unroll
  for (int it = 0; it < BWD_MAXQ; ++it) Tlo[it] = Thi[it] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
  int cur = INT32_MIN, flip = 0;
  const int dmax = min(D, W);

  // the run changes to taps (xn, xn + 1): contract what is finished.  xn == cur - 1: only the upper sum (d = cur + 1), the
  // lower one becomes the new upper one.  Anything else (first run excepted): both.  ONE copy of the contraction per call
  // site: a loop of one or two rounds with the operand selected by the (workgroup-uniform) round.
  auto retire = [&](int xn) __attribute__((always_inline)) {
    if (cur != INT32_MIN) {
      const bool roll = xn == cur - 1;
#pragma nounroll
      for (int f = 0; f < (roll ? 1 : 2); ++f) {
        const int d = f ? cur : cur + 1;
        if (d < 0 || d >= dmax) continue;                    // workgroup-uniform
        float* Sx = S + flip * W * G;
        flip ^= 1;
#pragma unroll
        for (int it = 0; it < BWD_MAXQ; ++it)
          reinterpret_cast<v4f*>(Sx)[tid + it * nthr] = f ? Tlo[it] : Thi[it];
        lds_barrier();                                       // readers of the other buffer are past their previous round
        // branch-free and clamp-free (the launcher guarantees MAXI * w_step == W): an out-of-range item reads initialised
        // floats of a neighbouring LDS region and contributes a selected 0, so that the LDS reads go out back to back
        // (guarded, every item was an exec-mask region with its own LDS latency: 22 of the kernel's 67 us).
        // (the asm makes the lane's offsets opaque: otherwise LICM precomputes base + it * step + sg[i] for every
        // (item, source) once per kernel and the register allocator spills the 24 of them)
        int oS = w_first * G, og[MAX_SRC];
        asm volatile("" : "+v"(oS));
#pragma unroll
        for (int i = 0; i < MAX_SRC; ++i) {
          og[i] = sg[i];
          asm volatile("" : "+v"(og[i]));
        }
        const float* pSl = Sx + oS;
        const float* pSr = pSl + d * G;
        const float* pR = Rrow + (w_first - d) * stride + s * CPG;
        const float* pL = Lrow + (w_first + d) * stride + s * CPG;
#pragma unroll
        for (int it = 0; it < MAXI; ++it) {
          const int w = w_first + it * w_step;
          float gl = 0.0f, gr = 0.0f;
#pragma unroll
          for (int i = 0; i < MAX_SRC; ++i)
            if (i < nsrc) {                                  // workgroup-uniform
              gl += sw[i] * pSl[it * w_step * G + og[i]];
              gr += sw[i] * pSr[it * w_step * G + og[i]];
            }
          gl = w >= d ? gl : 0.0f;
          gr = w + d < W ? gr : 0.0f;
#pragma unroll
          for (int c = 0; c < CPG; ++c) {
            accL[it][c] += gl * pR[it * w_step * stride + c];
            accR[it][c] += gr * pL[it * w_step * stride + c];
          }
        }
      }
#pragma unroll
      for (int it = 0; it < BWD_MAXQ; ++it) {
        Thi[it] = roll ? Tlo[it] : v4f{0.0f, 0.0f, 0.0f, 0.0f};
        Tlo[it] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
      }
    }
    cur = xn;
  };

  // batch = planes [k, min(k + UNR, ke)) of the run that ends at ke; slots past the batch repeat its last plane (so do the
  // two batches behind the chunk: a conditional load makes hipcc drain vmcnt in the loop, an L2 hit costs nothing).  The
  // batch's taps travel with it in registers (read when its loads are issued: their LDS latency hides behind the loads)
  struct Batch { v4f v[UNR][BWD_MAXQ]; int k, ke, x0; float w0[UNR], w1[UNR]; };
  Batch bA, bB;
  auto load = [&](Batch& bt, int k, int ke) __attribute__((always_inline)) {
    bt.k = k; bt.ke = ke;
    const int last = min(ke, k_end) - 1;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int kk = max(min(k + u, last), k_begin);
      const v4f* gk = grow + (size_t)kk * plane4;
#pragma unroll
      for (int it = 0; it < BWD_MAXQ; ++it) bt.v[u][it] = gk[tid + it * nthr];
      const BTap t = taps[kk - k_begin];
      if (u == 0) bt.x0 = t.x0;
      bt.w0[u] = t.w0; bt.w1[u] = t.w1;
    }
  };
  auto fold = [&](const Batch& bt) __attribute__((always_inline)) {
    if (bt.k >= k_end) return;
    if (bt.x0 != cur) retire(bt.x0);
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (bt.k + u < bt.ke) {
#pragma unroll
        for (int it = 0; it < BWD_MAXQ; ++it) {
          Tlo[it] += bt.w0[u] * bt.v[u][it];
          Thi[it] += bt.w1[u] * bt.v[u][it];
        }
      }
    }
  };
  auto advance = [&](int& k, int& ke) __attribute__((always_inline)) {   // the batch behind (k, ke)
    k = min(k + UNR, ke);
    if (k >= ke && k < k_end) ke = taps[k - k_begin].runend;
  };
  int kA = k_begin, keA = taps[0].runend, kB, keB;
  load(bA, kA, keA);
  while (kA < k_end) {
    kB = kA; keB = keA;
    advance(kB, keB);
    load(bB, kB, keB);
    fold(bA);
    kA = kB; keA = keB;
    advance(kA, keA);
    load(bA, kA, keA);
    fold(bB);
  }
  retire(INT32_MIN + 7);                                     // no plane follows: both sums are finished

  const float inv_cpg = 1.0f / (float)CPG;
  const size_t slab = (size_t)B * H * W * C;
  float* pl = part_l + blockIdx.y * slab + ((size_t)b * H + h) * W * C;
  float* pr = part_r + blockIdx.y * slab + ((size_t)b * H + h) * W * C;
#pragma unroll
  for (int it = 0; it < MAXI; ++it) {
    const int w = w_first + it * w_step;
#pragma unroll
    for (int c = 0; c < CPG; ++c) {
      pl[(size_t)w * C + s * CPG + c] = accL[it][c] * inv_cpg;
      pr[(size_t)w * C + s * CPG + c] = accR[it][c] * inv_cpg;
    }
  }
}

// out[i] = sum over chunks (ascending) of part[chunk][i]; n4 float4 elements per slab
__global__ void gwc_partial_reduce_kernel(const float* __restrict__ part0, float* __restrict__ out0,
                                          const float* __restrict__ part1, float* __restrict__ out1, int nchunks, size_t n4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float* part = blockIdx.y ? part1 : part0;             // blockIdx.y: left / right view
  float* out = blockIdx.y ? out1 : out0;
  v4f acc = reinterpret_cast<const v4f*>(part)[i];
  for (int c = 1; c < nchunks; ++c) acc += reinterpret_cast<const v4f*>(part)[(size_t)c * n4 + i];
  reinterpret_cast<v4f*>(out)[i] = acc;
}

// Plane chunks of (roughly) equal cost.  Cost model: one unit per plane plus `run_cost` units per run start, with the run
// starts distributed like the changes of x0(k) ~ X / (k + 1), X ~ D / 2 (KITTI: calib / 4 = 98 for D = 192): one per
// plane while X / (k+1)^2 >= 1, X / (k+1)^2 per plane behind.  The true taps depend on calib (device memory); this only
// balances the workgroups, any partition is correct.
PlaneChunks make_chunks(int D, int n, float run_cost, int maxlen) {
  PlaneChunks ch;
  n = std::max(1, std::min(n, std::min(D, MAX_CHUNKS)));
  const float X = 0.5f * (float)D;
  for (;; ++n) {
    double total = 0.0;
    for (int k = 0; k < D; ++k) total += 1.0 + run_cost * std::min(1.0, (double)X / ((double)(k + 1) * (k + 1)));
    ch.n = n;
    ch.start[0] = 0;
    double acc = 0.0;
    int c = 1;
    for (int k = 0; k < D && c < n; ++k) {
      acc += 1.0 + run_cost * std::min(1.0, (double)X / ((double)(k + 1) * (k + 1)));
      // chunk c starts behind plane k once the prefix reaches c/n of the total; keep >= 1 plane per remaining chunk
      while (c < n && (acc >= total * c / n || D - (k + 1) <= n - c)) {
        ch.start[c] = std::min(std::max(k + 1, ch.start[c - 1] + 1), D - (n - c));
        ++c;
      }
    }
    ch.start[n] = D;
    int longest = 0;
    for (int i = 0; i < n; ++i) longest = std::max(longest, ch.start[i + 1] - ch.start[i]);
    if (longest <= maxlen || n >= std::min(D, MAX_CHUNKS)) return ch;
  }
}

bool gwc_dims_ok(const ssbev_gwc_dims* d) {
  if (!d || d->B <= 0 || d->C <= 0 || d->G <= 0 || d->D <= 0 || d->H <= 0 || d->W <= 0) return false;
  if (d->C % d->G != 0 || d->C % 4 != 0 || d->down != 1.0f) return false;
  const int cpg = d->C / d->G;
  return cpg == 1 || cpg == 2 || cpg == 4 || cpg == 8;
}

size_t fwd_lds_bytes(const ssbev_gwc_dims* d) { return (size_t)d->W * (d->C + 4) * 4 + KCHUNK * sizeof(XTap); }
size_t bwd_lds_bytes(const ssbev_gwc_dims* d) {
  return (size_t)d->W * (d->C + 4) * 4 + d->D * sizeof(XTap) + (size_t)d->G * MAX_SRC * 8 + d->G * 4;
}

int env_int(const char* name, int dflt) {
  const char* v = ssbev_tune(name);          // tuning hooks only (common.h)
  return v ? atoi(v) : dflt;
}

// ---- forward: fwd4 (four groups per lane, run-cached correlations) when G % 4 == 0, else the per-plane kernel
template <int CPG>
int launch_fwd(const float* l, const float* r, const float* calib, float* vol, const ssbev_gwc_dims* d,
               hipStream_t st) {
  static const int variant = env_int("SSBEV_GWC_FWD", 3);          // 1 = per-plane kernel (r1), 2 = fwd4 (r2), 3 = fwd5
  if (variant >= 3 && d->G % 4 == 0) {
    static const int threads = env_int("SSBEV_GWC_FWD_THREADS", 256);
    static const int wg_target = env_int("SSBEV_GWC_FWD_WGS", 480);
    static const int nt = env_int("SSBEV_GWC_FWD_NT", 0);
    static const float run_cost = (float)env_int("SSBEV_GWC_FWD_RUNCOST", 4);
    int unit = 64, g4 = d->G / 4;                   // block size: a multiple of the wave and of G / 4
    while (unit % g4 != 0) unit += 64;
    const int nthreads = std::max(unit, std::min(512, std::max(64, threads)) / unit * unit);
    const int px = nthreads / g4, ntiles = (int)cdiv(d->W, px);
    const long groups = (long)d->B * d->H * ntiles;
    const PlaneChunks ch = make_chunks(d->D, (int)cdiv(wg_target, groups), run_cost, d->D);
    const size_t lds = (size_t)d->W * (d->C + 4) * 4 + (size_t)d->D * sizeof(XTap);
    if (lds <= 160 * 1024 && groups <= 0x7fffffffL) {
      auto kern = nt ? gwc_warp_fwd5_kernel<CPG, true> : gwc_warp_fwd5_kernel<CPG, false>;
      if (lds > 64 * 1024 &&
          hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
              hipSuccess)
        return SSBEV_ELAUNCH;
      hipLaunchKernelGGL(kern, dim3((unsigned)groups, ch.n), dim3(nthreads), lds, st, l, r, calib, vol, ch, d->B, d->C,
                         d->G, d->D, d->H, d->W, d->down, d->align_corners, ntiles);
      return ssbev_launch_status();
    }
  }
  if (variant != 1 && d->G % 4 == 0) {
    static const int threads = env_int("SSBEV_GWC_FWD_THREADS", 256);
    static const int wg_target = env_int("SSBEV_GWC_FWD_WGS", 768);
    static const int nt = env_int("SSBEV_GWC_FWD_NT", 1);
    static const float run_cost = (float)env_int("SSBEV_GWC_FWD_RUNCOST", 4);
    static const int rows_env = env_int("SSBEV_GWC_FWD_ROWS", 0);        // 0 = LDS-staged row per workgroup; n = n rows, unstaged
    int rows = rows_env;
    while (rows > 1 && d->H % rows != 0) --rows;                         // a workgroup's rows share one batch element
    const int nrow_groups = rows > 0 ? d->B * d->H / rows : d->B * d->H;
    const PlaneChunks ch = make_chunks(d->D, (int)cdiv(wg_target, nrow_groups), run_cost, FWD_MAXLEN);
    const size_t lds = (rows > 0 ? 0 : (size_t)d->W * (d->C + 4) * 4) + FWD_MAXLEN * sizeof(XTap);
    if (lds <= 160 * 1024) {
      auto kern = rows > 0 ? (nt ? gwc_warp_fwd4_kernel<CPG, true, false> : gwc_warp_fwd4_kernel<CPG, false, false>)
                           : (nt ? gwc_warp_fwd4_kernel<CPG, true, true> : gwc_warp_fwd4_kernel<CPG, false, true>);
      if (lds > 64 * 1024 &&
          hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
              hipSuccess)
        return SSBEV_ELAUNCH;
      int unit = 64, g4 = d->G / 4;                 // block size: a multiple of the wave and of G / 4 (see the kernel)
      while (unit % g4 != 0) unit += 64;
      const int nthreads = std::max(unit, std::min(512, std::max(64, threads)) / unit * unit);
      hipLaunchKernelGGL(kern, dim3(nrow_groups, ch.n), dim3(nthreads), lds, st, l, r, calib,
                         vol, ch, d->B, d->C, d->G, d->D, d->H, d->W, d->down, d->align_corners, std::max(rows, 1));
      return ssbev_launch_status();
    }
  }
  const size_t lds = fwd_lds_bytes(d);
  if (lds > 160 * 1024) return SSBEV_EINVAL;
  auto kern = gwc_warp_fwd_kernel<CPG>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
          hipSuccess)
    return SSBEV_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3(d->B * d->H, cdiv(d->D, KCHUNK)), dim3(256), lds, st, l, r, calib, vol, d->B, d->C,
                     d->G, d->D, d->H, d->W, d->down, d->align_corners);
  return ssbev_launch_status();
}

template <int CPG>
int launch_bwd(const float* gvol, const float* l, const float* r, const float* calib, float* gl, float* gr,
               const ssbev_gwc_dims* d, hipStream_t st) {
  const size_t lds = bwd_lds_bytes(d);
  if (lds > 160 * 1024) return SSBEV_EINVAL;
  auto kl = gwc_warp_bwd_kernel<CPG, true>;
  auto kr = gwc_warp_bwd_kernel<CPG, false>;
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kl), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(kr), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
      return SSBEV_ELAUNCH;
  }
  // one thread per (pixel, source group) of the tile: the 192-plane walk is the only serial loop left
  dim3 grid(d->B * d->H, cdiv(d->W, WTILE)), block(std::min(1024, std::max(256, WTILE * d->G)));
  hipLaunchKernelGGL(kl, grid, block, lds, st, gvol, r, calib, gl, d->B, d->C, d->G, d->D, d->H, d->W, d->down,
                     d->align_corners);
  hipLaunchKernelGGL(kr, grid, block, lds, st, gvol, l, calib, gr, d->B, d->C, d->G, d->D, d->H, d->W, d->down,
                     d->align_corners);
  return ssbev_launch_status();
}

// ---- fused backward (one read of the gradient volume) ------------------------------------------------------------
struct Bwd2Plan { bool ok; int threads; PlaneChunks ch; size_t lds; };

template <int CPG>
Bwd2Plan plan_bwd2(const ssbev_gwc_dims* d) {
  Bwd2Plan p;
  p.ok = false;
  if (d->G % 4 != 0 || d->G > 64 || 64 % d->G != 0) return p;
  const int items = d->W * d->G;
  int threads = (int)cdiv(cdiv(items, BwdCfg<CPG>::MAXI), 64) * 64;
  threads = std::max(threads, (int)cdiv(cdiv(items / 4, BWD_MAXQ), 64) * 64);
  threads = std::max(threads, 256);
  if (threads > BWD_MAXTHREADS) return p;
  p.threads = threads;
  p.lds = (size_t)2 * d->W * (d->C + 4) * 4 + (size_t)2 * d->W * d->G * 4 + (size_t)d->D * sizeof(BTap) +
          (size_t)d->G * MAX_SRC * 8 + (size_t)d->G * 16 + 16;
  if (p.lds > 160 * 1024) return p;
  static const int wg_target = env_int("SSBEV_GWC_BWD_WGS", 256);     // one workgroup per CU (LDS-bound occupancy)
  const int rows = d->B * d->H;
  static const float run_cost = (float)env_int("SSBEV_GWC_BWD_RUNCOST", 8);
  p.ch = make_chunks(d->D, std::max(1, wg_target / rows), run_cost, d->D);
  p.ok = true;
  return p;
}

template <int CPG>
int launch_bwd2(const float* gvol, const float* l, const float* r, const float* calib, float* gl, float* gr,
                const ssbev_gwc_dims* d, void* ws, size_t ws_bytes, hipStream_t st) {
  const Bwd2Plan p = plan_bwd2<CPG>(d);
  if (!p.ok) return launch_bwd<CPG>(gvol, l, r, calib, gl, gr, d, st);
  const size_t slab = (size_t)d->B * d->H * d->W * d->C;
  float *pl = gl, *pr = gr;
  if (p.ch.n > 1) {
    if (!ws || ws_bytes < 2 * p.ch.n * slab * sizeof(float)) return SSBEV_EWORKSPACE;
    pl = static_cast<float*>(ws);
    pr = pl + p.ch.n * slab;
  }
  static const int variant = env_int("SSBEV_GWC_BWD", 3);           // 2 = bwd2 (r2), 3 = bwd3
  static const int unr = env_int("SSBEV_GWC_BWD_UNR", variant >= 3 ? 1 : 2), nt = env_int("SSBEV_GWC_BWD_NT", 0);
  const dim3 grid(d->B * d->H, p.ch.n), block(p.threads);
  const bool exact = d->W * (d->G / 4) == BWD_MAXQ * p.threads && p.threads % d->G == 0 &&
                     BwdCfg<CPG>::MAXI * (p.threads / d->G) == d->W;       // bwd3 has no tails (KITTI: 640 threads)
  if (variant >= 3 && exact) {
    auto kern = unr >= 2 ? gwc_warp_bwd3_kernel<CPG, 2> : gwc_warp_bwd3_kernel<CPG, 1>;   // 4 planes per batch spill
    if (p.lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds) !=
            hipSuccess)
      return SSBEV_ELAUNCH;
    hipLaunchKernelGGL(kern, grid, block, p.lds, st, gvol, l, r, calib, pl, pr, p.ch, d->B, d->C, d->G, d->D, d->H, d->W,
                       d->down, d->align_corners);
  } else {
    auto kern = unr >= 4 ? (nt ? gwc_warp_bwd2_kernel<CPG, 4, true> : gwc_warp_bwd2_kernel<CPG, 4, false>)
                         : (nt ? gwc_warp_bwd2_kernel<CPG, 2, true> : gwc_warp_bwd2_kernel<CPG, 2, false>);
    if (p.lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds) !=
            hipSuccess)
      return SSBEV_ELAUNCH;
    hipLaunchKernelGGL(kern, grid, block, p.lds, st, gvol, l, r, calib, pl, pr, p.ch, d->B, d->C, d->G, d->D, d->H, d->W,
                       d->down, d->align_corners);
  }
  if (p.ch.n > 1) {
    const size_t n4 = slab / 4;
    hipLaunchKernelGGL(gwc_partial_reduce_kernel, dim3(cdiv(n4, 256), 2), dim3(256), 0, st, pl, gl, pr, gr, p.ch.n, n4);
  }
  return ssbev_launch_status();
}

template <int CPG>
size_t bwd2_workspace(const ssbev_gwc_dims* d) {
  const Bwd2Plan p = plan_bwd2<CPG>(d);
  if (!p.ok || p.ch.n <= 1) return 0;
  return (size_t)2 * p.ch.n * d->B * d->H * d->W * d->C * sizeof(float);
}

}  // namespace

extern "C" {

int ssbev_gwc_warp_fwd(const float* left, const float* right, const float* calib, float* vol,
                       const ssbev_gwc_dims* d, ssbev_stream_t stream) {
  if (!gwc_dims_ok(d) || !left || !right || !calib || !vol) return SSBEV_EINVAL;
  hipStream_t st = as_stream(stream);
  switch (d->C / d->G) {
    case 1: return launch_fwd<1>(left, right, calib, vol, d, st);
    case 2: return launch_fwd<2>(left, right, calib, vol, d, st);
    case 4: return launch_fwd<4>(left, right, calib, vol, d, st);
    default: return launch_fwd<8>(left, right, calib, vol, d, st);
  }
}

int ssbev_gwc_warp_bwd(const float* grad_vol, const float* left, const float* right,
                       const float* calib, float* grad_left, float* grad_right,
                       const ssbev_gwc_dims* d, ssbev_stream_t stream) {
  if (!gwc_dims_ok(d) || !grad_vol || !left || !right || !calib || !grad_left || !grad_right) return SSBEV_EINVAL;
  hipStream_t st = as_stream(stream);
  switch (d->C / d->G) {
    case 1: return launch_bwd<1>(grad_vol, left, right, calib, grad_left, grad_right, d, st);
    case 2: return launch_bwd<2>(grad_vol, left, right, calib, grad_left, grad_right, d, st);
    case 4: return launch_bwd<4>(grad_vol, left, right, calib, grad_left, grad_right, d, st);
    default: return launch_bwd<8>(grad_vol, left, right, calib, grad_left, grad_right, d, st);
  }
}

size_t ssbev_gwc_warp_bwd_workspace(const ssbev_gwc_dims* d) {
  if (!gwc_dims_ok(d)) return 0;
  switch (d->C / d->G) {
    case 1: return bwd2_workspace<1>(d);
    case 2: return bwd2_workspace<2>(d);
    case 4: return bwd2_workspace<4>(d);
    default: return bwd2_workspace<8>(d);
  }
}

int ssbev_gwc_warp_bwd_fused(const float* grad_vol, const float* left, const float* right, const float* calib,
                             float* grad_left, float* grad_right, const ssbev_gwc_dims* d, void* ws, size_t ws_bytes,
                             ssbev_stream_t stream) {
  if (!gwc_dims_ok(d) || !grad_vol || !left || !right || !calib || !grad_left || !grad_right) return SSBEV_EINVAL;
  hipStream_t st = as_stream(stream);
  switch (d->C / d->G) {
    case 1: return launch_bwd2<1>(grad_vol, left, right, calib, grad_left, grad_right, d, ws, ws_bytes, st);
    case 2: return launch_bwd2<2>(grad_vol, left, right, calib, grad_left, grad_right, d, ws, ws_bytes, st);
    case 4: return launch_bwd2<4>(grad_vol, left, right, calib, grad_left, grad_right, d, ws, ws_bytes, st);
    default: return launch_bwd2<8>(grad_vol, left, right, calib, grad_left, grad_right, d, ws, ws_bytes, st);
  }
}

}  // extern "C"
