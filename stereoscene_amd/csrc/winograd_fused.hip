// Depth-fused Winograd contraction for the wide stride-1 3x3x3 layers (voxel encoder 128..512 channels, FPN, the
// 384 -> 192 occupancy-head conv: R3D:18-32, FPN:53-69, OCC:100-107), gfx950.  Replaces the 144 batched library GEMMs of the
// F(2x4x4) pipeline and their 4.5x transformed tensors:
//
//   P  = wino43_2d_input(x)        [36][B*D*Thw][K]     F(4,3) x F(4,3) over (h, w) only: 2.25x the activation
//   Mo = THIS KERNEL (P, Wp)       [36][B*D*Thw][N]
//   y  = wino43_2d_output(Mo)
//
// The depth axis of F(2,3) never exists in memory: for a depth tile i (output planes 2i, 2i+1) the four input planes
// 2i-1 .. 2i+2 of P are staged through LDS, every wave forms the four depth frequencies v0 = p0 - p2, v1 = p1 + p2,
// v2 = p2 - p1, v3 = p1 - p3 on its A fragments in registers (three VALU adds per element), multiplies each with its own
// weight matrix U[f][xi_hw] (four accumulator sets) and applies the depth output transform o0 = m0 + m1 + m2,
// o1 = m1 - m2 - m3 in the epilogue.  Same 6x multiply-add reduction as F(2x4x4), half the HBM traffic of the GEMM stage
// (P and Mo are 2.25x, not 4.5x), no V / M tensors, no library call.
//
// Workgroup = NW waves (2..4), tile = 64 rows (hw-tiles of one (b, depth tile, xi_hw)) x NW*32 columns:
//   * A: the 4 x 64 x 32-channel slab of a k-stage is copied global -> LDS by global_load_lds_dwordx4 (whole 128-byte rows,
//     16-byte slots XOR-swizzled by (row & 7) on the GLOBAL side so that the per-lane ds_read_b128 of the MFMA A operand is
//     conflict-free); two stages in flight; every wave of the workgroup reads the same slab (A leaves LDS NW times per fetch);
//   * B: packed weights Wp[xi_hw][f][q][kh][n][4] (L2 resident), one coalesced float4 per lane per (f, 8-channel step),
//     prefetched one step ahead in registers;
//   * v_mfma_f32_32x32x2_f32 (exact fp32), 4 f x 2 row tiles = 8 accumulator tiles (128 VGPRs) per wave: two workgroups per CU.
// Workgroups are ordered xi_hw-major so that all XCDs work on one frequency's weight slab at a time, an XCD owns a contiguous
// range of row tasks (adjacent depth tiles share two of their four planes through its L2), column groups are adjacent.
//
// The weight gradient (wino_dfw_kernel) uses the same decomposition transposed: gU[f][xi_hw] = sum_rows v_f^T z_f with
// v from P (saved by the forward) and z0 = g0, z1 = g0 + g1, z2 = g0 - g1, z3 = -g1 from the (h, w)-adjoint planes of gy.
#include "common.h"

#include <algorithm>
#include <cstdlib>

namespace {

typedef float wf32x16 __attribute__((ext_vector_type(16)));
__device__ const float kDfZeros[4] = {0.f, 0.f, 0.f, 0.f};

constexpr int DF_BM = 64;       // rows (hw tiles) per workgroup
constexpr int DF_BK = 32;       // channels per stage: one 128-byte line per row
constexpr int DF_STAGE_FLOATS = 4 * DF_BM * DF_BK;   // 4 planes

struct DfGeom {
  int B, D, Thw, K, N, NPad;
  int ND;         // depth tiles = D / 2
  int nrowgrp;    // ceil(Thw / 64)
  int ncolgrp;    // column groups of NW * 32
  int NU, NU8;    // row tasks per frequency (B * ND * nrowgrp) and its per-XCD share
  int nxi;        // (h, w) frequencies (36)
};

template <int DUMMY>
__global__ void __launch_bounds__(256, 2)
wino_df_kernel(const float* __restrict__ P, const float* __restrict__ Wp, float* __restrict__ Mo, DfGeom g) {
  extern __shared__ __align__(16) float lds[];            // [2 stages][4 planes][64 rows][32 k]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = blockDim.x >> 6;
  const int li = lane & 31, lk = lane >> 5;
  // ---- task decode (XCD-aware: consecutive workgroup ids go round-robin over the 8 XCDs)
  const int id = blockIdx.x, xcd = id & 7, s = id >> 3;
  const int per_xi = g.NU8 * g.ncolgrp;
  const int xhw = s / per_xi, r = s - xhw * per_xi;
  const int cg = r % g.ncolgrp, u = xcd * g.NU8 + r / g.ncolgrp;
  if (u >= g.NU) return;
  const int i = u % g.ND, tg = (u / g.ND) % g.nrowgrp, b = u / (g.ND * g.nrowgrp);
  const int t0 = tg * DF_BM;
  const int n0 = (cg * NW + wave) * 32;
  const bool col_active = n0 < g.NPad;
  const long R = (long)g.B * g.D * g.Thw;
  const float* Px = P + (long)xhw * R * g.K;
  const int nst = g.K / DF_BK;

  // ---- A staging: a stage = 4 planes x 64 rows x 8 slots of 16 bytes = 32 wave instructions of 1 KiB
  auto issue = [&](int st, int buf) {
    for (int j = wave; j < 32; j += NW) {
      const int a = j >> 3, item = (j & 7) * 64 + lane;
      const int row = item >> 3, slot = item & 7;
      const int d = 2 * i - 1 + a;
      const int t = t0 + row;
      const float* src = (d >= 0 && d < g.D && t < g.Thw)
                             ? Px + (((long)b * g.D + d) * g.Thw + t) * g.K + st * DF_BK + ((slot ^ (row & 7)) << 2)
                             : kDfZeros;
      __builtin_amdgcn_global_load_lds(src, lds + buf * DF_STAGE_FLOATS + a * (DF_BM * DF_BK) + (j & 7) * 256, 16, 0, 0);
    }
  };

  wf32x16 acc[4][2];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) acc[f][mt][rr] = 0.0f;

  // packed weights: ((((xhw * 4 + f) * Q + q) * 2 + lk) * NPad + n) * 4 + t
  const int Q = g.K >> 3;
  const size_t fstride = (size_t)Q * 2 * g.NPad * 4;
  const float* wl = Wp + (size_t)xhw * 4 * fstride + ((size_t)lk * g.NPad + (col_active ? n0 : 0) + li) * 4;
  auto load_b = [&](int q, float4 (&bv)[4]) {
#pragma unroll
    for (int f = 0; f < 4; ++f)
      bv[f] = *reinterpret_cast<const float4*>(wl + (size_t)f * fstride + (size_t)q * 2 * g.NPad * 4);
  };

  float4 bcur[4], bnext[4];
  load_b(0, bcur);
  issue(0, 0);
  for (int st = 0; st < nst; ++st) {
    const int buf = st & 1;
    if (st + 1 < nst) issue(st + 1, buf ^ 1);
    // this stage's A slab has landed once at most the next stage's copies (issued after it) are outstanding
    if (st + 1 < nst) {
      if (NW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (NW == 3) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");      // waves 0,1: 11 copies, wave 2: 10
      else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const float* ab = lds + buf * DF_STAGE_FLOATS;
#pragma unroll
    for (int qq = 0; qq < DF_BK / 8; ++qq) {
      const int q = st * (DF_BK / 8) + qq;
      if (q + 1 < Q) load_b(q + 1, bnext);
      float4 v[4][2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row = mt * 32 + li;
        const int off = row * DF_BK + (((2 * qq + lk) ^ (row & 7)) << 2);
        const float4 p0 = *reinterpret_cast<const float4*>(ab + 0 * (DF_BM * DF_BK) + off);
        const float4 p1 = *reinterpret_cast<const float4*>(ab + 1 * (DF_BM * DF_BK) + off);
        const float4 p2 = *reinterpret_cast<const float4*>(ab + 2 * (DF_BM * DF_BK) + off);
        const float4 p3 = *reinterpret_cast<const float4*>(ab + 3 * (DF_BM * DF_BK) + off);
        v[0][mt] = make_float4(p0.x - p2.x, p0.y - p2.y, p0.z - p2.z, p0.w - p2.w);
        v[1][mt] = make_float4(p1.x + p2.x, p1.y + p2.y, p1.z + p2.z, p1.w + p2.w);
        v[2][mt] = make_float4(p2.x - p1.x, p2.y - p1.y, p2.z - p1.z, p2.w - p1.w);
        v[3][mt] = make_float4(p1.x - p3.x, p1.y - p3.y, p1.z - p3.z, p1.w - p3.w);
      }
#define SSBEV_DF_COMP(COMP)                                                                              \
      _Pragma("unroll") for (int f = 0; f < 4; ++f)                                                      \
      _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                   \
        acc[f][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[f][mt].COMP, bcur[f].COMP, acc[f][mt], 0, 0, 0);
      SSBEV_DF_COMP(x) SSBEV_DF_COMP(y) SSBEV_DF_COMP(z) SSBEV_DF_COMP(w)
#undef SSBEV_DF_COMP
#pragma unroll
      for (int f = 0; f < 4; ++f) bcur[f] = bnext[f];
    }
    __syncthreads();          // every wave is done with `buf` before the stage after next is copied into it
  }
  // ---- epilogue: depth output transform; accumulator row = (r & 3) + 8 (r >> 2) + 4 lk, column li
  if (!col_active) return;
  const int co = n0 + li;
  if (co >= g.N) return;
  float* Mx = Mo + (long)xhw * R * g.N;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    float* o0 = Mx + (((long)b * g.D + 2 * i) * g.Thw + t0 + mt * 32) * g.N + co;
    float* o1 = o0 + (long)g.Thw * g.N;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int row = (rr & 3) + 8 * (rr >> 2) + 4 * lk;
      if (t0 + mt * 32 + row < g.Thw) {
        const float m0 = acc[0][mt][rr], m1 = acc[1][mt][rr], m2 = acc[2][mt][rr], m3 = acc[3][mt][rr];
        o0[(long)row * g.N] = m0 + m1 + m2;
        o1[(long)row * g.N] = m1 - m2 - m3;
      }
    }
  }
}

// ---- packed weights ----------------------------------------------------------------------------------------------------
// Wp[xi_hw = e * 6 + f][fd][q][kh][n][t] = U[fd][e][f][k = 8q + 4kh + t][n],  U = G_d (x) G43_h (x) G43_w applied to
//   mode 0: w[n][k][kd][kh][kw]           (forward: K = Cin, N = Cout)
//   mode 1: w[k][n] with mirrored taps     (data gradient: K = Cout, N = Cin)
__device__ __forceinline__ void g43(const float* g, int s, float* o, int so) {      // G g, 3 -> 6 (Lavin & Gray F(4,3))
  const float g0 = g[0], g1 = g[s], g2 = g[2 * s];
  o[0] = 0.25f * g0;
  o[so] = -(g0 + g1 + g2) * (1.0f / 6.0f);
  o[2 * so] = -(g0 - g1 + g2) * (1.0f / 6.0f);
  o[3 * so] = g0 * (1.0f / 24.0f) + g1 * (1.0f / 12.0f) + g2 * (1.0f / 6.0f);
  o[4 * so] = g0 * (1.0f / 24.0f) - g1 * (1.0f / 12.0f) + g2 * (1.0f / 6.0f);
  o[5 * so] = g2;
}
__device__ __forceinline__ void g23(const float* g, int s, float* o, int so) {      // G g, 3 -> 4 (F(2,3))
  const float g0 = g[0], g1 = g[s], g2 = g[2 * s];
  o[0] = g0;
  o[so] = 0.5f * (g0 + g1 + g2);
  o[2 * so] = 0.5f * (g0 - g1 + g2);
  o[3 * so] = g2;
}

__global__ void __launch_bounds__(256)
wino_df_pack_kernel(const float* __restrict__ w, float* __restrict__ Wp, int Cout, int Cin, int mode) {
  const int K = mode == 0 ? Cin : Cout, N = mode == 0 ? Cout : Cin;
  const int KPad = (K + 7) & ~7, NPad = (N + 31) & ~31;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= KPad * NPad) return;
  const int n = idx % NPad, k = idx / NPad;
  float u[144];                                   // [fd][e][f]
  if (k < K && n < N) {
    float gk[27];                                 // [kd][kh][kw]
    const int co = mode == 0 ? n : k, ci = mode == 0 ? k : n;
#pragma unroll
    for (int t = 0; t < 27; ++t) gk[t] = w[((size_t)co * Cin + ci) * 27 + (mode == 0 ? t : 26 - t)];
    float a1[54];                                 // [kd][kh][6] after w
#pragma unroll
    for (int p = 0; p < 9; ++p) g43(gk + p * 3, 1, a1 + p * 6, 1);
    float a2[108];                                // [kd][6][6] after h
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int f = 0; f < 6; ++f) g43(a1 + d * 18 + f, 6, a2 + d * 36 + f, 6);
#pragma unroll
    for (int p = 0; p < 36; ++p) g23(a2 + p, 36, u + p, 36);                   // along d: 3 -> 4
  } else {
#pragma unroll
    for (int x = 0; x < 144; ++x) u[x] = 0.0f;
  }
  const int Q = KPad >> 3, q = k >> 3, kh = (k >> 2) & 1, t = k & 3;
#pragma unroll
  for (int fd = 0; fd < 4; ++fd)
#pragma unroll
    for (int xhw = 0; xhw < 36; ++xhw)
      Wp[(((((size_t)xhw * 4 + fd) * Q + q) * 2 + kh) * NPad + n) * 4 + t] = u[fd * 36 + xhw];
}

// ---- weight gradient -----------------------------------------------------------------------------------------------------
// gU[fd][xi_hw][k][n] = sum over (b, depth tile i, hw tile) of v_fd[row][k] * z_fd[row][n]
//   v from the four planes 2i-1..2i+2 of P (as in the forward), z0 = g0, z1 = g0 + g1, z2 = g0 - g1, z3 = -g1 from the two
//   planes 2i, 2i+1 of Zhw = wino43_2d_output_adjoint(gy)  ([36][B*D*Thw][N]).
// The reduction runs over rows, the MFMA k dimension: A fragment lane (li, lk) = v[row 2s + lk][k0 + li], B fragment =
// z[row 2s + lk][n0 + li] -- both operands are read row-wise, 32 consecutive channels per half wave, straight from the
// row-major tensors through LDS (no transposition anywhere).  Workgroup = 4 waves sharing one 32-row slab of P (4 planes x
// 32 rows x KT*32 channels... see below) and Z; wave w owns k-tile kt = w of a 128-channel K block and all NT n-tiles of an
// NB-column block: accumulators 4 fd x NT tiles.  Row slabs of 16 rows are double buffered by global_load_lds.  Partial
// sums over row chunks go to a workspace [nchunks][4][36][K][N] reduced in chunk order by wino_dfw_reduce_kernel (+ G^T).
constexpr int DFW_BR = 16;        // rows per stage
struct DfwGeom {
  int B, D, Thw, K, N;
  int ND, nrowstage;              // depth tiles, stages of DFW_BR rows per plane (ceil(Thw / 16))
  int nkb, nnb;                   // 128-channel K blocks, 64-column N blocks
  int nchunk, stages_per_chunk;   // split of the (b, i, row stage) reduction over workgroups
  int nxi;
};

__global__ void __launch_bounds__(256, 2)
wino_dfw_kernel(const float* __restrict__ P, const float* __restrict__ Z, float* __restrict__ part, DfwGeom g) {
  // LDS: [2 bufs][ P: 4 planes x 16 rows x 128 k | Z: 2 planes x 16 rows x 64 n ]
  extern __shared__ __align__(16) float lds[];
  constexpr int PF = 4 * DFW_BR * 128, ZF = 2 * DFW_BR * 64, SF = PF + ZF;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lk = lane >> 5;
  // XCD-aware order: consecutive LOGICAL ids (column blocks / K blocks of one row chunk: they share the P and Z slabs) run
  // on one XCD and meet in its L2; hardware deals consecutive workgroup ids round-robin over the 8 XCDs
  const int nwg = gridDim.x, xq = nwg >> 3, xr = nwg & 7, xcd = blockIdx.x & 7;
  int id = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
  const int nb = id % g.nnb; id /= g.nnb;
  const int kb = id % g.nkb; id /= g.nkb;
  const int chunk = id % g.nchunk;
  const int xhw = id / g.nchunk;
  const long R = (long)g.B * g.D * g.Thw;
  const float* Px = P + (long)xhw * R * g.K;
  const float* Zx = Z + (long)xhw * R * g.N;
  const int k0 = kb * 128, n0 = nb * 64;
  const int total_stages = g.B * g.ND * g.nrowstage;
  const int s_begin = chunk * g.stages_per_chunk, s_end = min(total_stages, s_begin + g.stages_per_chunk);

  // one stage = P: 4 x 16 rows x 512 B = 32 KiB?  no: 4 planes x 16 rows x 128 ch x 4 B = 32 KiB;  Z: 2 x 16 x 64 x 4 = 8 KiB
  auto issue = [&](int sidx, int buf) {
    const int rs = sidx % g.nrowstage, bi = sidx / g.nrowstage;
    const int i = bi % g.ND, b = bi / g.ND;
    const int t0 = rs * DFW_BR;
    // P: 4 planes x 16 rows x 32 slots(16 B) = 2048 items = 32 wave instructions, 8 per wave
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int j = wave + 4 * e, a = j >> 3, item = (j & 7) * 64 + lane;
      const int row = item >> 5, slot = item & 31;
      const int d = 2 * i - 1 + a, t = t0 + row;
      const int kk = k0 + slot * 4;
      const float* src = (d >= 0 && d < g.D && t < g.Thw && kk < g.K) ? Px + (((long)b * g.D + d) * g.Thw + t) * g.K + kk : kDfZeros;
      __builtin_amdgcn_global_load_lds(src, lds + buf * SF + a * (DFW_BR * 128) + (j & 7) * 256, 16, 0, 0);
    }
    // Z: 2 planes x 16 rows x 16 slots = 512 items = 8 wave instructions, 2 per wave
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int j = wave + 4 * e, a = j >> 2, item = (j & 3) * 64 + lane;
      const int row = item >> 4, slot = item & 15;
      const int t = t0 + row, nn = n0 + slot * 4;
      const float* src = (t < g.Thw && nn < g.N) ? Zx + (((long)b * g.D + 2 * i + a) * g.Thw + t) * g.N + nn : kDfZeros;
      __builtin_amdgcn_global_load_lds(src, lds + buf * SF + PF + a * (DFW_BR * 64) + (j & 3) * 256, 16, 0, 0);
    }
  };

  wf32x16 acc[4][2];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) acc[f][nt][rr] = 0.0f;

  const bool kt_active = k0 + wave * 32 < g.K;         // K = 192: the second 128-channel block has two live k-tiles
  if (s_begin < s_end) issue(s_begin, 0);
  for (int sidx = s_begin; sidx < s_end; ++sidx) {
    const int buf = (sidx - s_begin) & 1;
    if (sidx + 1 < s_end) {
      issue(sidx + 1, buf ^ 1);
      asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const float* pb = lds + buf * SF;
    const float* zb = pb + PF;
    if (kt_active)
#pragma unroll
    for (int rp = 0; rp < DFW_BR / 2; ++rp) {            // one MFMA k-step = 2 rows
      const int row = 2 * rp + lk;
      const int ko = row * 128 + wave * 32 + li;
      const float p0 = pb[0 * (DFW_BR * 128) + ko], p1 = pb[1 * (DFW_BR * 128) + ko];
      const float p2 = pb[2 * (DFW_BR * 128) + ko], p3 = pb[3 * (DFW_BR * 128) + ko];
      const float v0 = p0 - p2, v1 = p1 + p2, v2 = p2 - p1, v3 = p1 - p3;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int zo = row * 64 + nt * 32 + li;
        const float g0 = zb[zo], g1 = zb[DFW_BR * 64 + zo];
        acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, g0, acc[0][nt], 0, 0, 0);
        acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, g0 + g1, acc[1][nt], 0, 0, 0);
        acc[2][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v2, g0 - g1, acc[2][nt], 0, 0, 0);
        acc[3][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v3, -g1, acc[3][nt], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // partial tile: part[chunk][fd][xhw][k][n]
  const int kt = k0 + wave * 32;
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    float* dst = part + ((((size_t)chunk * 4 + f) * g.nxi + xhw) * g.K) * g.N;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int n = n0 + nt * 32 + li;
      if (n >= g.N) continue;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int k = kt + (rr & 3) + 8 * (rr >> 2) + 4 * lk;
        if (k < g.K) dst[(size_t)k * g.N + n] = acc[f][nt][rr];
      }
    }
  }
}

// gw[co][ci][27] = G_d^T G43_h^T G43_w^T ( sum_chunks part[chunk][fd][xhw][ci][co] )
__device__ __forceinline__ void g43t(const float* u, int s, float* o, int so) {     // G^T u, 6 -> 3
  const float u0 = u[0], u1 = u[s], u2 = u[2 * s], u3 = u[3 * s], u4 = u[4 * s], u5 = u[5 * s];
  o[0] = 0.25f * u0 - (u1 + u2) * (1.0f / 6.0f) + (u3 + u4) * (1.0f / 24.0f);
  o[so] = (u2 - u1) * (1.0f / 6.0f) + (u3 - u4) * (1.0f / 12.0f);
  o[2 * so] = -(u1 + u2) * (1.0f / 6.0f) + (u3 + u4) * (1.0f / 6.0f) + u5;
}
__device__ __forceinline__ void g23t(const float* u, int s, float* o, int so) {     // G^T u, 4 -> 3
  const float u0 = u[0], u1 = u[s], u2 = u[2 * s], u3 = u[3 * s];
  o[0] = u0 + 0.5f * (u1 + u2);
  o[so] = 0.5f * (u1 - u2);
  o[2 * so] = 0.5f * (u1 + u2) + u3;
}

__global__ void __launch_bounds__(256)
wino_dfw_reduce_kernel(const float* __restrict__ part, float* __restrict__ gw, int Cout, int Cin, int nchunk) {
  const int idx = blockIdx.x * 256 + threadIdx.x;          // (ci, co), co fastest: coalesced partial reads
  if (idx >= Cin * Cout) return;
  const int co = idx % Cout, ci = idx / Cout;
  float u[144];
  const size_t slab = (size_t)Cin * Cout;
#pragma unroll
  for (int x = 0; x < 144; ++x) u[x] = 0.0f;
  for (int c = 0; c < nchunk; ++c) {                 // chunk order: deterministic
    const float* pc = part + (size_t)c * 144 * slab + idx;
#pragma unroll
    for (int x = 0; x < 144; ++x) u[x] += pc[(size_t)x * slab];
  }
  float a1[72];                                     // [fd][6][3] after w
#pragma unroll
  for (int p = 0; p < 24; ++p) g43t(u + p * 6, 1, a1 + p * 3, 1);
  float a2[36];                                     // [fd][3][3] after h
#pragma unroll
  for (int fd = 0; fd < 4; ++fd)
#pragma unroll
    for (int f = 0; f < 3; ++f) g43t(a1 + fd * 18 + f, 3, a2 + fd * 9 + f, 3);
  float gk[27];
#pragma unroll
  for (int p = 0; p < 9; ++p) g23t(a2 + p, 9, gk + p, 9);                      // along d: 4 -> 3
#pragma unroll
  for (int t = 0; t < 27; ++t) gw[((size_t)co * Cin + ci) * 27 + t] = gk[t];
}

bool df_dims_ok(const ssbev_wino_dims* d, int N) {
  return d && d->B > 0 && d->C > 0 && d->D > 0 && d->H > 0 && d->W > 0 && d->H % 4 == 0 && d->W % 4 == 0 && d->D % 2 == 0 &&
         d->C % DF_BK == 0 && N > 0;
}

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

}  // namespace

extern "C" {

int ssbev_wino43_df_supported(const ssbev_wino_dims* d, int N) { return df_dims_ok(d, N) ? 1 : 0; }

size_t ssbev_wino43_df_packed_elems(int Cout, int Cin) {
  const size_t a = (size_t)((Cin + 7) & ~7) * ((Cout + 31) & ~31), b = (size_t)((Cout + 7) & ~7) * ((Cin + 31) & ~31);
  return 144 * (a > b ? a : b);
}

int ssbev_wino43_df_pack(const float* w, float* Wp, int Cout, int Cin, int mode, ssbev_stream_t stream) {
  if (!w || !Wp || Cout <= 0 || Cin <= 0 || (mode != 0 && mode != 1)) return SSBEV_EINVAL;
  const int K = mode == 0 ? Cin : Cout, N = mode == 0 ? Cout : Cin;
  const size_t total = (size_t)((K + 7) & ~7) * ((N + 31) & ~31);
  hipLaunchKernelGGL(wino_df_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), w, Wp, Cout, Cin, mode);
  return ssbev_launch_status();
}

// P [36][B*D*Thw][K] = ssbev_wino43_2d_input_transform(x), Wp = ssbev_wino43_df_pack(...), Mo [36][B*D*Thw][N];
// d = (B, D, H, W, C = K).  Requires H % 4 == W % 4 == 0, even D, K % 32 == 0 (ssbev_wino43_df_supported).
int ssbev_wino43_df_gemm(const float* P, const float* Wp, float* Mo, const ssbev_wino_dims* d, int N, ssbev_stream_t stream) {
  if (!df_dims_ok(d, N) || !P || !Wp || !Mo) return SSBEV_EINVAL;
  DfGeom g;
  g.B = d->B; g.D = d->D; g.Thw = (d->H / 4) * (d->W / 4); g.K = d->C; g.N = N; g.NPad = (N + 31) & ~31;
  g.ND = d->D / 2;
  g.nrowgrp = (g.Thw + DF_BM - 1) / DF_BM;
  const int ntile = g.NPad / 32;
  // waves per workgroup: the largest of 4, 3, 2 that wastes no column tile (N = 192 -> 3, 128 / 256 / 512 -> 4)
  static const int forced_nw = env_int("SSBEV_DF_NW", 0);
  int nw = ntile % 4 == 0 ? 4 : (ntile % 3 == 0 ? 3 : (ntile % 2 == 0 ? 2 : 4));
  if (forced_nw >= 2 && forced_nw <= 4) nw = forced_nw;
  g.ncolgrp = (ntile + nw - 1) / nw;
  g.NU = g.B * g.ND * g.nrowgrp;
  g.NU8 = (g.NU + 7) / 8;
  g.nxi = 36;
  const size_t lds = (size_t)2 * DF_STAGE_FLOATS * sizeof(float);        // 64 KiB
  auto kern = wino_df_kernel<0>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return SSBEV_ELAUNCH;
  const long nwg = (long)8 * g.nxi * g.NU8 * g.ncolgrp;
  hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(64 * nw), lds, as_stream(stream), P, Wp, Mo, g);
  return ssbev_launch_status();
}

// Weight gradient.  P [36][B*D*Thw][K] (saved by the forward), Z [36][B*D*Thw][N] = ssbev_wino43_2d_output_adjoint(gy),
// gw [N = Cout][K = Cin][27].  Workspace: ssbev_wino43_df_wgrad_workspace bytes.
static int dfw_chunks(const ssbev_wino_dims* d, int N) {
  const int Thw = (d->H / 4) * (d->W / 4);
  const int total_stages = d->B * (d->D / 2) * ((Thw + DFW_BR - 1) / DFW_BR);
  const int nkb = (d->C + 127) / 128, nnb = (N + 63) / 64;
  // enough workgroups for ~3 rounds of 512 slots (2 per CU), but at least 8 stages per chunk
  static const int target = env_int("SSBEV_DFW_WGS", 1536);
  int nchunk = std::max(1, target / (36 * nkb * nnb));
  nchunk = std::min(nchunk, std::max(1, total_stages / 8));
  return nchunk;
}

size_t ssbev_wino43_df_wgrad_workspace(const ssbev_wino_dims* d, int N) {
  if (!df_dims_ok(d, N)) return 0;
  return (size_t)dfw_chunks(d, N) * 144 * d->C * N * sizeof(float);
}

int ssbev_wino43_df_wgrad(const float* P, const float* Z, float* gw, const ssbev_wino_dims* d, int N, void* ws, size_t ws_bytes,
                          ssbev_stream_t stream) {
  if (!df_dims_ok(d, N) || !P || !Z || !gw || !ws) return SSBEV_EINVAL;
  if (N % 4 != 0) return SSBEV_EINVAL;
  if (ws_bytes < ssbev_wino43_df_wgrad_workspace(d, N)) return SSBEV_EWORKSPACE;
  DfwGeom g;
  g.B = d->B; g.D = d->D; g.Thw = (d->H / 4) * (d->W / 4); g.K = d->C; g.N = N;
  g.ND = d->D / 2; g.nrowstage = (g.Thw + DFW_BR - 1) / DFW_BR;
  g.nkb = (g.K + 127) / 128; g.nnb = (N + 63) / 64;
  g.nchunk = dfw_chunks(d, N);
  const int total_stages = g.B * g.ND * g.nrowstage;
  g.stages_per_chunk = (total_stages + g.nchunk - 1) / g.nchunk;
  g.nxi = 36;
  const size_t lds = (size_t)2 * (4 * DFW_BR * 128 + 2 * DFW_BR * 64) * sizeof(float);     // 80 KiB
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(wino_dfw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
      hipSuccess)
    return SSBEV_ELAUNCH;
  hipStream_t st = as_stream(stream);
  const long nwg = (long)g.nxi * g.nchunk * g.nkb * g.nnb;
  hipLaunchKernelGGL(wino_dfw_kernel, dim3((unsigned)nwg), dim3(256), lds, st, P, Z, static_cast<float*>(ws), g);
  hipLaunchKernelGGL(wino_dfw_reduce_kernel, dim3(cdiv((size_t)g.K * N, 256)), dim3(256), 0, st, static_cast<const float*>(ws), gw,
                     N, g.K, g.nchunk);
  return ssbev_launch_status();
}

}  // extern "C"
