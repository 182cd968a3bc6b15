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
//     16-byte slots XOR-swizzled by ((row >> 1) & 7) (r3; r2's key row & 7 left a 2-way bank conflict, see conv_mfma.hip) on the GLOBAL side so that the per-lane ds_read_b128 of the MFMA A operand is
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

#include <cstdio>
#include <vector>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

typedef float wf32x16 __attribute__((ext_vector_type(16)));
__device__ const float kDfZeros[4] = {0.f, 0.f, 0.f, 0.f};

constexpr int DF_BK = 32;       // channels per stage: one 128-byte line per row

struct DfGeom {
  int B, D, Thw, K, N, NPad;
  int ND;         // depth tiles = D / 2
  int nrowgrp;    // ceil(Thw / (32 * MT))
  int ncolgrp;    // column groups of NW * 32
  int NU;         // row tasks per frequency (B * ND * nrowgrp)
  int nxi;        // (h, w) frequencies (36)
  unsigned long long* dbg = nullptr;   // phase clocks (build with -DSSBEV_DF_CLOCKS, run with SSBEV_DF_TIMES=1)
};

typedef float v4f __attribute__((ext_vector_type(4)));

// Weight fragments are loaded by inline asm so that the compiler does not track them: with a global_load_lds in flight
// hipcc drains vmcnt to 0 at the first use of ANY ordinary load result, which serialises the LDS-DMA prefetch of the next
// A slab with the weight stream.  The waits below are counted by hand (vmcnt retires in order) and tied to the fragment
// registers ("+v") so that no MFMA can be scheduled above its wait.
__device__ __forceinline__ void df_load_b(v4f& dst, const float* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void df_wait_b(v4f (&b)[4]) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N) : "memory");
}

// MT = 32-row tiles per wave (workgroup tile = 32 MT rows x NW*32 columns); NW = waves per workgroup
template <int MT, int NW>
__global__ void __launch_bounds__(64 * NW, 2)
wino_df_kernel(const float* __restrict__ P, const float* __restrict__ Wp, float* __restrict__ Mo, DfGeom g) {
  constexpr int BM = 32 * MT;                       // rows per workgroup
  constexpr int PLANE = BM * DF_BK;                 // floats per plane per stage
  constexpr int STAGE = 4 * PLANE;
  constexpr int NI = 4 * BM / 8;                    // global_load_lds instructions per stage (1 KiB each)
  constexpr int IPP = BM / 8;                       // ... per plane
  constexpr int GL = NI / NW;                       // per wave (waves j < NI % NW issue one more: counts below are conservative)
  extern __shared__ __align__(16) float lds[];      // [2 stages][4 planes][BM rows][32 k]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: LDS-DMA bases go to M0 without a waterfall loop
  const int li = lane & 31, lk = lane >> 5;
  // ---- task decode.  Hardware deals consecutive workgroup ids round-robin over the 8 XCDs; the logical task list
  // (xi_hw slowest, row task u, column group fastest) is cut into 8 contiguous ranges, one per XCD: an XCD works through
  // whole frequencies (their weight slab and P planes meet in ITS L2), adjacent depth tiles / column groups run back to back.
  const int nwg = gridDim.x, xq = nwg >> 3, xr = nwg & 7, xcd = blockIdx.x & 7;
  const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (blockIdx.x >> 3);
  const int cg = logical % g.ncolgrp, u = (logical / g.ncolgrp) % g.NU, xhw = logical / (g.ncolgrp * g.NU);
  const int i = u % g.ND, tg = (u / g.ND) % g.nrowgrp, b = u / (g.ND * g.nrowgrp);
  const int t0 = tg * BM;
  const int n0 = (cg * NW + wave) * 32;
  const bool col_active = n0 < g.NPad;
  const long R = (long)g.B * g.D * g.Thw;
  const float* Px = P + (long)xhw * R * g.K;
  const int nst = g.K / DF_BK;

  // ---- A staging: a stage = 4 planes x BM rows x 8 slots of 16 bytes, 64 slots (1 KiB) per wave instruction
  auto issue = [&](int st, int buf) {
#pragma unroll
    for (int e = 0; e < (NI + NW - 1) / NW; ++e) {
      const int j = wave + NW * e;
      if (NI % NW != 0 && j >= NI) break;
      const int a = j / IPP, jj = j % IPP, item = jj * 64 + lane;
      const int row = item >> 3, slot = item & 7;
      const int d = 2 * i - 1 + a;
      const int t = t0 + row;
      const float* src = (d >= 0 && d < g.D && t < g.Thw)
                             ? Px + (((long)b * g.D + d) * g.Thw + t) * g.K + st * DF_BK + ((slot ^ ((row >> 1) & 7)) << 2)
                             : kDfZeros;
      __builtin_amdgcn_global_load_lds(src, lds + buf * STAGE + a * PLANE + jj * 256, 16, 0, 0);
    }
  };

  wf32x16 acc[4][MT];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) acc[f][mt][rr] = 0.0f;

  // packed weights: ((((xhw * 4 + f) * Q + q) * 2 + lk) * NPad + n) * 4 + t
  const int Q = g.K >> 3;
  const size_t fstride = (size_t)Q * 2 * g.NPad * 4;
  const float* wl = Wp + (size_t)xhw * 4 * fstride + ((size_t)lk * g.NPad + (col_active ? n0 : 0) + li) * 4;
  const size_t qstride = (size_t)2 * g.NPad * 4;
  auto load_b = [&](int q, v4f (&bv)[4]) {
#pragma unroll
    for (int f = 0; f < 4; ++f) df_load_b(bv[f], wl + (size_t)f * fstride + (size_t)q * qstride);
  };

  // ---- pipeline.  Per k-step (8 channels): issue the weight fragments of the NEXT step, wait for this step's (issued one
  // step earlier), 8 MT ds_read_b128 + depth transform + 16 MT MFMAs.  The LDS-DMA copies of stage st+1 are issued in
  // k-step 1 of stage st (after that step's weight loads) into the buffer last read in stage st-1 -- every wave has passed
  // this stage's barrier, so nobody reads it any more -- and are forced complete by the in-order wait of k-step 3, two
  // k-steps (~4000 MFMA cycles) later.  One raw s_barrier per stage (__syncthreads() would drain vmcnt at every barrier).
  //   in-flight queue (oldest first) at the wait of     k-step 0: B(q) B(q+1)             -> vmcnt(4)
  //                                                     k-step 1: B(q) B(q+1) A(st+1)     -> vmcnt(4 + GL)
  //                                                     k-step 2: B(q) A(st+1) B(q+1)     -> vmcnt(GL + 4)
  //                                                     k-step 3: B(q) B(q+1)             -> vmcnt(4)  (retires A(st+1) too)
  // The stage body is straight-line code and every fragment has landed (vmcnt(0)) before the loop's back edge: the
  // compiler may copy or spill fragment registers at control-flow joins, and a copy of a register whose load is still in
  // flight would read stale data (checked on the ISA: no v_mov of a fragment register between its load and its wait).
  v4f bb[2][4];
  auto stage = [&](int st, int buf, auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;
    __builtin_amdgcn_s_barrier();                           // stage st's slab is in LDS for every wave
    const float* ab = lds + buf * STAGE;
#pragma unroll
    for (int qq = 0; qq < DF_BK / 8; ++qq) {
      const int q = st * (DF_BK / 8) + qq;
      v4f (&bc)[4] = bb[qq & 1];
      v4f (&bn)[4] = bb[(qq + 1) & 1];
      load_b(min(q + 1, Q - 1), bn);                        // (the very last step re-reads its own fragments: unused)
      if (MORE && qq == 1) issue(st + 1, buf ^ 1);
      if (MORE && (qq == 1 || qq == 2)) df_wait_b<4 + GL>(bc);
      else df_wait_b<4>(bc);
      v4f v[4][MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = mt * 32 + li;
        const int off = row * DF_BK + (((2 * qq + lk) ^ ((row >> 1) & 7)) << 2);
        const v4f p0 = *reinterpret_cast<const v4f*>(ab + 0 * PLANE + off);
        const v4f p1 = *reinterpret_cast<const v4f*>(ab + 1 * PLANE + off);
        const v4f p2 = *reinterpret_cast<const v4f*>(ab + 2 * PLANE + off);
        const v4f p3 = *reinterpret_cast<const v4f*>(ab + 3 * PLANE + off);
        v[0][mt] = p0 - p2;
        v[1][mt] = p1 + p2;
        v[2][mt] = p2 - p1;
        v[3][mt] = p1 - p3;
      }
#define SSBEV_DF_COMP(COMP)                                                                              \
      _Pragma("unroll") for (int f = 0; f < 4; ++f)                                                      \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                  \
        acc[f][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[f][mt].COMP, bc[f].COMP, acc[f][mt], 0, 0, 0);
      SSBEV_DF_COMP(x) SSBEV_DF_COMP(y) SSBEV_DF_COMP(z) SSBEV_DF_COMP(w)
#undef SSBEV_DF_COMP
    }
    df_wait_b<0>(bb[0]);                                    // next stage's first fragments (and this wave's LDS-DMA) landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's LDS reads of `buf` are complete before it moves on
  };
#ifdef SSBEV_DF_CLOCKS
  const unsigned long long c0 = __builtin_readcyclecounter();
#endif
  load_b(0, bb[0]);
  issue(0, 0);
  df_wait_b<0>(bb[0]);
#ifdef SSBEV_DF_CLOCKS
  const unsigned long long c1 = __builtin_readcyclecounter();
#endif
  for (int st = 0; st + 1 < nst; ++st) stage(st, st & 1, std::true_type{});
  stage(nst - 1, (nst - 1) & 1, std::false_type{});
#ifdef SSBEV_DF_CLOCKS
  const unsigned long long c2 = __builtin_readcyclecounter();
  struct ClockOut { unsigned long long* p; unsigned long long a, b, c; int w; ~ClockOut() {} };
#endif
  // ---- epilogue: depth output transform; accumulator row = (r & 3) + 8 (r >> 2) + 4 lk, column li.
  // Round 4: the two output tiles of a wave (BM rows x 32 columns each) go through the wave's share of the (now idle) stage
  // buffers and leave as 16-byte stores, 8 rows x 128 bytes per wave instruction -- the r2 form issued 32 MT four-byte stores
  // per lane behind per-row bound branches and cost 13 k clocks per workgroup (phase clocks, -DSSBEV_DF_CLOCKS), as much as
  // 0.8 of a k-stage, with the matrix pipe idle for this wave.
  __builtin_amdgcn_s_barrier();                             // every wave has finished reading the stage buffers
  if (!col_active) return;
  if (n0 >= g.N) return;
  float* wb = lds + wave * (2 * BM * 32);                   // [2 outputs][BM rows][32 columns]
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int row = mt * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * lk;
      const float m0 = acc[0][mt][rr], m1 = acc[1][mt][rr], m2 = acc[2][mt][rr], m3 = acc[3][mt][rr];
      wb[row * 32 + li] = m0 + m1 + m2;
      wb[(BM + row) * 32 + li] = m1 - m2 - m3;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the wave reads back only what it wrote itself
  float* Mx = Mo + (long)xhw * R * g.N;
  const int cq = lane & 7, r8 = lane >> 3;
  const bool cok = n0 + 4 * cq < g.N;                       // N % 4 == 0 (ssbev_wino43_df_supported)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float* oj = Mx + (((long)b * g.D + 2 * i + j) * g.Thw + t0) * g.N + n0 + 4 * cq;
#pragma unroll
    for (int it = 0; it < BM / 8; ++it) {
      const int row = it * 8 + r8;
      const v4f v = *reinterpret_cast<const v4f*>(wb + (j * BM + row) * 32 + 4 * cq);
      if (cok && t0 + row < g.Thw) *reinterpret_cast<v4f*>(oj + (long)row * g.N) = v;
    }
  }
#ifdef SSBEV_DF_CLOCKS
  if (g.dbg && lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long* o = g.dbg + ((size_t)blockIdx.x * NW + wave) * 4;
    o[0] = c1 - c0; o[1] = c2 - c1; o[2] = __builtin_readcyclecounter() - c2; o[3] = c0;
  }
#endif
}

// (Round 6 built a persistent variant -- resident workgroups walking a tile list, the next tile's first slab and weight fragments
// requested inside the last k-stage, bit-identical output -- on the hypothesis that the 7.1 k-clock prologue per tile is what keeps
// the kernel at 0.6-0.7 of the pipe.  Measured SLOWER: 128 -> 128 0.590 -> 0.636 ms, 384 -> 192 1.92 -> 2.37 ms (its 200 registers
// cost the MT = 1 instances a wave per SIMD), step 70.8 -> 72.9 ms; resident workgroups run in phase, the dispatcher staggers
// one-tile workgroups.  Removed; profiles/r6d_wino_df_persistent_refuted.txt.)
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

// Workgroup = 64 columns x 4 consecutive k (one float4 of the packed layout per column): every thread transforms its
// (k, n) filter, the 144 results go through LDS 36 frequencies at a time and leave as coalesced 16-byte stores (1 KiB per
// wave instruction).  A direct store from the transforming thread would write 4-byte elements at a 16-byte stride.
__global__ void __launch_bounds__(256)
wino_df_pack_kernel(const float* __restrict__ w, float* __restrict__ Wp, int Cout, int Cin, int mode) {
  __shared__ float4 stage[36][64];
  const int K = mode == 0 ? Cin : Cout, N = mode == 0 ? Cout : Cin;
  const int KPad = (K + 7) & ~7, NPad = (N + 31) & ~31;
  const int t = threadIdx.x >> 6, nl = threadIdx.x & 63;            // k = 4 * kq + t, n = 64 * nb + nl
  const int nblocks = (NPad + 63) / 64;
  const int nb = blockIdx.x % nblocks, kq = blockIdx.x / nblocks;
  const int n = nb * 64 + nl, k = kq * 4 + t;
  float u[144];                                   // [fd][e][f]
  if (k < K && n < N) {
    float gk[27];                                 // [kd][kh][kw]
    const int co = mode == 0 ? n : k, ci = mode == 0 ? k : n;
#pragma unroll
    for (int x = 0; x < 27; ++x) gk[x] = w[((size_t)co * Cin + ci) * 27 + (mode == 0 ? x : 26 - x)];
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
  const int Q = KPad >> 3, q = kq >> 1, kh = kq & 1;
#pragma unroll
  for (int fd = 0; fd < 4; ++fd) {
    __syncthreads();
#pragma unroll
    for (int xhw = 0; xhw < 36; ++xhw) reinterpret_cast<float*>(&stage[xhw][nl])[t] = u[fd * 36 + xhw];
    __syncthreads();
    for (int e = threadIdx.x; e < 36 * 64; e += 256) {
      const int xhw = e >> 6, c = e & 63;
      if (nb * 64 + c < NPad)
        *reinterpret_cast<float4*>(Wp + (((((size_t)xhw * 4 + fd) * Q + q) * 2 + kh) * NPad + nb * 64 + c) * 4) = stage[xhw][c];
    }
  }
}

// ---- weight gradient -----------------------------------------------------------------------------------------------------
// gU[fd][xi_hw][k][n] = sum over (b, depth tile i, hw tile) of v_fd[row][k] * z_fd[row][n]
//   v from the four planes 2i-1..2i+2 of P (as in the forward), z0 = g0, z1 = g0 + g1, z2 = g0 - g1, z3 = -g1 from the two
//   planes 2i, 2i+1 of Zhw = wino43_2d_output_adjoint(gy)  ([36][B*D*Thw][N]).
// The reduction runs over rows, the MFMA k dimension: A fragment lane (li, lk) = v[row 2s + lk][k0 + li], B fragment =
// z[row 2s + lk][n0 + li] -- both operands are read row-wise, 32 consecutive channels per half wave, straight from the
// row-major tensors through LDS (no transposition anywhere).  Workgroup = 4 waves in a KW x (4 / KW) grid; a wave owns one
// 32-channel k-tile and NT 32-column n-tiles (accumulators 4 fd x NT tiles):
//   <KW = 4, NT = 3>: 128 k x  96 n   (N = 192, 384: the occupancy-head conv; P leaves L2 once per 96 columns)
//   <KW = 4, NT = 2>: 128 k x  64 n
//   <KW = 2, NT = 1>:  64 k x  64 n   (the 64-channel hourglass layers: all four waves busy)
// Row slabs of 16 rows (4 P planes + 2 Z planes) are double buffered by global_load_lds; the slab of stage s+1 is requested
// right after stage s's barrier.  Partial sums over row chunks go to a workspace [nchunks][4][36][K][N] reduced in chunk
// order (deterministic) by wino_dfw_sum_kernel, then wino_dfw_reduce_kernel applies G^T along the three axes.
struct DfwGeom {
  int B, D, Thw, K, N;
  int ND, nrowstage;              // depth tiles, stages of BR rows per plane (ceil(Thw / BR))
  int nkb, nnb;                   // K blocks of KW*32 channels, N blocks of (4/KW)*NT*32 columns
  int nchunk, stages_per_chunk;   // split of the (b, i, row stage) reduction over workgroups
  int nxi;
};

template <int KW, int NT, int DFW_BR>      // DFW_BR = rows per stage (16; 8 for NT = 3 so that two workgroups fit the 160 KiB of LDS)
__global__ void __launch_bounds__(256, 2)
wino_dfw_kernel(const float* __restrict__ P, const float* __restrict__ Z, float* __restrict__ part, DfwGeom g) {
  constexpr int NWv = 4 / KW;                       // waves along n
  constexpr int PC = KW * 32, ZC = NWv * NT * 32;   // channels / columns per workgroup
  constexpr int PF = 4 * DFW_BR * PC, ZF = 2 * DFW_BR * ZC, SF = PF + ZF;
  constexpr int PI = PF / 256, ZI = ZF / 256;       // global_load_lds instructions per stage (1 KiB each)
  constexpr int PIP = PI / 4, ZIP = ZI / 2;         // ... per plane
  constexpr int PSL = PC / 4, ZSL = ZC / 4;         // 16-byte slots per row
  extern __shared__ __align__(16) float lds[];      // [2 bufs][ P: 4 planes x 16 rows x PC | Z: 2 planes x 16 rows x ZC ]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
  const int li = lane & 31, lk = lane >> 5;
  const int kw = wave % KW, nw = wave / KW;
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
  const int k0 = kb * PC, n0 = nb * ZC;
  const int total_stages = g.B * g.ND * g.nrowstage;
  const int s_begin = chunk * g.stages_per_chunk, s_end = min(total_stages, s_begin + g.stages_per_chunk);

  // Out-of-range rows / planes / channels read a 16-byte zero constant.  32-bit element offsets (one frequency slab holds
  // < 2^31 elements, checked by the host): cheap enough that the compiler selects between the two sources instead of
  // branching around the address math (the copies must stay straight-line code).
  auto issue = [&](int sidx, int buf) {
    // depth tile fastest: consecutive stages share two of their four P planes (same rows) -> L2 hits
    const int i = sidx % g.ND, br = sidx / g.ND;
    const int rs = br % g.nrowstage, b = br / g.nrowstage;
    const int t0 = rs * DFW_BR;
#pragma unroll
    for (int e = 0; e < (PI + 3) / 4; ++e) {
      const int j = wave + 4 * e;
      if (PI % 4 != 0 && j >= PI) break;
      const int a = j / PIP, jj = j % PIP, item = jj * 64 + lane;
      const int row = item / PSL, slot = item % PSL;
      const int d = 2 * i - 1 + a, t = t0 + row;
      const int kk = k0 + slot * 4;
      const bool ok = d >= 0 && d < g.D && t < g.Thw && kk < g.K;
      const int off = ((b * g.D + d) * g.Thw + t) * g.K + kk;
      const float* src = ok ? Px + off : kDfZeros;
      __builtin_amdgcn_global_load_lds(src, lds + buf * SF + a * (DFW_BR * PC) + jj * 256, 16, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < (ZI + 3) / 4; ++e) {
      const int j = wave + 4 * e;
      if (ZI % 4 != 0 && j >= ZI) break;
      const int a = j / ZIP, jj = j % ZIP, item = jj * 64 + lane;
      const int row = item / ZSL, slot = item % ZSL;
      const int t = t0 + row, nn = n0 + slot * 4;
      const bool ok = t < g.Thw && nn < g.N;
      const int off = ((b * g.D + 2 * i + a) * g.Thw + t) * g.N + nn;
      const float* src = ok ? Zx + off : kDfZeros;
      __builtin_amdgcn_global_load_lds(src, lds + buf * SF + PF + a * (DFW_BR * ZC) + jj * 256, 16, 0, 0);
    }
  };

  wf32x16 acc[4][NT];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) acc[f][nt][rr] = 0.0f;

  const bool tile_active = k0 + kw * 32 < g.K && n0 + nw * NT * 32 < g.N;   // K = 192: the second 128-block has two live k-tiles
  if (s_begin < s_end) issue(s_begin, 0);
  for (int sidx = s_begin; sidx < s_end; ++sidx) {
    const int buf = (sidx - s_begin) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // stage sidx has landed (nothing else is in flight here)
    __builtin_amdgcn_s_barrier();                          // ... for every wave; and every wave is done with the other buffer
    const float* pb = lds + buf * SF;
    const float* zb = pb + PF;
    // the other buffer was last read in the previous stage and every wave has passed the barrier: refill it now, the copies
    // have this whole stage's MFMAs to land
    if (sidx + 1 < s_end) issue(sidx + 1, buf ^ 1);
    // (the tile_active test stays OUTSIDE the row-pair loop: one basic block)
    if (tile_active) {
      // operands of row pair rp+1 are read from LDS before the MFMAs of row pair rp are issued (explicit software
      // pipeline: hipcc otherwise puts an lgkmcnt(0) in front of every group of MFMAs)
      float pc[4], gc[NT][2], pn[4], gn[NT][2];
      auto fetch = [&](int rp, float (&pv)[4], float (&gz)[NT][2]) {
        const int row = 2 * rp + lk;
#pragma unroll
        for (int a = 0; a < 4; ++a) pv[a] = pb[a * (DFW_BR * PC) + row * PC + kw * 32 + li];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          gz[nt][0] = zb[row * ZC + (nw * NT + nt) * 32 + li];
          gz[nt][1] = zb[DFW_BR * ZC + row * ZC + (nw * NT + nt) * 32 + li];
        }
      };
      fetch(0, pc, gc);
#pragma unroll
      for (int rp = 0; rp < DFW_BR / 2; ++rp) {            // one MFMA k-step = 2 rows
        if (rp + 1 < DFW_BR / 2) fetch(rp + 1, pn, gn);
        __builtin_amdgcn_sched_barrier(0);                   // keep the reads ABOVE this row pair's MFMAs
        const float v0 = pc[0] - pc[2], v1 = pc[1] + pc[2], v2 = pc[2] - pc[1], v3 = pc[1] - pc[3];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float g0 = gc[nt][0], g1 = gc[nt][1];
          acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, g0, acc[0][nt], 0, 0, 0);
          acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, g0 + g1, acc[1][nt], 0, 0, 0);
          acc[2][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v2, g0 - g1, acc[2][nt], 0, 0, 0);
          acc[3][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v3, -g1, acc[3][nt], 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) pc[a] = pn[a];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { gc[nt][0] = gn[nt][0]; gc[nt][1] = gn[nt][1]; }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  // partial tile: part[chunk][fd][xhw][k][n]
  const int kt = k0 + kw * 32;
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    float* dst = part + ((((size_t)chunk * 4 + f) * g.nxi + xhw) * g.K) * g.N;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n0 + (nw * NT + nt) * 32 + li;
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

// part[0][x][idx] = sum over chunks (ascending: deterministic) of part[c][x][idx]
__global__ void __launch_bounds__(256)
wino_dfw_sum_kernel(float* __restrict__ part, size_t n4, int nchunk) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 a = reinterpret_cast<const float4*>(part)[i];
  for (int c = 1; c < nchunk; ++c) {
    const float4 o = reinterpret_cast<const float4*>(part)[(size_t)c * n4 + i];
    a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
  }
  reinterpret_cast<float4*>(part)[i] = a;
}

__global__ void __launch_bounds__(256)
wino_dfw_reduce_kernel(const float* __restrict__ gU, float* __restrict__ gw, int Cout, int Cin) {
  const int idx = blockIdx.x * 256 + threadIdx.x;          // (ci, co), co fastest: coalesced reads of every frequency
  if (idx >= Cin * Cout) return;
  const int co = idx % Cout, ci = idx / Cout;
  float u[144];
  const size_t slab = (size_t)Cin * Cout;
#pragma unroll
  for (int x = 0; x < 144; ++x) u[x] = gU[(size_t)x * slab + idx];
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
         d->C % DF_BK == 0 && N > 0 && N % 4 == 0;       // (N % 4: 16-byte stores of the output rows)
}

int env_int(const char* name, int dflt) {
  const char* v = ssbev_tune(name);          // tuning hooks only (common.h)
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
  const int KPad = (K + 7) & ~7, NPad = (N + 31) & ~31;
  hipLaunchKernelGGL(wino_df_pack_kernel, dim3((unsigned)((KPad / 4) * ((NPad + 63) / 64))), dim3(256), 0, as_stream(stream), w,
                     Wp, Cout, Cin, mode);
  return ssbev_launch_status();
}

// Template instance <MT, NW> of wino_df_kernel for a problem (also reported by ssbev_wino43_df_instance, so that a profiler can
// attribute launches to kernel symbols)
static void df_instance(const ssbev_wino_dims* d, int N, int& mt, int& nw) {
  const int ntile = ((N + 31) & ~31) / 32, Thw = (d->H / 4) * (d->W / 4), ND = d->D / 2;
  // waves per workgroup: the largest of 4, 3, 2 that wastes no column tile (N = 192 -> 3, 128 / 256 / 512 -> 4)
  static const int forced_nw = env_int("SSBEV_DF_NW", 0), forced_mt = env_int("SSBEV_DF_MT", 0);
  nw = ntile % 4 == 0 ? 4 : (ntile % 3 == 0 ? 3 : (ntile % 2 == 0 ? 2 : 4));
  if (forced_nw >= 2 && forced_nw <= 4) nw = forced_nw;
  const int ncolgrp = (ntile + nw - 1) / nw;
  // 64-row workgroup tiles unless that leaves the chip's 512 workgroup slots (2 per CU) less than four times covered
  mt = 2;
  if ((long)36 * d->B * ND * ((Thw + 63) / 64) * ncolgrp < 2048) mt = 1;      // (r4: 256 -> 256 at 8 x 64 x 64, 1152 workgroups of 64 rows: 0.249 -> 0.235 ms with 32-row tiles)
  // round 6: never pad a plane's hw-tiles by more than 10 %.  The voxel encoder's 256 -> 256 layers run on the 64 x 64 x 8 grid
  // with the SHORT axis last: Thw = 16 x 2 = 32 row tiles per plane, and 64-row workgroup tiles spent half of their MFMAs on
  // rows that do not exist (0.32 ms per launch at "0.38 of the pipe" -- the pipe was busy, with zeros)
  if (((Thw + 63) / 64) * 64 * 10 > ((Thw + 31) / 32) * 32 * 11) mt = 1;
  // three-wave workgroups: two 64 KiB workgroups per CU put 6 waves on 4 SIMDs (2, 2, 1, 1); four 32 KiB ones are balanced
  // (384 -> 192 head conv: 1.98 -> 1.72 ms)
  if (nw == 3) mt = 1;
  // two-wave workgroups (N = 64): at 64 KiB of LDS only two fit a CU = ONE wave per SIMD and nothing hides a wait; 32 KiB
  // tiles put two waves on every SIMD (SSBEV_DF_NW2_MT=2 restores the r2 choice)
  static const int nw2_mt = env_int("SSBEV_DF_NW2_MT", 1);
  if (nw == 2) mt = nw2_mt == 2 ? 2 : 1;
  if (forced_mt == 1 || forced_mt == 2) mt = forced_mt;
  const int key = mt * 10 + nw;
  if (key != 24 && key != 23 && key != 22 && key != 14 && key != 13 && key != 12) { mt = 1; nw = 2; }     // the launcher's default case
}

int ssbev_wino43_df_instance(const ssbev_wino_dims* d, int N) {
  if (!df_dims_ok(d, N)) return 0;
  int mt, nw;
  df_instance(d, N, mt, nw);
  return mt * 10 + nw;
}

// P [36][B*D*Thw][K] = ssbev_wino43_2d_input_transform(x), Wp = ssbev_wino43_df_pack(...), Mo [36][B*D*Thw][N];
// d = (B, D, H, W, C = K).  Requires H % 4 == W % 4 == 0, even D, K % 32 == 0, N % 4 == 0 (ssbev_wino43_df_supported).
int ssbev_wino43_df_gemm(const float* P, const float* Wp, float* Mo, const ssbev_wino_dims* d, int N, ssbev_stream_t stream) {
  if (!df_dims_ok(d, N) || !P || !Wp || !Mo) return SSBEV_EINVAL;
  DfGeom g;
  g.B = d->B; g.D = d->D; g.Thw = (d->H / 4) * (d->W / 4); g.K = d->C; g.N = N; g.NPad = (N + 31) & ~31;
  g.ND = d->D / 2;
  const int ntile = g.NPad / 32;
  int mt, nw;
  df_instance(d, N, mt, nw);
  g.ncolgrp = (ntile + nw - 1) / nw;
  g.nrowgrp = (g.Thw + 32 * mt - 1) / (32 * mt);
  g.NU = g.B * g.ND * g.nrowgrp;
  g.nxi = 36;
  const size_t lds = (size_t)2 * 4 * 32 * mt * DF_BK * sizeof(float);        // 64 KiB (MT = 2) / 32 KiB
  const long nwg = (long)g.nxi * g.NU * g.ncolgrp;
  hipStream_t st = as_stream(stream);
#ifdef SSBEV_DF_CLOCKS
#define DF_CLOCKS_BEGIN(NW_)                                                                                               \
  unsigned long long* dbg_dev = nullptr;                                                                                   \
  const bool dbg_on = ssbev_tune("SSBEV_DF_TIMES") != nullptr;                                                                \
  if (dbg_on) { if (hipMalloc(&dbg_dev, (size_t)nwg * NW_ * 32) != hipSuccess) return SSBEV_ELAUNCH; g.dbg = dbg_dev; }
#define DF_CLOCKS_END(NW_)                                                                                                 \
  if (dbg_on) {                                                                                                            \
    std::vector<unsigned long long> h((size_t)nwg * NW_ * 4);                                                              \
    (void)hipStreamSynchronize(st); (void)hipMemcpy(h.data(), dbg_dev, h.size() * 8, hipMemcpyDeviceToHost); (void)hipFree(dbg_dev); \
    double ph[3] = {0, 0, 0}; unsigned long long t0 = ~0ull, t1 = 0;                                                       \
    for (size_t w = 0; w < (size_t)nwg * NW_; ++w) { for (int i = 0; i < 3; ++i) ph[i] += (double)h[w * 4 + i];            \
      t0 = std::min(t0, h[w * 4 + 3]); t1 = std::max(t1, h[w * 4 + 3] + h[w * 4] + h[w * 4 + 1] + h[w * 4 + 2]); }         \
    const double nw_ = (double)nwg * NW_;                                                                                  \
    fprintf(stderr, "wino_df<mt %d, %d> K=%d N=%d: clocks per wave: prologue %.0f stages %.0f (%d stages) epilogue %.0f | span %.0f, %ld workgroups\n", \
            mt, NW_, g.K, g.N, ph[0] / nw_, ph[1] / nw_, g.K / DF_BK, ph[2] / nw_, (double)(t1 - t0), nwg);                \
  }
#else
#define DF_CLOCKS_BEGIN(NW_)
#define DF_CLOCKS_END(NW_)
#endif
#define SSBEV_DF_LAUNCH(MT_, NW_)                                                                                          \
  do {                                                                                                                     \
    auto kern = wino_df_kernel<MT_, NW_>;                                                                                  \
    if (lds > 64 * 1024 - 1 &&                                                                                             \
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=  \
            hipSuccess)                                                                                                    \
      return SSBEV_ELAUNCH;                                                                                                \
    DF_CLOCKS_BEGIN(NW_)                                                                                                    \
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(64 * NW_), lds, st, P, Wp, Mo, g);                                  \
    DF_CLOCKS_END(NW_)                                                                                                      \
  } while (0)
  switch (mt * 10 + nw) {
    case 24: SSBEV_DF_LAUNCH(2, 4); break;
    case 23: SSBEV_DF_LAUNCH(2, 3); break;
    case 22: SSBEV_DF_LAUNCH(2, 2); break;
    case 14: SSBEV_DF_LAUNCH(1, 4); break;
    case 13: SSBEV_DF_LAUNCH(1, 3); break;
    default: SSBEV_DF_LAUNCH(1, 2); break;
  }
#undef SSBEV_DF_LAUNCH
  return ssbev_launch_status();
}

// Weight gradient.  P [36][B*D*Thw][K] (saved by the forward), Z [36][B*D*Thw][N] = ssbev_wino43_2d_output_adjoint(gy),
// gw [N = Cout][K = Cin][27].  Workspace: ssbev_wino43_df_wgrad_workspace bytes.
struct DfwPlan { int kw, nt, br, pc, zc, nkb, nnb, nchunk, total_stages; };

static DfwPlan dfw_plan(const ssbev_wino_dims* d, int N) {
  DfwPlan p;
  const int K = d->C;
  if (K <= 64) { p.kw = 2; p.nt = 1; p.br = 16; }
  else if (N % 96 == 0) { p.kw = 4; p.nt = 3; p.br = 8; }
  else { p.kw = 4; p.nt = 2; p.br = 16; }
  p.pc = p.kw * 32;
  p.zc = (4 / p.kw) * p.nt * 32;
  p.nkb = (K + p.pc - 1) / p.pc;
  p.nnb = (N + p.zc - 1) / p.zc;
  const int Thw = (d->H / 4) * (d->W / 4);
  p.total_stages = d->B * (d->D / 2) * ((Thw + p.br - 1) / p.br);
  // enough workgroups for ~2 rounds of the 512 slots (2 per CU), but at least 8 stages per chunk
  static const int target = env_int("SSBEV_DFW_WGS", 1024);
  const int per = 36 * p.nkb * p.nnb;
  int nchunk = std::max(1, target / per);
  // whole rounds of the chip's 512 workgroup slots (2 per CU): among the chunk counts around the target take the one whose
  // last round is fullest (384 -> 192: 216 workgroups per chunk, 4 chunks = 1.7 rounds -> 7 chunks = 2.95 rounds, 1.97 -> 1.75 ms)
  {
    double best = -1.0;
    int best_n = nchunk;
    for (int n = std::max(1, nchunk * 2 / 3); n <= nchunk * 2; ++n) {
      const long tot = (long)per * n, rounds = (tot + 511) / 512;
      const double eff = (double)tot / (double)(rounds * 512) - 0.002 * std::abs(n - nchunk);
      if (eff > best) { best = eff; best_n = n; }
    }
    nchunk = best_n;
  }
  p.nchunk = std::min(nchunk, std::max(1, p.total_stages / 8));
  return p;
}

size_t ssbev_wino43_df_wgrad_workspace(const ssbev_wino_dims* d, int N) {
  if (!df_dims_ok(d, N)) return 0;
  return (size_t)dfw_plan(d, N).nchunk * 144 * d->C * N * sizeof(float);
}

int ssbev_wino43_df_wgrad(const float* P, const float* Z, float* gw, const ssbev_wino_dims* d, int N, void* ws, size_t ws_bytes,
                          ssbev_stream_t stream) {
  if (!df_dims_ok(d, N) || !P || !Z || !gw || !ws) return SSBEV_EINVAL;
  if (N % 4 != 0) return SSBEV_EINVAL;
  if ((long)d->B * d->D * (d->H / 4) * (d->W / 4) * std::max(d->C, N) >= (1L << 31)) return SSBEV_EINVAL;   // 32-bit slab offsets
  if (ws_bytes < ssbev_wino43_df_wgrad_workspace(d, N)) return SSBEV_EWORKSPACE;
  const DfwPlan pl = dfw_plan(d, N);
  DfwGeom g;
  g.B = d->B; g.D = d->D; g.Thw = (d->H / 4) * (d->W / 4); g.K = d->C; g.N = N;
  g.ND = d->D / 2; g.nrowstage = (g.Thw + pl.br - 1) / pl.br;
  g.nkb = pl.nkb; g.nnb = pl.nnb;
  g.nchunk = pl.nchunk;
  g.stages_per_chunk = (pl.total_stages + g.nchunk - 1) / g.nchunk;
  g.nxi = 36;
  const size_t lds = (size_t)2 * (4 * pl.br * pl.pc + 2 * pl.br * pl.zc) * sizeof(float);     // 80 / 44 / 48 KiB
  hipStream_t st = as_stream(stream);
  const long nwg = (long)g.nxi * g.nchunk * g.nkb * g.nnb;
#define SSBEV_DFW_LAUNCH(KW_, NT_, BR_)                                                                                       \
  do {                                                                                                                     \
    auto kern = wino_dfw_kernel<KW_, NT_, BR_>;                                                                               \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=  \
        hipSuccess)                                                                                                        \
      return SSBEV_ELAUNCH;                                                                                                \
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), lds, st, P, Z, static_cast<float*>(ws), g);                   \
  } while (0)
  if (pl.kw == 2) SSBEV_DFW_LAUNCH(2, 1, 16);
  else if (pl.nt == 3) SSBEV_DFW_LAUNCH(4, 3, 8);
  else SSBEV_DFW_LAUNCH(4, 2, 16);
#undef SSBEV_DFW_LAUNCH
  const size_t n4 = (size_t)144 * g.K * N / 4;
  if (g.nchunk > 1)
    hipLaunchKernelGGL(wino_dfw_sum_kernel, dim3(cdiv(n4, 256)), dim3(256), 0, st, static_cast<float*>(ws), n4, g.nchunk);
  hipLaunchKernelGGL(wino_dfw_reduce_kernel, dim3(cdiv((size_t)g.K * N, 256)), dim3(256), 0, st, static_cast<const float*>(ws), gw,
                     N, g.K);
  return ssbev_launch_status();
}

}  // extern "C"
