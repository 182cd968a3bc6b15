// Plain fp32 GEMMs of the hot path on the matrix cores, gfx950 -- the layers that ARE matrix products in the channels-last
// layout: kernel == stride transposed convolutions of SECONDFPN3D (second_fpn_3d.py:50-69), wide pointwise convolutions
// (ASPP's 3200 -> 640, BD:312-414), the group contractions of DepthNet's DCN (BD:490-498), the six products of a BRI block
// (attention.py:72-81) and the batched frequency products of the 2-D / weight-streaming Winograd layers.  Round 1 sent
// these to rocBLAS; this file is their own realisation (v_mfma_f32_32x32x2_f32, exact fp32).
//
//   NN:  C[b][m][n] = sum_k A[b][m][k] * B[b][k][n] (+ bias[n])         A, B, C row-major, arbitrary leading dimensions
//   NT:  C[b][m][n] = sum_k A[b][m][k] * W[b][n][k] (+ bias[n])         (the nn.Linear / pointwise-conv layout: no transposes)
//   TN:  C[b][k][n] = sum_r A[b][r][k] * B[b][r][n]                      (reduction over ROWS: weight gradients)
// A pointwise layer is NT forward, NN data gradient, TN weight gradient.  The kernel == stride transposed convolution is the
// same trio around a depth-to-space index map (ssbev_gemm_dims.d2s): NN scatters its result rows to y[(d kd + a, h kh + b,
// w kw + c)][co] for column (tap, co) -- the permute copy of the library realisation never happens --, NT gathers its A rows
// and TN its B rows from there.
//
// Every operand is read ROW-WISE (32 consecutive floats per half wave), so nothing is packed or transposed:
//   NT  B fragment = like A: W[n0 + li][8q + 4lk .. +3], one ds_read_b128 out of a swizzled [BN rows][32 k] slab;
//   NN  A fragment: lane (li, lk) holds A[m0 + li][8q + 4lk .. +3]  -- one ds_read_b128 out of a [rows][32 k] LDS slab whose
//       16-byte slots are XOR-swizzled by ((row >> 1) & 7) on the global side of the copy (conflict-free for the 16-lane groups
//       of ds_read_b128; r2's key row & 7 was a 2-way conflict, see conv_mfma.hip);
//       B fragment: B[8q + 4lk + t][n0 + li], t = 0..3 -- four ds_read_b32 out of a linear [32 k][BN] slab;
//   TN  A fragment: A[2s + lk][k0 + li], B fragment: B[2s + lk][n0 + li] -- one ds_read_b32 each.
// Slabs travel global -> LDS by global_load_lds_dwordx4 (no staging registers), double buffered: the copies of stage s+1 are
// issued right after stage s's barrier and have the whole stage's MFMAs to land; one raw s_barrier per stage; the only
// vmem traffic of a wave is its own LDS-DMA, so `s_waitcnt vmcnt(0)` at the top of a stage is exact.
// Workgroup = 4 waves in a 2 x 2 grid, wave tile 64 x (32 WN): tiles 128 x 128 (WN = 2) or 128 x 64 (WN = 1).
// Workgroup ids are cut into 8 contiguous logical ranges (one per XCD); column blocks of a row block are adjacent.
#include "common.h"

#include <algorithm>
#include <cstdlib>
#include <string>

namespace {

typedef float gf32x16 __attribute__((ext_vector_type(16)));
typedef float gv4f __attribute__((ext_vector_type(4)));
__device__ const float kGemmZeros[4] = {0.f, 0.f, 0.f, 0.f};

struct GemmGeom {
  int M, N, K, batch;
  long lda, ldb, ldc;          // leading dimensions (floats)
  long sa, sb, sc;             // batch strides (floats; 0 = shared operand)
  int mblocks, nblocks;
  int nchunk, rows_per_chunk;  // TN: split of the row reduction (partials [chunk][batch][K][N], then gemm_sum_kernel)
  int relu;
  // depth-to-space map of a kernel == stride transposed convolution (kd = 0: off): row m = (bb, d, h, w) of the coarse grid
  // [D, H, W], "wide" index j = tap * Co + co with tap = (a * kh + b) * kw + c  <->  fine[bb][d kd + a][h kh + b][w kw + c][co].
  // rowoff[m] = float offset of fine[bb][d kd][h kh][w kw][0] (built once per shape by gemm_d2s_rowoff_kernel: no integer
  // divisions in the GEMM loops); a tap adds ((a FH + b) FW + c) * Co.
  int d2s_D, d2s_H, d2s_W, d2s_kd, d2s_kh, d2s_kw, d2s_Co;
  const long* rowoff;
  // TN epilogue C = ep_mul (.) (A^T B - ep_rowsub[row]) (softmax backward of the BRI attention, see ssbev_gemm_dims)
  const float* ep_mul;
  const float* ep_rowsub;
};

__device__ __forceinline__ long d2s_tapoff(const GemmGeom& g, int tap) {
  const int c = tap % g.d2s_kw, ab = tap / g.d2s_kw;
  const int b = ab % g.d2s_kh, a = ab / g.d2s_kh;
  return (((long)a * (g.d2s_H * g.d2s_kh) + b) * (g.d2s_W * g.d2s_kw) + c) * g.d2s_Co;
}

__global__ void gemm_d2s_rowoff_kernel(long* __restrict__ rowoff, int M, int D, int H, int W, int kd, int kh, int kw, int Co) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const int w = m % W; int t = m / W;
  const int h = t % H; t /= H;
  const int d = t % D; const int bb = t / D;
  rowoff[m] = ((((long)bb * (D * kd) + (long)d * kd) * (H * kh) + (long)h * kh) * (W * kw) + (long)w * kw) * Co;
}

__device__ __forceinline__ int xcd_logical(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// ---------------------------------------------------------------------------------------------------------------- NN / NT
// BT = false: B is [K][N] (NN);  BT = true: B is [N][K] (NT).  D2S: 0 = plain, 1 = NN scatters its C rows, 2 = NT gathers its A rows.
// MW = 32-row tiles per wave along M: 2 (workgroup tile 128 rows) or 3 (192 rows: the BRI products have M = D = 192 rows, on
// which 128-row tiles spend a quarter of their MFMAs on padding)
// WGN = waves along N (2: the 2 x 2 wave grid; 1: four waves stacked along M, each owning the full tile width -- round 5: the
// 128 x 160 tile (MW = 1, WN = 5, WGN = 1) cuts the 16 x [1920 x 640 x 640] frequency products of DepthNet's Winograd layers into
// 960 tiles = 1.9 rounds of the 512 workgroup slots instead of 1200 = 2.3 -> three rounds).
// BKT = k depth of a stage (32, or 16: half the LDS per workgroup, so OCC = 3 workgroups per CU fit -- three waves per SIMD
// to hide a stage's barrier and operand reads behind, at twice the barriers).
template <int WN, bool BT, int D2S, int MW = 2, int WGN = 2, int BKT = 32, int OCC = 2>
__global__ void __launch_bounds__(256, OCC)
gemm_nn_kernel(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ bias, float* __restrict__ Cm,
               GemmGeom g) {
  constexpr int WGM = 4 / WGN;
  constexpr int BM = 32 * MW * WGM, BN = 32 * WN * WGN, BK = BKT;
  constexpr int SPR = BK / 4;                                     // 16-byte slots per operand row of a [rows][BK] slab
  constexpr int AF = BM * BK, BF = BK * BN, SF = AF + BF;         // floats per stage
  constexpr int AI = AF / 256, BI = BF / 256;                     // 1 KiB LDS-DMA instructions per stage
  constexpr int AE = (AI + 3) / 4, BE = (BI + 3) / 4;             // ... per wave (the last one may be partial: guarded)
  static_assert(AF % 256 == 0 && BF % 256 == 0, "slabs are whole 1 KiB LDS-DMA pieces");
  // XOR key of a row's 16-byte slots: conflict-free ds_read_b128 over the 16-lane groups (128-byte rows: (row >> 1) & 7, see
  // conv_mfma.hip; 64-byte rows: four rows span the 64 banks, so rows equal modulo 4 must differ in their slot: (row >> 2) & 3)
  auto swz = [](int row) { return BKT == 32 ? ((row >> 1) & 7) : ((row >> 2) & 3); };
  extern __shared__ __align__(16) float lds[];                    // [2][A slab | B slab]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lk = lane >> 5;
  const int wm = wave / WGN, wn = wave % WGN;
  int id = xcd_logical(blockIdx.x, gridDim.x);
  const int nb = id % g.nblocks; id /= g.nblocks;
  const int mb = id % g.mblocks; id /= g.mblocks;
  const int chunk = id % g.nchunk;                 // split-K: k stages [st_begin, st_end) of this chunk
  const int b = id / g.nchunk;
  const int m0 = mb * BM, n0 = nb * BN;
  const float* Ab = A + (long)b * g.sa;
  const float* Bb = B + (long)b * g.sb;
  const int nst_all = (g.K + BK - 1) / BK;
  const int st_begin = chunk * g.rows_per_chunk, st_end = min(nst_all, st_begin + g.rows_per_chunk);

  // Per-lane copy sources are formed ONCE: the rows / columns a lane copies are the same in every stage, a stage only adds
  // k0 (A, NT-B) or k0 * ldb (NN-B).  Out-of-range rows and columns read a 16-byte zero constant for the whole loop; the K
  // tail (K % 32 != 0) is one compare per copy.  Straight-line selects: no branch inside the stage loop.
  const float* ap[AE]; int ak[AE];
#pragma unroll
  for (int e = 0; e < AE; ++e) {                  // A: BM rows x SPR slots of 16 B, slots XOR-swizzled
    const int item = (wave + 4 * e) * 64 + lane, row = item / SPR, slot = item % SPR;
    const int m = m0 + row;
    ak[e] = (slot ^ swz(row)) << 2;
    ap[e] = (wave + 4 * e < AI && m < g.M) ? (D2S == 2 ? Ab + g.rowoff[m] : Ab + (long)m * g.lda) + ak[e] : nullptr;
  }
  const float* bp[BE]; int bk[BE];
#pragma unroll
  for (int e = 0; e < BE; ++e) {
    const int item = (wave + 4 * e) * 64 + lane;
    if (BT) {                                     // W: BN rows x SPR slots, same swizzle
      const int row = item / SPR, slot = item % SPR, n = n0 + row;
      bk[e] = (slot ^ swz(row)) << 2;
      bp[e] = (wave + 4 * e < BI && n < g.N) ? Bb + (long)n * g.ldb + bk[e] : nullptr;
    } else {                                      // B: BK k-rows x BN/4 slots of 16 B, linear
      const int kr = item / (BN / 4), slot = item % (BN / 4), n = n0 + slot * 4;
      bk[e] = kr;
      bp[e] = (wave + 4 * e < BI && n < g.N) ? Bb + (long)kr * g.ldb + n : nullptr;
    }
  }

  auto issue = [&](int st, int buf) {
    const int k0 = st * BK;
    long aoff = k0;
    if (D2S == 2) aoff = d2s_tapoff(g, k0 / g.d2s_Co) + (k0 % g.d2s_Co);    // a 32-wide stage lies inside one tap
    const long boff = BT ? (long)k0 : (long)k0 * g.ldb;
#pragma unroll
    for (int e = 0; e < AE; ++e) {
      if (AI % 4 != 0 && wave + 4 * e >= AI) break;
      const float* src = (ap[e] != nullptr && k0 + ak[e] < g.K) ? ap[e] + aoff : kGemmZeros;
      __builtin_amdgcn_global_load_lds(src, lds + buf * SF + (wave + 4 * e) * 256, 16, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < BE; ++e) {
      if (BI % 4 != 0 && wave + 4 * e >= BI) break;
      const float* src = (bp[e] != nullptr && k0 + bk[e] < g.K) ? bp[e] + boff : kGemmZeros;
      __builtin_amdgcn_global_load_lds(src, lds + buf * SF + AF + (wave + 4 * e) * 256, 16, 0, 0);
    }
  };

  gf32x16 acc[MW][WN];
#pragma unroll
  for (int mt = 0; mt < MW; ++mt)
#pragma unroll
    for (int nt = 0; nt < WN; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

  if (st_begin < st_end) issue(st_begin, 0);
  for (int st = st_begin; st < st_end; ++st) {
    const int buf = (st - st_begin) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // (requesting the next stage's copies behind the first quarter of the MFMAs instead -- what pays in the one-workgroup-per-CU
    // ring kernels -- measured +2 .. +6 % here: the second workgroup of the CU already fills the head of the stage)
    if (st + 1 < st_end) issue(st + 1, buf ^ 1);
    const float* as = lds + buf * SF;
    const float* bs = as + AF;
    // fragments of k-step q+1 are read from LDS before the MFMAs of k-step q are issued (explicit software pipeline)
    gv4f ac[MW], bc[WN], an[MW], bn[WN];
    auto fetch = [&](int q, gv4f (&a)[MW], gv4f (&bf)[WN]) {
#pragma unroll
      for (int mt = 0; mt < MW; ++mt) {
        const int row = (wm * MW + mt) * 32 + li;
        a[mt] = *reinterpret_cast<const gv4f*>(as + row * BK + (((2 * q + lk) ^ swz(row)) << 2));
      }
#pragma unroll
      for (int nt = 0; nt < WN; ++nt) {
        if (BT) {
          const int row = (wn * WN + nt) * 32 + li;
          bf[nt] = *reinterpret_cast<const gv4f*>(bs + row * BK + (((2 * q + lk) ^ swz(row)) << 2));
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) bf[nt][t] = bs[(8 * q + 4 * lk + t) * BN + (wn * WN + nt) * 32 + li];
        }
      }
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
  // epilogue: accumulator row = (r & 3) + 8 (r >> 2) + 4 lk, column li.
  // Every load of the epilogue (bias, the d2s row offsets) is issued AND waited for before the first store: on gfx9 stores
  // count in vmcnt too, and a load pending across the row-bound branches makes hipcc put `s_waitcnt vmcnt(0)` in front of
  // every store -- 64 serialised stores per lane (measured: 70 % -> the kernel's MFMA rate with the loads hoisted).
  float* Cb = Cm + ((long)chunk * g.batch + b) * g.sc;       // split-K partials: [chunk][batch][M][ldc] (bias / ReLU in the sum pass)
  float bv[WN];
  long coloff[WN];
#pragma unroll
  for (int nt = 0; nt < WN; ++nt) {
    const int n = min(n0 + (wn * WN + nt) * 32 + li, g.N - 1);
    const int co = D2S == 1 ? n % g.d2s_Co : n;
    bv[nt] = bias ? bias[co] : 0.0f;
    coloff[nt] = D2S == 1 ? d2s_tapoff(g, n / g.d2s_Co) + co : n;
  }
  long rowbase[MW][16];
#pragma unroll
  for (int mt = 0; mt < MW; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = min(m0 + (wm * MW + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, g.M - 1);
      rowbase[mt][r] = D2S == 1 ? g.rowoff[m] : (long)m * g.ldc;
    }
  if (D2S == 1) {
#pragma unroll
    for (int mt = 0; mt < MW; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(rowbase[mt][r]));     // all offsets have landed: plain registers from here
  }
#pragma unroll
  for (int nt = 0; nt < WN; ++nt) asm volatile("" : "+v"(bv[nt]));
#pragma unroll
  for (int nt = 0; nt < WN; ++nt) {
    const int n = n0 + (wn * WN + nt) * 32 + li;
    if (n >= g.N) continue;
#pragma unroll
    for (int mt = 0; mt < MW; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * MW + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (m < g.M) {
          float v = acc[mt][nt][r] + bv[nt];
          if (g.relu) v = fmaxf(v, 0.0f);
          Cb[rowbase[mt][r] + coloff[nt]] = v;
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------- TN
// C[b][k][n] = sum_r A[b][r][k] B[b][r][n]; workgroup tile 128 k x (64 WN) n, 32 rows per stage.
// KT = 32-row k tiles per wave, WGN = waves along n (2: 2 x 2 waves of 64 x 32 WN -- rounds 2-4; 1: four waves stacked along k, each
// 32 KT x 32 WN: the 128 x 160 tile for the 640-column frequency products, see gemm_nn_kernel)
template <int WN, int KT = 2, int WGN = 2>
__global__ void __launch_bounds__(256, 2)
gemm_tn_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ Cm, GemmGeom g) {
  constexpr int BKK = (4 / WGN) * 32 * KT, BN = WGN * 32 * WN, BR = 32;
  constexpr int AF = BR * BKK, BF = BR * BN, SF = AF + BF;
  constexpr int AI = AF / 256, BI = BF / 256;
  extern __shared__ __align__(16) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lk = lane >> 5;
  const int wk = wave / WGN, wn = wave % WGN;
  int id = xcd_logical(blockIdx.x, gridDim.x);
  const int nb = id % g.nblocks; id /= g.nblocks;
  const int kb = id % g.mblocks; id /= g.mblocks;
  const int chunk = id % g.nchunk;
  const int b = id / g.nchunk;
  const int k0 = kb * BKK, n0 = nb * BN;
  const float* Ab = A + (long)b * g.sa;
  const float* Bb = B + (long)b * g.sb;
  const int r_begin = chunk * g.rows_per_chunk, r_end = min(g.M, r_begin + g.rows_per_chunk);     // g.M = rows
  const int nst = (r_end - r_begin + BR - 1) / BR;

  const bool gather_b = g.d2s_kd > 0;
  // gathered B (k == s deconv weight gradient): the column block lies inside one tap (Co % BN == 0, host-checked); the row
  // offsets of a stage come from the rowoff table, fetched one stage ahead
  const long ncoloff = gather_b ? d2s_tapoff(g, n0 / g.d2s_Co) + (n0 % g.d2s_Co) - n0 : 0;
  long brow[BI / 4];
  auto load_rowoff = [&](int st) {
#pragma unroll
    for (int e = 0; e < BI / 4; ++e) {
      const int r = r_begin + st * BR + ((wave + 4 * e) * 64 + lane) / (BN / 4);
      brow[e] = r < r_end ? (gather_b ? g.rowoff[r] : (long)r * g.ldb) : -1;
    }
  };
  auto issue = [&](int st, int buf) {           // uses brow[] = row offsets of stage st
    const int r0 = r_begin + st * BR;
#pragma unroll
    for (int e = 0; e < AI / 4; ++e) {            // A: 32 rows x 32 slots
      const int j = wave + 4 * e, item = j * 64 + lane;
      const int row = item >> 5, slot = item & 31;
      const int r = r0 + row, k = k0 + slot * 4;
      const float* src = (r < r_end && k < g.K) ? Ab + (long)r * g.lda + k : kGemmZeros;
      __builtin_amdgcn_global_load_lds(src, lds + buf * SF + j * 256, 16, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < BI / 4; ++e) {            // B: 32 rows x BN/4 slots
      const int j = wave + 4 * e, item = j * 64 + lane;
      const int slot = item % (BN / 4);
      const int n = n0 + slot * 4;
      const float* src = (brow[e] >= 0 && n < g.N) ? Bb + brow[e] + ncoloff + n : kGemmZeros;
      __builtin_amdgcn_global_load_lds(src, lds + buf * SF + AF + j * 256, 16, 0, 0);
    }
  };

  gf32x16 acc[KT][WN];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int nt = 0; nt < WN; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[kt][nt][r] = 0.0f;

  if (nst > 0) { load_rowoff(0); issue(0, 0); load_rowoff(1); }
  for (int st = 0; st < nst; ++st) {
    const int buf = st & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // stage st's slabs AND the row offsets of stage st+1
    __builtin_amdgcn_s_barrier();
    const float* as = lds + buf * SF;
    const float* bs = as + AF;
    float ac[KT], bc[WN], an[KT], bn[WN];
    auto fetch = [&](int rp, float (&av)[KT], float (&bv)[WN]) {
      const int row = 2 * rp + lk;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) av[kt] = as[row * BKK + (wk * KT + kt) * 32 + li];
#pragma unroll
      for (int nt = 0; nt < WN; ++nt) bv[nt] = bs[row * BN + (wn * WN + nt) * 32 + li];
    };
    fetch(0, ac, bc);
#pragma unroll
    for (int rp = 0; rp < BR / 2; ++rp) {
      if (rp + 1 < BR / 2) fetch(rp + 1, an, bn);
      __builtin_amdgcn_sched_barrier(0);
      // the next stage's copies (address arithmetic from scratch: a 64-bit multiply per piece) are requested behind the first
      // quarter of this stage's MFMAs, not between the barrier and the first one (round 6: -1 .. -2 %, same bits)
      if (rp == 4 && st + 1 < nst) { issue(st + 1, buf ^ 1); load_rowoff(st + 2); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int nt = 0; nt < WN; ++nt)
          acc[kt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[kt], bc[nt], acc[kt][nt], 0, 0, 0);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) ac[kt] = an[kt];
#pragma unroll
      for (int nt = 0; nt < WN; ++nt) bc[nt] = bn[nt];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float* Cb = Cm + ((long)chunk * g.batch + b) * g.sc;
  if (g.ep_mul) {
    // C = ep_mul (.) (acc - ep_rowsub[row]): EVERY load of the epilogue is issued and waited for before the first store (round 4;
    // a load inside the bound branches makes hipcc wait vmcnt(0) in front of every store, stores count in vmcnt on gfx9: 64
    // serialised load -> store round trips per lane, 0.46 ms for the BRI softmax-backward product against 0.21 ms plain)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {             // two rounds of 16 x (1 + WN) loads (all 32 rows at once do not fit the registers)
      float rs[16], em[WN][16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = min(k0 + (wk * KT + kt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, g.K - 1);
        rs[r] = g.ep_rowsub[(long)b * g.K + k];
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) {
          const int n = min(n0 + (wn * WN + nt) * 32 + li, g.N - 1);
          em[nt][r] = g.ep_mul[(long)b * g.sc + (long)k * g.ldc + n];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        asm volatile("" : "+v"(rs[r]));
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) asm volatile("" : "+v"(em[nt][r]));
      }
#pragma unroll
      for (int nt = 0; nt < WN; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[kt][nt][r] = em[nt][r] * (acc[kt][nt][r] - rs[r]);
    }
  }
#pragma unroll
  for (int nt = 0; nt < WN; ++nt) {
    const int n = n0 + (wn * WN + nt) * 32 + li;
    if (n >= g.N) continue;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = k0 + (wk * KT + kt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (k < g.K) Cb[(long)k * g.ldc + n] = acc[kt][nt][r];
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------- TN, skinny
// K, N <= 64 with a very long row reduction (weight gradients of the 1x1x1 convolutions on the cost volume: 1.47 M rows of
// 32 channels): an HBM-streaming problem -- 256 bytes of operands per MFMA.  No LDS, no sharing: every WAVE walks its own
// contiguous run of rows, a half wave reads one 128-byte row segment per load (8 row pairs in flight), and leaves a partial
// [K][N] tile; gemm_sum_kernel adds the partials in wave order (deterministic).
// QUAD = false (K, N <= 64): the four waves of a workgroup walk DIFFERENT row runs and fold their tiles through LDS.
// QUAD = true  (K, N <= 128): the four waves walk the SAME rows, wave (qk, qn) owns the 64 x 64 quadrant (qk, qn) of the result
//              (each operand row is read by two waves: the second read hits L1).
template <int KT, int NT, bool QUAD>
__global__ void __launch_bounds__(256)
gemm_tn_skinny_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ part, GemmGeom g, int rows_per_run) {
  const int lane = threadIdx.x & 63, li = lane & 31, lk = lane >> 5;
  const int wave = threadIdx.x >> 6;
  const int run = QUAD ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;               // row run of this wave
  const int ko = QUAD ? (wave >> 1) * 64 : 0, no = QUAD ? (wave & 1) * 64 : 0;        // quadrant origin
  const int b = blockIdx.y;
  const float* Ab = A + (long)b * g.sa + ko;
  const float* Bb = B + (long)b * g.sb + no;
  const long r_begin = (long)run * rows_per_run, r_end = min((long)g.M, r_begin + rows_per_run);
  gf32x16 acc[KT][NT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[kt][nt][r] = 0.0f;
  const bool kok[2] = {ko + li < g.K, ko + 32 + li < g.K}, nok[2] = {no + li < g.N, no + 32 + li < g.N};
  constexpr int U = 8;
  for (long r0 = r_begin; r0 < r_end; r0 += 2 * U) {
    float a[U][KT], bb[U][NT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long r = r0 + 2 * u + lk;
      const bool rok = r < r_end;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) a[u][kt] = (rok && kok[kt]) ? Ab[r * g.lda + kt * 32 + li] : 0.0f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bb[u][nt] = (rok && nok[nt]) ? Bb[r * g.ldb + nt * 32 + li] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[kt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][kt], bb[u][nt], acc[kt][nt], 0, 0, 0);
  }
  float* dst = part + ((size_t)blockIdx.x * g.batch + b) * g.K * g.N;               // one partial per workgroup
  if (QUAD) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = no + nt * 32 + li;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = ko + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
          if (k < g.K && n < g.N) dst[(size_t)k * g.N + n] = acc[kt][nt][r];
        }
    }
    return;
  }
  // the four waves of the workgroup fold their tiles through LDS in wave order
  __shared__ float fold[QUAD ? 1 : 4][KT * NT * 1024];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) fold[wave][((kt * NT + nt) * 16 + r) * 64 + lane] = acc[kt][nt][r];
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = nt * 32 + li;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        const int o = ((kt * NT + nt) * 16 + r) * 64 + lane;
        const float v = ((fold[0][o] + fold[1][o]) + fold[2][o]) + fold[3][o];
        if (k < g.K && n < g.N) dst[(size_t)k * g.N + n] = v;
      }
  }
}

// Many partials, few elements (the skinny TN): a workgroup = 64 consecutive elements x 16 contiguous chunk groups; every
// group adds its chunks in ascending order, the 16 group sums are folded in group order (fixed order: deterministic).
__global__ void __launch_bounds__(1024)
gemm_sum_wide_kernel(const float* __restrict__ part, float* __restrict__ out, size_t n, int nchunk) {
  __shared__ float red[16][64];
  const int e = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const size_t i = (size_t)blockIdx.x * 64 + e;
  const int per = (nchunk + 15) / 16, c0 = grp * per, c1 = min(nchunk, c0 + per);
  float a = 0.0f;
  if (i < n)
    for (int c = c0; c < c1; ++c) a += part[(size_t)c * n + i];
  red[grp][e] = a;
  __syncthreads();
  if (grp == 0 && i < n) {
    float v = red[0][e];
#pragma unroll
    for (int q = 1; q < 16; ++q) v += red[q][e];
    out[i] = v;
  }
}

// out[i] = sum over chunks (ascending: deterministic) of part[c][i] (+ bias[i % N], ReLU)
__global__ void __launch_bounds__(256)
gemm_sum_kernel(const float* __restrict__ part, float* __restrict__ out, size_t n, int nchunk, const float* __restrict__ bias,
                int N, int relu, long MN = 0, long ldc = 0, long sc = 0) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float a = part[i];
  for (int c = 1; c < nchunk; ++c) a += part[(size_t)c * n + i];
  if (bias) a += bias[i % N];
  if (relu) a = fmaxf(a, 0.0f);
  // MN > 0: the result is a [batch][M][N] block structure inside a wider buffer (leading dimension ldc, batch stride sc):
  // round 6, the channel-slice outputs of a grouped pointwise layer
  if (MN > 0) {
    const size_t b = i / MN, r = i - b * MN;
    out[b * sc + (r / N) * ldc + r % N] = a;
  } else {
    out[i] = a;
  }
}

bool gemm_ok(const ssbev_gemm_dims* d) {
  if (!(d && d->M > 0 && d->N > 0 && d->K > 0 && d->batch > 0 && d->K % 4 == 0 && d->N % 4 == 0 && d->lda % 4 == 0 &&
        d->ldb % 4 == 0 && d->ldc >= 0))
    return false;
  if (d->d2s_kd > 0)      // column / k blocks must not straddle a tap: Co a multiple of 64 (and of 128 when 128-wide tiles are used)
    return d->d2s_kh > 0 && d->d2s_kw > 0 && d->d2s_D > 0 && d->d2s_H > 0 && d->d2s_W > 0 && d->d2s_Co > 0 && d->d2s_Co % 64 == 0 &&
           d->M % (d->d2s_D * d->d2s_H * d->d2s_W) == 0 && d->d2s_rowoff != nullptr;
  return true;
}

void fill_geom(GemmGeom& g, const ssbev_gemm_dims* d) {
  g.M = d->M; g.N = d->N; g.K = d->K; g.batch = d->batch;
  g.lda = d->lda; g.ldb = d->ldb; g.ldc = d->ldc; g.sa = d->sa; g.sb = d->sb; g.sc = d->sc;
  g.nchunk = 1; g.rows_per_chunk = 0; g.relu = d->relu;
  g.d2s_D = d->d2s_D; g.d2s_H = d->d2s_H; g.d2s_W = d->d2s_W; g.d2s_kd = d->d2s_kd; g.d2s_kh = d->d2s_kh; g.d2s_kw = d->d2s_kw;
  g.d2s_Co = d->d2s_Co;
  g.rowoff = reinterpret_cast<const long*>(d->d2s_rowoff);
  g.ep_mul = nullptr; g.ep_rowsub = nullptr;
}

// 128-wide column tiles unless 64-wide ones pad less (N = 192 -> 3 x 64, N = 160 -> 3 x 64, N = 640 -> 5 x 128)
int pick_wn(int N, int tap_width = 0) {
  const int pad2 = (N + 127) / 128 * 128 - N, pad1 = (N + 63) / 64 * 64 - N;
  if (tap_width && tap_width % 128 != 0) return 1;          // d2s: a column block stays inside one tap
  return pad2 <= pad1 ? 2 : 1;
}

// the chunk count that minimises the rows the busiest CU walks plus the partial-sum pass (see tn_chunks)
int tn_chunks_model(const ssbev_gemm_dims* d, int tiles) {
  const int cmax = std::min(16, std::max(1, d->M / 256));
  int best = 1;
  double best_cost = 1e300;
  for (int c = 1; c <= cmax; ++c) {
    const long per_cu = std::max(2L, ((long)tiles * d->batch * c + 255) / 256);      // (a CU needs two workgroups to run at full rate)
    const double rows = (double)((d->M + c - 1) / c) + 64.0;
    const double cost = per_cu * rows + (c > 1 ? 12.0 * (c + 1) : 0.0);        // (the sum pass reads c and writes 1 result-sized buffers)
    if (cost < best_cost * 0.999) { best_cost = cost; best = c; }
  }
  return best;
}

// round 5: 128 x 160 tiles (four waves stacked along k) + the chunk model for the batched frequency products whose column count is
// a multiple of 160 (16 x [1920 rows -> 640 x 640]: 320 tiles x 4 chunks = 5.0 per CU instead of 400 x 2 = 3.1 -> 4 rounds of
// twice the rows); SSBEV_GEMM_TN_WIDE=0 keeps the 128 x 128 tiles
bool tn_wide(const ssbev_gemm_dims* d) {
  static const bool enabled = !(ssbev_tune("SSBEV_GEMM_TN_WIDE") && atoi(ssbev_tune("SSBEV_GEMM_TN_WIDE")) == 0);
  return enabled && d->batch >= 8 && d->N % 160 == 0 && d->d2s_kd == 0 && !d->ep_mul;
}

int tn_chunks(const ssbev_gemm_dims* d, int tiles) {
  // TN: M = reduction rows.  Enough workgroups for ~2 rounds of the 512 slots, at least 256 rows per chunk.
  // (Round 5 tried the chunk count that minimises the rows the busiest CU walks -- 3 chunks instead of 2 on 16 x [1920 rows ->
  // 640 x 640], 4 instead of 8 on the BRI energy product: +2 ... +9 % alone on the device, but +1.3 ms per step next to the side
  // stream's kernels, where many small workgroups fill the gaps better (profiles/r5_gemm_cfg_probe.txt); removed in round 6.)
  return std::min(std::max(1, 1024 / std::max(1, tiles * d->batch)), std::max(1, d->M / 256));
}

// split-K: when the output tiles alone leave the 512 workgroup slots under-filled and K is deep (BRI's 192 x 7680 products,
// the k4 deconv's data gradient), the k stages are cut into chunks whose partial tiles are summed in chunk order
int nn_chunks(const ssbev_gemm_dims* d, int tiles) {
  if (d->d2s_kd > 0 && d->N != d->K) { /* scatter epilogue writes final values: no split */ }
  const int nst = (d->K + 31) / 32;
  int nchunk = std::max(1, 512 / std::max(1, tiles * d->batch));
  nchunk = std::min(nchunk, std::max(1, nst / 16));          // at least 16 stages (512 k) per chunk
  return std::min(nchunk, 16);
}

// rows of the workgroup tile: 192 when that wastes fewer padded rows than 128 (M = 192: 0 % instead of 25 %)
int pick_bm(const ssbev_gemm_dims* d, int wn) {
  if (d->d2s_kd > 0 || wn != 2) return 128;
  static const bool enabled = !(ssbev_tune("SSBEV_GEMM_BM192") && atoi(ssbev_tune("SSBEV_GEMM_BM192")) == 0);     // A/B hook
  if (!enabled) return 128;
  const long p128 = (long)((d->M + 127) / 128) * 128, p192 = (long)((d->M + 191) / 192) * 192;
  return p192 * 8 <= p128 * 7 ? 192 : 128;              // at least 1/8 fewer MFMA rows
}

// ---- tile configurations of gemm_nn_kernel (plain, non-d2s problems choose among all of them; d2s keeps the 2 x 2 wave grid)
struct NnCfg { int bm, bn, bk, occ; };
constexpr NnCfg kNnCfgs[] = {
    {128, 128, 32, 2},     // 0: 2 x 2 waves of 64 x 64 (rounds 2-4)
    {128, 64, 32, 2},      // 1
    {192, 128, 32, 2},     // 2: 96-row wave tiles (M = 192)
    {128, 160, 32, 2},     // 3: 4 x 1 waves of 32 x 160
    {128, 128, 16, 3},     // 4: three workgroups per CU (probing only: +9 % on 16 x [1920 x 640 x 640], -3 % elsewhere)
};
constexpr int kNnCfgCount = sizeof(kNnCfgs) / sizeof(kNnCfgs[0]);

int nn_forced_cfg() {          // SSBEV_GEMM_CFG=<n>: probing hook (tools/gemm_probe.py)
  const char* e = ssbev_tune("SSBEV_GEMM_CFG");
  if (!e || !*e) return -1;
  const int c = atoi(e);
  return c >= 0 && c < kNnCfgCount ? c : -1;
}

// Model of a configuration's run time on the 256 CUs: padded MFMA work of the busiest CU (the tiles of one launch are equal:
// whole tiles per CU), a per-tile constant (prologue fill + epilogue) worth ~2 k-stages, and the measured efficiency of the
// 4 x 1 wave grid relative to 2 x 2 (profiles/r5_gemm_cfg_probe.txt: the model picks the measured-best configuration on the
// nine shapes of the path)
double nn_cfg_cost(const ssbev_gemm_dims* d, const NnCfg& c, int nchunk) {
  const long tiles = (long)((d->M + c.bm - 1) / c.bm) * ((d->N + c.bn - 1) / c.bn) * d->batch * nchunk;
  const long per_cu = (tiles + 255) / 256;
  const double stages = (double)((d->K + 31) / 32) / nchunk + 2.0;
  return (double)per_cu * c.bm * c.bn * stages * (c.bn == 160 ? 0.93 : 1.0);
}

template <bool BT>
int nn_pick_cfg(const ssbev_gemm_dims* d, int* nchunk_out) {
  int cfg;
  if (d->d2s_kd > 0) {
    cfg = pick_wn(d->N, BT ? 0 : d->d2s_Co) == 2 ? 0 : 1;
  } else {
    const int forced = nn_forced_cfg();
    if (forced >= 0) {
      cfg = forced;
    } else {
      // rounds 2-4 choice (tiles that pad least), unless the model says another shape is >= 5 % cheaper
      const int wn = pick_wn(d->N, 0);
      cfg = wn == 2 ? (pick_bm(d, wn) == 192 ? 2 : 0) : 1;
      static const bool wide = !(ssbev_tune("SSBEV_GEMM_WIDE_TILES") && atoi(ssbev_tune("SSBEV_GEMM_WIDE_TILES")) == 0);     // A/B hook
      if (wide) {
        const NnCfg& c0 = kNnCfgs[cfg];
        const int t0 = ((d->M + c0.bm - 1) / c0.bm) * ((d->N + c0.bn - 1) / c0.bn);
        double best = nn_cfg_cost(d, c0, nn_chunks(d, t0));
        const double base = best;
        for (int k : {1, 3}) {
          const NnCfg& c = kNnCfgs[k];
          const int t = ((d->M + c.bm - 1) / c.bm) * ((d->N + c.bn - 1) / c.bn);
          const double cost = nn_cfg_cost(d, c, nn_chunks(d, t));
          if (cost < 0.95 * base && cost < best) { best = cost; cfg = k; }
        }
      }
    }
  }
  const NnCfg& c = kNnCfgs[cfg];
  const int tiles = ((d->M + c.bm - 1) / c.bm) * ((d->N + c.bn - 1) / c.bn);
  *nchunk_out = (!BT && d->d2s_kd > 0) ? 1 : nn_chunks(d, tiles);
  return cfg;
}

template <bool BT>
size_t nn_workspace(const ssbev_gemm_dims* d) {
  int nchunk;
  nn_pick_cfg<BT>(d, &nchunk);
  return nchunk > 1 ? (size_t)nchunk * d->batch * d->M * d->N * sizeof(float) : 0;
}

template <bool BT>
int launch_nn(const float* A, const float* B, const float* bias, float* Cm, const ssbev_gemm_dims* d, void* ws, size_t ws_bytes,
              hipStream_t st) {
  GemmGeom g;
  fill_geom(g, d);
  int nchunk;
  const int cfg = nn_pick_cfg<BT>(d, &nchunk);
  const NnCfg& c = kNnCfgs[cfg];
  g.mblocks = (d->M + c.bm - 1) / c.bm;
  g.nblocks = (d->N + c.bn - 1) / c.bn;
  g.nchunk = nchunk;
  const int nst = (d->K + c.bk - 1) / c.bk;
  g.rows_per_chunk = (nst + g.nchunk - 1) / g.nchunk;         // k stages per chunk
  float* dst = Cm;
  const float* kbias = bias;
  if (g.nchunk > 1) {
    if (!ws || ws_bytes < nn_workspace<BT>(d)) return SSBEV_EWORKSPACE;
    dst = static_cast<float*>(ws);
    g.ldc = d->N; g.sc = (long)d->M * d->N; g.relu = 0;
    kbias = nullptr;
  }
  const long nwg = (long)g.batch * g.nchunk * g.mblocks * g.nblocks;
  const size_t lds = (size_t)2 * (c.bm * c.bk + c.bk * c.bn) * sizeof(float);
  constexpr int DM = BT ? 2 : 1;          // what d2s means for this form
#define SSBEV_GEMM_LAUNCH(...)                                                                                             \
  do {                                                                                                                     \
    auto kern = gemm_nn_kernel<__VA_ARGS__>;                                                                               \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=  \
        hipSuccess)                                                                                                        \
      return SSBEV_ELAUNCH;                                                                                                \
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), lds, st, A, B, kbias, dst, g);                                \
  } while (0)
  if (d->d2s_kd > 0) {
    if (cfg == 0) SSBEV_GEMM_LAUNCH(2, BT, DM); else SSBEV_GEMM_LAUNCH(1, BT, DM);
  } else {
    switch (cfg) {
      case 0: SSBEV_GEMM_LAUNCH(2, BT, 0); break;
      case 1: SSBEV_GEMM_LAUNCH(1, BT, 0); break;
      case 2: SSBEV_GEMM_LAUNCH(2, BT, 0, 3); break;
      case 3: SSBEV_GEMM_LAUNCH(5, BT, 0, 1, 1); break;
      case 4: SSBEV_GEMM_LAUNCH(2, BT, 0, 2, 2, 16, 3); break;
      default: return SSBEV_EINVAL;
    }
  }
#undef SSBEV_GEMM_LAUNCH
  if (g.nchunk > 1) {
    const size_t n = (size_t)g.batch * d->M * d->N;
    const bool dense = d->ldc == d->N && (d->batch == 1 || d->sc == (long)d->M * d->N);
    hipLaunchKernelGGL(gemm_sum_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, dst, Cm, n, g.nchunk, bias, d->N, d->relu,
                       dense ? 0L : (long)d->M * d->N, (long)d->ldc, (long)d->sc);
  }
  return ssbev_launch_status();
}

}  // namespace

extern "C" {

// rowoff[m] for the d2s modes: M = B * D * H * W coarse voxels (int64 float offsets into the fine tensor)
int ssbev_gemm_d2s_rowoff(int64_t* rowoff, int M, int D, int H, int W, int kd, int kh, int kw, int Co, ssbev_stream_t stream) {
  if (!rowoff || M <= 0 || D <= 0 || H <= 0 || W <= 0 || kd <= 0 || kh <= 0 || kw <= 0 || Co <= 0 || M % (D * H * W) != 0)
    return SSBEV_EINVAL;
  hipLaunchKernelGGL(gemm_d2s_rowoff_kernel, dim3(cdiv(M, 256)), dim3(256), 0, as_stream(stream), reinterpret_cast<long*>(rowoff),
                     M, D, H, W, kd, kh, kw, Co);
  return ssbev_launch_status();
}

// C[b] = A[b] x B[b] (+ bias, ReLU).  M rows, K inner, N columns; lda / ldb / ldc leading dimensions and sa / sb / sc batch
// strides in floats (sb = 0: one B for every batch).  K, N, lda, ldb multiples of 4 (16-byte LDS-DMA granules).
// d2s set: the result row m, column (tap, co) is stored at the depth-to-space position of the fine grid (ldc / sc unused,
// C = the fine tensor of batch element b = 0; bias indexed by co).
size_t ssbev_gemm_nn_workspace(const ssbev_gemm_dims* d) { return gemm_ok(d) ? nn_workspace<false>(d) : 0; }
size_t ssbev_gemm_nt_workspace(const ssbev_gemm_dims* d) { return gemm_ok(d) ? nn_workspace<true>(d) : 0; }

int ssbev_gemm_nn(const float* A, const float* B, const float* bias, float* Cm, const ssbev_gemm_dims* d, void* ws, size_t ws_bytes,
                  ssbev_stream_t stream) {
  if (!gemm_ok(d) || !A || !B || !Cm || d->lda < d->K || d->ldb < d->N) return SSBEV_EINVAL;
  if (d->d2s_kd > 0 ? (d->batch != 1 || d->N != d->d2s_kd * d->d2s_kh * d->d2s_kw * d->d2s_Co) : d->ldc < d->N) return SSBEV_EINVAL;
  return launch_nn<false>(A, B, bias, Cm, d, ws, ws_bytes, as_stream(stream));
}

// C[b][m][n] = sum_k A[b][m][k] W[b][n][k] (+ bias, ReLU): W is [N][K] with leading dimension ldb >= K.
// d2s set: row m of A is GATHERED from the fine grid (K = taps * Co wide; lda / sa unused, batch = 1).
int ssbev_gemm_nt(const float* A, const float* W, const float* bias, float* Cm, const ssbev_gemm_dims* d, void* ws, size_t ws_bytes,
                  ssbev_stream_t stream) {
  if (!gemm_ok(d) || !A || !W || !Cm || d->ldb < d->K || d->ldc < d->N) return SSBEV_EINVAL;
  if (d->d2s_kd > 0 ? (d->batch != 1 || d->K != d->d2s_kd * d->d2s_kh * d->d2s_kw * d->d2s_Co) : d->lda < d->K) return SSBEV_EINVAL;
  return launch_nn<true>(A, W, bias, Cm, d, ws, ws_bytes, as_stream(stream));
}

// C[b][k][n] = sum_r A[b][r][k] B[b][r][n]: d->M = rows (the reduction), d->K x d->N the dense result (ldc = N).
// d2s set: row r of B is gathered from the fine grid (N = taps * Co wide).  Workspace: partial results of the row chunks
// (ssbev_gemm_tn_workspace bytes; 0 when one chunk suffices), summed in chunk order (deterministic).
static bool tn_skinny(const ssbev_gemm_dims* d) { return d->K <= 128 && d->N <= 128 && d->M >= 32768 && d->d2s_kd == 0; }
static bool tn_quad(const ssbev_gemm_dims* d) { return d->K > 64 || d->N > 64; }
static int tn_skinny_wgs(const ssbev_gemm_dims* d) {
  // ~2 workgroups per CU (8 waves streaming per CU), at least 256 rows per run
  const long runs = tn_quad(d) ? 512 : 2048;
  const long w = std::min<long>(runs, std::max<long>(4, d->M / 256));
  return tn_quad(d) ? (int)w : (int)((w + 3) / 4);
}

size_t ssbev_gemm_tn_workspace(const ssbev_gemm_dims* d) {
  if (!gemm_ok(d)) return 0;
  if (tn_skinny(d)) return (size_t)tn_skinny_wgs(d) * d->batch * d->K * d->N * sizeof(float);
  const int wn = pick_wn(d->N, d->d2s_kd > 0 ? d->d2s_Co : 0);
  const bool wide = tn_wide(d);
  const int tiles = ((d->K + 127) / 128) * (wide ? (d->N + 159) / 160 : (d->N + 64 * wn - 1) / (64 * wn));
  const int nchunk = d->ep_mul ? 1 : (wide ? tn_chunks_model(d, tiles) : tn_chunks(d, tiles));
  return nchunk > 1 ? (size_t)nchunk * d->batch * d->K * d->N * sizeof(float) : 0;
}

int ssbev_gemm_tn(const float* A, const float* B, float* Cm, const ssbev_gemm_dims* d, void* ws, size_t ws_bytes,
                  ssbev_stream_t stream) {
  if (!gemm_ok(d) || !A || !B || !Cm || d->lda < d->K) return SSBEV_EINVAL;
  if (d->d2s_kd > 0 ? (d->batch != 1 || d->N != d->d2s_kd * d->d2s_kh * d->d2s_kw * d->d2s_Co) : d->ldb < d->N) return SSBEV_EINVAL;
  GemmGeom g;
  fill_geom(g, d);
  g.ldc = d->N; g.sc = (long)d->K * d->N; g.relu = 0;
  g.ep_mul = d->ep_mul; g.ep_rowsub = d->ep_rowsub;
  if ((d->ep_mul != nullptr) != (d->ep_rowsub != nullptr)) return SSBEV_EINVAL;
  if (d->ep_mul && (tn_skinny(d) || d->d2s_kd > 0)) return SSBEV_EINVAL;
  if (tn_skinny(d)) {
    if (!ws || ws_bytes < ssbev_gemm_tn_workspace(d)) return SSBEV_EWORKSPACE;
    const int wgs = tn_skinny_wgs(d);
    const bool quad = tn_quad(d);
    const long runs = quad ? wgs : (long)wgs * 4;
    const int rows_per_run = (int)(((long)d->M + runs - 1) / runs + 1) / 2 * 2;
    hipStream_t st = as_stream(stream);
    dim3 grid(wgs, d->batch), block(256);
    float* part = static_cast<float*>(ws);
    if (quad) {
      hipLaunchKernelGGL((gemm_tn_skinny_kernel<2, 2, true>), grid, block, 0, st, A, B, part, g, rows_per_run);
    } else {
      const int kt = (d->K + 31) / 32, nt = (d->N + 31) / 32;
      if (kt == 1 && nt == 1) hipLaunchKernelGGL((gemm_tn_skinny_kernel<1, 1, false>), grid, block, 0, st, A, B, part, g, rows_per_run);
      else if (kt == 1) hipLaunchKernelGGL((gemm_tn_skinny_kernel<1, 2, false>), grid, block, 0, st, A, B, part, g, rows_per_run);
      else if (nt == 1) hipLaunchKernelGGL((gemm_tn_skinny_kernel<2, 1, false>), grid, block, 0, st, A, B, part, g, rows_per_run);
      else hipLaunchKernelGGL((gemm_tn_skinny_kernel<2, 2, false>), grid, block, 0, st, A, B, part, g, rows_per_run);
    }
    const size_t n = (size_t)d->batch * d->K * d->N;
    hipLaunchKernelGGL(gemm_sum_wide_kernel, dim3(cdiv(n, 64)), dim3(1024), 0, st, part, Cm, n, wgs);
    return ssbev_launch_status();
  }
  const int wn = pick_wn(d->N, d->d2s_kd > 0 ? d->d2s_Co : 0);
  const bool wide = tn_wide(d);
  const int BN = wide ? 160 : 64 * wn;
  g.mblocks = (d->K + 127) / 128;
  g.nblocks = (d->N + BN - 1) / BN;
  g.nchunk = d->ep_mul ? 1 : (wide ? tn_chunks_model(d, g.mblocks * g.nblocks) : tn_chunks(d, g.mblocks * g.nblocks));      // the fused epilogue needs the complete row reduction
  g.rows_per_chunk = ((d->M + g.nchunk - 1) / g.nchunk + 31) / 32 * 32;
  if (g.nchunk > 1 && (!ws || ws_bytes < ssbev_gemm_tn_workspace(d))) return SSBEV_EWORKSPACE;
  float* dst = g.nchunk > 1 ? static_cast<float*>(ws) : Cm;
  const long nwg = (long)g.batch * g.nchunk * g.mblocks * g.nblocks;
  const size_t lds = (size_t)2 * (32 * 128 + 32 * BN) * sizeof(float);       // 64 / 48 KiB
  hipStream_t st = as_stream(stream);
  if (wide) {
    auto kern = gemm_tn_kernel<5, 1, 1>;
    const size_t ldsw = (size_t)2 * (32 * 128 + 32 * 160) * sizeof(float);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw) != hipSuccess)
      return SSBEV_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), ldsw, st, A, B, dst, g);
  } else if (wn == 2) {
    auto kern = gemm_tn_kernel<2>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return SSBEV_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), lds, st, A, B, dst, g);
  } else {
    hipLaunchKernelGGL(gemm_tn_kernel<1>, dim3((unsigned)nwg), dim3(256), lds, st, A, B, dst, g);
  }
  if (g.nchunk > 1) {
    const size_t n = (size_t)g.batch * d->K * d->N;
    hipLaunchKernelGGL(gemm_sum_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, dst, Cm, n, g.nchunk, (const float*)nullptr, 1, 0);
  }
  return ssbev_launch_status();
}

}  // extern "C"
