// Frustum -> voxel scatter for the LSS "splat" (SURVEY a9-a11), gfx950.
//
// Design (MI355X-first, not a translation of BEVFusion's bev_pool CUDA op):
//   * the scatter is turned into a GATHER over a CSR table (voxel -> ascending point list) that
//     is built on the device by an own two-level stable counting sort of (voxel id, point id) pairs
//     ("CSR build" below; no library primitive in this file); no float atomics anywhere, sums are sequential fp32 in a fixed order => bit-reproducible
//     and bit-identical to the CPU oracle;
//   * one wavefront owns one voxel and streams its C channels with 8-byte lanes (C=128 ->
//     one 512-B line per point, one 512-B store per voxel): the kernel is bound by the
//     [B,nx,ny,nz,C] output write (134 MB at the KITTI config), everything else stays in L2;
//   * Lift (depth x feature outer product, 755 MB at D=192) is fused in: never materialised.
#include "common.h"

#include <cstdlib>

namespace {

// ---------------------------------------------------------------- voxel index (VT:441-451)
__global__ void voxel_index_kernel(const float* __restrict__ geom, int32_t* __restrict__ vox,
                                   int32_t* __restrict__ idx3, long total, int P, int nx, int ny, int nz,
                                   float ox, float oy, float oz, float dx, float dy, float dz) {
  long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= total) return;
  const float gx = geom[3 * p + 0], gy = geom[3 * p + 1], gz = geom[3 * p + 2];
  // fp32 subtract, IEEE divide, no contraction: identical to the torch CPU sequence.
  const float fx = __fdiv_rn(__fsub_rn(gx, ox), dx);
  const float fy = __fdiv_rn(__fsub_rn(gy, oy), dy);
  const float fz = __fdiv_rn(__fsub_rn(gz, oz), dz);
  // trunc-toward-zero then 0 <= i < n  <=>  -1 < f < n  (NaN fails both, as INT64_MIN does).
  const bool kept = (fx > -1.0f) && (fx < (float)nx) && (fy > -1.0f) && (fy < (float)ny) &&
                    (fz > -1.0f) && (fz < (float)nz);
  int v = -1;
  if (kept) {
    const int b = (int)(p / P);
    v = ((b * nx + (int)fx) * ny + (int)fy) * nz + (int)fz;
  }
  vox[p] = v;
  if (idx3) {
    auto sat = [](float f) -> int32_t {
      if (!(f == f)) return INT32_MIN;
      if (f >= 2147483648.0f) return INT32_MAX;
      if (f <= -2147483648.0f) return INT32_MIN;
      return (int32_t)f;
    };
    idx3[3 * p + 0] = sat(fx);
    idx3[3 * p + 1] = sat(fy);
    idx3[3 * p + 2] = sat(fz);
  }
}

// ---------------------------------------------------------------- frustum -> ego frame (BD:123-156)
// The per-point chain of get_geometry -- un-do the image augmentation, lift by depth, camera -> ego, BEV augmentation -- as ONE
// kernel instead of ~22 broadcast ATen passes over the [B, N, D, H, W, 3] point cloud.  Every product and sum is the separately
// rounded fp32 operation of the reference's tensor expression, in its order ((m0 x + m1 y) + m2 z; this file is built without FMA
// contraction), so the points -- and the voxel indices derived from them -- are bit for bit those of the ATen path.
struct GeomArgs {
  const float *frustum, *m1, *t0, *m2, *t2, *tr, *m3, *t3;
  float* out;
  int B, N, D, H, W;
};

__device__ __forceinline__ void mat3_apply(const float* __restrict__ m, float x, float y, float z, float& ox, float& oy,
                                           float& oz) {
  ox = __fadd_rn(__fadd_rn(__fmul_rn(m[0], x), __fmul_rn(m[1], y)), __fmul_rn(m[2], z));
  oy = __fadd_rn(__fadd_rn(__fmul_rn(m[3], x), __fmul_rn(m[4], y)), __fmul_rn(m[5], z));
  oz = __fadd_rn(__fadd_rn(__fmul_rn(m[6], x), __fmul_rn(m[7], y)), __fmul_rn(m[8], z));
}

__global__ void __launch_bounds__(256) frustum_geometry_kernel(GeomArgs a) {
  const long dhw = (long)a.D * a.H * a.W, total = (long)a.B * a.N * dhw;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= total) return;
  const long f = p % dhw;
  const int bn = (int)(p / dhw), b = bn / a.N;
  float x = __fsub_rn(a.frustum[3 * f + 0], a.t0[3 * bn + 0]);
  float y = __fsub_rn(a.frustum[3 * f + 1], a.t0[3 * bn + 1]);
  float z = __fsub_rn(a.frustum[3 * f + 2], a.t0[3 * bn + 2]);
  float u, v, w;
  mat3_apply(a.m1 + 9 * bn, x, y, z, u, v, w);
  x = __fmul_rn(u, w); y = __fmul_rn(v, w); z = w;
  if (a.t2) {
    x = __fsub_rn(x, a.t2[3 * bn + 0]); y = __fsub_rn(y, a.t2[3 * bn + 1]); z = __fsub_rn(z, a.t2[3 * bn + 2]);
  }
  mat3_apply(a.m2 + 9 * bn, x, y, z, u, v, w);
  x = __fadd_rn(u, a.tr[3 * bn + 0]); y = __fadd_rn(v, a.tr[3 * bn + 1]); z = __fadd_rn(w, a.tr[3 * bn + 2]);
  mat3_apply(a.m3 + 9 * b, x, y, z, u, v, w);
  if (a.t3) {
    u = __fadd_rn(u, a.t3[3 * b + 0]); v = __fadd_rn(v, a.t3[3 * b + 1]); w = __fadd_rn(w, a.t3[3 * b + 2]);
  }
  a.out[3 * p + 0] = u; a.out[3 * p + 1] = v; a.out[3 * p + 2] = w;
}

__global__ void coords_to_vox_kernel(const int32_t* __restrict__ coords, int n, int32_t* __restrict__ vox,
                                     int B, int nx, int ny, int nz) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int ix = coords[4 * i], iy = coords[4 * i + 1], iz = coords[4 * i + 2], b = coords[4 * i + 3];
  const bool ok = ix >= 0 && ix < nx && iy >= 0 && iy < ny && iz >= 0 && iz < nz && b >= 0 && b < B;
  vox[i] = ok ? ((b * nx + ix) * ny + iy) * nz + iz : -1;
}

// ---------------------------------------------------------------- CSR build
// The table (starts[NV + 1], order[n]: the points of voxel v in ASCENDING point id) is a stable sort of the points by voxel
// id.  Own two-level counting sort (round 4; hipCUB's three onesweep passes + a global-atomics histogram took 166 us for the
// 1.47 M points of the KITTI frustum):
//   level 1  stable partition by the HIGH digit of the voxel id (<= 11 bits): per-tile digit counts in LDS (csr_hist), one
//            workgroup per digit turns its row of tile counts into output offsets (csr_scan), then every tile places its
//            points (csr_partition): a wave owns a contiguous quarter of the tile and walks it 64 points at a time; the rank
//            of a point among the lanes with the same digit is a popcount of a match mask (one ballot per digit bit), the
//            running offsets live in LDS.  No sorting network, no global atomics on the data path, stable by construction.
//   level 2  one WAVE per high-digit bucket (512 voxels at the KITTI grid, a few thousand points): counts its points per
//            LOW digit in LDS, scans them -- that IS starts[] for its voxels, written as coalesced lines, so the 262 144-bin
//            histogram of the old path is gone -- and places the point ids with the same match-mask ranking.
// Voxel grids above 2^22 cells take more than one level-1 pass (LSD over the high part; bucket bounds by binary search).
constexpr int POOL_LONG = 32;              // lists longer than this are "long": whole-wave path of the gather kernels
constexpr int CSR_T = 256;                 // threads of a level-1 workgroup
constexpr int CSR_WAVES = CSR_T / 64;
constexpr int CSR_MAX_DIGIT_BITS = 11;

int key_bits(long long maxkey) {
  int b = 1;
  while ((1ll << b) <= maxkey) ++b;
  return b;
}

struct CsrPlan {
  int lo_bits, npass, shift[16], nbits[16];
  int items, nblk, nbuckets;
};

// SSBEV_POOL_MAX_DIGIT_BITS (2..11, tests only) narrows the level-1 digit so that small inputs exercise the multi-pass path
int csr_max_digit_bits() {
  const char* e = ssbev_env("SSBEV_POOL_MAX_DIGIT_BITS");
  const int x = e ? atoi(e) : CSR_MAX_DIGIT_BITS;
  return x < 2 ? 2 : (x > CSR_MAX_DIGIT_BITS ? CSR_MAX_DIGIT_BITS : x);
}

bool csr_plan(long long nv, long long n, CsrPlan* p) {
  const int maxd = csr_max_digit_bits();
  const int bits = key_bits(nv - 1);
  int hi = bits - 9 > (bits + 1) / 2 ? bits - 9 : (bits + 1) / 2;
  if (hi > maxd) hi = maxd;
  int lo = bits - hi;
  if (lo > maxd) lo = maxd;
  const int hi_total = bits - lo;
  p->lo_bits = lo;
  p->npass = (hi_total + maxd - 1) / maxd;
  if (p->npass > 16) return false;
  int sh = lo, left = hi_total;
  for (int j = 0; j < p->npass; ++j) {
    const int nb = (left + (p->npass - j) - 1) / (p->npass - j);
    p->shift[j] = sh; p->nbits[j] = nb;
    sh += nb; left -= nb;
  }
  long long items = 8;
  // (measured on the KITTI frustum, 1.47 M points: 8 rows per wave 35.5 us for the four kernels, 16 rows 43 us, 32 rows 60 us
  // -- the placing phase is a chain of dependent 64-point steps, so small tiles / many workgroups win)
  while ((n + CSR_T * items - 1) / (CSR_T * items) > 1024) items *= 2;
  p->items = (int)items;
  p->nblk = (int)((n + CSR_T * items - 1) / (CSR_T * items));
  if (p->nblk < 1) p->nblk = 1;
  p->nbuckets = (int)(((nv - 1) >> lo) + 1);
  return true;
}

// lanes of this wave that hold the same digit (among the valid ones): one ballot per digit bit
__device__ __forceinline__ unsigned long long match_digit(int digit, bool valid, int nbits) {
  unsigned long long m = __ballot(valid);
  for (int b = 0; b < nbits; ++b) {
    const bool bit = (digit >> b) & 1;
    const unsigned long long bal = __ballot(bit);
    m &= bit ? bal : ~bal;
  }
  return m;
}

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(v, off, 64);
    if (lane >= off) v += t;
  }
  return v;
}

// hist[d * nblk + tile] = points of the tile whose digit is d; totals[d] += the same (zeroed by the host)
__global__ void __launch_bounds__(CSR_T)
csr_hist_kernel(const int32_t* __restrict__ keys, int n, int nv, int shift, int nbits, int items, int nblk,
                int32_t* __restrict__ hist, int32_t* __restrict__ totals, int32_t* __restrict__ long_list) {
  extern __shared__ int cnt[];
  const int nd = 1 << nbits, tid = threadIdx.x;
  if (long_list && blockIdx.x == 0 && tid == 0) long_list[0] = 0;      // (the level-2 kernel of this call appends behind it)
  for (int d = tid; d < nd; d += CSR_T) cnt[d] = 0;
  __syncthreads();
  const long long base = (long long)blockIdx.x * CSR_T * items;
  for (int s0 = 0; s0 < items; s0 += 8) {         // items is a multiple of 8: eight loads in flight per thread
    int k[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long long e = base + (long long)(s0 + j) * CSR_T + tid;
      k[j] = e < n ? keys[e] : -1;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if ((unsigned)k[j] < (unsigned)nv) atomicAdd(&cnt[(k[j] >> shift) & (nd - 1)], 1);
  }
  __syncthreads();
  for (int d = tid; d < nd; d += CSR_T) {
    const int c = cnt[d];
    hist[(size_t)d * nblk + blockIdx.x] = c;
    if (c) atomicAdd(&totals[d], c);
  }
}

// one workgroup per digit d: hist[d][*] -> exclusive offsets, starting at the number of points with a smaller digit
__global__ void __launch_bounds__(256)
csr_scan_kernel(int32_t* __restrict__ hist, const int32_t* __restrict__ totals, int nd, int nblk,
                int32_t* __restrict__ bucket_base) {
  __shared__ int wsum[4];
  __shared__ int carry_s;
  const int d = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int part = 0;
  for (int i = tid; i < d; i += 256) part += totals[i];
  part = wave_incl_scan(part, lane);
  if (lane == 63) wsum[wave] = part;
  __syncthreads();
  const int base = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  if (tid == 0) {
    carry_s = base;
    bucket_base[d] = base;
    if (d == nd - 1) bucket_base[nd] = base + totals[d];
  }
  __syncthreads();
  int32_t* row = hist + (size_t)d * nblk;
  for (int c0 = 0; c0 < nblk; c0 += 256) {
    const int i = c0 + tid;
    const int v = i < nblk ? row[i] : 0;
    const int incl = wave_incl_scan(v, lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int before = carry_s;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (i < nblk) row[i] = before + incl - v;
    __syncthreads();
    if (tid == 0) carry_s += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
}

// stable placement of a tile's points behind the offsets of csr_scan_kernel.  FIRST: keys = vox[], id = position.
// KEEP: items == 8, the wave's eight key rows stay in registers between the counting and the placing phase.
template <bool FIRST, bool KEEP>
__global__ void __launch_bounds__(CSR_T)
csr_partition_kernel(const int32_t* __restrict__ keys, const int32_t* __restrict__ ids, int n, int nv, int shift, int nbits,
                     int items, int nblk, const int32_t* __restrict__ offsets, int32_t* __restrict__ keys_out,
                     int32_t* __restrict__ ids_out) {
  extern __shared__ int cnt[];                   // [4 waves][nd]
  const int nd = 1 << nbits, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int d = tid; d < CSR_WAVES * nd; d += CSR_T) cnt[d] = 0;
  __syncthreads();
  const long long sub = ((long long)blockIdx.x * CSR_WAVES + wave) * 64 * items;
  int* mine = cnt + wave * nd;
  int kk[8], ii[8];
  unsigned long long mm[8];                     // KEEP: the match masks of the counting phase serve the placing phase too
  for (int s0 = 0; s0 < items; s0 += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long long e = sub + (s0 + j) * 64 + lane;
      kk[j] = e < n ? keys[e] : -1;
      if (KEEP && !FIRST) ii[j] = e < n ? ids[e] : 0;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool valid = (unsigned)kk[j] < (unsigned)nv;
      const int digit = valid ? (kk[j] >> shift) & (nd - 1) : 0;
      mm[j] = match_digit(digit, valid, nbits);
      if (valid && lane == 63 - __clzll(mm[j])) atomicAdd(&mine[digit], __popcll(mm[j]));
    }
  }
  __syncthreads();
  for (int d = tid; d < nd; d += CSR_T) {
    int run = offsets[(size_t)d * nblk + blockIdx.x];
#pragma unroll
    for (int w = 0; w < CSR_WAVES; ++w) {
      const int c = cnt[w * nd + d];
      cnt[w * nd + d] = run;
      run += c;
    }
  }
  __syncthreads();
  volatile int* run = mine;
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int s0 = 0; s0 < items; s0 += 8) {
    if (!KEEP) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const long long e = sub + (s0 + j) * 64 + lane;
        kk[j] = e < n ? keys[e] : -1;
        if (!FIRST) ii[j] = e < n ? ids[e] : 0;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kk[j];
      const bool valid = (unsigned)k < (unsigned)nv;
      const int digit = valid ? (k >> shift) & (nd - 1) : 0;
      const unsigned long long m = KEEP ? mm[j] : match_digit(digit, valid, nbits);
      int old = 0;
      if (valid) {
        old = run[digit];
        const int pos = old + __popcll(m & below);
        keys_out[pos] = k;
        ids_out[pos] = FIRST ? (int)(sub + (s0 + j) * 64 + lane) : ii[j];
      }
      __builtin_amdgcn_wave_barrier();
      if (valid && lane == 63 - __clzll(m)) run[digit] = old + __popcll(m);
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// keys[n_valid ..] = -1 (between two level-1 passes: the compacted buffer's tail is stale)
__global__ void csr_invalidate_tail_kernel(int32_t* __restrict__ keys, int n, const int32_t* __restrict__ n_valid) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && i >= *n_valid) keys[i] = -1;
}

// level 2: one workgroup of 16 waves per bucket (= value of key >> lo_bits); keys / ids are sorted by bucket, ascending id
// inside a bucket.  Wave w owns the w-th contiguous 16th of the bucket (rounded up to whole 64-point rows): it counts its
// points per low digit, the workgroup turns the [16][2^lo] counts into offsets (scan over digits = starts[], prefix over
// waves inside a digit), and every wave places its points with the match-mask ranking of level 1.  A wave's share of up to
// 256 points stays in registers between the two phases (the KITTI frustum's largest bucket holds 3936 points: one load
// round per wave; a first version with one wave per bucket walked 62 dependent 64-point rows and took 72 us).
constexpr int CSR_BW = 16;
__global__ void __launch_bounds__(CSR_BW * 64)
csr_bucket_kernel(const int32_t* __restrict__ keys, const int32_t* __restrict__ ids, const int32_t* __restrict__ bucket_base,
                  const int32_t* __restrict__ n_valid, int nv, int lo_bits, int32_t* __restrict__ starts,
                  int32_t* __restrict__ order, int32_t* __restrict__ long_list) {
  extern __shared__ int cnt[];                   // [16 waves][1 << lo_bits] | wave sums [16]
  const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nl = 1 << lo_bits;
  int* wsum = cnt + CSR_BW * nl;
  int b0, b1;
  if (bucket_base) {
    b0 = bucket_base[h]; b1 = bucket_base[h + 1];
  } else {                                        // several level-1 passes: bounds of the bucket by binary search
    const int nval = *n_valid;
    int lo = 0, hi = nval;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if ((keys[mid] >> lo_bits) < h) lo = mid + 1; else hi = mid; }
    b0 = lo; hi = nval;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if ((keys[mid] >> lo_bits) <= h) lo = mid + 1; else hi = mid; }
    b1 = lo;
  }
  for (int l = tid; l < CSR_BW * nl; l += CSR_BW * 64) cnt[l] = 0;
  __syncthreads();
  const int share = (((b1 - b0 + CSR_BW - 1) / CSR_BW) + 63) & ~63;      // points per wave, whole rows
  const int w0 = b0 + wave * share, w1 = min(b1, w0 + share);
  const bool keep = share <= 256;
  int* mine = cnt + wave * nl;
  int kk[4], ii[4];
  unsigned long long mm[4];
  for (int e0 = w0; e0 < w1; e0 += 256) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = e0 + j * 64 + lane;
      kk[j] = e < w1 ? keys[e] : -1;
      ii[j] = e < w1 ? ids[e] : 0;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mm[j] = match_digit(kk[j] >= 0 ? kk[j] & (nl - 1) : 0, kk[j] >= 0, lo_bits);
      if (kk[j] >= 0 && lane == 63 - __clzll(mm[j])) atomicAdd(&mine[kk[j] & (nl - 1)], __popcll(mm[j]));
    }
  }
  __syncthreads();
  // per digit: prefix over the waves; then the exclusive scan over the digits (each thread owns `per` consecutive digits)
  const int per = (nl + CSR_BW * 64 - 1) / (CSR_BW * 64);
  int tot[2] = {0, 0};                            // nl <= 2048 = 2 per thread
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int l = tid * per + j;
    if (j < per && l < nl) {
      int run = 0;
#pragma unroll
      for (int w = 0; w < CSR_BW; ++w) { const int c = cnt[w * nl + l]; cnt[w * nl + l] = run; run += c; }
      tot[j] = run;
    }
  }
  const int tsum = tot[0] + tot[1];
  const int incl = wave_incl_scan(tsum, lane);
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int dbase = b0 + incl - tsum;
  for (int w = 0; w < wave; ++w) dbase += wsum[w];
  const long long v0 = (long long)h << lo_bits;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int l = tid * per + j;
    if (j < per && l < nl) {
      if (v0 + l < nv) starts[v0 + l] = dbase;
      // compacted list of the voxels with more than POOL_LONG points (long_list[0] = count, any order): the gather sums them
      // on dedicated waves (ssbev_lift_splat_fwd2)
      if (long_list && tot[j] > POOL_LONG && v0 + l < nv) long_list[1 + atomicAdd(&long_list[0], 1)] = (int32_t)(v0 + l);
#pragma unroll
      for (int w = 0; w < CSR_BW; ++w) cnt[w * nl + l] += dbase;
      dbase += tot[j];
    }
  }
  if (h == (int)gridDim.x - 1 && tid == 0) starts[nv] = b1;
  __syncthreads();
  volatile int* run = mine;
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int e0 = w0; e0 < w1; e0 += 256) {
    if (!keep) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = e0 + j * 64 + lane;
        kk[j] = e < w1 ? keys[e] : -1;
        ii[j] = e < w1 ? ids[e] : 0;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool valid = kk[j] >= 0;
      const int digit = valid ? kk[j] & (nl - 1) : 0;
      const unsigned long long m = keep ? mm[j] : match_digit(digit, valid, lo_bits);
      int old = 0;
      if (valid) {
        old = run[digit];
        order[old + __popcll(m & below)] = ii[j];
      }
      __builtin_amdgcn_wave_barrier();
      if (valid && lane == 63 - __clzll(m)) run[digit] = old + __popcll(m);
      __builtin_amdgcn_wave_barrier();
    }
  }
}


// r2 version of the gather: the same sums in the same order, with the latency chain of a voxel's point list cut from three
// dependent loads per point (order[j] -> depth[p] / feature row -> add) to ~6 load round trips per 64 points:
//   * a wave owns VPW consecutive voxels and fetches their segment bounds with one load;
//   * per 64-point block of a list the lanes fetch order[], depth[] and the row indices in parallel (one coalesced load +
//     one gather), then the block is walked with wave-uniform v_readlane broadcasts and the feature-row loads of U points
//     are issued back to back before their (sequential, fp32, ascending point id => bit-exact) accumulation.
// The first version ran at 1.05 TB/s because the kernel's duration was the serial walk of the longest (near-camera, up to
// ~350 points) lists; empty voxels (88 % of the grid) are a bounds load and a 512-byte zero store.
constexpr int POOL_VPW = 4;      // voxels per wave
constexpr int POOL_U = 16;       // feature rows in flight per wave

template <bool FUSED, int VEC>
__global__ void __launch_bounds__(256)
pool_gather2_kernel(const float* __restrict__ depth, const float* __restrict__ feat,
                    const int32_t* __restrict__ starts, const int32_t* __restrict__ order,
                    float* __restrict__ out, int nv, int C, int P, int vox_per_batch, int N, int D, int HW) {
  const int lane = threadIdx.x & 63;
  const int v0 = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * POOL_VPW;
  if (v0 >= nv) return;
  const int st = (lane <= POOL_VPW && v0 + lane <= nv) ? starts[v0 + lane] : 0;
  for (int i = 0; i < POOL_VPW; ++i) {
    const int v = v0 + i;
    if (v >= nv) break;
    const int s = __builtin_amdgcn_readlane(st, i), e = __builtin_amdgcn_readlane(st, i + 1);
    const int b = v / vox_per_batch;
    for (int cb = 0; cb < C; cb += 64 * VEC) {        // wave-uniform loop: every lane takes part in the broadcasts below
      const int c0 = cb + lane * VEC;
      const bool active = c0 < C;
      float acc[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] = 0.0f;
      for (int base = s; base < e; base += 64) {
        const int cnt = min(64, e - base);
        const int p = lane < cnt ? order[base + lane] : 0;
        int row = p;
        float wgt = 1.0f;
        if (FUSED) {
          const int q = p - b * P;              // point index inside the batch element
          const int n = q / (D * HW);
          row = (b * N + n) * HW + (q % HW);
          wgt = lane < cnt ? depth[p] : 0.0f;
        }
        for (int j0 = 0; j0 < cnt; j0 += POOL_U) {
          float f[POOL_U][VEC], ww[POOL_U];
#pragma unroll
          for (int u = 0; u < POOL_U; ++u) {
            const int jj = min(j0 + u, cnt - 1);          // wave-uniform; the clamped tail re-reads the last row (unused)
            const int r = __builtin_amdgcn_readlane(row, jj);
            ww[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wgt), jj));
            const float* src = feat + (size_t)r * C + (active ? c0 : 0);
            if (VEC == 4) {
              const float4 t = *reinterpret_cast<const float4*>(src);
              f[u][0] = t.x; f[u][1 % VEC] = t.y; f[u][2 % VEC] = t.z; f[u][3 % VEC] = t.w;
            } else if (VEC == 2) {
              const float2 t = *reinterpret_cast<const float2*>(src);
              f[u][0] = t.x; f[u][1 % VEC] = t.y;
            } else {
              f[u][0] = src[0];
            }
          }
#pragma unroll
          for (int u = 0; u < POOL_U; ++u) {
            if (j0 + u < cnt) {
#pragma unroll
              for (int k = 0; k < VEC; ++k) acc[k] = __fadd_rn(acc[k], FUSED ? __fmul_rn(ww[u], f[u][k]) : f[u][k]);
            }
          }
        }
      }
      if (!active) continue;
      float* dst = out + (size_t)v * C + c0;
      if (VEC == 4) {
        *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1 % VEC], acc[2 % VEC], acc[3 % VEC]);
      } else if (VEC == 2) {
        *reinterpret_cast<float2*>(dst) = make_float2(acc[0], acc[1 % VEC]);
      } else {
        dst[0] = acc[0];
      }
    }
  }
}


// Fifth layout = the third one plus a WIDE path for long lists.  What bounds gather3 is not its thousands of short voxels but
// the handful of near-camera ones: the longest list of the KITTI frustum has 342 points, a 16-lane group keeps 8 feature rows
// in flight, so that one voxel is a serial chain of 43 load latencies (~43 us of the kernel's 66).  The additions themselves
// must stay sequential (ascending point id: the oracle's order, bit for bit) but they are cheap; the loads are not ordered.
// Lists longer than POOL_LONG points are therefore summed by the WHOLE wave: a lane owns C / 64 channels, one instruction
// loads one feature row, and 32 rows are in flight -- a quarter of the registers per row, four times the depth.
// (Measured, tools/gather_probe.py: 66 -> 61 us.  The near-camera voxels are NEIGHBOURS, so a wave usually owns four long lists
// and walks them one after the other; with lists cut to <= 64 points the kernel takes 58 us, to <= 1 point 51 us, and the bare
// 134 MB zero store 28 us.  A 79-register variant (4 rows per batch, 6 waves per SIMD) brings the short lists to 34-40 us but
// then spends 72 us on the real CSR: the long lists want registers, the short ones want occupancy, one kernel has one budget.)

template <bool FUSED, int NV4>
__global__ void __launch_bounds__(256)
pool_gather5_kernel(const float* __restrict__ depth, const float* __restrict__ feat,
                    const int32_t* __restrict__ starts, const int32_t* __restrict__ order,
                    float* __restrict__ out, int nv, int P, int vox_per_batch, int N, int D, int HW) {
  constexpr int C = 64 * NV4, U = 8, UW = 32;
  const int lane = threadIdx.x & 63, gl = lane & 15, gbase = lane & 48;
  const int v = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 4 + (lane >> 4);
  const bool vok = v < nv;
  const int s = vok ? starts[v] : 0, e0 = vok ? starts[v + 1] : 0;
  const bool is_long = e0 - s > POOL_LONG;
  const int e = is_long ? s : e0;                          // long lists are skipped here and summed by the whole wave below
  const int b = vok ? v / vox_per_batch : 0;
  float4 acc[NV4];
#pragma unroll
  for (int k = 0; k < NV4; ++k) acc[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  int p_next = (s + gl < e) ? order[s + gl] : 0;
  for (int base = s; base < e; base += 16) {
    const int cnt = min(16, e - base);
    const int p = p_next;
    p_next = (base + 16 + gl < e) ? order[base + 16 + gl] : 0;
    int row = p;
    float wgt = 1.0f;
    if (FUSED) {
      const int q = p - b * P;
      const int n = q / (D * HW);
      row = (b * N + n) * HW + (q % HW);
      wgt = gl < cnt ? depth[p] : 0.0f;
    }
    for (int j0 = 0; j0 < cnt; j0 += U) {
      float4 f[U][NV4];
      float ww[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int jj = gbase + min(j0 + u, cnt - 1);
        const int r = __shfl(row, jj, 64);
        ww[u] = __shfl(wgt, jj, 64);
        const float4* src = reinterpret_cast<const float4*>(feat + (size_t)r * C) + gl * NV4;
#pragma unroll
        for (int k = 0; k < NV4; ++k) f[u][k] = src[k];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (j0 + u < cnt) {
#pragma unroll
          for (int k = 0; k < NV4; ++k) {
            if (FUSED) {
              acc[k].x = __fadd_rn(acc[k].x, __fmul_rn(ww[u], f[u][k].x));
              acc[k].y = __fadd_rn(acc[k].y, __fmul_rn(ww[u], f[u][k].y));
              acc[k].z = __fadd_rn(acc[k].z, __fmul_rn(ww[u], f[u][k].z));
              acc[k].w = __fadd_rn(acc[k].w, __fmul_rn(ww[u], f[u][k].w));
            } else {
              acc[k].x = __fadd_rn(acc[k].x, f[u][k].x);
              acc[k].y = __fadd_rn(acc[k].y, f[u][k].y);
              acc[k].z = __fadd_rn(acc[k].z, f[u][k].z);
              acc[k].w = __fadd_rn(acc[k].w, f[u][k].w);
            }
          }
        }
      }
    }
  }
  if (vok && !is_long) {
    float4* dst = reinterpret_cast<float4*>(out + (size_t)v * C) + gl * NV4;
#pragma unroll
    for (int k = 0; k < NV4; ++k) dst[k] = acc[k];
  }
  // ---- long lists: one voxel at a time on all 64 lanes, lane = NV4 consecutive channels
  unsigned long long todo = __ballot(is_long && gl == 0);
  while (todo) {
    const int src_lane = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int vs = __shfl(s, src_lane, 64), ve = __shfl(e0, src_lane, 64), vv = __shfl(v, src_lane, 64);
    const int vb = __shfl(b, src_lane, 64);
    float a[NV4];
#pragma unroll
    for (int k = 0; k < NV4; ++k) a[k] = 0.0f;
    for (int base = vs; base < ve; base += 64) {
      const int cnt = min(64, ve - base);
      const int p = lane < cnt ? order[base + lane] : 0;
      int row = p;
      float wgt = 1.0f;
      if (FUSED) {
        const int q = p - vb * P;
        const int n = q / (D * HW);
        row = (vb * N + n) * HW + (q % HW);
        wgt = lane < cnt ? depth[p] : 0.0f;
      }
      for (int j0 = 0; j0 < cnt; j0 += UW) {
        float f[UW][NV4], ww[UW];
#pragma unroll
        for (int u = 0; u < UW; ++u) {
          const int jj = min(j0 + u, cnt - 1);
          const int r = __shfl(row, jj, 64);
          ww[u] = __shfl(wgt, jj, 64);
          const float* src = feat + (size_t)r * C + lane * NV4;
          if (NV4 == 4) {
            const float4 t = *reinterpret_cast<const float4*>(src);
            f[u][0] = t.x; f[u][1 % NV4] = t.y; f[u][2 % NV4] = t.z; f[u][3 % NV4] = t.w;
          } else if (NV4 == 2) {
            const float2 t = *reinterpret_cast<const float2*>(src);
            f[u][0] = t.x; f[u][1 % NV4] = t.y;
          } else {
            f[u][0] = src[0];
          }
        }
#pragma unroll
        for (int u = 0; u < UW; ++u) {
          if (j0 + u < cnt) {
#pragma unroll
            for (int k = 0; k < NV4; ++k) a[k] = __fadd_rn(a[k], FUSED ? __fmul_rn(ww[u], f[u][k]) : f[u][k]);
          }
        }
      }
    }
    float* dst = out + (size_t)vv * C + lane * NV4;
#pragma unroll
    for (int k = 0; k < NV4; ++k) dst[k] = a[k];
  }
}

// Sixth layout (round 4) = the fifth one split by ROLE over a compacted list of the long voxels (ssbev_pool_prepare2).  The
// r3 experiment (profiles/r3y_gather_role_split_refuted.txt) showed the short lists alone can be written in 32.5 us at 48 VGPRs /
// 8 waves per SIMD, and that the long ones need a work list: found by scanning starts[], the ~1100 long voxels of the KITTI
// frustum are neighbours and pile up on a few workgroups.  pool_gather_long_kernel: wave i of a fixed pool takes entries i,
// i + nwaves, ... of the list -- one long voxel on all 64 lanes, 32 feature rows in flight (the whole-wave path of gather5);
// pool_gather_short_kernel: gather3's four voxels per wave, long voxels skipped.  Same sums in the same order: bit-exact.
template <bool FUSED, int NV4>
__global__ void __launch_bounds__(256)
pool_gather_long_kernel(const float* __restrict__ depth, const float* __restrict__ feat, const int32_t* __restrict__ starts,
                        const int32_t* __restrict__ order, const int32_t* __restrict__ long_list, float* __restrict__ out,
                        int P, int vox_per_batch, int N, int D, int HW) {
  constexpr int C = 64 * NV4, UW = 32;
  const int lane = threadIdx.x & 63;
  const int nwaves = (int)((size_t)gridDim.x * blockDim.x >> 6);
  const int nlong = long_list[0];
  for (int i = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6); i < nlong; i += nwaves) {
    const int vv = long_list[1 + i];
    const int vs = starts[vv], ve = starts[vv + 1], vb = vv / vox_per_batch;
    float a[NV4];
#pragma unroll
    for (int k = 0; k < NV4; ++k) a[k] = 0.0f;
    for (int base = vs; base < ve; base += 64) {
      const int cnt = min(64, ve - base);
      const int p = lane < cnt ? order[base + lane] : 0;
      int row = p;
      float wgt = 1.0f;
      if (FUSED) {
        const int q = p - vb * P;
        const int n = q / (D * HW);
        row = (vb * N + n) * HW + (q % HW);
        wgt = lane < cnt ? depth[p] : 0.0f;
      }
      for (int j0 = 0; j0 < cnt; j0 += UW) {
        float f[UW][NV4], ww[UW];
#pragma unroll
        for (int u = 0; u < UW; ++u) {
          const int jj = min(j0 + u, cnt - 1);
          const int r = __shfl(row, jj, 64);
          ww[u] = __shfl(wgt, jj, 64);
          const float* src = feat + (size_t)r * C + lane * NV4;
          if (NV4 == 4) {
            const float4 t = *reinterpret_cast<const float4*>(src);
            f[u][0] = t.x; f[u][1 % NV4] = t.y; f[u][2 % NV4] = t.z; f[u][3 % NV4] = t.w;
          } else if (NV4 == 2) {
            const float2 t = *reinterpret_cast<const float2*>(src);
            f[u][0] = t.x; f[u][1 % NV4] = t.y;
          } else {
            f[u][0] = src[0];
          }
        }
#pragma unroll
        for (int u = 0; u < UW; ++u) {
          if (j0 + u < cnt) {
#pragma unroll
            for (int k = 0; k < NV4; ++k) a[k] = __fadd_rn(a[k], FUSED ? __fmul_rn(ww[u], f[u][k]) : f[u][k]);
          }
        }
      }
    }
    float* dst = out + (size_t)vv * C + lane * NV4;
#pragma unroll
    for (int k = 0; k < NV4; ++k) dst[k] = a[k];
  }
}

template <bool FUSED, int NV4>
__global__ void __launch_bounds__(256)
pool_gather_short_kernel(const float* __restrict__ depth, const float* __restrict__ feat,
                         const int32_t* __restrict__ starts, const int32_t* __restrict__ order,
                         float* __restrict__ out, int nv, int P, int vox_per_batch, int N, int D, int HW) {
  constexpr int C = 64 * NV4, U = 4;
  const int lane = threadIdx.x & 63, gl = lane & 15, gbase = lane & 48;
  const int v = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 4 + (lane >> 4);
  const bool vok = v < nv;
  const int s = vok ? starts[v] : 0, e0 = vok ? starts[v + 1] : 0;
  const bool is_long = e0 - s > POOL_LONG;
  const int e = is_long ? s : e0;                          // long lists belong to pool_gather_long_kernel
  const int b = vok ? v / vox_per_batch : 0;
  float4 acc[NV4];
#pragma unroll
  for (int k = 0; k < NV4; ++k) acc[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  int p_next = (s + gl < e) ? order[s + gl] : 0;
  for (int base = s; base < e; base += 16) {
    const int cnt = min(16, e - base);
    const int p = p_next;
    p_next = (base + 16 + gl < e) ? order[base + 16 + gl] : 0;
    int row = p;
    float wgt = 1.0f;
    if (FUSED) {
      const int q = p - b * P;
      const int n = q / (D * HW);
      row = (b * N + n) * HW + (q % HW);
      wgt = gl < cnt ? depth[p] : 0.0f;
    }
    for (int j0 = 0; j0 < cnt; j0 += U) {
      float4 f[U][NV4];
      float ww[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int jj = gbase + min(j0 + u, cnt - 1);
        const int r = __shfl(row, jj, 64);
        ww[u] = __shfl(wgt, jj, 64);
        const float4* src = reinterpret_cast<const float4*>(feat + (size_t)r * C) + gl * NV4;
#pragma unroll
        for (int k = 0; k < NV4; ++k) f[u][k] = src[k];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (j0 + u < cnt) {
#pragma unroll
          for (int k = 0; k < NV4; ++k) {
            if (FUSED) {
              acc[k].x = __fadd_rn(acc[k].x, __fmul_rn(ww[u], f[u][k].x));
              acc[k].y = __fadd_rn(acc[k].y, __fmul_rn(ww[u], f[u][k].y));
              acc[k].z = __fadd_rn(acc[k].z, __fmul_rn(ww[u], f[u][k].z));
              acc[k].w = __fadd_rn(acc[k].w, __fmul_rn(ww[u], f[u][k].w));
            } else {
              acc[k].x = __fadd_rn(acc[k].x, f[u][k].x);
              acc[k].y = __fadd_rn(acc[k].y, f[u][k].y);
              acc[k].z = __fadd_rn(acc[k].z, f[u][k].z);
              acc[k].w = __fadd_rn(acc[k].w, f[u][k].w);
            }
          }
        }
      }
    }
  }
  if (vok && !is_long) {
    float4* dst = reinterpret_cast<float4*>(out + (size_t)v * C) + gl * NV4;
#pragma unroll
    for (int k = 0; k < NV4; ++k) dst[k] = acc[k];
  }
}

// Seventh layout (round 4, C = 128): both roles in ONE launch.  Two launches (long 26 us: the longest list's chain of dependent
// 32-row batches; short 35 us) only tie gather5's 59 us, and a kernel holding gather5's 32-rows-in-registers long path cannot keep
// the short path at 8 waves per SIMD.  Here the first POOL7_LONG_WGS workgroups walk the long-voxel list with the rows of a
// 16-point chunk travelling global -> LDS by LDS-DMA (glds16: 2 x 1 KB per wave, no registers, 2 x 8 KB of LDS per
// workgroup so that the short role keeps 7 waves per SIMD -- with 64- / 32-point chunks it ran at 2 / 5 and the kernel took 47 us; the next chunk is requested
// before this one is summed), two of their waves then add the chunk's products in ascending point order out of LDS (lane =
// channel; product rounded, then added: the oracle's arithmetic); all other workgroups are gather3's four short voxels per
// wave.  The long workgroups are dispatched first and hide under the short ones' store stream: 58.5 -> 39.1 us on the KITTI
// frustum (144 MB of algorithmic traffic: 3.7 TB/s), bit-identical output.
constexpr int POOL7_LONG_WGS = 1024;       // (KITTI frustum: 1004 long voxels -> one each; measured 512: 44.1 us, 1024: 39.1 us, 2048: 38.4 us)

template <bool FUSED>
__global__ void __launch_bounds__(256)
pool_gather7_kernel(const float* __restrict__ depth, const float* __restrict__ feat,
                    const int32_t* __restrict__ starts, const int32_t* __restrict__ order,
                    const int32_t* __restrict__ long_list, float* __restrict__ out, int nv, int P, int vox_per_batch, int N,
                    int D, int HW, int nlw) {
  constexpr int C = 128, NV4 = 2, U = 4;
  constexpr int CH = 16;                                   // list entries per chunk (2 x 8 KB of LDS: the short role keeps 7 waves per SIMD)
  __shared__ __align__(16) float rows[2][CH * C];
  __shared__ float wl[2][CH];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if ((int)blockIdx.x < nlw) {
    const int nlong = long_list[0];
    for (int i = blockIdx.x; i < nlong; i += nlw) {
      const int vv = long_list[1 + i];
      const int vs = starts[vv], ve = starts[vv + 1], vb = vv / vox_per_batch;
      // a chunk = CH consecutive list entries; wave w stages the rows 8 w .. 8 w + 7 (two rows per LDS-DMA instruction)
      auto stage = [&](int base, int buf) {
        const int cnt = min(CH, ve - base);
        const int p = lane < cnt ? order[base + lane] : 0;
        int row = p;
        float wgt = 1.0f;
        if (FUSED) {
          const int q = p - vb * P;
          const int n = q / (D * HW);
          row = (vb * N + n) * HW + (q % HW);
          wgt = lane < cnt ? depth[p] : 0.0f;
        }
        if (wave == 0 && lane < CH) wl[buf][lane] = wgt;
#pragma unroll
        for (int e = 0; e < CH / 8; ++e) {
          const int j = (CH / 4) * wave + 2 * e + (lane >> 5);
          const int r = __shfl(row, min(j, cnt - 1), 64);
          if ((CH / 4) * wave + 2 * e < cnt) glds16(feat + (size_t)r * C + (lane & 31) * 4, rows[buf] + ((CH / 4) * wave + 2 * e) * C);
        }
      };
      float a = 0.0f;
      int buf = 0;
      stage(vs, 0);
      for (int base = vs; base < ve; base += CH, buf ^= 1) {
        wait_vm0();
        __syncthreads();                                    // chunk `base` is in rows[buf] for every wave; rows[buf ^ 1] is free
        if (base + CH < ve) stage(base + CH, buf ^ 1);
        if (wave < 2) {
          const int cnt = min(CH, ve - base);
          const float* rp = rows[buf] + 64 * wave + lane;
          for (int j0 = 0; j0 < cnt; j0 += 8) {
            float f[8], w8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { f[u] = rp[min(j0 + u, cnt - 1) * C]; w8[u] = wl[buf][min(j0 + u, cnt - 1)]; }
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (j0 + u < cnt) a = __fadd_rn(a, FUSED ? __fmul_rn(w8[u], f[u]) : f[u]);
          }
        }
      }
      if (wave < 2) out[(size_t)vv * C + 64 * wave + lane] = a;
      __syncthreads();                                      // the buffers are restaged for the next voxel
    }
    return;
  }
  const int gl = lane & 15, gbase = lane & 48;
  const int v = (int)(((size_t)(blockIdx.x - nlw) * blockDim.x + threadIdx.x) >> 6) * 4 + (lane >> 4);
  const bool vok = v < nv;
  const int s = vok ? starts[v] : 0, e0 = vok ? starts[v + 1] : 0;
  const bool is_long = e0 - s > POOL_LONG;
  const int e = is_long ? s : e0;
  const int b = vok ? v / vox_per_batch : 0;
  float4 acc[NV4];
#pragma unroll
  for (int k = 0; k < NV4; ++k) acc[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  int p_next = (s + gl < e) ? order[s + gl] : 0;
  for (int base = s; base < e; base += 16) {
    const int cnt = min(16, e - base);
    const int p = p_next;
    p_next = (base + 16 + gl < e) ? order[base + 16 + gl] : 0;
    int row = p;
    float wgt = 1.0f;
    if (FUSED) {
      const int q = p - b * P;
      const int n = q / (D * HW);
      row = (b * N + n) * HW + (q % HW);
      wgt = gl < cnt ? depth[p] : 0.0f;
    }
    for (int j0 = 0; j0 < cnt; j0 += U) {
      float4 f[U][NV4];
      float ww[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int jj = gbase + min(j0 + u, cnt - 1);
        const int r = __shfl(row, jj, 64);
        ww[u] = __shfl(wgt, jj, 64);
        const float4* src = reinterpret_cast<const float4*>(feat + (size_t)r * C) + gl * NV4;
#pragma unroll
        for (int k = 0; k < NV4; ++k) f[u][k] = src[k];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (j0 + u < cnt) {
#pragma unroll
          for (int k = 0; k < NV4; ++k) {
            if (FUSED) {
              acc[k].x = __fadd_rn(acc[k].x, __fmul_rn(ww[u], f[u][k].x));
              acc[k].y = __fadd_rn(acc[k].y, __fmul_rn(ww[u], f[u][k].y));
              acc[k].z = __fadd_rn(acc[k].z, __fmul_rn(ww[u], f[u][k].z));
              acc[k].w = __fadd_rn(acc[k].w, __fmul_rn(ww[u], f[u][k].w));
            } else {
              acc[k].x = __fadd_rn(acc[k].x, f[u][k].x);
              acc[k].y = __fadd_rn(acc[k].y, f[u][k].y);
              acc[k].z = __fadd_rn(acc[k].z, f[u][k].z);
              acc[k].w = __fadd_rn(acc[k].w, f[u][k].w);
            }
          }
        }
      }
    }
  }
  if (vok && !is_long) {
    float4* dst = reinterpret_cast<float4*>(out + (size_t)v * C) + gl * NV4;
#pragma unroll
    for (int k = 0; k < NV4; ++k) dst[k] = acc[k];
  }
}


// grad_feats[n,:] = grad_out[vox[n],:]
__global__ void bev_pool_bwd_kernel(const float* __restrict__ gout, const int32_t* __restrict__ vox,
                                    float* __restrict__ gfeat, long total, int C) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long n = i / C;
  const int c = (int)(i - n * C);
  const int v = vox[n];
  gfeat[i] = v >= 0 ? gout[(size_t)v * C + c] : 0.0f;
}

// Backward of the fused lift-splat.  One wave per feature row (b, n, pixel); the wave walks the D
// depth planes four at a time (16 lanes x float4 cover a 64-channel slab of one voxel row):
//   grad_depth[p]   = <grad_out[vox[p], :], feat[row, :]>
//   grad_feat[row]  = sum_d depth[p] * grad_out[vox[p], :]
// Every output element is written exactly once: no atomics, deterministic.
__global__ void __launch_bounds__(256)
lift_splat_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ depth,
                      const float* __restrict__ feat, const int32_t* __restrict__ vox,
                      float* __restrict__ gdepth, float* __restrict__ gfeat, int rows, int C, int P, int N,
                      int D, int HW) {
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= rows) return;
  const int slot = lane >> 4, l16 = lane & 15;
  const int bn = row / HW, pix = row - bn * HW;
  const int b = bn / N, n = bn - b * N;
  const size_t pbase = (size_t)b * P + (size_t)n * D * HW + pix;
  const float* frow = feat + (size_t)row * C;
  for (int cb = 0; cb < C; cb += 64) {   // 64-channel slabs (C=128 -> 2 passes over the planes)
    const int c = cb + l16 * 4;
    const bool cok = c < C;
    float4 f = cok ? *reinterpret_cast<const float4*>(frow + c) : make_float4(0, 0, 0, 0);
    float4 acc = make_float4(0, 0, 0, 0);
    for (int d0 = 0; d0 < D; d0 += 4) {
      const int d = d0 + slot;
      float dot = 0.0f;
      size_t p = 0;
      if (d < D) {
        p = pbase + (size_t)d * HW;
        const int v = vox[p];
        if (v >= 0 && cok) {
          const float4 g = *reinterpret_cast<const float4*>(gout + (size_t)v * C + c);
          const float w = depth[p];
          dot = g.x * f.x + g.y * f.y + g.z * f.z + g.w * f.w;
          acc.x += w * g.x; acc.y += w * g.y; acc.z += w * g.z; acc.w += w * g.w;
        }
      }
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
      if (d < D && l16 == 0) {
        if (cb == 0) gdepth[p] = dot; else gdepth[p] += dot;
      }
    }
    // fold the four depth slots together
    acc.x += __shfl_xor(acc.x, 16, 64); acc.y += __shfl_xor(acc.y, 16, 64);
    acc.z += __shfl_xor(acc.z, 16, 64); acc.w += __shfl_xor(acc.w, 16, 64);
    acc.x += __shfl_xor(acc.x, 32, 64); acc.y += __shfl_xor(acc.y, 32, 64);
    acc.z += __shfl_xor(acc.z, 32, 64); acc.w += __shfl_xor(acc.w, 32, 64);
    if (slot == 0 && cok) *reinterpret_cast<float4*>(gfeat + (size_t)row * C + c) = acc;
  }
}

// r2 version for C == 4 * LPS (LPS = 16 / 32 / 64 lanes per voxel row; the path's C = 128 -> LPS = 32): one WORKGROUP of four
// waves per feature row, each wave owning a quarter of the depth planes.  The first version walked all D planes with one wave
// per row -- 7680 waves, each a serial chain of 2 x D/4 dependent (vox -> grad_out) loads: 107 us at 1.25 TB/s.  Here
//   * a wave fetches vox[] and depth[] of up to 64 of its planes with one strided gather (lane = plane), then walks them
//     64/LPS planes per step with readlane broadcasts, four steps' grad_out rows in flight;
//   * the per-plane dots are collected lane-wise and stored with one strided store per 64 planes;
//   * the four partial grad_feat rows are folded through LDS in wave order (deterministic).
template <int LPS>
__global__ void __launch_bounds__(256)
lift_splat_bwd2_kernel(const float* __restrict__ gout, const float* __restrict__ depth,
                       const float* __restrict__ feat, const int32_t* __restrict__ vox,
                       float* __restrict__ gdepth, float* __restrict__ gfeat, int rows, int P, int N, int D, int HW) {
  constexpr int C = 4 * LPS, SLOTS = 64 / LPS, UNR = 4;
  __shared__ float fold[4][C];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int row = blockIdx.x;
  const int slot = lane / LPS, ls = lane % LPS;
  const int bn = row / HW, pix = row - bn * HW;
  const int b = bn / N, n = bn - b * N;
  const size_t pbase = (size_t)b * P + (size_t)n * D * HW + pix;
  const int per = (D + 3) >> 2;
  const int dbeg = min(D, wv * per), dend = min(D, dbeg + per);
  const float4 f = *reinterpret_cast<const float4*>(feat + (size_t)row * C + 4 * ls);
  float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  for (int d0 = dbeg; d0 < dend; d0 += 64) {
    const int cnt = min(64, dend - d0);
    const size_t pl = pbase + (size_t)(d0 + lane) * HW;
    const int vl = lane < cnt ? vox[pl] : -1;
    const float wl = lane < cnt ? depth[pl] : 0.0f;
    float dotl = 0.0f;                                   // lane j ends up with the dot of plane d0 + j
    // only ~27 % of the frustum points fall inside the grid: walk the KEPT planes of this block (ballot + bit scan, all
    // wave-uniform scalar work), SLOTS * UNR of them per batch; dropped planes keep a zero dot
    unsigned long long mask = __ballot(vl >= 0);
    while (mask) {
      int jsel[UNR];
      float4 g[UNR];
      float w[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        int mine = -1;
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
          const int j = mask ? (int)__builtin_ctzll(mask) : -1;
          mask &= mask - 1;                              // (0 & anything) stays 0
          if (sl == slot) mine = j;
        }
        jsel[u] = mine;
        const int vv = __shfl(vl, max(mine, 0), 64);
        w[u] = __shfl(wl, max(mine, 0), 64);
        g[u] = mine >= 0 ? *reinterpret_cast<const float4*>(gout + (size_t)vv * C + 4 * ls) : make_float4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        float dot = g[u].x * f.x + g[u].y * f.y + g[u].z * f.z + g[u].w * f.w;
#pragma unroll
        for (int off = LPS / 2; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
        if (jsel[u] >= 0) { acc.x += w[u] * g[u].x; acc.y += w[u] * g[u].y; acc.z += w[u] * g[u].z; acc.w += w[u] * g[u].w; }
        // hand the dot of plane j to lane j
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
          const float dsl = __shfl(dot, sl * LPS, 64);
          const int jl = __shfl(jsel[u], sl * LPS, 64);
          if (lane == jl) dotl = dsl;
        }
      }
    }
    if (lane < cnt) gdepth[pl] = dotl;
  }
  // fold the slots of this wave, then the four waves through LDS in wave order
#pragma unroll
  for (int off = LPS; off < 64; off <<= 1) {
    acc.x += __shfl_xor(acc.x, off, 64); acc.y += __shfl_xor(acc.y, off, 64);
    acc.z += __shfl_xor(acc.z, off, 64); acc.w += __shfl_xor(acc.w, off, 64);
  }
  if (slot == 0) *reinterpret_cast<float4*>(&fold[wv][4 * ls]) = acc;
  __syncthreads();
  if (wv == 0 && slot == 0) {
    float4 t = *reinterpret_cast<const float4*>(&fold[0][4 * ls]);
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const float4 o = *reinterpret_cast<const float4*>(&fold[k][4 * ls]);
      t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
    }
    *reinterpret_cast<float4*>(gfeat + (size_t)row * C + 4 * ls) = t;
  }
}

bool pool_dims_ok(const ssbev_pool_dims* d) {
  return d && d->B > 0 && d->P >= 0 && d->C > 0 && d->nx > 0 && d->ny > 0 && d->nz > 0 &&
         (long)d->B * d->nx * d->ny * d->nz < (1L << 31) && (long)d->B * d->P < (1L << 31);
}

template <bool FUSED>
int launch_gather(const float* depth, const float* feat, const int32_t* starts, const int32_t* order, float* out,
                  const ssbev_pool_dims* d, int N, int D, int HW, hipStream_t st, const int32_t* long_list = nullptr) {
  const int nv = d->B * d->nx * d->ny * d->nz;
  const int vpb = d->nx * d->ny * d->nz;
  if (long_list && d->C == 128) {                         // one launch: long-voxel workgroups first, short role behind them
    static const int nlw = ssbev_tune("SSBEV_POOL7_LONG_WGS") ? std::max(1, atoi(ssbev_tune("SSBEV_POOL7_LONG_WGS"))) : POOL7_LONG_WGS;   // (tuning hook)
    dim3 g7(nlw + cdiv((size_t)cdiv(nv, 4) * 64, 256));
    hipLaunchKernelGGL((pool_gather7_kernel<FUSED>), g7, dim3(256), 0, st, depth, feat, starts, order, long_list, out, nv, d->P, vpb,
                       N, D, HW, nlw);
    return ssbev_launch_status();
  }
  if (long_list && (d->C == 64 || d->C == 128 || d->C == 256)) {      // role split over the compacted long-voxel list (round 4)
    dim3 gl_(512), gs_(cdiv((size_t)cdiv(nv, 4) * 64, 256)), blk(256);
#define SSBEV_GATHER_SPLIT(NV4_)                                                                                            \
    hipLaunchKernelGGL((pool_gather_long_kernel<FUSED, NV4_>), gl_, blk, 0, st, depth, feat, starts, order, long_list, out, d->P, \
                       vpb, N, D, HW);                                                                                       \
    hipLaunchKernelGGL((pool_gather_short_kernel<FUSED, NV4_>), gs_, blk, 0, st, depth, feat, starts, order, out, nv, d->P,  \
                       vpb, N, D, HW)
    if (d->C == 64) { SSBEV_GATHER_SPLIT(1); } else if (d->C == 128) { SSBEV_GATHER_SPLIT(2); } else { SSBEV_GATHER_SPLIT(4); }
#undef SSBEV_GATHER_SPLIT
    return ssbev_launch_status();
  }
  // (the r1 one-voxel-per-wave kernel, the four-side-by-side one without the whole-wave long-list path and its persistent
  // variant were superseded in rounds 2-4 and removed in round 6)
  if (d->C == 64 || d->C == 128 || d->C == 256) {
    dim3 grid5(cdiv((size_t)cdiv(nv, 4) * 64, 256)), block5(256);
    if (d->C == 64)
      hipLaunchKernelGGL((pool_gather5_kernel<FUSED, 1>), grid5, block5, 0, st, depth, feat, starts, order, out, nv, d->P,
                         vpb, N, D, HW);
    else if (d->C == 128)
      hipLaunchKernelGGL((pool_gather5_kernel<FUSED, 2>), grid5, block5, 0, st, depth, feat, starts, order, out, nv, d->P,
                         vpb, N, D, HW);
    else
      hipLaunchKernelGGL((pool_gather5_kernel<FUSED, 4>), grid5, block5, 0, st, depth, feat, starts, order, out, nv, d->P,
                         vpb, N, D, HW);
    return ssbev_launch_status();
  }
  dim3 grid2(cdiv((size_t)cdiv(nv, POOL_VPW) * 64, 256)), block2(256);      // any other channel count
  if (d->C % 256 == 0)
    hipLaunchKernelGGL((pool_gather2_kernel<FUSED, 4>), grid2, block2, 0, st, depth, feat, starts, order, out, nv, d->C,
                       d->P, vpb, N, D, HW);
  else if (d->C % 2 == 0)
    hipLaunchKernelGGL((pool_gather2_kernel<FUSED, 2>), grid2, block2, 0, st, depth, feat, starts, order, out, nv, d->C,
                       d->P, vpb, N, D, HW);
  else
    hipLaunchKernelGGL((pool_gather2_kernel<FUSED, 1>), grid2, block2, 0, st, depth, feat, starts, order, out, nv, d->C,
                       d->P, vpb, N, D, HW);
  return ssbev_launch_status();
}

}  // namespace

extern "C" {

int ssbev_voxel_index(const float* geom, int32_t* vox, int32_t* idx3, const ssbev_pool_dims* d,
                      ssbev_stream_t stream) {
  if (!pool_dims_ok(d) || !geom || !vox) return SSBEV_EINVAL;
  const long total = (long)d->B * d->P;
  if (total == 0) return SSBEV_OK;
  hipLaunchKernelGGL(voxel_index_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), geom, vox, idx3,
                     total, d->P, d->nx, d->ny, d->nz, d->origin[0], d->origin[1], d->origin[2], d->dx[0], d->dx[1],
                     d->dx[2]);
  return ssbev_launch_status();
}

int ssbev_frustum_geometry(const float* frustum, const float* m1, const float* t0, const float* m2, const float* t2,
                           const float* tr, const float* m3, const float* t3, float* geom, const ssbev_geom_dims* d,
                           ssbev_stream_t stream) {
  if (!d || d->B <= 0 || d->N <= 0 || d->D <= 0 || d->H <= 0 || d->W <= 0 || !frustum || !m1 || !t0 || !m2 || !tr || !m3 || !geom)
    return SSBEV_EINVAL;
  const long total = (long)d->B * d->N * d->D * d->H * d->W;
  if (total >= (1L << 31)) return SSBEV_EINVAL;
  GeomArgs a{frustum, m1, t0, m2, t2, tr, m3, t3, geom, d->B, d->N, d->D, d->H, d->W};
  hipLaunchKernelGGL(frustum_geometry_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), a);
  return ssbev_launch_status();
}

int ssbev_coords_to_vox(const int32_t* coords, int n, int32_t* vox, const ssbev_pool_dims* d,
                        ssbev_stream_t stream) {
  if (!pool_dims_ok(d) || n < 0 || (n && (!coords || !vox))) return SSBEV_EINVAL;
  if (n == 0) return SSBEV_OK;
  hipLaunchKernelGGL(coords_to_vox_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), coords, n, vox, d->B,
                     d->nx, d->ny, d->nz);
  return ssbev_launch_status();
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// workspace of ssbev_pool_prepare: totals[npass][2048] | bucket_base[2049] | hist[2^digit bits][tiles] | (keys, ids) x 1 or 2
struct CsrWs { size_t totals, base, hist, pairs, total; };
static CsrWs csr_ws(const CsrPlan& p, size_t n) {
  CsrWs w;
  int nd_max = 1;
  for (int j = 0; j < p.npass; ++j) nd_max = nd_max > (1 << p.nbits[j]) ? nd_max : (1 << p.nbits[j]);
  w.totals = 0;
  w.base = w.totals + align256((size_t)p.npass * 2048 * 4);
  w.hist = w.base + align256(2049 * 4);
  w.pairs = w.hist + align256((size_t)nd_max * p.nblk * 4);
  w.total = w.pairs + (p.npass > 1 ? 4 : 2) * align256(n * 4 + 4);
  return w;
}

size_t ssbev_pool_prepare_workspace(int n_points, const ssbev_pool_dims* d) {
  if (!pool_dims_ok(d) || n_points < 0) return 0;
  CsrPlan p;
  if (!csr_plan((long long)d->B * d->nx * d->ny * d->nz, n_points, &p)) return 0;
  return csr_ws(p, (size_t)n_points).total;
}

int ssbev_pool_prepare2(const int32_t* vox, int n_points, int32_t* starts, int32_t* order, int32_t* long_list,
                        const ssbev_pool_dims* d, void* ws, size_t ws_bytes, ssbev_stream_t stream);

int ssbev_pool_prepare(const int32_t* vox, int n_points, int32_t* starts, int32_t* order,
                       const ssbev_pool_dims* d, void* ws, size_t ws_bytes, ssbev_stream_t stream) {
  return ssbev_pool_prepare2(vox, n_points, starts, order, nullptr, d, ws, ws_bytes, stream);
}

size_t ssbev_pool_long_list_elems(int n_points) { return n_points < 0 ? 0 : (size_t)n_points / (POOL_LONG + 1) + 2; }

int ssbev_pool_prepare2(const int32_t* vox, int n_points, int32_t* starts, int32_t* order, int32_t* long_list,
                        const ssbev_pool_dims* d, void* ws, size_t ws_bytes, ssbev_stream_t stream) {
  if (!pool_dims_ok(d) || n_points < 0 || !starts || !ws || (n_points && (!vox || !order))) return SSBEV_EINVAL;
  const long long nvl = (long long)d->B * d->nx * d->ny * d->nz;
  CsrPlan p;
  if (nvl >= (1ll << 31) || !csr_plan(nvl, n_points, &p)) return SSBEV_EINVAL;
  const CsrWs w = csr_ws(p, (size_t)n_points);
  if (ws_bytes < w.total) return SSBEV_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  const int nv = (int)nvl;
  if (n_points == 0) {
    if (hipMemsetAsync(starts, 0, ((size_t)nv + 1) * 4, st) != hipSuccess) return SSBEV_ELAUNCH;
    if (long_list && hipMemsetAsync(long_list, 0, 4, st) != hipSuccess) return SSBEV_ELAUNCH;
    return SSBEV_OK;
  }
  char* base = static_cast<char*>(ws);
  int32_t* totals = reinterpret_cast<int32_t*>(base + w.totals);
  int32_t* bucket_base = reinterpret_cast<int32_t*>(base + w.base);
  int32_t* hist = reinterpret_cast<int32_t*>(base + w.hist);
  const size_t nb = align256((size_t)n_points * 4 + 4);
  int32_t* kbuf[2] = {reinterpret_cast<int32_t*>(base + w.pairs), reinterpret_cast<int32_t*>(base + w.pairs + 2 * nb)};
  int32_t* ibuf[2] = {reinterpret_cast<int32_t*>(base + w.pairs + nb), reinterpret_cast<int32_t*>(base + w.pairs + 3 * nb)};
  if (hipMemsetAsync(totals, 0, (size_t)p.npass * 2048 * 4, st) != hipSuccess) return SSBEV_ELAUNCH;
  const int32_t* kin = vox;
  const int32_t* iin = nullptr;
  for (int j = 0; j < p.npass; ++j) {
    const int nd = 1 << p.nbits[j];
    int32_t* tot = totals + (size_t)j * 2048;
    hipLaunchKernelGGL(csr_hist_kernel, dim3(p.nblk), dim3(CSR_T), (size_t)nd * 4, st, kin, n_points, nv, p.shift[j],
                       p.nbits[j], p.items, p.nblk, hist, tot, j == 0 ? long_list : nullptr);
    hipLaunchKernelGGL(csr_scan_kernel, dim3(nd), dim3(256), 0, st, hist, tot, nd, p.nblk, bucket_base);
    const size_t lds = (size_t)CSR_WAVES * nd * 4;
    auto part = j == 0 ? (p.items == 8 ? csr_partition_kernel<true, true> : csr_partition_kernel<true, false>)
                       : (p.items == 8 ? csr_partition_kernel<false, true> : csr_partition_kernel<false, false>);
    hipLaunchKernelGGL(part, dim3(p.nblk), dim3(CSR_T), lds, st, kin, iin, n_points, nv, p.shift[j], p.nbits[j], p.items,
                       p.nblk, hist, kbuf[j & 1], ibuf[j & 1]);
    kin = kbuf[j & 1]; iin = ibuf[j & 1];
    // the passes behind the first see only the valid points, packed at the front; the tail of the buffers is stale
    // (keys_out of pass j has bucket_base[nd] entries) -- the kernels bound their reads by n_points and by the key range,
    // so the stale tail must not look valid:
    if (j + 1 < p.npass) {
      // n_valid is on the device; mark the tail invalid instead of reading it back
      hipLaunchKernelGGL(csr_invalidate_tail_kernel, dim3(cdiv((size_t)n_points, 256)), dim3(256), 0, st,
                         kbuf[j & 1], n_points, bucket_base + nd);
    }
  }
  const int nd_last = 1 << p.nbits[p.npass - 1];
  const size_t blds = ((size_t)CSR_BW * (1 << p.lo_bits) + CSR_BW) * 4;
  if (blds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(csr_bucket_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)blds) != hipSuccess)
    return SSBEV_ELAUNCH;
  hipLaunchKernelGGL(csr_bucket_kernel, dim3(p.nbuckets), dim3(CSR_BW * 64), blds, st, kin, iin,
                     p.npass == 1 ? bucket_base : nullptr, bucket_base + nd_last, nv, p.lo_bits, starts, order, long_list);
  return ssbev_launch_status();
}

int ssbev_bev_pool_fwd(const float* feats, const int32_t* starts, const int32_t* order, float* out,
                       const ssbev_pool_dims* d, ssbev_stream_t stream) {
  if (!pool_dims_ok(d) || !starts || !out) return SSBEV_EINVAL;
  return launch_gather<false>(nullptr, feats, starts, order, out, d, 1, 1, 1, as_stream(stream));
}

int ssbev_bev_pool_bwd(const float* grad_out, const int32_t* vox, int n_points, float* grad_feats,
                       const ssbev_pool_dims* d, ssbev_stream_t stream) {
  if (!pool_dims_ok(d) || n_points < 0) return SSBEV_EINVAL;
  if (n_points == 0) return SSBEV_OK;
  if (!grad_out || !vox || !grad_feats) return SSBEV_EINVAL;
  const long total = (long)n_points * d->C;
  hipLaunchKernelGGL(bev_pool_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), grad_out, vox,
                     grad_feats, total, d->C);
  return ssbev_launch_status();
}

int ssbev_lift_splat_fwd(const float* depth, const float* feat, const int32_t* starts,
                         const int32_t* order, float* out, const ssbev_pool_dims* d,
                         const ssbev_lift_dims* l, ssbev_stream_t stream) {
  if (!pool_dims_ok(d) || !l || !depth || !feat || !starts || !out) return SSBEV_EINVAL;
  if (l->N <= 0 || l->D <= 0 || l->HW <= 0 || (long)l->N * l->D * l->HW != d->P) return SSBEV_EINVAL;
  return launch_gather<true>(depth, feat, starts, order, out, d, l->N, l->D, l->HW, as_stream(stream));
}

int ssbev_lift_splat_fwd2(const float* depth, const float* feat, const int32_t* starts, const int32_t* order,
                          const int32_t* long_list, float* out, const ssbev_pool_dims* d, const ssbev_lift_dims* l,
                          ssbev_stream_t stream) {
  if (!pool_dims_ok(d) || !l || !depth || !feat || !starts || !out) return SSBEV_EINVAL;
  if (l->N <= 0 || l->D <= 0 || l->HW <= 0 || (long)l->N * l->D * l->HW != d->P) return SSBEV_EINVAL;
  return launch_gather<true>(depth, feat, starts, order, out, d, l->N, l->D, l->HW, as_stream(stream), long_list);
}

int ssbev_lift_splat_bwd(const float* grad_out, const float* depth, const float* feat,
                         const int32_t* vox, float* grad_depth, float* grad_feat,
                         const ssbev_pool_dims* d, const ssbev_lift_dims* l, ssbev_stream_t stream) {
  if (!pool_dims_ok(d) || !l || !grad_out || !depth || !feat || !vox || !grad_depth || !grad_feat)
    return SSBEV_EINVAL;
  if (l->N <= 0 || l->D <= 0 || l->HW <= 0 || (long)l->N * l->D * l->HW != d->P || d->C % 4 != 0)
    return SSBEV_EINVAL;
  const int rows = d->B * l->N * l->HW;
  if (d->C == 64 || d->C == 128 || d->C == 256) {
    hipStream_t st = as_stream(stream);
    if (d->C == 64)
      hipLaunchKernelGGL(lift_splat_bwd2_kernel<16>, dim3(rows), dim3(256), 0, st, grad_out, depth, feat, vox, grad_depth,
                         grad_feat, rows, d->P, l->N, l->D, l->HW);
    else if (d->C == 128)
      hipLaunchKernelGGL(lift_splat_bwd2_kernel<32>, dim3(rows), dim3(256), 0, st, grad_out, depth, feat, vox, grad_depth,
                         grad_feat, rows, d->P, l->N, l->D, l->HW);
    else
      hipLaunchKernelGGL(lift_splat_bwd2_kernel<64>, dim3(rows), dim3(256), 0, st, grad_out, depth, feat, vox, grad_depth,
                         grad_feat, rows, d->P, l->N, l->D, l->HW);
    return ssbev_launch_status();
  }
  hipLaunchKernelGGL(lift_splat_bwd_kernel, dim3(cdiv((size_t)rows * 64, 256)), dim3(256), 0, as_stream(stream),
                     grad_out, depth, feat, vox, grad_depth, grad_feat, rows, d->C, d->P, l->N, l->D, l->HW);
  return ssbev_launch_status();
}

}  // extern "C"
