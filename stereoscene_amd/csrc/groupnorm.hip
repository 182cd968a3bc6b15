// GroupNorm / BatchNorm(train) over channels-last volumes with fused residual-add + ReLU, gfx950.
//
// The reference applies GN(2) / GN(32) / GN(1) / BatchNorm3d after (almost) every 3-D convolution
// (VT:66-88, ATT:94-111, resnet3d.py:42-45, second_fpn_3d.py:68, occhead.py:104).  ATen's GroupNorm
// launches one workgroup per (sample, group): with B=1 and G=2 on the 1.47 M-voxel cost volume that
// is TWO workgroups on a 256-CU chip (344 ms/step measured).  Here the statistics are a two-stage
// reduction spread over the whole chip, and normalise + affine + residual + ReLU is one streaming
// pass: HBM-bound, 3 tensor passes forward (read, read, write), 5 backward.
//
// x, y, residual: [B, S, C] channels-last fp32 (S = D*H*W).  Group g owns channels [g*C/G, (g+1)*C/G).
// BatchNorm in training mode is the same computation with G = C and (B, S) -> (1, B*S).
#include "common.h"

#include <algorithm>

namespace {

constexpr int NT = 256;
// The statistics kernels: PNT threads per workgroup, ~kStatBlocks workgroups (three per CU).  Round 5: a workgroup covers a
// chunk of voxels x a SLAB of kSlabLanes lanes (32 fp32 / 64 bf16 channels), so that every problem, whatever its channel count,
// leaves the same ~24.6 k (sum, sum-of-squares) records for the finalize kernels.
// (One workgroup per CU with 768 threads and 256 chunks was measured and dropped: the streaming passes lost 40-70 %, and a
// 12-wave workgroup waits for a whole free CU next to the side stream's kernels.)
constexpr int PNT = 256;
constexpr int kStatBlocks = 768;
constexpr int kSlabLanes = 8;

struct GnGeom {
  int B, C, G;
  long S;
  int chunks;        // chunks per sample
  long chunk_len;    // voxels per chunk
  float eps;
  int relu;
  int slab_q;        // lanes (of VW channels) per channel slab of the statistics kernels (blockIdx.z)
  int pre;           // 1: u = gelu(x) (exact, erf) is what gets normalised; x is still the tensor in memory
  long ldy, ldg;     // row strides (floats) of y (forward) and gy (backward): C when dense, the width of the concatenation
                     // when the operator writes / reads a channel slice of a wider channels-last tensor (ssbev_norm_dims.ld_*)
};

// GELU in its exact (erf) form, nn.GELU's default, and its derivative from ONE exponential: with z = x / sqrt(2),
// Phi(x) = 1 - q / 2 (z >= 0) or q / 2 (z < 0), q = erfc(|z|) = poly(t) exp(-z^2), t = 1 / (1 + p |z|) (Abramowitz &
// Stegun 7.1.26, |error| <= 1.5e-7, no cancellation on the negative side); gelu = x Phi, gelu' = Phi + x phi with
// phi = exp(-z^2) / sqrt(2 pi).  libm's erff costs ~40 VALU instructions per element -- enough to make the streaming
// normalisation passes compute-bound (measured: +0.15 ms per 189 MB pass).
__device__ __forceinline__ void gelu_both(float x, float& u, float& du) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __frcp_rn(1.0f + 0.3275911f * z);
  const float e = __expf(-z * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float hq = 0.5f * poly * e;                       // erfc(|z|) / 2
  const float cdf = x >= 0.0f ? 1.0f - hq : hq;
  u = x * cdf;
  du = cdf + x * 0.39894228040143268f * e;
}
__device__ __forceinline__ float gelu_f(float x) {
  float u, du;
  gelu_both(x, u, du);
  return u;
}

// VW consecutive channels per lane: 4 (one 16-byte fp32 access, or 8 bytes of bf16) or -- bf16 tensors with C % 8 == 0 -- 8
// (one 16-byte bf16 access).  With 8-byte accesses the bf16 passes ran at half the bytes per second of the fp32 ones (round 4:
// 17.3 -> 14.3 ms per two-sample step instead of -> 9); the ReLU bit mask keeps one bit per element, VW words per 64 lanes.
template <int VW, typename T>
__device__ __forceinline__ void ldn(const T* p, float (&o)[VW]) {
  if constexpr (VW == 4) {
    const float4 v = ld4(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  } else {
    static_assert(sizeof(T) == 2, "8 channels per lane: bf16 tensors only");
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    o[0] = __uint_as_float(u.x << 16); o[1] = __uint_as_float(u.x & 0xffff0000u);
    o[2] = __uint_as_float(u.y << 16); o[3] = __uint_as_float(u.y & 0xffff0000u);
    o[4] = __uint_as_float(u.z << 16); o[5] = __uint_as_float(u.z & 0xffff0000u);
    o[6] = __uint_as_float(u.w << 16); o[7] = __uint_as_float(u.w & 0xffff0000u);
  }
}
template <int VW, typename T>
__device__ __forceinline__ void stn(T* p, const float (&v)[VW]) {
  if constexpr (VW == 4) {
    st4(p, make_float4(v[0], v[1], v[2], v[3]));
  } else {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
  }
}


// A chunk record = (sum, sum of squares) of one channel = ONE aligned 8-byte store (round 5: whatever its channel count, a problem
// leaves ~24.6 k records for the finalize kernels, which request all of them at once).
// (Round 5 also built a finalize-in-the-tail variant -- last-arriving workgroup behind agent-scope tickets folds the records,
// no finalize launch.  Measured +0.7 ms per step (one workgroup folds 200 KB while the stream's other 255 CUs wait; ~770
// serialised ticket atomics per problem) and removed in round 6; profiles/r5z_summary*.txt)
__device__ __forceinline__ void st_record(float* p, float a, float q) {
  *reinterpret_cast<float2*>(p) = make_float2(a, q);
}

// Per-channel partial sums over one chunk of voxels.  MODE 0: (sum x, sum x^2).
// MODE 1 (backward): (sum g, sum g * xhat) with g = gy * [y > 0] when relu is fused.
template <int MODE, bool PRE, int RM = 0, typename T = float, int VW = 4>   // RM: source of the fused ReLU's sign (0 none, 1 bit mask, 2 y)
__global__ void __launch_bounds__(PNT, 3)
gn_partial_kernel(const T* __restrict__ x, const T* __restrict__ gy, const T* __restrict__ y,
                  const unsigned long long* __restrict__ mask, const float* __restrict__ mean,
                  const float* __restrict__ rstd, float* __restrict__ partial, GnGeom g) {
  extern __shared__ __align__(16) float lds[];         // [rows][Cs][2]
  // channel slab of this block (blockIdx.z): g.slab_q lanes of VW channels
  const int q0 = blockIdx.z * g.slab_q;                // first lane of the slab
  const int q = min(g.C / VW - q0, g.slab_q);          // lanes per voxel in this slab
  const int Cs = q * VW;
  const int rows = PNT / q > 0 ? PNT / q : 1;          // voxels handled per block iteration
  const int tid = threadIdx.x;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const long s0 = (long)chunk * g.chunk_len;
  const long s1 = min(g.S, s0 + g.chunk_len);
  float a0[VW], a1[VW];
#pragma unroll
  for (int k = 0; k < VW; ++k) { a0[k] = 0.0f; a1[k] = 0.0f; }
  const int c4 = tid % q, r = tid / q;
  if (r < rows) {
    const int c = (q0 + c4) * VW;
    const int cpg = g.C / g.G;
    float mu4[VW], rs4[VW];                             // this thread's channels never change
#pragma unroll
    for (int k = 0; k < VW; ++k) { mu4[k] = 0.0f; rs4[k] = 1.0f; }
    if (MODE == 1) {
#pragma unroll
      for (int k = 0; k < VW; ++k) { mu4[k] = mean[b * g.G + (c + k) / cpg]; rs4[k] = rstd[b * g.G + (c + k) / cpg]; }
    }
    const size_t base = (size_t)b * g.S * g.C;
    // MODE 0 accumulates SHIFTED sums, sum (u - p) and sum (u - p)^2 with the pivot p = the channel's value at voxel 0 of
    // the sample: the fp32 partials then hold no large common offset, and the finalize kernels rebuild sum u and sum u^2
    // in double -- E[u^2] - mean^2 from plain fp32 partials cancels catastrophically when |mean| >> std
    float pv[VW];
#pragma unroll
    for (int k = 0; k < VW; ++k) pv[k] = 0.0f;
    if (MODE == 0) {
      ldn<VW>(x + base + c, pv);
      if (PRE) {
#pragma unroll
        for (int k = 0; k < VW; ++k) pv[k] = gelu_f(pv[k]);
      }
    }
    // UB voxels per trip: all their loads are issued before the first sum (the plain loop kept ~2 loads per thread in flight
    // and ran at 3.4 TB/s); the sums themselves stay in voxel order, so the partials are bit for bit what they were
    constexpr int UB = 4;
    struct Item { size_t off; float xv[VW], gv[VW], yv[VW]; unsigned long long mw[4]; };
    auto accumulate = [&](const Item& it) __attribute__((always_inline)) {
      float xs[VW];
#pragma unroll
      for (int k = 0; k < VW; ++k) xs[k] = PRE ? gelu_f(it.xv[k]) : it.xv[k];
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < VW; ++k) { const float dv = xs[k] - pv[k]; a0[k] += dv; a1[k] += dv * dv; }
      } else {
        float gs[VW];
#pragma unroll
        for (int k = 0; k < VW; ++k) gs[k] = it.gv[k];
        if (RM == 1) {                          // 1 bit per element instead of re-reading y (see gn_apply_fwd_kernel)
          // the mask is laid out per channel QUAD (word (quad / 64) * 4 + component, bit quad % 64) whatever VW is: an
          // 8-channel lane reads two neighbouring bits (its first quad index is even) of the same four words
          const int sh = (int)((it.off >> 2) & 63);
          // (two 16-byte loads for the four words + one shift per word were measured: 115 -> 123 us on the bf16 pass; not kept)
#pragma unroll
          for (int k = 0; k < VW; ++k) gs[k] = ((it.mw[k & 3] >> (sh + (k >> 2))) & 1ull) ? gs[k] : 0.0f;
        } else if (RM == 2) {
#pragma unroll
          for (int k = 0; k < VW; ++k) gs[k] = it.yv[k] > 0.0f ? gs[k] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < VW; ++k) {
          a0[k] += gs[k];
          a1[k] += gs[k] * (xs[k] - mu4[k]) * rs4[k];
        }
      }
    };
    auto fetch = [&](long s, Item& it) __attribute__((always_inline)) {
      it.off = base + (size_t)s * g.C + c;
      ldn<VW>(x + it.off, it.xv);
      if (MODE == 1) {
        ldn<VW>(gy + ((size_t)b * g.S + s) * g.ldg + c, it.gv);
        if (RM == 1) {
          const unsigned long long* mp = mask + ((it.off >> 2) >> 6) * 4;
#pragma unroll
          for (int k = 0; k < 4; ++k) it.mw[k] = mp[k];
        } else if (RM == 2) {
          ldn<VW>(y + it.off, it.yv);
        }
      }
    };
    long s = s0 + r;
    for (; s + (long)(UB - 1) * rows < s1; s += (long)UB * rows) {
      Item it[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) fetch(s + (long)u * rows, it[u]);
#pragma unroll
      for (int u = 0; u < UB; ++u) accumulate(it[u]);
    }
    for (; s < s1; s += rows) {
      Item it;
      fetch(s, it);
      accumulate(it);
    }
#pragma unroll
    for (int k = 0; k < VW; ++k) {
      lds[((size_t)r * Cs + c4 * VW + k) * 2 + 0] = a0[k];
      lds[((size_t)r * Cs + c4 * VW + k) * 2 + 1] = a1[k];
    }
  }
  __syncthreads();
  // fold the `rows` voxel lanes: thread t < Cs sums the two columns of channel t of the slab and publishes the pair as one
  // 8-byte write-through record (st_record)
  for (int t = tid; t < Cs; t += PNT) {
    float s0 = 0.0f, s1 = 0.0f;
    for (int rr = 0; rr < rows; ++rr) { s0 += lds[((size_t)rr * Cs + t) * 2]; s1 += lds[((size_t)rr * Cs + t) * 2 + 1]; }
    st_record(partial + ((size_t)(b * g.chunks + chunk) * g.C + q0 * VW + t) * 2, s0, s1);
  }
}

constexpr int FT = 1024;     // threads of the (tiny, latency-bound) finalize kernels

// block-wide sum of two doubles (fixed tree: deterministic).  Round 4: butterflies inside the waves, ONE barrier for the 16
// wave totals (the r1 form walked a 1024-entry LDS tree with ten barriers: most of the ~10 us these latency-bound
// kernels took, 104 launches per step).  s0 / s1 need FT / 64 entries.
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ void block_sum2(double& a, double& q, double* s0, double* s1) {
  a = wave_sum_d(a); q = wave_sum_d(q);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s0[w] = a; s1[w] = q; }
  __syncthreads();
  double ta = 0.0, tq = 0.0;
#pragma unroll
  for (int i = 0; i < FT / 64; ++i) { ta += s0[i]; tq += s1[i]; }
  a = ta; q = tq;
}

// forward finalize: one thread block per (b, group): mean / rstd in double.  Only B*G workgroups exist (2 for GN(2) at
// B = 1), so the kernel is pure latency: 1024 threads, 8-byte loads, four independent partial sums per thread.
template <typename T>
__global__ void __launch_bounds__(FT)
gn_finalize_fwd_kernel(const float* __restrict__ partial, const T* __restrict__ x, float* __restrict__ mean,
                       float* __restrict__ rstd, GnGeom g) {
  __shared__ double s0[FT], s1[FT];
  __shared__ float piv[FT];                           // pivots of this group's channels (when cpg <= FT)
  const int b = blockIdx.x / g.G, grp = blockIdx.x % g.G;
  const int cpg = g.C / g.G, n_el = g.chunks * cpg;
  const bool lds_piv = cpg <= FT;
  // Round 5: the kernel is a chain of round trips (pivot, then records four at a time: ~10 us for 100 KB).  Now the pivot
  // loads and the first kFinBatch records of every thread are requested together, and combined after one barrier.
  constexpr int kFinBatch = 16;
  float pf_own = 0.0f;
  if (lds_piv && (int)threadIdx.x < cpg) pf_own = ld1(x + (size_t)b * g.S * g.C + grp * cpg + threadIdx.x);
  double a[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  bool first = true;
  for (int i0 = threadIdx.x; i0 < n_el || first; i0 += kFinBatch * FT) {
    float2 rec[kFinBatch];
#pragma unroll
    for (int k = 0; k < kFinBatch; ++k) {
      const int i = i0 + k * FT;
      rec[k] = make_float2(0.f, 0.f);
      if (i < n_el) {
        const int chunk = i / cpg, c = grp * cpg + i % cpg;
        rec[k] = *reinterpret_cast<const float2*>(partial + ((size_t)(b * g.chunks + chunk) * g.C + c) * 2);
      }
    }
    if (first) {
      if (lds_piv && (int)threadIdx.x < cpg) piv[threadIdx.x] = g.pre ? gelu_f(pf_own) : pf_own;
      __syncthreads();
      first = false;
    }
#pragma unroll
    for (int k = 0; k < kFinBatch; ++k) {
      const int i = i0 + k * FT;
      if (i < n_el) {
        const int chunk = i / cpg, c = grp * cpg + i % cpg;
        // shifted partials (pivot = the channel's value at voxel 0, see gn_partial_kernel): sum u = sum d + n p,
        // sum u^2 = sum d^2 + 2 p sum d + n p^2, with n = voxels of this chunk
        float pf = lds_piv ? piv[i % cpg] : ld1(x + (size_t)b * g.S * g.C + c);
        if (!lds_piv && g.pre) pf = gelu_f(pf);
        const double pd = pf;
        const double n = (double)(min(g.S, (long)(chunk + 1) * g.chunk_len) - (long)chunk * g.chunk_len);
        a[k & 3] += (double)rec[k].x + n * pd;
        q[k & 3] += (double)rec[k].y + 2.0 * pd * (double)rec[k].x + n * pd * pd;
      }
    }
  }
  double sa = (a[0] + a[1]) + (a[2] + a[3]), sq = (q[0] + q[1]) + (q[2] + q[3]);
  block_sum2(sa, sq, s0, s1);
  if (threadIdx.x == 0) {
    const double n = (double)g.S * cpg;
    const double m = sa / n;
    double var = sq / n - m * m;
    if (var < 0.0) var = 0.0;
    mean[blockIdx.x] = (float)m;
    rstd[blockIdx.x] = (float)(1.0 / sqrt(var + (double)g.eps));
  }
}

// Per-channel statistics (BatchNorm: one channel per group): one WAVE per (b, channel), lanes striding over the chunk
// partials, butterfly sum.  The image branch runs ~170 BatchNorms per pass, most of them on maps of a few thousand
// pixels where the workgroup-per-group kernels above (1024 threads, ten barriers) are pure latency.
template <typename T>
__global__ void __launch_bounds__(256)
bn_finalize_fwd_flat_kernel(const float* __restrict__ partial, const T* __restrict__ x, float* __restrict__ mean,
                            float* __restrict__ rstd, GnGeom g, float* __restrict__ rm, float* __restrict__ rv, float mom,
                            float unbias) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= g.B * g.C) return;
  const int b = i / g.C, c = i % g.C;
  double a = 0.0, q = 0.0;
  float pf = ld1(x + (size_t)b * g.S * g.C + c);          // the pivot travels with the first batch of records
  constexpr int kFlatBatch = 12;                           // 768 chunks / 64 lanes: one batch
  for (int chunk0 = lane; chunk0 < g.chunks; chunk0 += 64 * kFlatBatch) {
    float2 rec[kFlatBatch];
#pragma unroll
    for (int k = 0; k < kFlatBatch; ++k) {
      const int chunk = chunk0 + 64 * k;
      rec[k] = chunk < g.chunks ? *reinterpret_cast<const float2*>(partial + ((size_t)(b * g.chunks + chunk) * g.C + c) * 2)
                                : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < kFlatBatch; ++k) { a += rec[k].x; q += rec[k].y; }
  }
  a = wave_sum_d(a); q = wave_sum_d(q);
  if (lane == 0) {
    // one channel per group: the statistics of the SHIFTED values d = u - p directly (mean = p + E[d], var = var(d))
    if (g.pre) pf = gelu_f(pf);
    const double n = (double)g.S, md = a / n;
    double var = q / n - md * md;
    if (var < 0.0) var = 0.0;
    const double m = (double)pf + md;
    const float mf = (float)m, rs = (float)(1.0 / sqrt(var + (double)g.eps));
    mean[i] = mf;
    rstd[i] = rs;
    if (rm && b == 0) {                                 // BatchNorm running statistics (round 5: was a launch of its own)
      const float vf = 1.0f / (rs * rs) - g.eps;        // as bn_update_running_kernel computes it from the stored rstd
      rm[c] = (1.0f - mom) * rm[c] + mom * mf;
      rv[c] = (1.0f - mom) * rv[c] + mom * (vf * unbias);
    }
  }
}

__global__ void __launch_bounds__(256)
bn_finalize_bwd_flat_kernel(const float* __restrict__ partial, const float* __restrict__ gamma, float* __restrict__ coef,
                            float* __restrict__ dgamma, float* __restrict__ dbeta, GnGeom g) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= g.C) return;
  double tb = 0.0, ts = 0.0;
  const double gm = gamma[c], n = (double)g.S;
  for (int b = 0; b < g.B; ++b) {
    double sb = 0.0, ss = 0.0;
    constexpr int kFlatBatch = 12;
    for (int chunk0 = lane; chunk0 < g.chunks; chunk0 += 64 * kFlatBatch) {
      float2 rec[kFlatBatch];
#pragma unroll
      for (int k = 0; k < kFlatBatch; ++k) {
        const int chunk = chunk0 + 64 * k;
        rec[k] = chunk < g.chunks ? *reinterpret_cast<const float2*>(partial + ((size_t)(b * g.chunks + chunk) * g.C + c) * 2)
                                  : make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int k = 0; k < kFlatBatch; ++k) { sb += rec[k].x; ss += rec[k].y; }
    }
    sb = wave_sum_d(sb); ss = wave_sum_d(ss);
    if (lane == 0) {
      coef[(b * g.C + c) * 2 + 0] = (float)(gm * ss / n);
      coef[(b * g.C + c) * 2 + 1] = (float)(gm * sb / n);
    }
    tb += sb; ts += ss;
  }
  if (lane == 0) { dbeta[c] = (float)tb; dgamma[c] = (float)ts; }
}

// ---- ReLU bit mask of the streaming apply kernels (VW = 4 lanes) ------------------------------------------------------------------
// Word (i / 64) * 4 + k holds bit (i % 64) = [component k of lane-vector i > 0]; the 64 lanes of a wave own 64 consecutive,
// 64-aligned vectors (block offsets and the grid stride are multiples of 256), so a mask word is exactly a wave's LANE MASK of one
// component.  Round 5: the writer stores the four ballots of a step as ONE 8-byte store from lanes 0..3 (was: four predicated
// stores from lane 0 behind four branches): forward apply on the 189 MB activation (tools/gn_dtype_probe.py) bf16 94.6 -> 82.4 us,
// fp32 65.8 -> 60.5 us (without any mask: 75.0 / 58.3).
__device__ __forceinline__ void relu_mask_store4(unsigned long long* __restrict__ mask, long i, const float (&v)[4]) {
  unsigned long long bal[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) bal[k] = __ballot(v[k] > 0.0f);
  const int ln = threadIdx.x & 63;
  const unsigned long long act = __ballot(1);
  if (act == ~0ull || __popcll(act) >= 4) {          // lanes 0..3 are active (a partial wave is a prefix of lanes)
    if (ln < 4) {
      unsigned long long w = bal[0];
      w = ln == 1 ? bal[1] : w; w = ln == 2 ? bal[2] : w; w = ln == 3 ? bal[3] : w;
      mask[(i >> 6) * 4 + ln] = w;
    }
  } else if (ln == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) mask[(i >> 6) * 4 + k] = bal[k];
  }
}
// Reader: the word index is made wave-uniform with v_readfirstlane on the INDEX (no memory wait), so the four words of a step
// come through the scalar cache (s_load, no vector-memory instruction: the bf16 passes issue twice the VMEM instructions of the
// fp32 ones for the same bytes) straight into SGPR pairs, and each is the select mask of ONE v_cndmask_b32.
__device__ __forceinline__ float lane_select(unsigned long long wave_mask, float v) {     // wave_mask: wave-uniform (SGPR pair)
  float o;
  asm volatile("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(o) : "v"(v), "s"(wave_mask));
  return o;
}
__device__ __forceinline__ void relu_mask_apply4(const unsigned long long* __restrict__ mask, long i, float (&gs)[4]) {
  const long wq = i >> 6;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)wq), hi = __builtin_amdgcn_readfirstlane((unsigned)(wq >> 32));
  const unsigned long long* mw = mask + ((((unsigned long long)hi << 32) | lo) << 2);
  const unsigned long long w0 = mw[0], w1 = mw[1], w2 = mw[2], w3 = mw[3];      // one s_load_dwordx8, one wait
  gs[0] = lane_select(w0, gs[0]); gs[1] = lane_select(w1, gs[1]); gs[2] = lane_select(w2, gs[2]); gs[3] = lane_select(w3, gs[3]);
}

// (The reader side was tried the same way -- each word into an SGPR pair through v_readfirstlane, applied as the select mask of one
// v_cndmask_b32 -- and LOSES: bf16 backward apply 129.7 -> 165 us, fp32 92.7 -> 101.5: the readfirstlanes wait for the mask loads in
// front of everything else.  The shift / and / select form stays.)

// Sample index of lane-vector i along a grid-stride walk: one 64-bit division per THREAD instead of one per step (round 5: the
// division was ~60 VALU instructions in front of every 8 / 16 bytes -- the bf16 apply passes, with twice the elements per byte,
// were VALU-bound on it: 42 / 60 us where the fp32 passes take 29 / 36 us for the same bytes).
struct SampleWalk {
  long per, rem;
  int b;
  __device__ __forceinline__ SampleWalk() : per(1), rem(0), b(0) {}
  __device__ __forceinline__ SampleWalk(long i, long per_) : per(per_) { b = (int)(i / per_); rem = i - (long)b * per_; }
  __device__ __forceinline__ void step(long stride) {
    rem += stride;
    while (rem >= per) { rem -= per; ++b; }
  }
};

// y = (x - mean) * rstd * gamma + beta (+ residual) (ReLU)
template <bool PRE, typename T = float, int VW = 4>
__global__ void __launch_bounds__(NT)
gn_apply_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                    const T* __restrict__ res, const float* __restrict__ mean, const float* __restrict__ rstd,
                    T* __restrict__ y, unsigned long long* __restrict__ mask, GnGeom g, long totalv) {
  // bf16 tensors: the per-channel constants are folded to one fma per element (the stored result is rounded to 8 bits of
  // mantissa anyway); fp32 tensors keep the reference's operation order (x - mean) * rstd * gamma + beta
  constexpr bool FOLD = sizeof(T) == 2;
  const int q = g.C / VW, cpg = g.C / g.G;
  const long stride = (long)gridDim.x * NT;
  const bool fixed = stride % q == 0;            // see gn_apply_bwd_kernel
  long i = (long)blockIdx.x * NT + threadIdx.x;
  int c = (int)(i % q) * VW, bcur = -1;
  float gam[VW], bet[VW], mu[VW], rs[VW];
#pragma unroll
  for (int k = 0; k < VW; ++k) { gam[k] = gamma[c + k]; bet[k] = beta[c + k]; mu[k] = 0.f; rs[k] = 0.f; }
  SampleWalk sw(i, (long)q * g.S);
  for (; i < totalv; i += stride, sw.step(stride)) {
    float v[VW], rr[VW];
    ldn<VW>(x + (size_t)VW * i, v);
#pragma unroll
    for (int k = 0; k < VW; ++k) rr[k] = 0.0f;
    if (res) ldn<VW>(res + (size_t)VW * i, rr);
    const int b = sw.b;
    if (!fixed) {
      c = (int)(i % q) * VW;
#pragma unroll
      for (int k = 0; k < VW; ++k) { gam[k] = gamma[c + k]; bet[k] = beta[c + k]; }
    }
    if (!fixed || b != bcur) {
      bcur = b;
#pragma unroll
      for (int k = 0; k < VW; ++k) {
        const int grp = b * g.G + (c + k) / cpg;
        mu[k] = mean[grp]; rs[k] = rstd[grp];
        if (FOLD) { rs[k] *= gam[k]; mu[k] = bet[k] - mu[k] * rs[k]; }       // y = rs' x + mu'
      }
    }
    if (PRE) {
#pragma unroll
      for (int k = 0; k < VW; ++k) v[k] = gelu_f(v[k]);
    }
#pragma unroll
    for (int k = 0; k < VW; ++k) {
      const float o = FOLD ? fmaf(v[k], rs[k], mu[k]) + rr[k] : (v[k] - mu[k]) * rs[k] * gam[k] + bet[k] + rr[k];
      v[k] = g.relu ? fmaxf(o, 0.0f) : o;
    }
    if (mask) {
      // ReLU mask for the backward pass: word (i / 64) * VW + k holds bit (i % 64) = [component k of lane-vector i is > 0].
      // The 64 lanes of a wave own 64 consecutive vectors (block offsets and the grid stride are multiples of 256), so one
      // ballot per component is exactly one mask word; backward then reads 1 bit per element instead of the 189 MB of y.
      if constexpr (VW == 4) relu_mask_store4(mask, i, v);
      else {
#pragma unroll
        for (int k = 0; k < VW; ++k) {
          const unsigned long long bal = __ballot(v[k] > 0.0f);
          if ((threadIdx.x & 63) == 0) mask[(i >> 6) * VW + k] = bal;
        }
      }
    }
    if (g.ldy == g.C) stn<VW>(y + (size_t)VW * i, v);
    else stn<VW>(y + (i / q) * g.ldy + (i % q) * VW, v);
  }
}

// backward finalize, ONE launch for both small reductions over the chunk partials:
//   workgroups [0, B*G):      per (b, group)  ds/n = sum_c gamma_c * sum(g * xhat), db/n = sum_c gamma_c * sum(g)
//   workgroups [B*G, B*G+C):  per channel     dbeta[c] = sum g, dgamma[c] = sum g * xhat over samples and chunks
__global__ void __launch_bounds__(FT)
gn_finalize_bwd_kernel(const float* __restrict__ partial, const float* __restrict__ gamma, float* __restrict__ coef,
                       float* __restrict__ dgamma, float* __restrict__ dbeta, GnGeom g) {
  __shared__ double r0[FT], r1[FT];
  const int nbg = g.B * g.G;
  if ((int)blockIdx.x < nbg) {
    const int b = blockIdx.x / g.G, grp = blockIdx.x % g.G;
    const int cpg = g.C / g.G, n_el = g.chunks * cpg;
    double ds[4] = {0, 0, 0, 0}, db[4] = {0, 0, 0, 0};
    constexpr int kFinBatch = 16;                     // all records of a thread in flight together (see gn_finalize_fwd_kernel)
    for (int i0 = threadIdx.x; i0 < n_el; i0 += kFinBatch * FT) {
      float2 rec[kFinBatch];
      float gmf[kFinBatch];
#pragma unroll
      for (int k = 0; k < kFinBatch; ++k) {
        const int i = i0 + k * FT;
        rec[k] = make_float2(0.f, 0.f);
        gmf[k] = 0.0f;
        if (i < n_el) {
          const int chunk = i / cpg, c = grp * cpg + i % cpg;
          rec[k] = *reinterpret_cast<const float2*>(partial + ((size_t)(b * g.chunks + chunk) * g.C + c) * 2);
          gmf[k] = gamma[c];
        }
      }
#pragma unroll
      for (int k = 0; k < kFinBatch; ++k) {
        const double gm = gmf[k];
        db[k & 3] += gm * rec[k].x;
        ds[k & 3] += gm * rec[k].y;
      }
    }
    double s = (ds[0] + ds[1]) + (ds[2] + ds[3]), t = (db[0] + db[1]) + (db[2] + db[3]);
    block_sum2(s, t, r0, r1);
    if (threadIdx.x == 0) {
      const double n = (double)g.S * cpg;
      coef[blockIdx.x * 2 + 0] = (float)(s / n);
      coef[blockIdx.x * 2 + 1] = (float)(t / n);
    }
  } else {
    const int c = blockIdx.x - nbg, n_el = g.B * g.chunks;
    double a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
    for (int i0 = threadIdx.x; i0 < n_el; i0 += 4 * FT) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * FT;
        if (i < n_el) {
          const float2 p = *reinterpret_cast<const float2*>(partial + ((size_t)i * g.C + c) * 2);
          a0[k] += p.x; a1[k] += p.y;
        }
      }
    }
    double s = (a0[0] + a0[1]) + (a0[2] + a0[3]), t = (a1[0] + a1[1]) + (a1[2] + a1[3]);
    block_sum2(s, t, r0, r1);
    if (threadIdx.x == 0) { dbeta[c] = (float)s; dgamma[c] = (float)t; }
  }
}

// gx = (gamma * g - xhat * ds/n - db/n) * rstd ;  gres = g  (g = gy masked by the fused ReLU)
template <bool PRE, typename T = float, int VW = 4>
__global__ void __launch_bounds__(NT)
gn_apply_bwd_kernel(const T* __restrict__ gy, const T* __restrict__ x, const T* __restrict__ y,
                    const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
                    const float* __restrict__ coef, T* __restrict__ gx, T* __restrict__ gres,
                    const unsigned long long* __restrict__ mask, GnGeom g, long totalv) {
  const int q = g.C / VW, cpg = g.C / g.G;
  const long stride = (long)gridDim.x * NT;
  // the launch makes `stride` a multiple of q whenever it can: a thread then always owns the same channels and
  // the per-channel / per-group constants leave the streaming loop
  constexpr bool FOLD = sizeof(T) == 2;          // bf16 tensors: folded constants, see gn_apply_fwd_kernel
  const bool fixed = stride % q == 0;
  long i = (long)blockIdx.x * NT + threadIdx.x;
  int c = (int)(i % q) * VW, bcur = -1;
  float gam[VW], mu[VW], rs[VW], c0[VW], c1[VW];
#pragma unroll
  for (int k = 0; k < VW; ++k) { gam[k] = gamma[c + k]; mu[k] = 0.f; rs[k] = 0.f; c0[k] = 0.f; c1[k] = 0.f; }
  constexpr bool PF = sizeof(T) == 2;            // prefetch one step ahead (bf16 tensors), see gn_apply_fwd_kernel
  float nx[VW], ng[VW];
#pragma unroll
  for (int k = 0; k < VW; ++k) { nx[k] = 0.0f; ng[k] = 0.0f; }
  if (PF && i < totalv) {
    ldn<VW>(x + (size_t)VW * i, nx);
    if (g.ldg == g.C) ldn<VW>(gy + (size_t)VW * i, ng);
    else ldn<VW>(gy + (i / q) * g.ldg + (i % q) * VW, ng);
  }
  SampleWalk sw(i, (long)q * g.S);
  for (; i < totalv; i += stride, sw.step(stride)) {
    const int b = sw.b;
    if (!fixed) {
      c = (int)(i % q) * VW;
#pragma unroll
      for (int k = 0; k < VW; ++k) gam[k] = gamma[c + k];
    }
    if (!fixed || b != bcur) {
      bcur = b;
#pragma unroll
      for (int k = 0; k < VW; ++k) {
        const int grp = b * g.G + (c + k) / cpg;
        mu[k] = mean[grp]; rs[k] = rstd[grp]; c0[k] = coef[grp * 2]; c1[k] = coef[grp * 2 + 1];
        if (FOLD) { c0[k] = -rs[k] * rs[k] * c0[k]; c1[k] = -c1[k] * rs[k]; rs[k] *= gam[k]; }   // gx = rs' g + c0' (u - mu) + c1'
      }
    }
    float xs[VW], gs[VW];
    if (PF) {
#pragma unroll
      for (int k = 0; k < VW; ++k) { xs[k] = nx[k]; gs[k] = ng[k]; }
      const long in = i + stride;
      if (in < totalv) {
        ldn<VW>(x + (size_t)VW * in, nx);
        if (g.ldg == g.C) ldn<VW>(gy + (size_t)VW * in, ng);
        else ldn<VW>(gy + (in / q) * g.ldg + (in % q) * VW, ng);
      }
    } else {
      ldn<VW>(x + (size_t)VW * i, xs);
      if (g.ldg == g.C) ldn<VW>(gy + (size_t)VW * i, gs);
      else ldn<VW>(gy + (i / q) * g.ldg + (i % q) * VW, gs);
    }
    if (g.relu && mask) {
      if constexpr (VW == 4) relu_mask_apply4(mask, i, gs);
      else {
        const unsigned long long* mw = mask + (i >> 6) * VW;
        const int sh = (int)(i & 63);
#pragma unroll
        for (int k = 0; k < VW; ++k) gs[k] = ((mw[k] >> sh) & 1ull) ? gs[k] : 0.0f;
      }
    } else if (g.relu) {
      float yv[VW];
      ldn<VW>(y + (size_t)VW * i, yv);
#pragma unroll
      for (int k = 0; k < VW; ++k) gs[k] = yv[k] > 0.0f ? gs[k] : 0.0f;
    }
    float o[VW];
#pragma unroll
    for (int k = 0; k < VW; ++k) {
      float u = xs[k], du = 1.0f;
      if (PRE) gelu_both(xs[k], u, du);
      if (FOLD) {
        const float t = fmaf(rs[k], gs[k], fmaf(c0[k], u - mu[k], c1[k]));
        o[k] = PRE ? t * du : t;
      } else {
        const float xh = (u - mu[k]) * rs[k];
        o[k] = (gam[k] * gs[k] - xh * c0[k] - c1[k]) * rs[k] * du;
      }
    }
    stn<VW>(gx + (size_t)VW * i, o);
    if (gres) stn<VW>(gres + (size_t)VW * i, gs);
  }
}

// grid of the streaming apply kernels: <= 16384 workgroups, and (workgroups * NT) a multiple of the float4 count per
// voxel q so that every thread keeps its channels for the whole grid-stride loop
unsigned apply_blocks(long total4, int q) {
  long blocks = min((long)cdiv(total4, NT), 16384L);
  long m = q;                                   // smallest block multiple: q / gcd(q, NT)
  for (long a = NT, bb = q; bb;) { const long t = a % bb; a = bb; bb = t; m = q / a; }
  if (blocks >= m) blocks -= blocks % m;
  return (unsigned)(blocks > 0 ? blocks : 1);
}

bool gn_ok(const ssbev_norm_dims* d) {
  if (d && ((d->ld_y != 0 && (d->ld_y < d->C || d->ld_y % 4 != 0)) || (d->ld_gy != 0 && (d->ld_gy < d->C || d->ld_gy % 4 != 0))))
    return false;
  return d && d->B > 0 && d->S > 0 && d->C > 0 && d->G > 0 && d->C % d->G == 0 && d->C % 4 == 0 && (d->pre_act == 0 || d->pre_act == 1) &&
         (d->io_dtype == 0 || d->io_dtype == 1);
}

// channels per lane of the STATISTICS passes: 8 for bf16 tensors whose channel count allows 16-byte lanes, else 4 (round 4,
// measured per launch on the two-sample step: statistics passes 47 -> 32 us and 18.6 -> 19.6 us with 8; the apply passes got
// SLOWER with 8 -- 41 -> 56 us forward, 60 -> 74 us backward: twice the registers for the per-channel constants, half the
// waves in flight -- and stay at 4.  The ReLU bit mask has ONE layout, per channel quad, read by both widths.)
int gn_vw(int io_dtype, int C) { return (io_dtype == 1 && C % 8 == 0) ? 8 : 4; }

GnGeom make_geom(const ssbev_norm_dims* d) {
  GnGeom g;
  g.B = d->B; g.C = d->C; g.G = d->G; g.S = d->S; g.eps = d->eps; g.relu = d->relu; g.pre = d->pre_act;
  g.ldy = d->ld_y > 0 ? d->ld_y : d->C;
  g.ldg = d->ld_gy > 0 ? d->ld_gy : d->C;
  // ~768 workgroups over the chip (3 per CU) = samples x chunks x channel slabs, each chunk at least 64 voxels
  const int vw = gn_vw(d->io_dtype, d->C);
  g.slab_q = std::min(d->C / vw, kSlabLanes);
  const long slabs = (d->C / vw + g.slab_q - 1) / g.slab_q;
  long chunks = kStatBlocks / ((long)d->B * slabs);
  if (chunks < 1) chunks = 1;
  long len = (d->S + chunks - 1) / chunks;
  if (len < 64) len = 64;
  g.chunk_len = len;
  g.chunks = (int)((d->S + len - 1) / len);
  return g;
}

size_t lds_bytes(const GnGeom& g, int vw = 4) {
  const int q = std::min(g.C / vw, g.slab_q);         // widest slab; narrower ones need rows * q <= PNT entries too
  const int rows = PNT / q > 0 ? PNT / q : 1;
  return (size_t)std::max(rows * q, PNT) * vw * 2 * sizeof(float);
}

unsigned gn_slabs(const GnGeom& g, int vw = 4) { return cdiv((size_t)(g.C / vw), g.slab_q); }


// running statistics of a training-mode BatchNorm (nn.BatchNorm semantics: unbiased variance in the running buffer)
__global__ void bn_update_running_kernel(const float* __restrict__ mean, const float* __restrict__ rstd,
                                         float* __restrict__ rm, float* __restrict__ rv, int C, float m, float eps,
                                         float unbias) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float var = 1.0f / (rstd[c] * rstd[c]) - eps;
  rm[c] = (1.0f - m) * rm[c] + m * mean[c];
  rv[c] = (1.0f - m) * rv[c] + m * (var * unbias);
}


// ------------------------------------------------------------------------------------------------
// Two normalisations, one sum: y = relu?(N_a(xa) + N_b(xb)).  The tail of every hourglass level (VT:92-95) is
//     conv5 = relu(BatchNorm(deconv(conv4)) + GN(redir2(conv2)))        conv6 = relu(BatchNorm(deconv(conv5)) + GN(redir1(x)))
// Run as two normalisation operators the BatchNorm output makes a round trip through HBM only to be added (write 189 MB,
// read 189 MB at full resolution) and the backward pass reads the incoming gradient four times.  Here the apply pass reads
// the two RAW tensors and writes the sum once; backward reads (gy, mask, xa, xb) once for the three per-channel sums
// (sum g, sum g xhat_a, sum g xhat_b: sum g is shared) and once more to emit both input gradients.
// Each of the two norms is either per-sample GroupNorm (statistics [B][G]) or BatchNorm over the batch (statistics [C]).
struct Gn2Geom {
  int B, C, Ga, Gb;
  long S;
  int chunks;
  long chunk_len;
  int relu, a_batch, b_batch;
};

template <bool BWD, int VW>
__device__ __forceinline__ void gn2_load_stats(const Gn2Geom& g, int b, int c, const float* mean_a, const float* rstd_a,
                                               const float* mean_b, const float* rstd_b, const float* coef_a, const float* coef_b,
                                               float (&mua)[VW], float (&rsa)[VW], float (&mub)[VW], float (&rsb)[VW],
                                               float (&c0a)[VW], float (&c1a)[VW], float (&c0b)[VW], float (&c1b)[VW]) {
  const int cpa = g.C / g.Ga, cpb = g.C / g.Gb;
#pragma unroll
  for (int k = 0; k < VW; ++k) {
    const int ia = (g.a_batch ? 0 : b) * g.Ga + (c + k) / cpa, ib = (g.b_batch ? 0 : b) * g.Gb + (c + k) / cpb;
    mua[k] = mean_a[ia]; rsa[k] = rstd_a[ia]; mub[k] = mean_b[ib]; rsb[k] = rstd_b[ib];
    if (BWD) { c0a[k] = coef_a[ia * 2]; c1a[k] = coef_a[ia * 2 + 1]; c0b[k] = coef_b[ib * 2]; c1b[k] = coef_b[ib * 2 + 1]; }
  }
}

template <typename T, int VW>
__global__ void __launch_bounds__(NT)
gn2_apply_fwd_kernel(const T* __restrict__ xa, const float* __restrict__ gamma_a, const float* __restrict__ beta_a,
                     const float* __restrict__ mean_a, const float* __restrict__ rstd_a, const T* __restrict__ xb,
                     const float* __restrict__ gamma_b, const float* __restrict__ beta_b, const float* __restrict__ mean_b,
                     const float* __restrict__ rstd_b, T* __restrict__ y, unsigned long long* __restrict__ mask, Gn2Geom g,
                     long totalv) {
  const int q = g.C / VW;
  const long stride = (long)gridDim.x * NT;
  const bool fixed = stride % q == 0;
  long i = (long)blockIdx.x * NT + threadIdx.x;
  int c = (int)(i % q) * VW, bcur = -1;
  float ga[VW], ba[VW], gb[VW], bb[VW], mua[VW], rsa[VW], mub[VW], rsb[VW], d0[VW], d1[VW], d2[VW], d3[VW];
#pragma unroll
  for (int k = 0; k < VW; ++k) { ga[k] = gamma_a[c + k]; ba[k] = beta_a[c + k]; gb[k] = gamma_b[c + k]; bb[k] = beta_b[c + k]; }
  SampleWalk sw(i, (long)q * g.S);
  for (; i < totalv; i += stride, sw.step(stride)) {
    const int b = sw.b;
    if (!fixed) {
      c = (int)(i % q) * VW;
#pragma unroll
      for (int k = 0; k < VW; ++k) { ga[k] = gamma_a[c + k]; ba[k] = beta_a[c + k]; gb[k] = gamma_b[c + k]; bb[k] = beta_b[c + k]; }
    }
    if (!fixed || b != bcur) {
      bcur = b;
      gn2_load_stats<false, VW>(g, b, c, mean_a, rstd_a, mean_b, rstd_b, nullptr, nullptr, mua, rsa, mub, rsb, d0, d1, d2, d3);
    }
    float as[VW], bs[VW], v[VW];
    ldn<VW>(xa + (size_t)VW * i, as);
    ldn<VW>(xb + (size_t)VW * i, bs);
#pragma unroll
    for (int k = 0; k < VW; ++k) {
      // the same association as gn_apply_fwd_kernel with the other norm's output as its residual
      const float o = (as[k] - mua[k]) * rsa[k] * ga[k] + ba[k] + ((bs[k] - mub[k]) * rsb[k] * gb[k] + bb[k]);
      v[k] = g.relu ? fmaxf(o, 0.0f) : o;
    }
    if (mask) {
      if constexpr (VW == 4) relu_mask_store4(mask, i, v);
      else {
#pragma unroll
        for (int k = 0; k < VW; ++k) {
          const unsigned long long bal = __ballot(v[k] > 0.0f);
          if ((threadIdx.x & 63) == 0) mask[(i >> 6) * VW + k] = bal;
        }
      }
    }
    stn<VW>(y + (size_t)VW * i, v);
  }
}

// per-chunk (sum g, sum g xhat_a) -> pa, (sum g, sum g xhat_b) -> pb, both in gn_partial_kernel<1>'s layout (so the
// finalize kernels of the single-norm operator serve unchanged); g = gy masked by the fused ReLU
template <typename T, int VW>
__global__ void __launch_bounds__(PNT, 3)
gn2_partial_bwd_kernel(const T* __restrict__ gy, const unsigned long long* __restrict__ mask, const T* __restrict__ xa,
                       const float* __restrict__ mean_a, const float* __restrict__ rstd_a, const T* __restrict__ xb,
                       const float* __restrict__ mean_b, const float* __restrict__ rstd_b, float* __restrict__ pa,
                       float* __restrict__ pb, Gn2Geom g) {
  extern __shared__ __align__(16) float lds[];         // [rows][C][3]
  const int q = g.C / VW, rows = PNT / q > 0 ? PNT / q : 1;
  const int tid = threadIdx.x, b = blockIdx.y, chunk = blockIdx.x;
  const long s0 = (long)chunk * g.chunk_len, s1 = min(g.S, s0 + g.chunk_len);
  const int c4 = tid % q, r = tid / q;
  if (r < rows) {
    const int c = c4 * VW;
    float mua[VW], rsa[VW], mub[VW], rsb[VW], d0[VW], d1[VW], d2[VW], d3[VW];
    gn2_load_stats<false, VW>(g, b, c, mean_a, rstd_a, mean_b, rstd_b, nullptr, nullptr, mua, rsa, mub, rsb, d0, d1, d2, d3);
    float a0[VW], a1[VW], a2[VW];
#pragma unroll
    for (int k = 0; k < VW; ++k) { a0[k] = 0.0f; a1[k] = 0.0f; a2[k] = 0.0f; }
    const size_t base = (size_t)b * g.S * g.C;
    for (long s = s0 + r; s < s1; s += rows) {
      const size_t off = base + (size_t)s * g.C + c;
      float gs[VW], as[VW], bs[VW];
      ldn<VW>(gy + off, gs);
      ldn<VW>(xa + off, as);
      ldn<VW>(xb + off, bs);
      if (g.relu) {                                     // quad-layout mask, see gn_partial_kernel
        const size_t i4 = off >> 2;
        const unsigned long long* mw = mask + (i4 >> 6) * 4;
        const int sh = (int)(i4 & 63);
#pragma unroll
        for (int k = 0; k < VW; ++k) gs[k] = ((mw[k & 3] >> (sh + (k >> 2))) & 1ull) ? gs[k] : 0.0f;
      }
#pragma unroll
      for (int k = 0; k < VW; ++k) {
        a0[k] += gs[k];
        a1[k] += gs[k] * (as[k] - mua[k]) * rsa[k];
        a2[k] += gs[k] * (bs[k] - mub[k]) * rsb[k];
      }
    }
#pragma unroll
    for (int k = 0; k < VW; ++k) {
      float* l = lds + ((size_t)r * g.C + c + k) * 3;
      l[0] = a0[k]; l[1] = a1[k]; l[2] = a2[k];
    }
  }
  __syncthreads();
  for (int t = tid; t < g.C; t += PNT) {
    float s0v = 0.0f, s1v = 0.0f, s2v = 0.0f;
    for (int rr = 0; rr < rows; ++rr) {
      const float* l = lds + ((size_t)rr * g.C + t) * 3;
      s0v += l[0]; s1v += l[1]; s2v += l[2];
    }
    const size_t o = ((size_t)(b * g.chunks + chunk) * g.C + t) * 2;
    st_record(pa + o, s0v, s1v);
    st_record(pb + o, s0v, s2v);
  }
}

template <typename T, int VW>
__global__ void __launch_bounds__(NT)
gn2_apply_bwd_kernel(const T* __restrict__ gy, const unsigned long long* __restrict__ mask, const T* __restrict__ xa,
                     const float* __restrict__ gamma_a, const float* __restrict__ mean_a, const float* __restrict__ rstd_a,
                     const float* __restrict__ coef_a, const T* __restrict__ xb, const float* __restrict__ gamma_b,
                     const float* __restrict__ mean_b, const float* __restrict__ rstd_b, const float* __restrict__ coef_b,
                     T* __restrict__ gxa, T* __restrict__ gxb, Gn2Geom g, long totalv) {
  const int q = g.C / VW;
  const long stride = (long)gridDim.x * NT;
  const bool fixed = stride % q == 0;
  long i = (long)blockIdx.x * NT + threadIdx.x;
  int c = (int)(i % q) * VW, bcur = -1;
  float ga[VW], gb[VW], mua[VW], rsa[VW], mub[VW], rsb[VW], c0a[VW], c1a[VW], c0b[VW], c1b[VW];
#pragma unroll
  for (int k = 0; k < VW; ++k) { ga[k] = gamma_a[c + k]; gb[k] = gamma_b[c + k]; }
  SampleWalk sw(i, (long)q * g.S);
  for (; i < totalv; i += stride, sw.step(stride)) {
    const int b = sw.b;
    if (!fixed) {
      c = (int)(i % q) * VW;
#pragma unroll
      for (int k = 0; k < VW; ++k) { ga[k] = gamma_a[c + k]; gb[k] = gamma_b[c + k]; }
    }
    if (!fixed || b != bcur) {
      bcur = b;
      gn2_load_stats<true, VW>(g, b, c, mean_a, rstd_a, mean_b, rstd_b, coef_a, coef_b, mua, rsa, mub, rsb, c0a, c1a, c0b, c1b);
    }
    float gs[VW], as[VW], bs[VW];
    ldn<VW>(gy + (size_t)VW * i, gs);
    ldn<VW>(xa + (size_t)VW * i, as);
    ldn<VW>(xb + (size_t)VW * i, bs);
    if (g.relu) {
      if constexpr (VW == 4) relu_mask_apply4(mask, i, gs);
      else {
        const unsigned long long* mw = mask + (i >> 6) * VW;
        const int sh = (int)(i & 63);
#pragma unroll
        for (int k = 0; k < VW; ++k) gs[k] = ((mw[k] >> sh) & 1ull) ? gs[k] : 0.0f;
      }
    }
    float oa[VW], ob[VW];
#pragma unroll
    for (int k = 0; k < VW; ++k) {
      oa[k] = (ga[k] * gs[k] - (as[k] - mua[k]) * rsa[k] * c0a[k] - c1a[k]) * rsa[k];
      ob[k] = (gb[k] * gs[k] - (bs[k] - mub[k]) * rsb[k] * c0b[k] - c1b[k]) * rsb[k];
    }
    stn<VW>(gxa + (size_t)VW * i, oa);
    stn<VW>(gxb + (size_t)VW * i, ob);
  }
}

bool gn2_ok(const ssbev_norm2_dims* d) {
  return d && d->B > 0 && d->S > 0 && d->C > 0 && d->C % 4 == 0 && d->C <= 4 * NT && d->Ga > 0 && d->Gb > 0 &&
         d->C % d->Ga == 0 && d->C % d->Gb == 0;
}

// the single-norm view of one side (statistics passes and finalize kernels are those of the single-norm operator)
ssbev_norm_dims gn2_side(const ssbev_norm2_dims* d, int side) {
  ssbev_norm_dims n;
  const bool batch = side ? d->b_batch : d->a_batch;
  n.B = batch ? 1 : d->B; n.C = d->C; n.G = side ? d->Gb : d->Ga; n.S = batch ? d->S * d->B : d->S;
  n.eps = side ? d->eps_b : d->eps_a; n.relu = d->relu; n.stats_given = 0; n.pre_act = 0; n.ld_y = 0; n.ld_gy = 0;
  n.io_dtype = d->io_dtype;
  return n;
}

Gn2Geom make_geom2(const ssbev_norm2_dims* d) {
  Gn2Geom g;
  g.B = d->B; g.C = d->C; g.Ga = d->Ga; g.Gb = d->Gb; g.S = d->S; g.relu = d->relu; g.a_batch = d->a_batch; g.b_batch = d->b_batch;
  long chunks = kStatBlocks / d->B;
  if (chunks < 1) chunks = 1;
  long len = (d->S + chunks - 1) / chunks;
  if (len < 64) len = 64;
  g.chunk_len = len;
  g.chunks = (int)((d->S + len - 1) / len);
  return g;
}

// backward finalize of one side on the [B * chunks] partial records: a batch norm folds the sample axis into the chunk axis
GnGeom gn2_bwd_side_geom(const Gn2Geom& g2, const ssbev_norm2_dims* d, int side) {
  const bool batch = side ? d->b_batch : d->a_batch;
  GnGeom g;
  g.B = batch ? 1 : g2.B; g.C = g2.C; g.G = side ? g2.Gb : g2.Ga; g.S = batch ? g2.S * g2.B : g2.S;
  g.chunks = batch ? g2.chunks * g2.B : g2.chunks; g.chunk_len = g2.chunk_len; g.eps = 0.f; g.relu = g2.relu; g.pre = 0;
  g.ldy = g.ldg = g.C; g.slab_q = 0;
  return g;
}

void gn2_finalize_bwd(const Gn2Geom& g2, const ssbev_norm2_dims* d, int side, const float* partial, const float* gamma, float* coef,
                      float* dgamma, float* dbeta, hipStream_t st) {
  const GnGeom g = gn2_bwd_side_geom(g2, d, side);
  if (g.G == g.C)
    hipLaunchKernelGGL(bn_finalize_bwd_flat_kernel, dim3(cdiv((size_t)g.C, 4)), dim3(256), 0, st, partial, gamma, coef, dgamma, dbeta, g);
  else
    hipLaunchKernelGGL(gn_finalize_bwd_kernel, dim3(g.B * g.G + g.C), dim3(FT), 0, st, partial, gamma, coef, dgamma, dbeta, g);
}

}  // namespace

extern "C" {

size_t ssbev_groupnorm_workspace(const ssbev_norm_dims* d) {
  if (!gn_ok(d)) return 0;
  const GnGeom g = make_geom(d);
  return ((size_t)g.B * g.chunks * g.C * 2 + (size_t)g.B * g.G * 2 + 64) * sizeof(float);
}

size_t ssbev_groupnorm_mask_words(const ssbev_norm_dims* d) {
  if (!gn_ok(d)) return 0;
  const size_t total4 = (size_t)d->B * d->S * (d->C / 4);      // one bit per element, four words per 64 channel quads
  return ((total4 + 63) / 64) * 4;
}

extern "C++" {
// statistics of one normalisation: chunk partials + finalize
template <typename T, int VW>
static int gn_stats_fwd(const T* x, float* mean, float* rstd, const GnGeom& g, float* partial, const ssbev_norm_ext* ext,
                        hipStream_t st) {
  const size_t lds = lds_bytes(g, VW);
  if (lds > 96 * 1024) return SSBEV_EINVAL;
  const dim3 grid(g.chunks, g.B, gn_slabs(g, VW));
  if (g.pre)
    hipLaunchKernelGGL((gn_partial_kernel<0, true, 0, T, VW>), grid, dim3(PNT), lds, st, x, (const T*)nullptr,
                       (const T*)nullptr, nullptr, nullptr, nullptr, partial, g);
  else
    hipLaunchKernelGGL((gn_partial_kernel<0, false, 0, T, VW>), grid, dim3(PNT), lds, st, x, (const T*)nullptr,
                       (const T*)nullptr, nullptr, nullptr, nullptr, partial, g);
  if (g.G == g.C) {
    const bool run = ext && ext->running_mean && ext->running_var;       // running statistics: updated by the finalize kernel
    hipLaunchKernelGGL(bn_finalize_fwd_flat_kernel<T>, dim3(cdiv((size_t)g.B * g.C, 4)), dim3(256), 0, st, partial, x, mean, rstd, g,
                       run ? ext->running_mean : nullptr, run ? ext->running_var : nullptr, run ? ext->momentum : 0.0f,
                       run ? (float)((double)ext->n / (double)(ext->n > 1 ? ext->n - 1 : 1)) : 0.0f);
  } else {
    hipLaunchKernelGGL(gn_finalize_fwd_kernel<T>, dim3(g.B * g.G), dim3(FT), 0, st, partial, x, mean, rstd, g);
  }
  return SSBEV_OK;
}

template <typename T, int VW>
static int groupnorm_fwd_t(const T* x, const float* gamma, const float* beta, const T* residual, T* y,
                           float* mean, float* rstd, unsigned long long* mask, const ssbev_norm_dims* d, void* ws,
                           ssbev_stream_t stream, const ssbev_norm_ext* ext = nullptr) {
  const GnGeom g = make_geom(d);
  hipStream_t st = as_stream(stream);
  float* partial = static_cast<float*>(ws);
  if (!d->stats_given) {
    const int rc = gn_stats_fwd<T, VW>(x, mean, rstd, g, partial, ext, st);
    if (rc != SSBEV_OK) return rc;
  }
  const long totalv = (long)g.B * g.S * (g.C / 4);
  const unsigned blocks = apply_blocks(totalv, g.C / 4);
  // (measured and removed in round 6: two lane-vectors in flight per thread -- fp32 66.7 -> 76.8 us, bf16 96.2 -> 103.4 us on the
  // 189 MB activation -- and 16-byte bf16 lanes with fused constants -- 73.6 vs 72.5 ms per two-sample step)
  if (g.pre)
    hipLaunchKernelGGL((gn_apply_fwd_kernel<true, T, 4>), dim3(blocks), dim3(NT), 0, st, x, gamma, beta, residual, mean, rstd, y,
                       d->relu ? mask : nullptr, g, totalv);
  else
    hipLaunchKernelGGL((gn_apply_fwd_kernel<false, T, 4>), dim3(blocks), dim3(NT), 0, st, x, gamma, beta, residual, mean, rstd, y,
                       d->relu ? mask : nullptr, g, totalv);
  return ssbev_launch_status();
}
}  // extern "C++"

// 16-byte lanes of bf16 tensors need 16-byte aligned rows
static bool gn_rows16(const ssbev_norm_dims* d, const void* const* ptrs, int n) {
  if ((d->ld_y != 0 && d->ld_y % 8 != 0) || (d->ld_gy != 0 && d->ld_gy % 8 != 0)) return false;
  for (int i = 0; i < n; ++i)
    if (ptrs[i] && reinterpret_cast<uintptr_t>(ptrs[i]) % 16 != 0) return false;
  return true;
}

static int groupnorm_fwd_impl(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                              float* mean, float* rstd, unsigned long long* mask, const ssbev_norm_dims* d, void* ws,
                              size_t ws_bytes, ssbev_stream_t stream, const ssbev_norm_ext* ext = nullptr) {
  if (!gn_ok(d) || !x || !gamma || !beta || !y || !mean || !rstd || !ws) return SSBEV_EINVAL;
  // a strided output (ld_y) is a slice of a concatenation: the residual operand has no stride of its own -> refused together
  if (residual && d->ld_y != 0 && d->ld_y != d->C) return SSBEV_EINVAL;
  if (ws_bytes < ssbev_groupnorm_workspace(d)) return SSBEV_EWORKSPACE;
  if (d->io_dtype == 1) {   // x, residual, y hold bf16 bit patterns (statistics, affine parameters and arithmetic stay fp32)
    const bf16_t* x16 = reinterpret_cast<const bf16_t*>(x);
    const bf16_t* r16 = reinterpret_cast<const bf16_t*>(residual);
    bf16_t* y16 = reinterpret_cast<bf16_t*>(y);
    if (gn_vw(1, d->C) == 8) {
      const void* ps[1] = {x};
      if (!gn_rows16(d, ps, 1)) return groupnorm_fwd_t<bf16_t, 4>(x16, gamma, beta, r16, y16, mean, rstd, mask, d, ws, stream, ext);
      return groupnorm_fwd_t<bf16_t, 8>(x16, gamma, beta, r16, y16, mean, rstd, mask, d, ws, stream, ext);
    }
    return groupnorm_fwd_t<bf16_t, 4>(x16, gamma, beta, r16, y16, mean, rstd, mask, d, ws, stream, ext);
  }
  return groupnorm_fwd_t<float, 4>(x, gamma, beta, residual, y, mean, rstd, mask, d, ws, stream, ext);
}

int ssbev_groupnorm_fwd_ext(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                            float* mean, float* rstd, uint64_t* relu_mask, const ssbev_norm_dims* d,
                            const ssbev_norm_ext* ext, void* ws, size_t ws_bytes, ssbev_stream_t stream) {
  return groupnorm_fwd_impl(x, gamma, beta, residual, y, mean, rstd, reinterpret_cast<unsigned long long*>(relu_mask), d, ws,
                            ws_bytes, stream, ext);
}

int ssbev_groupnorm_fwd(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                        float* mean, float* rstd, const ssbev_norm_dims* d, void* ws, size_t ws_bytes,
                        ssbev_stream_t stream) {
  return groupnorm_fwd_impl(x, gamma, beta, residual, y, mean, rstd, nullptr, d, ws, ws_bytes, stream);
}

int ssbev_groupnorm_fwd_mask(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                             float* mean, float* rstd, uint64_t* relu_mask, const ssbev_norm_dims* d, void* ws,
                             size_t ws_bytes, ssbev_stream_t stream) {
  if (d && d->relu && !relu_mask) return SSBEV_EINVAL;
  return groupnorm_fwd_impl(x, gamma, beta, residual, y, mean, rstd, reinterpret_cast<unsigned long long*>(relu_mask), d, ws,
                            ws_bytes, stream);
}

int ssbev_bn_update_running(const float* mean, const float* rstd, float* running_mean, float* running_var, int C,
                            float momentum, float eps, int64_t n, ssbev_stream_t stream) {
  if (!mean || !rstd || !running_mean || !running_var || C <= 0 || n <= 0) return SSBEV_EINVAL;
  hipLaunchKernelGGL(bn_update_running_kernel, dim3(cdiv((size_t)C, 256)), dim3(256), 0, as_stream(stream), mean, rstd,
                     running_mean, running_var, C, momentum, eps, (float)((double)n / (double)(n > 1 ? n - 1 : 1)));
  return ssbev_launch_status();
}

extern "C++" {
template <typename T, int VW>
static int groupnorm_bwd_t(const T* gy, const T* x, const T* y, const unsigned long long* mask,
                           const float* gamma, const float* mean, const float* rstd, T* gx, T* gresidual,
                           float* ggamma, float* gbeta, const ssbev_norm_dims* d, void* ws, ssbev_stream_t stream,
                           const ssbev_norm_ext* ext = nullptr) {
  const GnGeom g = make_geom(d);
  hipStream_t st = as_stream(stream);
  float* partial = static_cast<float*>(ws);
  float* coef = partial + (size_t)g.B * g.chunks * g.C * 2;
  const size_t lds = lds_bytes(g, VW);
  if (lds > 96 * 1024) return SSBEV_EINVAL;
  {
    const dim3 grid(g.chunks, g.B, gn_slabs(g, VW));
    const int rm = !g.relu ? 0 : (mask ? 1 : 2);
#define SSBEV_GNP(PRE_, RM_) \
    hipLaunchKernelGGL((gn_partial_kernel<1, PRE_, RM_, T, VW>), grid, dim3(PNT), lds, st, x, gy, y, mask, mean, rstd, partial, g)
    if (g.pre) { if (rm == 0) SSBEV_GNP(true, 0); else if (rm == 1) SSBEV_GNP(true, 1); else SSBEV_GNP(true, 2); }
    else { if (rm == 0) SSBEV_GNP(false, 0); else if (rm == 1) SSBEV_GNP(false, 1); else SSBEV_GNP(false, 2); }
#undef SSBEV_GNP
  }
  if (g.G == g.C)
    hipLaunchKernelGGL(bn_finalize_bwd_flat_kernel, dim3(cdiv((size_t)g.C, 4)), dim3(256), 0, st, partial, gamma, coef, ggamma, gbeta, g);
  else
    hipLaunchKernelGGL(gn_finalize_bwd_kernel, dim3(g.B * g.G + g.C), dim3(FT), 0, st, partial, gamma, coef, ggamma, gbeta, g);
  const long totalv = (long)g.B * g.S * (g.C / 4);
  const unsigned blocks = apply_blocks(totalv, g.C / 4);
  if (g.pre)
    hipLaunchKernelGGL((gn_apply_bwd_kernel<true, T, 4>), dim3(blocks), dim3(NT), 0, st, gy, x, y, gamma, mean, rstd, coef, gx,
                       gresidual, mask, g, totalv);
  else
    hipLaunchKernelGGL((gn_apply_bwd_kernel<false, T, 4>), dim3(blocks), dim3(NT), 0, st, gy, x, y, gamma, mean, rstd, coef, gx,
                       gresidual, mask, g, totalv);
  return ssbev_launch_status();
}
}  // extern "C++"

static int groupnorm_bwd_impl(const float* gy, const float* x, const float* y, const unsigned long long* mask,
                              const float* gamma, const float* mean, const float* rstd, float* gx, float* gresidual,
                              float* ggamma, float* gbeta, const ssbev_norm_dims* d, void* ws, size_t ws_bytes,
                              ssbev_stream_t stream, const ssbev_norm_ext* ext = nullptr) {
  if (!gn_ok(d) || !gy || !x || !gamma || !mean || !rstd || !gx || !ggamma || !gbeta || !ws) return SSBEV_EINVAL;
  if (d->relu && !y && !mask) return SSBEV_EINVAL;
  // row strides are honoured for gy (ld_gy) on the paths that exist for them: the saved y of the non-mask ReLU path and the
  // residual gradient are dense tensors -> a strided call that would need them is refused instead of mis-read
  const bool strided = (d->ld_gy != 0 && d->ld_gy != d->C) || (d->ld_y != 0 && d->ld_y != d->C);
  if (strided && ((d->relu && !mask) || gresidual)) return SSBEV_EINVAL;
  if (ws_bytes < ssbev_groupnorm_workspace(d)) return SSBEV_EWORKSPACE;
  if (d->io_dtype == 1) {
    typedef const bf16_t* cb;
    if (gn_vw(1, d->C) == 8) {
      const void* ps[3] = {gy, x, y};
      if (gn_rows16(d, ps, 3))
        return groupnorm_bwd_t<bf16_t, 8>(reinterpret_cast<cb>(gy), reinterpret_cast<cb>(x), reinterpret_cast<cb>(y), mask, gamma, mean,
                                        rstd, reinterpret_cast<bf16_t*>(gx), reinterpret_cast<bf16_t*>(gresidual), ggamma, gbeta, d, ws,
                                        stream, ext);
    }
    return groupnorm_bwd_t<bf16_t, 4>(reinterpret_cast<cb>(gy), reinterpret_cast<cb>(x), reinterpret_cast<cb>(y), mask, gamma, mean, rstd,
                                      reinterpret_cast<bf16_t*>(gx), reinterpret_cast<bf16_t*>(gresidual), ggamma, gbeta, d, ws, stream, ext);
  }
  return groupnorm_bwd_t<float, 4>(gy, x, y, mask, gamma, mean, rstd, gx, gresidual, ggamma, gbeta, d, ws, stream, ext);
}

int ssbev_groupnorm_bwd_ext(const float* gy, const float* x, const float* y, const uint64_t* relu_mask, const float* gamma,
                            const float* mean, const float* rstd, float* gx, float* gresidual, float* ggamma, float* gbeta,
                            const ssbev_norm_dims* d, const ssbev_norm_ext* ext, void* ws, size_t ws_bytes,
                            ssbev_stream_t stream) {
  return groupnorm_bwd_impl(gy, x, y, reinterpret_cast<const unsigned long long*>(relu_mask), gamma, mean, rstd, gx, gresidual,
                            ggamma, gbeta, d, ws, ws_bytes, stream, ext);
}

int ssbev_groupnorm_bwd(const float* gy, const float* x, const float* y, const float* gamma, const float* mean,
                        const float* rstd, float* gx, float* gresidual, float* ggamma, float* gbeta,
                        const ssbev_norm_dims* d, void* ws, size_t ws_bytes, ssbev_stream_t stream) {
  return groupnorm_bwd_impl(gy, x, y, nullptr, gamma, mean, rstd, gx, gresidual, ggamma, gbeta, d, ws, ws_bytes, stream);
}

int ssbev_groupnorm_bwd_mask(const float* gy, const float* x, const uint64_t* relu_mask, const float* gamma,
                             const float* mean, const float* rstd, float* gx, float* gresidual, float* ggamma, float* gbeta,
                             const ssbev_norm_dims* d, void* ws, size_t ws_bytes, ssbev_stream_t stream) {
  return groupnorm_bwd_impl(gy, x, nullptr, reinterpret_cast<const unsigned long long*>(relu_mask), gamma, mean, rstd, gx,
                            gresidual, ggamma, gbeta, d, ws, ws_bytes, stream);
}


size_t ssbev_groupnorm2_workspace(const ssbev_norm2_dims* d) {
  if (!gn2_ok(d)) return 0;
  const ssbev_norm_dims na = gn2_side(d, 0), nb = gn2_side(d, 1);
  const Gn2Geom g = make_geom2(d);
  // forward: the two single-norm statistics passes (their own workspaces); backward: two partial buffers + two coefficient vectors
  const size_t fwd = ssbev_groupnorm_workspace(&na) + ssbev_groupnorm_workspace(&nb);
  const size_t part = (size_t)g.B * g.chunks * g.C * 2;
  const size_t bwd = (2 * part + 2 * (size_t)(g.B * g.C) * 2 + 128) * sizeof(float);
  return fwd > bwd ? fwd : bwd;
}

extern "C++" {
template <typename T, int VW>
static int groupnorm2_fwd_t(const T* xa, const float* gamma_a, const float* beta_a, float* mean_a, float* rstd_a,
                            const T* xb, const float* gamma_b, const float* beta_b, float* mean_b, float* rstd_b, T* y,
                            uint64_t* relu_mask, const ssbev_norm2_dims* d, void* ws, ssbev_stream_t stream,
                            const ssbev_norm2_ext* ext = nullptr) {
  hipStream_t st = as_stream(stream);
  const T* xs[2] = {xa, xb};
  float* means[2] = {mean_a, mean_b};
  float* rstds[2] = {rstd_a, rstd_b};
  char* wsp = static_cast<char*>(ws);
  for (int side = 0; side < 2; ++side) {
    const ssbev_norm_dims n = gn2_side(d, side);
    const GnGeom g = make_geom(&n);
    float* partial = reinterpret_cast<float*>(wsp);
    ssbev_norm_ext e1 = {};
    if (ext) {
      e1.running_mean = side ? ext->running_mean_b : ext->running_mean_a;
      e1.running_var = side ? ext->running_var_b : ext->running_var_a;
      e1.momentum = side ? ext->momentum_b : ext->momentum_a;
      e1.n = (int64_t)d->B * d->S;
    }
    const int rc = gn_stats_fwd<T, VW>(xs[side], means[side], rstds[side], g, partial, ext ? &e1 : nullptr, st);
    if (rc != SSBEV_OK) return rc;
    wsp += ssbev_groupnorm_workspace(&n);
  }
  const Gn2Geom g2 = make_geom2(d);
  const long totalv = (long)g2.B * g2.S * (g2.C / 4);
  hipLaunchKernelGGL((gn2_apply_fwd_kernel<T, 4>), dim3(apply_blocks(totalv, g2.C / 4)), dim3(NT), 0, st, xa, gamma_a, beta_a, mean_a,
                     rstd_a, xb, gamma_b, beta_b, mean_b, rstd_b, y, d->relu ? reinterpret_cast<unsigned long long*>(relu_mask) : nullptr,
                     g2, totalv);
  return ssbev_launch_status();
}
}  // extern "C++"

static int groupnorm2_fwd_impl(const float* xa, const float* gamma_a, const float* beta_a, float* mean_a, float* rstd_a,
                               const float* xb, const float* gamma_b, const float* beta_b, float* mean_b, float* rstd_b, float* y,
                               uint64_t* relu_mask, const ssbev_norm2_dims* d, void* ws, size_t ws_bytes, ssbev_stream_t stream,
                               const ssbev_norm2_ext* ext) {
  if (!gn2_ok(d) || !xa || !gamma_a || !beta_a || !mean_a || !rstd_a || !xb || !gamma_b || !beta_b || !mean_b || !rstd_b || !y || !ws)
    return SSBEV_EINVAL;
  if (d->relu && !relu_mask) return SSBEV_EINVAL;
  if (ws_bytes < ssbev_groupnorm2_workspace(d)) return SSBEV_EWORKSPACE;
  if (d->io_dtype == 1) {
    typedef const bf16_t* cb;
    const bool al = reinterpret_cast<uintptr_t>(xa) % 16 == 0 && reinterpret_cast<uintptr_t>(xb) % 16 == 0 &&
                    reinterpret_cast<uintptr_t>(y) % 16 == 0;
    if (gn_vw(1, d->C) == 8 && al) {
      return groupnorm2_fwd_t<bf16_t, 8>(reinterpret_cast<cb>(xa), gamma_a, beta_a, mean_a, rstd_a, reinterpret_cast<cb>(xb), gamma_b,
                                         beta_b, mean_b, rstd_b, reinterpret_cast<bf16_t*>(y), relu_mask, d, ws, stream, ext);
    }
    return groupnorm2_fwd_t<bf16_t, 4>(reinterpret_cast<cb>(xa), gamma_a, beta_a, mean_a, rstd_a, reinterpret_cast<cb>(xb), gamma_b,
                                       beta_b, mean_b, rstd_b, reinterpret_cast<bf16_t*>(y), relu_mask, d, ws, stream, ext);
  }
  return groupnorm2_fwd_t<float, 4>(xa, gamma_a, beta_a, mean_a, rstd_a, xb, gamma_b, beta_b, mean_b, rstd_b, y, relu_mask, d, ws, stream, ext);
}

int ssbev_groupnorm2_fwd(const float* xa, const float* gamma_a, const float* beta_a, float* mean_a, float* rstd_a,
                         const float* xb, const float* gamma_b, const float* beta_b, float* mean_b, float* rstd_b, float* y,
                         uint64_t* relu_mask, const ssbev_norm2_dims* d, void* ws, size_t ws_bytes, ssbev_stream_t stream) {
  return groupnorm2_fwd_impl(xa, gamma_a, beta_a, mean_a, rstd_a, xb, gamma_b, beta_b, mean_b, rstd_b, y, relu_mask, d, ws, ws_bytes,
                             stream, nullptr);
}

int ssbev_groupnorm2_fwd_ext(const float* xa, const float* gamma_a, const float* beta_a, float* mean_a, float* rstd_a,
                             const float* xb, const float* gamma_b, const float* beta_b, float* mean_b, float* rstd_b, float* y,
                             uint64_t* relu_mask, const ssbev_norm2_dims* d, const ssbev_norm2_ext* ext, void* ws, size_t ws_bytes,
                             ssbev_stream_t stream) {
  return groupnorm2_fwd_impl(xa, gamma_a, beta_a, mean_a, rstd_a, xb, gamma_b, beta_b, mean_b, rstd_b, y, relu_mask, d, ws, ws_bytes,
                             stream, ext);
}

extern "C++" {
template <typename T, int VW>
static int groupnorm2_bwd_t(const T* gy, const uint64_t* relu_mask, const T* xa, const float* gamma_a, const float* mean_a,
                            const float* rstd_a, const T* xb, const float* gamma_b, const float* mean_b, const float* rstd_b,
                            T* gxa, T* gxb, float* ggamma_a, float* gbeta_a, float* ggamma_b, float* gbeta_b,
                            const ssbev_norm2_dims* d, void* ws, ssbev_stream_t stream, const ssbev_norm2_ext* ext = nullptr) {
  hipStream_t st = as_stream(stream);
  const Gn2Geom g = make_geom2(d);
  const size_t part = (size_t)g.B * g.chunks * g.C * 2;
  float* pa = static_cast<float*>(ws);
  float* pb = pa + part;
  float* coef_a = pb + part;
  float* coef_b = coef_a + (size_t)g.B * g.C * 2 + 64;
  const int q = g.C / VW, rows = PNT / q > 0 ? PNT / q : 1;
  size_t lds = (size_t)rows * g.C * 3 * sizeof(float);
  const unsigned long long* mk = reinterpret_cast<const unsigned long long*>(relu_mask);
  if (lds > 48 * 1024) {                    // above the default dynamic-LDS limit of a launch: raise it for this instance (once)
    static bool raised = false;
    if (!raised) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(gn2_partial_bwd_kernel<T, VW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              96 * 1024) != hipSuccess)
        return SSBEV_ELAUNCH;
      raised = true;
    }
  }
  hipLaunchKernelGGL((gn2_partial_bwd_kernel<T, VW>), dim3(g.chunks, g.B), dim3(PNT), lds, st, gy, mk, xa, mean_a, rstd_a, xb, mean_b,
                     rstd_b, pa, pb, g);
  gn2_finalize_bwd(g, d, 0, pa, gamma_a, coef_a, ggamma_a, gbeta_a, st);
  gn2_finalize_bwd(g, d, 1, pb, gamma_b, coef_b, ggamma_b, gbeta_b, st);
  const long totalv = (long)g.B * g.S * (g.C / 4);
  hipLaunchKernelGGL((gn2_apply_bwd_kernel<T, 4>), dim3(apply_blocks(totalv, g.C / 4)), dim3(NT), 0, st, gy, mk, xa, gamma_a, mean_a,
                     rstd_a, coef_a, xb, gamma_b, mean_b, rstd_b, coef_b, gxa, gxb, g, totalv);
  return ssbev_launch_status();
}
}  // extern "C++"

static int groupnorm2_bwd_impl(const float* gy, const uint64_t* relu_mask, const float* xa, const float* gamma_a, const float* mean_a,
                               const float* rstd_a, const float* xb, const float* gamma_b, const float* mean_b, const float* rstd_b,
                               float* gxa, float* gxb, float* ggamma_a, float* gbeta_a, float* ggamma_b, float* gbeta_b,
                               const ssbev_norm2_dims* d, void* ws, size_t ws_bytes, ssbev_stream_t stream,
                               const ssbev_norm2_ext* ext) {
  if (!gn2_ok(d) || !gy || !xa || !gamma_a || !mean_a || !rstd_a || !xb || !gamma_b || !mean_b || !rstd_b || !gxa || !gxb ||
      !ggamma_a || !gbeta_a || !ggamma_b || !gbeta_b || !ws)
    return SSBEV_EINVAL;
  if (d->relu && !relu_mask) return SSBEV_EINVAL;
  if (ws_bytes < ssbev_groupnorm2_workspace(d)) return SSBEV_EWORKSPACE;
  if (d->io_dtype == 1) {
    typedef const bf16_t* cb;
    bool al = true;
    for (const void* q : {(const void*)gy, (const void*)xa, (const void*)xb, (const void*)gxa, (const void*)gxb})
      al = al && reinterpret_cast<uintptr_t>(q) % 16 == 0;
    if (gn_vw(1, d->C) == 8 && al) {
      return groupnorm2_bwd_t<bf16_t, 8>(reinterpret_cast<cb>(gy), relu_mask, reinterpret_cast<cb>(xa), gamma_a, mean_a, rstd_a,
                                         reinterpret_cast<cb>(xb), gamma_b, mean_b, rstd_b, reinterpret_cast<bf16_t*>(gxa),
                                         reinterpret_cast<bf16_t*>(gxb), ggamma_a, gbeta_a, ggamma_b, gbeta_b, d, ws, stream, ext);
    }
    return groupnorm2_bwd_t<bf16_t, 4>(reinterpret_cast<cb>(gy), relu_mask, reinterpret_cast<cb>(xa), gamma_a, mean_a, rstd_a,
                                       reinterpret_cast<cb>(xb), gamma_b, mean_b, rstd_b, reinterpret_cast<bf16_t*>(gxa),
                                       reinterpret_cast<bf16_t*>(gxb), ggamma_a, gbeta_a, ggamma_b, gbeta_b, d, ws, stream, ext);
  }
  return groupnorm2_bwd_t<float, 4>(gy, relu_mask, xa, gamma_a, mean_a, rstd_a, xb, gamma_b, mean_b, rstd_b, gxa, gxb, ggamma_a,
                                    gbeta_a, ggamma_b, gbeta_b, d, ws, stream, ext);
}

int ssbev_groupnorm2_bwd(const float* gy, const uint64_t* relu_mask, const float* xa, const float* gamma_a, const float* mean_a,
                         const float* rstd_a, const float* xb, const float* gamma_b, const float* mean_b, const float* rstd_b,
                         float* gxa, float* gxb, float* ggamma_a, float* gbeta_a, float* ggamma_b, float* gbeta_b,
                         const ssbev_norm2_dims* d, void* ws, size_t ws_bytes, ssbev_stream_t stream) {
  return groupnorm2_bwd_impl(gy, relu_mask, xa, gamma_a, mean_a, rstd_a, xb, gamma_b, mean_b, rstd_b, gxa, gxb, ggamma_a, gbeta_a,
                             ggamma_b, gbeta_b, d, ws, ws_bytes, stream, nullptr);
}

int ssbev_groupnorm2_bwd_ext(const float* gy, const uint64_t* relu_mask, const float* xa, const float* gamma_a, const float* mean_a,
                             const float* rstd_a, const float* xb, const float* gamma_b, const float* mean_b, const float* rstd_b,
                             float* gxa, float* gxb, float* ggamma_a, float* gbeta_a, float* ggamma_b, float* gbeta_b,
                             const ssbev_norm2_dims* d, const ssbev_norm2_ext* ext, void* ws, size_t ws_bytes,
                             ssbev_stream_t stream) {
  return groupnorm2_bwd_impl(gy, relu_mask, xa, gamma_a, mean_a, rstd_a, xb, gamma_b, mean_b, rstd_b, gxa, gxb, ggamma_a, gbeta_a,
                             ggamma_b, gbeta_b, d, ws, ws_bytes, stream, ext);
}

}  // extern "C"
