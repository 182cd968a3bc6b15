// BRI cross attention (ATT:63-81) as a flash-style kernel family on the fp32 matrix cores:
//
//     out[:, i] = sum_j softmax_j(q[:, i] . k[:, j]) * conf[j] * v[:, j]
//
// The reference materialises the T x T energy and attention matrices (T = fH*fW = 7680: 236 MB each,
// per direction, plus their gradients).  Here they only ever exist as 32x32 register tiles.
//
// Layout: q, k, v, out, gradients are [B, Dh, T] with the TOKEN axis contiguous -- exactly the memory of
// the reference's [B,1,D,H,W] volumes, so nothing is transposed on the way in or out.
//
// MFMA orientation (v_mfma_f32_32x32x2_f32, C[row][col], a lane owns one column):
//   forward / dQ kernels: tiles are S^T[key j][query i]  -> a lane owns ONE query: running max / sum /
//                         rescale are per-lane scalars, and P^T is directly the B operand of the
//                         P.V product (no cross-lane movement, no LDS round trip for P);
//   dK/dV kernel:         tiles are S[query i][key j]    -> a lane owns ONE key, P / dS feed the B
//                         operand of the two accumulations into dK^T, dV^T.
// Operands whose MFMA row index is the head dimension (V in forward, K in dQ, Q and dO in dK/dV) are
// staged per 32-token tile in LDS with a +1 padded stride (conflict-free column reads).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int TS = 33;   // LDS row stride of a [rows][32] tile

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int crow(int r, int lk) { return (r & 3) + 8 * (r >> 2) + 4 * lk; }

// Stage src[d][t0 .. t0+31] (d < DH) * scale[t0 + col] into tile[d][col]; rows DH..DHP-1 are zero.
template <int DH, int DHP>
__device__ __forceinline__ void stage_tile(const float* __restrict__ src, int T, int t0, const float* scale, float* tile,
                                           int lane) {
  const int li = lane & 31, lk = lane >> 5;
  const float sc = scale ? scale[t0 + li] : 1.0f;
#pragma unroll 4
  for (int d = lk; d < DHP; d += 2) tile[d * TS + li] = d < DH ? src[(size_t)d * T + t0 + li] * sc : 0.0f;
}

// ------------------------------------------------------------------------------------------ forward
// block = 4 waves, 32 queries; wave w walks key tiles w, w+4, ...; partial (m, l, O^T) merged through LDS.
template <int DH>
__global__ void __launch_bounds__(256)
bri_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
               const float* __restrict__ conf, float* __restrict__ out, float* __restrict__ lse, int T) {
  constexpr int NDT = (DH + 31) / 32, DHP = NDT * 32, DH2 = DH / 2;
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const int b = blockIdx.y, i0 = blockIdx.x * 32;
  const float* qb = q + (size_t)b * DH * T;
  const float* kb = k + (size_t)b * DH * T;
  const float* vb = v + (size_t)b * DH * T;
  const float* cb = conf + (size_t)b * T;
  float* qs = lds;                                               // [DH][32] Q of the block, shared by the waves
  float* vt = lds + DH * 32 + (size_t)wave * (DHP * TS + 64);    // this wave's V' tile (later: partial O^T, m, l)
  for (int e = threadIdx.x; e < DH * 32; e += 256) qs[e] = qb[(size_t)(e >> 5) * T + i0 + (e & 31)];

  f32x16 o[NDT];
#pragma unroll
  for (int t = 0; t < NDT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.0f;
  float m = -1e30f, l = 0.0f;

  const int ntiles = T / 32;
  for (int jt = wave; jt < ntiles; jt += 4) {
    const int j0 = jt * 32;
    __syncthreads();                                      // previous tile's LDS reads are done
    stage_tile<DH, DHP>(vb, T, j0, cb, vt, lane);
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll 8
    for (int st = 0; st < DH2; ++st)
      s = mfma32(kb[(size_t)(2 * st + lk) * T + j0 + li], qs[(2 * st + lk) * 32 + li], s);
    // online softmax over this lane's query column (its 16 rows + the partner half-wave's 16)
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx);
    const float alpha = __expf(m - mn);
    float ps = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = __expf(s[r] - mn); ps += s[r]; }
    ps += __shfl_xor(ps, 32, 64);
    l = l * alpha + ps;
    m = mn;
#pragma unroll
    for (int t = 0; t < NDT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
    __syncthreads();                                      // V' tile visible
#pragma unroll
    for (int t = 0; t < NDT; ++t)
#pragma unroll
      for (int st = 0; st < 16; ++st) o[t] = mfma32(vt[(t * 32 + li) * TS + crow(st, lk)], s[st], o[t]);
  }
  // the loop trip counts differ by at most one between waves: align the barrier count
  if (((ntiles - wave + 3) / 4) < ((ntiles + 3) / 4)) { __syncthreads(); __syncthreads(); }
  __syncthreads();
  // ---- merge the four partial results -----------------------------------------------------------
#pragma unroll
  for (int t = 0; t < NDT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) vt[(t * 32 + crow(r, lk)) * TS + li] = o[t][r];
  if (lk == 0) { vt[DHP * TS + li] = m; vt[DHP * TS + 32 + li] = l; }
  __syncthreads();
  const int stride = DHP * TS + 64;
  const float* parts = lds + DH * 32;
  for (int e = threadIdx.x; e < DH * 32; e += 256) {
    const int d = e >> 5, i = e & 31;
    float M = -1e30f;
#pragma unroll
    for (int w = 0; w < 4; ++w) M = fmaxf(M, parts[w * stride + DHP * TS + i]);
    float O = 0.0f, L = 0.0f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = __expf(parts[w * stride + DHP * TS + i] - M);
      O += parts[w * stride + d * TS + i] * f;
      L += parts[w * stride + DHP * TS + 32 + i] * f;
    }
    out[((size_t)b * DH + d) * T + i0 + i] = O / L;
    if (d == 0) lse[(size_t)b * T + i0 + i] = M + __logf(L);
  }
}

// ------------------------------------------------------------------------------------------ dQ
// block = 4 waves, 32 queries; Q and dO of the block live in LDS (shared by the waves); every wave stages
// its K tile.  dQ^T[d][i] += sum_j K[d][j] * dS^T[j][i],  dS^T = P^T * (dP^T - Dd[i]).
template <int DH>
__global__ void __launch_bounds__(256)
bri_bwd_dq_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                  const float* __restrict__ conf, const float* __restrict__ go, const float* __restrict__ lse,
                  const float* __restrict__ dd, float* __restrict__ gq, int T) {
  constexpr int NDT = (DH + 31) / 32, DHP = NDT * 32, DH2 = DH / 2;
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const int b = blockIdx.y, i0 = blockIdx.x * 32;
  const float* kb = k + (size_t)b * DH * T;
  const float* vb = v + (size_t)b * DH * T;
  const float* cb = conf + (size_t)b * T;
  float* qs = lds;                       // [DH][32]  Q  of the block (row stride 32: read along the row)
  float* gs = lds + DH * 32;             // [DH][32]  dO of the block
  float* kt = lds + 2 * DH * 32 + (size_t)wave * (DHP * TS);
  for (int e = threadIdx.x; e < DH * 32; e += 256) {
    const int d = e >> 5, i = e & 31;
    qs[e] = q[((size_t)b * DH + d) * T + i0 + i];
    gs[e] = go[((size_t)b * DH + d) * T + i0 + i];
  }
  const float lse_i = lse[(size_t)b * T + i0 + li];
  const float dd_i = dd[(size_t)b * T + i0 + li];
  f32x16 acc[NDT];
#pragma unroll
  for (int t = 0; t < NDT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  const int ntiles = T / 32;
  for (int jt = wave; jt < ntiles; jt += 4) {
    const int j0 = jt * 32;
    __syncthreads();
    stage_tile<DH, DHP>(kb, T, j0, nullptr, kt, lane);
    __syncthreads();
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
    const float cj = cb[j0 + li];
#pragma unroll 8
    for (int st = 0; st < DH2; ++st) {
      const int d = 2 * st + lk;
      s = mfma32(kt[d * TS + li], qs[d * 32 + li], s);
      dp = mfma32(vb[(size_t)d * T + j0 + li] * cj, gs[d * 32 + li], dp);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = __expf(s[r] - lse_i) * (dp[r] - dd_i);      // dS^T
#pragma unroll
    for (int t = 0; t < NDT; ++t)
#pragma unroll
      for (int st = 0; st < 16; ++st) acc[t] = mfma32(kt[(t * 32 + li) * TS + crow(st, lk)], s[st], acc[t]);
  }
  if (((ntiles - wave + 3) / 4) < ((ntiles + 3) / 4)) { __syncthreads(); __syncthreads(); }
  __syncthreads();
  // merge: every wave parks its partial dQ^T in its K-tile region, then all threads sum the four
#pragma unroll
  for (int t = 0; t < NDT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) kt[(t * 32 + crow(r, lk)) * TS + li] = acc[t][r];
  __syncthreads();
  const float* base = lds + 2 * DH * 32;
  for (int e = threadIdx.x; e < DH * 32; e += 256) {
    const int d = e >> 5, i = e & 31;
    float sum = 0.0f;
#pragma unroll
    for (int w = 0; w < 4; ++w) sum += base[(size_t)w * (DHP * TS) + d * TS + i];
    gq[((size_t)b * DH + d) * T + i0 + i] = sum;
  }
}

// ------------------------------------------------------------------------------------------ dK, dV, dconf
// block = 4 waves = 4 consecutive 32-key blocks; the Q and dO tiles of the current 32-query tile are staged
// ONCE per block in LDS and serve all four waves, both as the row operand of S / dP (rows = queries) and as
// the row operand of the two accumulations (rows = head dim).  K and V' of a wave's own keys come from
// L1/L2 as coalesced B-operand loads.  The query range is split over blockIdx.z; every split writes its
// partial dK / dV' / dconf slab, summed in split order by bri_dkv_finish_kernel (deterministic).
//   dV'^T[d][j] += sum_i dO[d][i] * P[i][j]        dK^T[d][j] += sum_i Q[d][i] * dS[i][j]
template <int DH>
__global__ void __launch_bounds__(256)
bri_bwd_dkv_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                   const float* __restrict__ conf, const float* __restrict__ go, const float* __restrict__ lse,
                   const float* __restrict__ dd, float* __restrict__ part, int T, int nsplit) {
  constexpr int NDT = (DH + 31) / 32, DHP = NDT * 32, DH2 = DH / 2;
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const int b = blockIdx.y, split = blockIdx.z;
  const int j0 = (blockIdx.x * 4 + wave) * 32;
  const bool jok = j0 < T;
  const float* qb = q + (size_t)b * DH * T;
  const float* gb = go + (size_t)b * DH * T;
  const float* kb = k + (size_t)b * DH * T;
  const float* vb = v + (size_t)b * DH * T;
  const float* lb = lse + (size_t)b * T;
  const float* db = dd + (size_t)b * T;
  float* qt = lds;                    // [DHP][TS]  Q  tile of the current queries
  float* gt = lds + DHP * TS;         // [DHP][TS]  dO tile
  float* lt = lds + 2 * DHP * TS;     // [64]       lse | Dd of the 32 queries
  const float cj = jok ? conf[(size_t)b * T + j0 + li] : 0.0f;
  const int jc = jok ? j0 + li : 0;
  f32x16 ak[NDT], av[NDT];
#pragma unroll
  for (int t = 0; t < NDT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { ak[t][r] = 0.0f; av[t][r] = 0.0f; }

  const int ntiles = T / 32;
  const int per = (ntiles + nsplit - 1) / nsplit;
  const int it_end = min(ntiles, (split + 1) * per);
  for (int it = split * per; it < it_end; ++it) {
    const int i0 = it * 32;
    __syncthreads();
    // cooperative staging: 4 waves x (rows d = wave*2 + lk, +8, ...)
    for (int d = wave * 2 + lk; d < DHP; d += 8) {
      qt[d * TS + li] = d < DH ? qb[(size_t)d * T + i0 + li] : 0.0f;
      gt[d * TS + li] = d < DH ? gb[(size_t)d * T + i0 + li] : 0.0f;
    }
    if (threadIdx.x < 32) lt[threadIdx.x] = lb[i0 + threadIdx.x];
    else if (threadIdx.x < 64) lt[threadIdx.x] = db[i0 + threadIdx.x - 32];
    __syncthreads();
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
#pragma unroll 8
    for (int st = 0; st < DH2; ++st) {
      const int d = 2 * st + lk;
      s = mfma32(qt[d * TS + li], kb[(size_t)d * T + jc], s);           // rows = queries, cols = keys
      dp = mfma32(gt[d * TS + li], vb[(size_t)d * T + jc] * cj, dp);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = crow(r, lk);
      const float p = __expf(s[r] - lt[i]);
      s[r] = p;                                   // P[i][j]
      dp[r] = p * (dp[r] - lt[32 + i]);           // dS[i][j]
    }
#pragma unroll
    for (int t = 0; t < NDT; ++t)
#pragma unroll
      for (int st = 0; st < 16; ++st) {
        av[t] = mfma32(gt[(t * 32 + li) * TS + crow(st, lk)], s[st], av[t]);
        ak[t] = mfma32(qt[(t * 32 + li) * TS + crow(st, lk)], dp[st], ak[t]);
      }
  }
  if (!jok) return;
  // partial slabs: part[split][b][{dK, dV'}][DH][T]
  float* pk = part + (((size_t)split * gridDim.y + b) * 2 + 0) * DH * T;
  float* pv = part + (((size_t)split * gridDim.y + b) * 2 + 1) * DH * T;
#pragma unroll
  for (int t = 0; t < NDT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d = t * 32 + crow(r, lk);
      if (d < DH) {
        pk[(size_t)d * T + j0 + li] = ak[t][r];
        pv[(size_t)d * T + j0 + li] = av[t][r];
      }
    }
}

// gk = sum_s dK_s ; gv = conf * sum_s dV'_s ; gconf[j] = sum_d v[d][j] * sum_s dV'_s[d][j]
__global__ void bri_dkv_finish_kernel(const float* __restrict__ part, const float* __restrict__ v,
                                      const float* __restrict__ conf, float* __restrict__ gk, float* __restrict__ gv,
                                      float* __restrict__ gconf, int Dh, int T, int nsplit, int B) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= T) return;
  const float cj = conf[(size_t)b * T + j];
  float gc = 0.0f;
  for (int d = 0; d < Dh; ++d) {
    float dk = 0.0f, dv = 0.0f;
    for (int s = 0; s < nsplit; ++s) {
      dk += part[((((size_t)s * B + b) * 2 + 0) * Dh + d) * T + j];
      dv += part[((((size_t)s * B + b) * 2 + 1) * Dh + d) * T + j];
    }
    gk[((size_t)b * Dh + d) * T + j] = dk;
    gv[((size_t)b * Dh + d) * T + j] = dv * cj;
    gc += dv * v[((size_t)b * Dh + d) * T + j];
  }
  gconf[(size_t)b * T + j] = gc;
}

// Dd[i] = sum_d dO[d][i] * O[d][i]
__global__ void bri_rowdot_kernel(const float* __restrict__ go, const float* __restrict__ o, float* __restrict__ dd,
                                  int Dh, int T) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T) return;
  float s = 0.0f;
  for (int d = 0; d < Dh; ++d) s += go[((size_t)b * Dh + d) * T + i] * o[((size_t)b * Dh + d) * T + i];
  dd[(size_t)b * T + i] = s;
}

template <int DH> size_t fwd_lds() { constexpr int DHP = ((DH + 31) / 32) * 32; return ((size_t)DH * 32 + 4 * (DHP * TS + 64)) * 4; }
template <int DH> size_t dq_lds() { constexpr int DHP = ((DH + 31) / 32) * 32; return ((size_t)2 * DH * 32 + 4 * DHP * TS) * 4; }
template <int DH> size_t dkv_lds() { constexpr int DHP = ((DH + 31) / 32) * 32; return ((size_t)2 * DHP * TS + 64) * 4; }
inline int dkv_splits(int T, int B) { int kb = (T / 32 + 3) / 4 * B; int s = (512 + kb - 1) / kb; return s < 1 ? 1 : (s > 16 ? 16 : s); }

template <typename K>
bool set_lds(K kern, size_t bytes) {
  if (bytes > 160 * 1024) return false;
  if (bytes <= 64 * 1024) return true;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)bytes) == hipSuccess;
}

template <int DH>
int run_fwd(const float* q, const float* k, const float* v, const float* conf, float* out, float* lse,
            const ssbev_attn_dims* d, hipStream_t st) {
  auto kern = bri_fwd_kernel<DH>;
  if (!set_lds(kern, fwd_lds<DH>())) return SSBEV_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3(d->T / 32, d->B), dim3(256), fwd_lds<DH>(), st, q, k, v, conf, out, lse, d->T);
  return ssbev_launch_status();
}

template <int DH>
int run_bwd(const float* q, const float* k, const float* v, const float* conf, const float* out, const float* lse,
            const float* gout, float* gq, float* gk, float* gv, float* gconf, const ssbev_attn_dims* d, float* dd,
            hipStream_t st) {
  hipLaunchKernelGGL(bri_rowdot_kernel, dim3(cdiv(d->T, 256), d->B), dim3(256), 0, st, gout, out, dd, DH, d->T);
  auto k1 = bri_bwd_dq_kernel<DH>;
  auto k2 = bri_bwd_dkv_kernel<DH>;
  if (!set_lds(k1, dq_lds<DH>()) || !set_lds(k2, dkv_lds<DH>())) return SSBEV_ELAUNCH;
  hipLaunchKernelGGL(k1, dim3(d->T / 32, d->B), dim3(256), dq_lds<DH>(), st, q, k, v, conf, gout, lse, dd, gq, d->T);
  const int ns = dkv_splits(d->T, d->B);
  float* part = dd + (size_t)d->B * d->T;
  hipLaunchKernelGGL(k2, dim3((d->T / 32 + 3) / 4, d->B, ns), dim3(256), dkv_lds<DH>(), st, q, k, v, conf, gout, lse, dd,
                     part, d->T, ns);
  hipLaunchKernelGGL(bri_dkv_finish_kernel, dim3(cdiv(d->T, 256), d->B), dim3(256), 0, st, part, v, conf, gk, gv, gconf,
                     DH, d->T, ns, d->B);
  return ssbev_launch_status();
}

bool attn_ok(const ssbev_attn_dims* d) {
  return d && d->B > 0 && d->T > 0 && d->T % 32 == 0 &&
         (d->Dh == 16 || d->Dh == 48 || d->Dh == 112 || d->Dh == 192);
}

}  // namespace

extern "C" {

int ssbev_bri_attention_supported(const ssbev_attn_dims* d) { return attn_ok(d) ? 1 : 0; }

size_t ssbev_bri_attention_workspace(const ssbev_attn_dims* d) {
  if (!attn_ok(d)) return 0;
  // Dd[B][T] + dK/dV' partial slabs of the query splits
  return ((size_t)d->B * d->T + (size_t)dkv_splits(d->T, d->B) * d->B * 2 * d->Dh * d->T) * sizeof(float);
}

int ssbev_bri_attention_fwd(const float* q, const float* k, const float* v, const float* conf, float* out, float* lse,
                            const ssbev_attn_dims* d, ssbev_stream_t stream) {
  if (!attn_ok(d) || !q || !k || !v || !conf || !out || !lse) return SSBEV_EINVAL;
  hipStream_t st = as_stream(stream);
  switch (d->Dh) {
    case 16: return run_fwd<16>(q, k, v, conf, out, lse, d, st);
    case 48: return run_fwd<48>(q, k, v, conf, out, lse, d, st);
    case 112: return run_fwd<112>(q, k, v, conf, out, lse, d, st);
    default: return run_fwd<192>(q, k, v, conf, out, lse, d, st);
  }
}

int ssbev_bri_attention_bwd(const float* q, const float* k, const float* v, const float* conf, const float* out,
                            const float* lse, const float* gout, float* gq, float* gk, float* gv, float* gconf,
                            const ssbev_attn_dims* d, void* ws, size_t ws_bytes, ssbev_stream_t stream) {
  if (!attn_ok(d) || !q || !k || !v || !conf || !out || !lse || !gout || !gq || !gk || !gv || !gconf || !ws)
    return SSBEV_EINVAL;
  if (ws_bytes < ssbev_bri_attention_workspace(d)) return SSBEV_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  float* dd = static_cast<float*>(ws);
  switch (d->Dh) {
    case 16: return run_bwd<16>(q, k, v, conf, out, lse, gout, gq, gk, gv, gconf, d, dd, st);
    case 48: return run_bwd<48>(q, k, v, conf, out, lse, gout, gq, gk, gv, gconf, d, dd, st);
    case 112: return run_bwd<112>(q, k, v, conf, out, lse, gout, gq, gk, gv, gconf, d, dd, st);
    default: return run_bwd<192>(q, k, v, conf, out, lse, gout, gq, gk, gv, gconf, d, dd, st);
  }
}

}  // extern "C"
