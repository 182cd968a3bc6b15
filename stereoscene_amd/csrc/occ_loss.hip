// Fused occupancy-head epilogue (SURVEY build-plan step 8), gfx950:
//   trilinear x2 upsample of the logits (align_corners=False) -> softmax over 20 classes ->
//   every reduction the three SemanticKITTI losses and the train-time metric need, in ONE pass,
//   without ever writing the 168 MB up-sampled logits / probabilities / one-hot volumes the reference
//   materialises (occhead.py:291-361; utils/semkitti.py:67-149).
//
// Forward kernel -> per-sample-batch sums S (doubles):
//   ce_num = sum_v w[t_v] * (-log p_v[t_v])      ce_den = sum_v w[t_v]           (v: label != 255)
//   sum_p[c] = sum_v p_v[c]     nom[c] = sum_v p_v[c] [t_v == c]     cnt[c] = #{t_v == c}     M = #valid
//   conf[t][a] = #{t_v == t, argmax_v == a}      (sc_iou / ssc_miou of occhead.py:345-359)
// The scalar losses are tiny functions of S (evaluated by the caller, e.g. under autograd); their gradient
// dL/dS comes back as coefficient vectors and the backward kernel turns them into dL/dlogits:
//   g_v[k] = a_ce * w[t] * (p_k - [k == t]) + p_k * (C_k - sum_c C_c p_c),   C_c = A[c] + B[c] [t == c]
// written once at the fine resolution (a scratch buffer of the backward pass only) and pulled back to the coarse
// grid by the gather-form x2 kernel of trilinear.hip: no atomics, deterministic.
#include "common.h"

namespace {

constexpr int NC = 20;                       // classes (asserted by the host wrapper)
constexpr int NS = 3 + 3 * NC + NC * NC;     // ce_num, ce_den, M, sum_p[NC], nom[NC], cnt[NC], conf[NC][NC]

__device__ __forceinline__ void src_taps2(int o, int in_size, int* i0, int* i1, float* l0, float* l1) {
  float s = 0.5f * ((float)o + 0.5f) - 0.5f;
  s = s < 0.0f ? 0.0f : s;
  const int a = (int)s;
  *i0 = a;
  *i1 = a + (a < in_size - 1 ? 1 : 0);
  *l1 = s - (float)a;
  *l0 = 1.0f - *l1;
}

// up-sampled logits of fine voxel (od, oh, ow) -> z[NC]
__device__ __forceinline__ void upsampled_logits(const float* __restrict__ x, int b, int D, int H, int W, int od, int oh,
                                                 int ow, float* z) {
  int d0, d1, h0, h1, w0, w1;
  float ld0, ld1, lh0, lh1, lw0, lw1;
  src_taps2(od, D, &d0, &d1, &ld0, &ld1);
  src_taps2(oh, H, &h0, &h1, &lh0, &lh1);
  src_taps2(ow, W, &w0, &w1, &lw0, &lw1);
#pragma unroll
  for (int c = 0; c < NC; ++c) z[c] = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int dd = (k & 4) ? d1 : d0, hh = (k & 2) ? h1 : h0, ww = (k & 1) ? w1 : w0;
    const float wt = ((k & 4) ? ld1 : ld0) * ((k & 2) ? lh1 : lh0) * ((k & 1) ? lw1 : lw0);
    const float4* src = reinterpret_cast<const float4*>(x + ((((size_t)b * D + dd) * H + hh) * W + ww) * NC);
#pragma unroll
    for (int q = 0; q < NC / 4; ++q) {
      const float4 v = src[q];
      z[4 * q + 0] += wt * v.x; z[4 * q + 1] += wt * v.y; z[4 * q + 2] += wt * v.z; z[4 * q + 3] += wt * v.w;
    }
  }
}

__device__ __forceinline__ float softmax_inplace(float* z, int* amax) {
  float m = z[0];
  int am = 0;
#pragma unroll
  for (int c = 1; c < NC; ++c) if (z[c] > m) { m = z[c]; am = c; }
  float s = 0.0f;
#pragma unroll
  for (int c = 0; c < NC; ++c) { z[c] = __expf(z[c] - m); s += z[c]; }
  const float inv = 1.0f / s;
#pragma unroll
  for (int c = 0; c < NC; ++c) z[c] *= inv;
  *amax = am;
  return m + __logf(s);     // logsumexp (of the raw logits)
}

__global__ void __launch_bounds__(256)
occ_loss_fwd_kernel(const float* __restrict__ x, const uint8_t* __restrict__ label, const float* __restrict__ cw,
                    double* __restrict__ partial, int B, int D, int H, int W) {
  constexpr int NF = 3 + 2 * NC;                 // float sums: ce_num, ce_den, M, sum_p[NC], nom[NC]
  __shared__ float wred[4][NF];                  // one slot per wave, folded in wave order (deterministic)
  __shared__ int ired[NC + NC * NC];             // cnt[NC], conf[NC][NC]: integer LDS atomics are exact
  for (int i = threadIdx.x; i < NC + NC * NC; i += 256) ired[i] = 0;
  __syncthreads();
  const long total = (long)B * 8 * D * H * W;
  float ce_num = 0.0f, ce_den = 0.0f, mvalid = 0.0f;
  float sp[NC], nm[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) { sp[c] = 0.0f; nm[c] = 0.0f; }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int t = label[i];
    if (t >= NC) continue;                       // 255 = ignore
    long r = i;
    const int ow = (int)(r % (2 * W)); r /= 2 * W;
    const int oh = (int)(r % (2 * H)); r /= 2 * H;
    const int od = (int)(r % (2 * D));
    const int b = (int)(r / (2 * D));
    float z[NC];
    upsampled_logits(x, b, D, H, W, od, oh, ow, z);
    float zt_raw = 0.0f;
#pragma unroll
    for (int c = 0; c < NC; ++c) zt_raw = (c == t) ? z[c] : zt_raw;
    int am;
    const float lse = softmax_inplace(z, &am);
    const float w = cw[t];
    ce_num += w * (lse - zt_raw);
    ce_den += w;
    mvalid += 1.0f;
#pragma unroll
    for (int c = 0; c < NC; ++c) { sp[c] += z[c]; nm[c] += (c == t) ? z[c] : 0.0f; }
    atomicAdd(&ired[t], 1);
    atomicAdd(&ired[NC + t * NC + am], 1);
  }
  ce_num = wave_sum(ce_num); ce_den = wave_sum(ce_den); mvalid = wave_sum(mvalid);
#pragma unroll
  for (int c = 0; c < NC; ++c) { sp[c] = wave_sum(sp[c]); nm[c] = wave_sum(nm[c]); }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    wred[wave][0] = ce_num; wred[wave][1] = ce_den; wred[wave][2] = mvalid;
#pragma unroll
    for (int c = 0; c < NC; ++c) { wred[wave][3 + c] = sp[c]; wred[wave][3 + NC + c] = nm[c]; }
  }
  __syncthreads();
  double* out = partial + (size_t)blockIdx.x * NS;
  for (int i = threadIdx.x; i < NF; i += 256)
    out[i] = (double)wred[0][i] + (double)wred[1][i] + (double)wred[2][i] + (double)wred[3][i];
  for (int i = threadIdx.x; i < NC + NC * NC; i += 256) out[NF + i] = (double)ired[i];
}

// sums[i] = sum over blocks in block order (deterministic given the per-block values)
__global__ void occ_loss_reduce_kernel(const double* __restrict__ partial, int nblocks, double* __restrict__ sums) {
  const int i = blockIdx.x;
  __shared__ double red[256];
  double a = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) a += partial[(size_t)b * NS + i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[i] = red[0];
}

// coef = [a_ce, A[NC], Bc[NC]] (dL/dce_num, dL/dsum_p[c], dL/dnom[c]).  One thread per FINE voxel writes its
// logit gradient (zero for ignored voxels); the x2 pull-back to the coarse grid is ssbev_trilinear2x_bwd.
__global__ void __launch_bounds__(256)
occ_loss_bwd_kernel(const float* __restrict__ x, const uint8_t* __restrict__ label, const float* __restrict__ cw,
                    const float* __restrict__ coef, float* __restrict__ gfine, int B, int D, int H, int W) {
  __shared__ float cf[1 + 2 * NC], wsh[NC];
  if (threadIdx.x < 1 + 2 * NC) cf[threadIdx.x] = coef[threadIdx.x];
  if (threadIdx.x < NC) wsh[threadIdx.x] = cw[threadIdx.x];
  __syncthreads();
  const long total = (long)B * 8 * D * H * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int t = label[i];
    float4* dst = reinterpret_cast<float4*>(gfine + (size_t)i * NC);
    if (t >= NC) {
#pragma unroll
      for (int q = 0; q < NC / 4; ++q) dst[q] = make_float4(0, 0, 0, 0);
      continue;
    }
    long r = i;
    const int ow = (int)(r % (2 * W)); r /= 2 * W;
    const int oh = (int)(r % (2 * H)); r /= 2 * H;
    const int od = (int)(r % (2 * D));
    const int b = (int)(r / (2 * D));
    float z[NC];
    upsampled_logits(x, b, D, H, W, od, oh, ow, z);
    int am;
    softmax_inplace(z, &am);
    float dot = 0.0f, wt = 0.0f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      dot += (cf[1 + c] + (c == t ? cf[1 + NC + c] : 0.0f)) * z[c];
      wt = (c == t) ? wsh[c] : wt;
    }
    const float wce = cf[0] * wt;
    float g[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float Cc = cf[1 + c] + (c == t ? cf[1 + NC + c] : 0.0f);
      g[c] = wce * (z[c] - (c == t ? 1.0f : 0.0f)) + z[c] * (Cc - dot);
    }
#pragma unroll
    for (int q = 0; q < NC / 4; ++q) dst[q] = make_float4(g[4 * q], g[4 * q + 1], g[4 * q + 2], g[4 * q + 3]);
  }
}

constexpr int FWD_BLOCKS = 2048;

// ---- the scalar algebra behind the sums (occhead.py:291-361, semkitti.py:67-149), one 64-thread block.
// losses: CE = ce_num / ce_den; sem_scal = mean over the present classes of nll(precision) + nll(recall) + nll(specificity);
// geo_scal = the same three terms for "occupied" (class 0 = empty complemented); nll(v) = -max(log v, -100)
// (F.binary_cross_entropy against a target of ones).  Written out as ~110 double-precision ATen ops (forward + autograd) this
// sat on the turning point of every step with the device idle; here the forward also emits the Jacobian of the three losses
// w.r.t. the 41 differentiable sums (ce_num, sum_p[20], nom[20]), so backward is three scaled rows.
__device__ __forceinline__ double nll1(double v, double& dv) {       // value and derivative of -clamp(log v, min = -100)
  const double l = log(v);
  if (l > -100.0) { dv = -1.0 / v; return -l; }
  dv = 0.0;                                     // clamped (or NaN: torch.clamp passes it on with a zero gradient)
  return l != l ? l : 100.0;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ void __launch_bounds__(64)
occ_loss_tail_kernel(const double* __restrict__ sums, float w_ce, float w_sem, float w_geo, float* __restrict__ out,
                     double* __restrict__ jac) {
  constexpr int ND = 1 + 2 * NC;                 // differentiable sums: ce_num, sum_p[NC], nom[NC]
  const int c = threadIdx.x;
  const double ce_num = sums[0], ce_den = sums[1], M = sums[2];
  const double* sum_p = sums + 3;
  const double* nom = sums + 3 + NC;
  const double* cnt = sums + 3 + 2 * NC;
  const double* conf = sums + 3 + 3 * NC;        // [gt][pred]
  // jac[k][j]: k = 0 CE, 1 sem_scal, 2 geo_scal; j = 0 ce_num, 1 + c sum_p[c], 1 + NC + c nom[c]
  for (int j = c; j < 3 * ND; j += 64) jac[j] = 0.0;
  __syncthreads();
  // ---- sem_scal
  double lc = 0.0, d_sp = 0.0, d_nom = 0.0, present = 0.0;
  if (c < NC) {
    const double sp = sum_p[c], nm = nom[c], ct = cnt[c], neg = M - ct;
    present = ct > 0.0 ? 1.0 : 0.0;
    double dv;
    if (sp > 0.0) {                                              // precision nom / sum_p
      lc += nll1(nm / sp, dv);
      d_nom += dv / sp; d_sp += dv * (-nm / (sp * sp));
    }
    {                                                            // recall nom / max(cnt, 1)
      const double den = ct < 1.0 ? 1.0 : ct;
      lc += nll1(nm / den, dv);
      d_nom += dv / den;
    }
    if (neg > 0.0) {                                             // specificity (neg - (sum_p - nom)) / max(neg, 1)
      const double den = neg < 1.0 ? 1.0 : neg;
      lc += nll1((neg - (sp - nm)) / den, dv);
      d_sp += -dv / den; d_nom += dv / den;
    }
  }
  const double npres = wave_sum(present);
  const double sem = wave_sum(lc * present) / npres;
  if (c < NC) {
    jac[1 * ND + 1 + c] = (double)w_sem * present * d_sp / npres;
    jac[1 * ND + 1 + NC + c] = (double)w_sem * present * d_nom / npres;
  }
  // ---- metric (no gradient): completion IoU over "occupied", mean IoU over classes 1..NC-1
  double tp = 0.0, fp = 0.0, fn = 0.0, iou_c = 0.0;
  if (c < NC) {
    double row = 0.0, col = 0.0, rocc = 0.0;                     // row: gt == c, col: pred == c
    for (int k = 0; k < NC; ++k) {
      row += conf[c * NC + k]; col += conf[k * NC + c];
      if (k > 0) rocc += conf[c * NC + k];
    }
    const double tpc = conf[c * NC + c];
    if (c > 0) { tp = rocc; fn = conf[c * NC + 0]; iou_c = tpc / (tpc + (col - tpc) + (row - tpc) + 1e-5); }
    else fp = rocc;                                              // gt empty, predicted occupied
  }
  tp = wave_sum(tp); fp = wave_sum(fp); fn = wave_sum(fn);
  const double miou = wave_sum(iou_c) / (double)(NC - 1);
  if (c == 0) {
    // ---- CE
    out[0] = (float)(ce_num / ce_den) * w_ce;
    jac[0] = (double)w_ce / ce_den;
    out[1] = (float)sem * w_sem;
    // ---- geo_scal
    const double sp0 = sum_p[0], nm0 = nom[0], ct0 = cnt[0];
    const double occ_t = M - ct0, inter = occ_t - (sp0 - nm0);
    double d1, d2, d3;
    const double g1 = nll1(inter / (M - sp0), d1), g2 = nll1(inter / occ_t, d2), g3 = nll1(nm0 / ct0, d3);
    out[2] = (float)(g1 + g2 + g3) * w_geo;
    // d inter / d sum_p0 = -1, d inter / d nom0 = +1
    const double den1 = M - sp0;
    jac[2 * ND + 1 + 0] = (double)w_geo * (d1 * (-1.0 / den1 + inter / (den1 * den1)) + d2 * (-1.0 / occ_t));
    jac[2 * ND + 1 + NC + 0] = (double)w_geo * (d1 / den1 + d2 / occ_t + d3 / ct0);
    out[3] = (float)(tp / (tp + fp + fn));
    out[4] = (float)miou;
  }
}

bool occ_ok(const ssbev_occloss_dims* d) {
  return d && d->B > 0 && d->D > 0 && d->H > 0 && d->W > 0 && d->C == NC;
}

}  // namespace

extern "C" {

int ssbev_occ_loss_num_sums(void) { return NS; }

size_t ssbev_occ_loss_workspace(const ssbev_occloss_dims* d) {
  return occ_ok(d) ? (size_t)FWD_BLOCKS * NS * sizeof(double) : 0;
}

int ssbev_occ_loss_fwd(const float* logits, const uint8_t* label, const float* class_weight, double* sums,
                       const ssbev_occloss_dims* d, void* ws, size_t ws_bytes, ssbev_stream_t stream) {
  if (!occ_ok(d) || !logits || !label || !class_weight || !sums || !ws) return SSBEV_EINVAL;
  if (ws_bytes < ssbev_occ_loss_workspace(d)) return SSBEV_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  double* partial = static_cast<double*>(ws);
  hipLaunchKernelGGL(occ_loss_fwd_kernel, dim3(FWD_BLOCKS), dim3(256), 0, st, logits, label, class_weight, partial, d->B,
                     d->D, d->H, d->W);
  hipLaunchKernelGGL(occ_loss_reduce_kernel, dim3(NS), dim3(256), 0, st, partial, FWD_BLOCKS, sums);
  return ssbev_launch_status();
}

int ssbev_occ_loss_tail(const double* sums, float w_ce, float w_sem, float w_geo, float* out5, double* jac,
                        ssbev_stream_t stream) {
  if (!sums || !out5 || !jac) return SSBEV_EINVAL;
  hipLaunchKernelGGL(occ_loss_tail_kernel, dim3(1), dim3(64), 0, as_stream(stream), sums, w_ce, w_sem, w_geo, out5, jac);
  return ssbev_launch_status();
}

size_t ssbev_occ_loss_bwd_workspace(const ssbev_occloss_dims* d) {
  return occ_ok(d) ? (size_t)d->B * 8 * d->D * d->H * d->W * NC * sizeof(float) : 0;
}

int ssbev_occ_loss_bwd(const float* logits, const uint8_t* label, const float* class_weight, const float* coef,
                       float* grad_logits, const ssbev_occloss_dims* d, void* ws, size_t ws_bytes,
                       ssbev_stream_t stream) {
  if (!occ_ok(d) || !logits || !label || !class_weight || !coef || !grad_logits || !ws) return SSBEV_EINVAL;
  if (ws_bytes < ssbev_occ_loss_bwd_workspace(d)) return SSBEV_EWORKSPACE;
  float* gfine = static_cast<float*>(ws);
  hipLaunchKernelGGL(occ_loss_bwd_kernel, dim3(4096), dim3(256), 0, as_stream(stream), logits, label, class_weight, coef,
                     gfine, d->B, d->D, d->H, d->W);
  ssbev_upsample_dims u = {d->B, d->D, d->H, d->W, NC};
  const int rc = ssbev_trilinear2x_bwd(gfine, grad_logits, &u, stream);
  return rc != SSBEV_OK ? rc : ssbev_launch_status();
}

}  // extern "C"
