// The elementwise shell of a BRI attention block (attention.py:45-86) around its six products: the three scalar-affine 1x1x1
// convolutions (query / key / value), the key-side confidence re-weight, `gamma * out + x`, and all of their gradients.  Written
// as tensor expressions these are ~9 launches forward and ~30 backward per block on 5.9 MB operands -- ~80 launches per step
// in a serial section of the graph where nothing else runs.  Here: two launches forward, two backward (+ four small sums).
// Operands are [B, D, T] with the token axis T contiguous (the [B, 1, D, H, W] volumes as they are).  Built without FMA
// contraction: every product and sum is the separately rounded fp32 operation of the tensor expression.
#include "common.h"

namespace {

constexpr int SH_DCH = 8;       // depth chunks: 7680 tokens alone are 30 workgroups

struct ShellGeom { int B, D, T; };

// Q = q wq + bq, K = kv wk + bk, Vc = (kv wv + bv) conf
__global__ void __launch_bounds__(256)
bri_shell_pre_fwd_kernel(const float* __restrict__ q, const float* __restrict__ kv, const float* __restrict__ conf,
                         const float* __restrict__ wq, const float* __restrict__ bq, const float* __restrict__ wk,
                         const float* __restrict__ bk, const float* __restrict__ wv, const float* __restrict__ bv,
                         float* __restrict__ Q, float* __restrict__ K, float* __restrict__ Vc, ShellGeom g) {
  const size_t n = (size_t)g.B * g.D * g.T;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int t = (int)(i % g.T), b = (int)(i / ((size_t)g.D * g.T));
  const float x = q[i], y = kv[i];
  Q[i] = __fadd_rn(__fmul_rn(x, wq[0]), bq[0]);
  K[i] = __fadd_rn(__fmul_rn(y, wk[0]), bk[0]);
  Vc[i] = __fmul_rn(__fadd_rn(__fmul_rn(y, wv[0]), bv[0]), conf[(size_t)b * g.T + t]);
}

// y = gamma out + kv
__global__ void __launch_bounds__(256)
bri_shell_post_fwd_kernel(const float* __restrict__ out, const float* __restrict__ kv, const float* __restrict__ gamma,
                          float* __restrict__ y, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = __fadd_rn(__fmul_rn(gamma[0], out[i]), kv[i]);
}

__device__ __forceinline__ double block_sum_d(double v, double* red) {       // fixed tree: deterministic
  red[threadIdx.x] = v;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

// gout = g gamma; delta_part[chunk][b, t] = sum_{d in chunk} gout out; part[block] = sum g out  (-> d gamma)
__global__ void __launch_bounds__(256)
bri_shell_post_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ out, const float* __restrict__ gamma,
                          float* __restrict__ gout, float* __restrict__ delta_part, double* __restrict__ part, ShellGeom g) {
  __shared__ double red[256];
  const int bt = blockIdx.x * 256 + threadIdx.x, nbt = g.B * g.T;
  const int per = (g.D + SH_DCH - 1) / SH_DCH, d0 = blockIdx.y * per, d1 = min(g.D, d0 + per);
  const float gm = gamma[0];
  float dl = 0.0f;
  double sg = 0.0;
  if (bt < nbt) {
    const int b = bt / g.T, t = bt - b * g.T;
    for (int d = d0; d < d1; ++d) {
      const size_t i = ((size_t)b * g.D + d) * g.T + t;
      const float gv = gy[i], o = out[i], go = __fmul_rn(gv, gm);
      gout[i] = go;
      dl = __fadd_rn(dl, __fmul_rn(go, o));
      sg += (double)__fmul_rn(gv, o);
    }
    delta_part[(size_t)blockIdx.y * nbt + bt] = dl;
  }
  const double s = block_sum_d(sg, red);
  if (threadIdx.x == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
}

// gq = gQ wq; gkv = (gK wk + (gVc conf) wv) + gres; gconf_part[chunk][b, t] = sum_d gVc (kv wv + bv);
// part[block][6] = sum gQ q, sum gQ, sum gK kv, sum gK, sum (gVc conf) kv, sum gVc conf   (-> d wq, bq, wk, bk, wv, bv)
__global__ void __launch_bounds__(256)
bri_shell_pre_bwd_kernel(const float* __restrict__ gQ, const float* __restrict__ gK, const float* __restrict__ gVc,
                         const float* __restrict__ q, const float* __restrict__ kv, const float* __restrict__ conf,
                         const float* __restrict__ gres, const float* __restrict__ wq, const float* __restrict__ wk,
                         const float* __restrict__ wv, const float* __restrict__ bv, float* __restrict__ gq,
                         float* __restrict__ gkv, float* __restrict__ gconf_part, double* __restrict__ part, ShellGeom g) {
  __shared__ double red[256];
  const int bt = blockIdx.x * 256 + threadIdx.x, nbt = g.B * g.T;
  const int per = (g.D + SH_DCH - 1) / SH_DCH, d0 = blockIdx.y * per, d1 = min(g.D, d0 + per);
  const float a = wq[0], bw = wk[0], c = wv[0], cb = bv[0];
  double s[6] = {0, 0, 0, 0, 0, 0};
  float gc = 0.0f;
  if (bt < nbt) {
    const int b = bt / g.T, t = bt - b * g.T;
    const float cf = conf[bt];
    for (int d = d0; d < d1; ++d) {
      const size_t i = ((size_t)b * g.D + d) * g.T + t;
      const float x = q[i], y = kv[i], u = gQ[i], v = gK[i], w = gVc[i];
      const float wc = __fmul_rn(w, cf);                         // gradient w.r.t. V
      gq[i] = __fmul_rn(u, a);
      gkv[i] = __fadd_rn(__fadd_rn(__fmul_rn(v, bw), __fmul_rn(wc, c)), gres[i]);
      gc = __fadd_rn(gc, __fmul_rn(w, __fadd_rn(__fmul_rn(y, c), cb)));
      s[0] += (double)__fmul_rn(u, x); s[1] += (double)u;
      s[2] += (double)__fmul_rn(v, y); s[3] += (double)v;
      s[4] += (double)__fmul_rn(wc, y); s[5] += (double)wc;
    }
    gconf_part[(size_t)blockIdx.y * nbt + bt] = gc;
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double r = block_sum_d(s[k], red);
    if (threadIdx.x == 0) part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 6 + k] = r;
  }
}

bool shell_ok(int B, int D, int T) { return B > 0 && D > 0 && T > 0 && (long)B * D * T < (1L << 31); }

}  // namespace

extern "C" {

int ssbev_bri_shell_chunks(void) { return SH_DCH; }

int ssbev_bri_shell_pre_fwd(const float* q, const float* kv, const float* conf, const float* wq, const float* bq,
                            const float* wk, const float* bk, const float* wv, const float* bv, float* Q, float* K, float* Vc,
                            int B, int D, int T, ssbev_stream_t stream) {
  if (!shell_ok(B, D, T) || !q || !kv || !conf || !wq || !bq || !wk || !bk || !wv || !bv || !Q || !K || !Vc) return SSBEV_EINVAL;
  const size_t n = (size_t)B * D * T;
  hipLaunchKernelGGL(bri_shell_pre_fwd_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream), q, kv, conf, wq, bq,
                     wk, bk, wv, bv, Q, K, Vc, ShellGeom{B, D, T});
  return ssbev_launch_status();
}

int ssbev_bri_shell_post_fwd(const float* out, const float* kv, const float* gamma, float* y, int B, int D, int T,
                             ssbev_stream_t stream) {
  if (!shell_ok(B, D, T) || !out || !kv || !gamma || !y) return SSBEV_EINVAL;
  const size_t n = (size_t)B * D * T;
  hipLaunchKernelGGL(bri_shell_post_fwd_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream), out, kv, gamma, y, n);
  return ssbev_launch_status();
}

/* delta_part [chunks][B*T] floats, part [chunks * ceil(B*T / 256)] doubles */
int ssbev_bri_shell_post_bwd(const float* gy, const float* out, const float* gamma, float* gout, float* delta_part, double* part,
                             int B, int D, int T, ssbev_stream_t stream) {
  if (!shell_ok(B, D, T) || !gy || !out || !gamma || !gout || !delta_part || !part) return SSBEV_EINVAL;
  hipLaunchKernelGGL(bri_shell_post_bwd_kernel, dim3((unsigned)cdiv((size_t)B * T, 256), SH_DCH), dim3(256), 0, as_stream(stream),
                     gy, out, gamma, gout, delta_part, part, ShellGeom{B, D, T});
  return ssbev_launch_status();
}

/* gconf_part [chunks][B*T] floats, part [chunks * ceil(B*T / 256)][6] doubles */
int ssbev_bri_shell_pre_bwd(const float* gQ, const float* gK, const float* gVc, const float* q, const float* kv, const float* conf,
                            const float* gres, const float* wq, const float* wk, const float* wv, const float* bv, float* gq,
                            float* gkv, float* gconf_part, double* part, int B, int D, int T, ssbev_stream_t stream) {
  if (!shell_ok(B, D, T) || !gQ || !gK || !gVc || !q || !kv || !conf || !gres || !wq || !wk || !wv || !bv || !gq || !gkv ||
      !gconf_part || !part)
    return SSBEV_EINVAL;
  hipLaunchKernelGGL(bri_shell_pre_bwd_kernel, dim3((unsigned)cdiv((size_t)B * T, 256), SH_DCH), dim3(256), 0, as_stream(stream),
                     gQ, gK, gVc, q, kv, conf, gres, wq, wk, wv, bv, gq, gkv, gconf_part, part, ShellGeom{B, D, T});
  return ssbev_launch_status();
}

}  // extern "C"
