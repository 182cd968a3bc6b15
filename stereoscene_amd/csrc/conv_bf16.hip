// bf16-STORAGE convolution family for gfx950 (BASELINE configs[3]: "bf16 mixed precision, MFMA 3D-conv path").
//
// Rounds 1-3 had a "bf16 mode" that kept every tensor fp32 in HBM and only rounded the MFMA operands in registers: the
// direct kernels stayed operand-delivery-bound and nothing HBM-bound got cheaper.  Here the activations and their
// gradients ARE bf16 in memory (channels-last [B, D, H, W, C], C a multiple of 8), which is what the 32x32x16 bf16 MFMA
// wants to eat: ONE 16-byte load (global or LDS) = 8 consecutive channels = the k-slice a lane supplies to
// v_mfma_f32_32x32x16_bf16 -- no v_cvt, half the bytes, 4x the multiply-adds per operand byte of the fp32 path.
// Accumulation is fp32; weights stay fp32 masters and are packed to bf16 operand order per launch (a few microseconds);
// weight gradients are produced in fp32.  The reference mechanism this realises is mmcv's Fp16OptimizerHook / auto_fp16
// (reference: projects/mmdet3d_plugin/occupancy/apis/mmdet_train.py:131-134, tools/fp16/train.py:224-226), with bf16 in
// place of fp16 (same 16-bit storage, fp32 exponent range: no loss scaling needed, kept available in train.LossScaler).
//
// Kernels:
//   conv_gather16_kernel<MT,NT,YT>   forward / data gradient of ANY conv / transposed conv (stride, dilation, 1x1, k == s):
//                                    implicit GEMM, A operand straight from L1/L2 as 16-byte voxel-line pieces (the
//                                    design of conv_gather_kernel in conv_mfma.hip, parity-class walk for the transposed form).
//   conv_tap16_kernel<YT>            the <= 32-channel stride-1 3x3x3 layers of the cost-volume stack: input rows in an LDS
//                                    ring of bf16 voxel lines (global_load_lds), weights in registers, taps split over 4 waves.
//   wgrad16_kernel<MQ,MP,TH,TW>      weight gradient of any of them.  The reduction axis is the VOXEL axis, i.e. the MFMA
//                                    operands are the transposes of what lies in memory: every wave stages 16-voxel x
//                                    32-channel tiles through its own slice of LDS (one 16-byte load + one ds_write_b128 per
//                                    lane) and reads them back with ds_read_b64_tr_b16, gfx950's transposing LDS read,
//                                    which hands each lane 4 consecutive VOXELS of its channel: two of them = one operand.
#include "conv_bf16.h"

#include <algorithm>
#include <cstdlib>

namespace ssbev_detail {
void wgrad_reduce(float* partial, float* gw, int nchunks, int taps, int Cq, int Cp, hipStream_t st);   // conv_mfma.hip
}

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short i16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma16(const uint4 a, const uint4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ const uint4 kZero16 = {0u, 0u, 0u, 0u};

int pad16(int c) { return (c + 15) & ~15; }
int pad32(int c) { return (c + 31) & ~31; }

struct Geom16 {
  int B, Cin, Cout, KP, CoutPad;       // K = Cin of THIS gather (multiple of 8), KP = ceil(Cin / 16) operand groups, N = Cout
  int Di, Hi, Wi, Do, Ho, Wo;
  int kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw;
  int form;                            // 0 conv gather, 1 transposed gather (parity classes)
  int relu, accumulate;
  long bsx = 0, bsy = 0, bsw = 0;      // conv_igemm16_kernel as a batched plain GEMM (ssbev_gemm16_nn): element strides of x / y / packed weights per blockIdx.z
};

// ------------------------------------------------------------------------------------------------ weight packing
// dst (bf16) [tap][p][lk][n][t] = W[k = 16 p + 8 lk + t][n][tap]: lane (n, lk) of the B operand reads 16 bytes.
//   layout 0: src [A0, A1, taps] with K = A1, N = A0;   layout 1: K = A0, N = A1   (as pack_weight_kernel, conv_mfma.hip)
__global__ void pack16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int K, int N, int KP, int NPad,
                              int taps, int layout, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  long r = i;
  const int t = (int)(r & 7); r >>= 3;
  const int n = (int)(r % NPad); r /= NPad;
  const int lk = (int)(r & 1); r >>= 1;
  const int p = (int)(r % KP);
  const int tap = (int)(r / KP);
  const int k = 16 * p + 8 * lk + t;
  float v = 0.0f;
  if (k < K && n < N) {
    const size_t a0 = layout == 0 ? n : k, a1 = layout == 0 ? k : n;
    const size_t A1 = layout == 0 ? K : N;
    v = src[(a0 * A1 + a1) * taps + tap];
  }
  dst[i] = f2bf(v);
}

// ------------------------------------------------------------------------------------------------ generic gather
// One wave: (MT * 32 voxels) x (NT * 32 channels); see conv_gather_kernel (conv_mfma.hip) for the tap walk, the parity
// classes of the transposed form and the XCD-aware tile order -- this is that kernel for bf16 tensors.
template <int MT, int NT, typename YT>
__global__ void __launch_bounds__(256)
conv_gather16_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp, const float* __restrict__ bias,
                     YT* __restrict__ y, Geom16 g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lk = lane >> 5;
  int par_d = 0, par_h = 0, par_w = 0;
  int Dc = g.Do, Hc = g.Ho, Wc = g.Wo;
  if (g.form == 1) {
    int cls = (int)gridDim.z - 1 - (int)blockIdx.z;       // heaviest parity class first
    par_w = cls % g.sw; cls /= g.sw;
    par_h = cls % g.sh; cls /= g.sh;
    par_d = cls;
    Dc = (g.Do - par_d + g.sd - 1) / g.sd; Hc = (g.Ho - par_h + g.sh - 1) / g.sh; Wc = (g.Wo - par_w + g.sw - 1) / g.sw;
  }
  const long Mtot = (long)g.B * Dc * Hc * Wc;
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
  const long m_wave = ((long)bx * 4 + wave) * (MT * 32);
  const int n0 = by * (NT * 32);
  if (m_wave >= Mtot) return;

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

  const int P = g.KP;
  const size_t w_tap_stride = (size_t)P * 2 * g.CoutPad * 8;
  const bf16_t* wlane = wp + ((size_t)lk * g.CoutPad + n0 + li) * 8;

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
  const bf16_t* pbase[MT];
  unsigned vmask[MT];
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
    pbase[mt] = x + ((((long)ob[mt] * g.Di + bd) * g.Hi + bh) * g.Wi + bw) * (long)g.Cin + 8 * lk;
  }

  for (int ti = 0; ti < ntaps; ++ti) {
    const int ic = ti % nkw, ib = (ti / nkw) % nkh, ia = ti / (nkw * nkh);
    const int c = kw0 + ic * kws, bq = kh0 + ib * khs, a = kd0 + ia * kds;
    const int tap = (a * g.kh + bq) * g.kw + c;
    const long toff = (((long)ia * step_d * g.Hi + (long)ib * step_h) * g.Wi + (long)ic * step_w) * g.Cin;   // wave-uniform
    const unsigned need = (1u << ia) | (1u << (8 + ib)) | (1u << (16 + ic));
    const bf16_t* ap[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) ap[mt] = ((vmask[mt] & need) == need) ? pbase[mt] + toff : nullptr;
    const bf16_t* wt = wlane + (size_t)tap * w_tap_stride;
    for (int p0 = 0; p0 < P; p0 += 2) {
      uint4 av[2][MT], bv[2][NT];
      const bool two = p0 + 1 < P;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int p = p0 + u;
        const bool pok = u == 0 || two;
        const bool cok = pok && (16 * p + 8 * lk) < g.Cin;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          av[u][mt] = (ap[mt] && cok) ? *reinterpret_cast<const uint4*>(ap[mt] + 16 * p) : kZero16;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          bv[u][nt] = (pok && n0 + nt * 32 < g.CoutPad)
                          ? *reinterpret_cast<const uint4*>(wt + ((size_t)p * 2 * g.CoutPad + nt * 32) * 8) : kZero16;
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma16(av[0][mt], bv[0][nt], acc[mt][nt]);
      if (two) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma16(av[1][mt], bv[1][nt], acc[mt][nt]);
      }
    }
  }

  // epilogue: accumulator row r of sub-tile mt = voxel of lane (r & 3) + 8 (r >> 2) + 4 lk, column = channel n0 + 32 nt + li
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
            oldv[r][nt] = (rok && co < g.Cout) ? ld1(y + vox * g.Cout + co) : 0.0f;
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
          st1(y + vox * g.Cout + co, v);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ LDS-ring tap kernel
// 3x3x3 / stride 1 / pad 1, K <= 32 input and N <= 32 output channels (multiples of 8): conv_tap_kernel (conv_mfma.hip) with
// bf16 voxel lines.  Ring [3 planes][4 row slots][34 voxels][32 channels] of bf16: a voxel line is 64 bytes = four 16-byte
// items; MFMA roles rows = output channel (A = this wave's taps, in registers), columns = voxel (B = one ds_read_b128 per
// 8 channels), k = input channel.  Four voxel lines share a 256-byte bank row, so the physical item of (voxel u, channel
// octet q) is q ^ ((u >> 2) & 3): the 16 lanes of a ds_read_b128 lane group then hit 16 different 16-byte slots for every
// kw shift (applied on the GLOBAL side of the LDS load, whose LDS side must stay lane-contiguous).
struct Tap16Geom {
  int B, D, H, W, K, N;
  int nseg, NG, gpc;
  int relu, has_bias, accumulate;
};

constexpr int kT16Wseg = 32, kT16Cols = kT16Wseg + 2;
constexpr int kT16RowB = kT16Cols * 64;                  // bytes per ring row
constexpr int kT16Slots = 4, kT16PlaneB = kT16Slots * kT16RowB, kT16RingB = 3 * kT16PlaneB;
constexpr size_t kT16LdsBytes = (size_t)kT16RingB + 4 * 16 * 64 * sizeof(float);
constexpr int kT16PackedU4 = 4 * 7 * 2 * 64;             // uint4 items

// wp (uint4 items) [((wave * 7 + tt) * 2 + j) * 64 + lane] = 8 bf16: Weff[n = lane & 31][k = 16 j + 8 (lane >> 5) + t][tap = wave + 4 tt]
//   mode 0: Weff[n][k][tap] = w[n][k][tap];  mode 1 (data gradient): Weff[n][k][tap] = w[k][n][26 - tap]
__global__ void __launch_bounds__(256)
pack_tap16_kernel(const float* __restrict__ w, bf16_t* __restrict__ wp, int Cout, int Cin, int mode) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= kT16PackedU4 * 8) return;
  const int t = i & 7, lane = (i >> 3) & 63, j = (i >> 9) & 1, wt = i >> 10;
  const int tt = wt % 7, wave = wt / 7;
  const int tap = wave + 4 * tt, n = lane & 31, k = 16 * j + 8 * (lane >> 5) + t;
  const int K = mode == 0 ? Cin : Cout, N = mode == 0 ? Cout : Cin;
  float v = 0.0f;
  if (tap < 27 && n < N && k < K)
    v = mode == 0 ? w[((size_t)n * Cin + k) * 27 + tap] : w[((size_t)k * Cin + n) * 27 + (26 - tap)];
  wp[i] = f2bf(v);
}

template <typename YT>
__global__ void __launch_bounds__(256)
conv_tap16_kernel(const bf16_t* __restrict__ X, const uint4* __restrict__ wp, const float* __restrict__ bias,
                  YT* __restrict__ Y, Tap16Geom g) {
  extern __shared__ __align__(16) unsigned char tl[];
  unsigned char* ring = tl;
  float* red = reinterpret_cast<float*>(tl + kT16RingB);      // [4 waves][16 rows][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lk = lane >> 5;

  uint4 wb[7][2];
#pragma unroll
  for (int tt = 0; tt < 7; ++tt)
#pragma unroll
    for (int j = 0; j < 2; ++j) wb[tt][j] = wp[((wave * 7 + tt) * 2 + j) * 64 + lane];
  const int ntap = wave < 3 ? 7 : 6;

  unsigned chunk_id;
  {
    const unsigned n = gridDim.x, L = blockIdx.x;
    const unsigned xcd = L & 7, q = n >> 3, r = n & 7;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    chunk_id = base + (L >> 3);
  }
  const int seg = chunk_id % g.nseg, range = chunk_id / g.nseg;
  const int w0 = seg * kT16Wseg;
  const int g_begin = range * g.gpc, g_end = min(g.NG, g_begin + g.gpc);

  // staging entries: one entry = 64 consecutive 16-byte items of one (plane, row); item j = (voxel u = j >> 2, physical
  // octet j & 3), source octet (j & 3) ^ ((u >> 2) & 3)
  constexpr int items = kT16Cols * 4, nxc = (items + 63) / 64;           // 136 items, 3 entries per row
  int xoff[3], xmeta[3];
  const int plane_g = g.H * g.W * g.K;
#pragma unroll
  for (int n = 0; n < 3; ++n) {
    const int q = wave + n * 4;
    int off = -2, meta = -1;
    if (q < 3 * nxc) {
      const int pl = q / nxc, ch = q % nxc;
      const int j = ch * 64 + lane;
      if (j < items) {
        const int u = j >> 2, c = (((j & 3) ^ ((u >> 2) & 3)) << 3), wsrc = w0 + u - 1;
        off = (wsrc >= 0 && wsrc < g.W && c < g.K) ? pl * plane_g + wsrc * g.K + c : -1;
      }
      meta = pl | ((pl * kT16PlaneB + ch * 1024) << 4);
    }
    xoff[n] = off;
    xmeta[n] = __builtin_amdgcn_readfirstlane(meta);
  }
  auto stage_row = [&](int b, int d, int hp) {
    const bf16_t* base = X + ((long)(b * g.D + d - 1) * g.H + (hp - 1)) * (long)(g.W * g.K);
    const int h = hp - 1;
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      const int meta = xmeta[n];
      if (meta < 0) break;
      const int pl = meta & 3, dp = d - 1 + pl;
      const bool rowok = h >= 0 && h < g.H && dp >= 0 && dp < g.D;
      unsigned char* dst = ring + (meta >> 4) + (hp & 3) * kT16RowB;
      const int off = xoff[n];
      const void* src = (rowok && off >= 0) ? static_cast<const void*>(base + off) : static_cast<const void*>(&kZero16);
      if (off != -2) __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
    }
  };

  int tp_plane[7], tp_kh[7], tp_kw[7];
#pragma unroll
  for (int tt = 0; tt < 7; ++tt) {
    const int t = min(wave + 4 * tt, 26);
    tp_plane[tt] = (t / 9) * kT16PlaneB; tp_kh[tt] = (t / 3) % 3; tp_kw[tt] = t % 3;
  }
  const int nb = 8 * wave + 4 * lk;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (g.has_bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = nb + i < g.N ? bias[nb + i] : 0.0f;
  }

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

    f32x16 acc2[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[q][r] = 0.0f;
    uint4 xa[2], xb2[2];
    auto fetch = [&](int tt, uint4 (&xv)[2]) {
      const int u = li + tp_kw[tt];
      const unsigned char* rowp = ring + tp_plane[tt] + ((h + tp_kh[tt]) & 3) * kT16RowB + u * 64;
      const int key = (u >> 2) & 3;
#pragma unroll
      for (int j = 0; j < 2; ++j) xv[j] = *reinterpret_cast<const uint4*>(rowp + (((2 * j + lk) ^ key) << 4));
    };
    fetch(0, xa);
#pragma unroll
    for (int tt = 0; tt < 7; tt += 2) {
      if (tt < ntap) {
        acc2[0] = mfma16(wb[tt][0], xa[0], acc2[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (tt + 1 < 7 && tt + 1 < ntap) fetch(tt + 1, xb2);
        __builtin_amdgcn_sched_barrier(0);
        acc2[1] = mfma16(wb[tt][1], xa[1], acc2[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (tt + 1 < 7 && tt + 1 < ntap) {
        acc2[0] = mfma16(wb[tt + 1][0], xb2[0], acc2[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (tt + 2 < 7 && tt + 2 < ntap) fetch(tt + 2, xa);
        __builtin_amdgcn_sched_barrier(0);
        acc2[1] = mfma16(wb[tt + 1][1], xb2[1], acc2[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc2[0][r] + acc2[1][r];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
      if (wv < g.W && nb < g.N) {
        YT* dst = Y + (((long)(b * g.D + d) * g.H + h) * g.W + wv) * g.N + nb;
        if (g.accumulate) { const float4 t = ld4(dst); o[0] += t.x; o[1] += t.y; o[2] += t.z; o[3] += t.w; }
        st4(dst, make_float4(o[0], o[1], o[2], o[3]));
      }
    }
    __syncthreads();
    fresh = !same_plane;
    if (++h == g.H) {
      h = 0;
      if (++d == g.D) { d = 0; ++b; }
    }
  }
}

// ------------------------------------------------------------------------------------------------ wide LDS-ring kernel
// 3x3x3 / stride 1 / pad 1 with K % 32 == 0 input and N % 64 == 0 output channels on grids whose W is a multiple of 16: the
// voxel encoder's 128 -> 128 layers and the 384 -> 192 head conv at 128 x 128 x 16, the 64 -> 64 layers of the cost-volume
// hourglasses.  In the bf16 mode these ran on the F(2,3)^3 Winograd pipeline + library GEMMs: HBM-bound on its 8x larger
// transformed tensors (4.5 GB per 128 -> 128 layer against 0.27 GB of operands, 334 TF/s operator rate), while the direct
// contraction is MFMA-bound (232 GF = 0.09 ms at the bf16 peak).  This kernel is the implicit GEMM with the A operand in LDS:
//   * a workgroup owns 16 rows x 16 voxels of one depth plane x NTL * 32 output channels; wave w owns rows 4w .. 4w+3 as two
//     32-voxel MFMA row groups (2 rows x 16 voxels), i.e. a 64 voxel x NTL * 32 channel register tile (2 x NTL accumulators);
//   * the input arrives 32 channels (one 64-byte voxel line piece) at a time: 3 planes x 18 rows x 18 voxels x 64 bytes = 61 KB
//     by global_load_lds, item swizzle as in conv_tap16_kernel; each staged chunk feeds 27 taps x 2 k-steps x 2 NTL MFMAs per
//     wave (432 for NTL = 4) between two barriers; two workgroups per CU, so one stages while the other contracts;
//   * the B operand (weights, pack16 layout) is read straight from L1/L2 -- 16 bytes per lane feeding two MFMAs -- one k-step
//     ahead of its use (ping-pong registers).
// Data gradient = the same walk with the channel roles swapped by the packing and the taps mirrored (26 - tap).
struct Wide16Geom {
  int B, D, H, W, K, N, KP, NPad;
  int nth, ntw, ntn;               // tiles along h, along w, along the output channels
  int relu, accumulate, mirror;
};

constexpr int kW16Rows = 16, kW16RR = kW16Rows + 2, kW16RC = 18;
constexpr int kW16RowB = kW16RC * 64, kW16PlaneB = kW16RR * kW16RowB, kW16RingB = 3 * kW16PlaneB;      // 1152, 20736, 62208
constexpr int kW16Items = 3 * kW16RR * kW16RC * 4, kW16Entries = (kW16Items + 63) / 64;                // 3888, 61

template <int NTL, typename YT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
conv_wide16_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ wp, const float* __restrict__ bias,
                   YT* __restrict__ Y, Wide16Geom g) {
  extern __shared__ __align__(16) unsigned char ring[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lk = lane >> 5;
  int mtile, ntile;
  {   // XCD-aware order: the column tiles of one voxel tile, then neighbouring voxel tiles, share an L2
    const unsigned n = gridDim.x, L = blockIdx.x;
    const unsigned xcd = L & 7, q = n >> 3, r = n & 7;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const unsigned Lp = base + (L >> 3);
    mtile = (int)(Lp / g.ntn); ntile = (int)(Lp % g.ntn);
  }
  int t = mtile;
  const int wt_ = t % g.ntw; t /= g.ntw;
  const int ht = t % g.nth; t /= g.nth;
  const int d = t % g.D;
  const int b = t / g.D;
  const int h0 = ht * kW16Rows, w0 = wt_ * 16;
  const int n0 = ntile * (NTL * 32);

  // ---- staging: entry e = wave + 4 n covers LDS items [64 e, 64 e + 64); item = (plane, row, voxel u, physical octet).  The
  // decode is redone per chunk (a few dozen VALU instructions against 432 MFMAs) rather than kept in 16 registers
  const bf16_t* xbase = X + ((((long)b * g.D + (d - 1)) * g.H + (h0 - 1)) * g.W + (w0 - 1)) * (long)g.K;
  auto stage = [&](int c) {
#pragma unroll 4
    for (int n = 0; n < (kW16Entries + 3) / 4; ++n) {
      const int e = wave + 4 * n;
      if (e >= kW16Entries) break;
      const int L = e * 64 + lane;
      if (L < kW16Items) {
        const int pl = L / (kW16RR * kW16RC * 4), rem = L % (kW16RR * kW16RC * 4);
        const int row = rem / (kW16RC * 4), it = rem % (kW16RC * 4);
        const int u = it >> 2, so = (it & 3) ^ ((u >> 2) & 3);
        const int ds = d - 1 + pl, hs = h0 - 1 + row, ws = w0 - 1 + u;
        const bool ok = ds >= 0 && ds < g.D && hs >= 0 && hs < g.H && ws >= 0 && ws < g.W;
        const void* src = ok ? static_cast<const void*>(xbase + (((long)pl * g.H + row) * g.W + u) * g.K + so * 8 + c * 32)
                             : static_cast<const void*>(&kZero16);
        __builtin_amdgcn_global_load_lds(src, ring + e * 1024, 16, 0, 0);
      }
    }
  };

  // ---- A operand: lane = (row r2 of its 2-row group, voxel wq)
  const int r2 = li >> 4, wq = li & 15;
  const int rowb0 = (4 * wave + r2) * kW16RowB;             // row group mt: + 2 mt rows

  const size_t w_tap_stride = (size_t)g.KP * 2 * g.NPad * 8;
  const size_t w_p_stride = (size_t)2 * g.NPad * 8;
  const bf16_t* wlane = wp + ((size_t)lk * g.NPad + n0 + li) * 8;

  f32x16 acc[2][NTL];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

  const int nchunks = g.K >> 5;
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();                                   // every wave is done with the previous chunk
    stage(c);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // 27 taps x 2 halves of the 32-channel chunk; the operands of a step are requested one step ahead (ping-pong registers)
    uint4 a0[2], b0[NTL], a1[2], b1[NTL];
    auto load = [&](int kd, int kh, int kw, int tap, int j, uint4 (&av)[2], uint4 (&bv)[NTL]) __attribute__((always_inline)) {
      const int u = wq + kw;
      const unsigned char* ap = ring + kd * kW16PlaneB + kh * kW16RowB + rowb0 + u * 64 + (((2 * j + lk) ^ ((u >> 2) & 3)) << 4);
      av[0] = *reinterpret_cast<const uint4*>(ap);
      av[1] = *reinterpret_cast<const uint4*>(ap + 2 * kW16RowB);
      const bf16_t* wt = wlane + (size_t)(g.mirror ? 26 - tap : tap) * w_tap_stride + (size_t)(2 * c + j) * w_p_stride;
#pragma unroll
      for (int nt = 0; nt < NTL; ++nt) bv[nt] = *reinterpret_cast<const uint4*>(wt + nt * 32 * 8);
    };
    auto mma = [&](const uint4 (&av)[2], const uint4 (&bv)[NTL]) __attribute__((always_inline)) {
#pragma unroll
      for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[mt][nt] = mfma16(av[mt], bv[nt], acc[mt][nt]);
    };
    load(0, 0, 0, 0, 0, a0, b0);
    int kd = 0, kh = 0, kw = 0;
    for (int tap = 0; tap < 27; ++tap) {
      load(kd, kh, kw, tap, 1, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      mma(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      // next tap (uniform counters)
      if (++kw == 3) { kw = 0; if (++kh == 3) { kh = 0; ++kd; } }
      if (tap + 1 < 27) load(kd, kh, kw, tap + 1, 0, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      mma(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: accumulator row rr of group mt = voxel (row 4 wave + 2 mt + (vi >> 4), w = vi & 15), vi = (rr & 3) + 8 (rr >> 2) + 4 lk
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += 8) {
      float oldv[8][NTL];
      if (g.accumulate) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int vi = ((r0 + r) & 3) + 8 * ((r0 + r) >> 2) + 4 * lk;
          const int h = h0 + 4 * wave + 2 * mt + (vi >> 4), w = w0 + (vi & 15);
          const size_t vox = (((size_t)b * g.D + d) * g.H + h) * g.W + w;
#pragma unroll
          for (int nt = 0; nt < NTL; ++nt)
            oldv[r][nt] = h < g.H ? ld1(Y + vox * g.N + n0 + nt * 32 + li) : 0.0f;
        }
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int vi = ((r0 + r) & 3) + 8 * ((r0 + r) >> 2) + 4 * lk;
        const int h = h0 + 4 * wave + 2 * mt + (vi >> 4), w = w0 + (vi & 15);
        const size_t vox = (((size_t)b * g.D + d) * g.H + h) * g.W + w;
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
          const int co = n0 + nt * 32 + li;
          if (h < g.H) {
            float v = acc[mt][nt][r0 + r];
            if (g.accumulate) v += oldv[r][nt];
            if (bias) v += bias[co];
            if (g.relu) v = fmaxf(v, 0.0f);
            st1(Y + vox * g.N + co, v);
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// gw[tap][cq][cp] = sum_m Q[pos(m, tap)][cq] * P[m][cp]   (m over the small grid; conv: P = gy, Q = x; transposed: P = x, Q = gy)
// One wave: (32 MQ q-channels) x (32 MP p-channels) x (TH x TW taps of one kd slice) over its own chunk of voxels, 16 voxels
// (one MFMA k-step) per trip.  Lane l loads 16 bytes = 8 channels of voxel l >> 2 for every operand tile (a whole 1 KB tile
// per wave instruction), parks it in the wave's LDS slice as [16 voxels][32 channels], and the transposing read returns
// [channel = l & 31][voxels 8 (l >> 5) .. + 7] -- the MFMA operand.  LDS is in-order per wave: no barrier anywhere.
struct Wg16Geom {
  int B, Cp, Cq;
  int Ds, Hs, Ws;
  int Dq, Hq, Wq;
  int kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw;
  int chunk, nchunks;
  int xcd_order;                    // 1: [chunk group][channel tiles][tap groups] per XCD; 0: the plain x-fastest order (A/B)
};

__device__ __forceinline__ uint4 tr_operand(const unsigned char* tile, int trbase) {
  typedef __attribute__((address_space(3))) i16x4 lds_i16x4;
  const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4*)(tile + trbase));
  const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4*)(tile + trbase + 4 * 64));
  uint4 r;
  r.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
  r.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
  r.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
  r.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
  return r;
}

template <int MQ, int MP, int TH, int TW>
__global__ void __launch_bounds__(256)
wgrad16_kernel(const bf16_t* __restrict__ P, const bf16_t* __restrict__ Q, float* __restrict__ ws, Wg16Geom g) {
  constexpr int NQ = MQ * TH * TW, NTILE = MP + NQ;
  extern __shared__ __align__(16) unsigned char wl[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lk = lane >> 5;
  // 1-D grid in the XCD-aware order [chunk group][channel tiles][tap groups]: the workgroups that read the SAME voxel chunk (one
  // per channel tile and tap group) are neighbours on one XCD -- the chunk comes from HBM once and from that L2 afterwards
  unsigned Lp;
  {
    const unsigned n = gridDim.x, L = blockIdx.x;
    const unsigned xcd = L & 7, q = n >> 3, r = n & 7;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    Lp = g.xcd_order ? base + (L >> 3) : L;
  }
  const int kw_groups = (g.kw + TW - 1) / TW, kh_groups = (g.kh + TH - 1) / TH;
  const unsigned ztiles = (unsigned)(g.kd * kh_groups * kw_groups);
  const int nqt = (g.Cq + 32 * MQ - 1) / (32 * MQ);
  const unsigned ytiles = (unsigned)(nqt * ((g.Cp + 32 * MP - 1) / (32 * MP)));
  unsigned bx, by, bz;
  if (g.xcd_order) { bz = Lp % ztiles; Lp /= ztiles; by = Lp % ytiles; bx = Lp / ytiles; }
  else { const unsigned nx = gridDim.x / (ytiles * ztiles); bx = Lp % nx; Lp /= nx; by = Lp % ytiles; bz = Lp / ytiles; }
  const int chunk_id = (int)bx * 4 + wave;
  if (chunk_id >= g.nchunks) return;
  unsigned char* my = wl + (size_t)wave * NTILE * 1024;
  const int qt = (int)by % nqt, pt = (int)by / nqt;
  int tg = (int)bz;
  const int kwg = tg % kw_groups; tg /= kw_groups;
  const int khg = tg % kh_groups;
  const int kdi = tg / kh_groups;

  f32x16 acc[MQ][MP][TH][TW];
#pragma unroll
  for (int a = 0; a < MQ; ++a)
#pragma unroll
    for (int e = 0; e < MP; ++e)
#pragma unroll
      for (int c = 0; c < TH; ++c)
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][e][c][t][r] = 0.0f;

  const long Mtot = (long)g.B * g.Ds * g.Hs * g.Ws;
  const long m_begin = (long)chunk_id * g.chunk;
  const long m_end = min(Mtot, m_begin + g.chunk);
  const int j = lane >> 2, c8 = (lane & 3) * 8;           // staging role: voxel j of the trip, channel octet c8
  // transposing-read address of this lane inside a tile (+ 4 rows for the second half)
  const int trbase = (8 * lk + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  int w, h, d, b;
  {
    long r = min(m_begin + j, Mtot - 1);
    w = (int)(r % g.Ws); r /= g.Ws;
    h = (int)(r % g.Hs); r /= g.Hs;
    d = (int)(r % g.Ds);
    b = (int)(r / g.Ds);
  }
  int pch[MP], qch[MQ];
#pragma unroll
  for (int e = 0; e < MP; ++e) pch[e] = (pt * MP + e) * 32 + c8;
#pragma unroll
  for (int a = 0; a < MQ; ++a) qch[a] = (qt * MQ + a) * 32 + c8;

  for (long m0 = m_begin; m0 < m_end; m0 += 16) {
    const long m = m0 + j;
    const bool mok = m < m_end;
    uint4 pv[MP], qv[TH][TW][MQ];
#pragma unroll
    for (int e = 0; e < MP; ++e)
      pv[e] = (mok && pch[e] < g.Cp) ? *reinterpret_cast<const uint4*>(P + (size_t)m * g.Cp + pch[e]) : kZero16;
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
        for (int a = 0; a < MQ; ++a)
          qv[c][t][a] = (ok && qch[a] < g.Cq) ? *reinterpret_cast<const uint4*>(Q + (rowbase + iw) * g.Cq + qch[a]) : kZero16;
      }
    }
    // advance this lane's voxel by 16
    w += 16;
    while (w >= g.Ws) {
      w -= g.Ws;
      if (++h >= g.Hs) { h = 0; if (++d >= g.Ds) { d = 0; ++b; } }
    }
    // park the tiles (lane-contiguous 1 KB each), read them back transposed
#pragma unroll
    for (int e = 0; e < MP; ++e) *reinterpret_cast<uint4*>(my + e * 1024 + lane * 16) = pv[e];
#pragma unroll
    for (int c = 0; c < TH; ++c)
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int a = 0; a < MQ; ++a)
          *reinterpret_cast<uint4*>(my + (MP + (c * TW + t) * MQ + a) * 1024 + lane * 16) = qv[c][t][a];
    uint4 pb[MP];
#pragma unroll
    for (int e = 0; e < MP; ++e) pb[e] = tr_operand(my + e * 1024, trbase);
#pragma unroll
    for (int c = 0; c < TH; ++c)
#pragma unroll
      for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int a = 0; a < MQ; ++a) {
          const uint4 qa = tr_operand(my + (MP + (c * TW + t) * MQ + a) * 1024, trbase);
#pragma unroll
          for (int e = 0; e < MP; ++e) acc[a][e][c][t] = mfma16(qa, pb[e], acc[a][e][c][t]);
        }
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

// ------------------------------------------------------------------------------------------------ LDS-ring weight gradient
// Stride-1 3x3x3 "same" layers on grids with W % 16 == 0 (the 32 -> 32 cost-volume layers, the wide encoder / head layers):
// wgrad16_kernel stages every operand tile of every tap per wave (global -> registers -> LDS -> transposing read).  Here the
// workgroup's four waves share ONE staged copy: the Q rows (the tensor the taps slide over) of an 8-row x 16-voxel tile plus
// halo and the matching P rows arrive by global_load_lds, and every wave takes its operands straight out of that ring with
// ds_read_b64_tr_b16 -- a tap is an LDS address offset.  One MFMA k-step = the 16 voxels of one row.
//   NARROW (Cq, Cp <= 32): three Q planes staged, the 27 taps dealt to the waves (7 / 7 / 7 / 6 accumulators);
//   wide: blockIdx.z = kd (one Q plane staged), blockIdx.y = (32-channel Q tile, group of four 32-channel P tiles), wave = P tile,
//         nine (kh, kw) accumulators per wave.
// A workgroup walks a chunk of tiles with its accumulators live and leaves ONE partial slab entry; wgrad_reduce folds the chunks.
struct WgRingGeom {
  int B, D, H, W, Cp, Cq;
  int nth, ntw, ntiles, tpc, nchunks;
  int ncqt;                        // wide: 32-channel Q tiles (blockIdx.y = cqt + ncqt * cp group)
  int xcd_order;                   // wide: 1-D grid, XCD-aware [chunk][cp group][cqt][kd] order
};

constexpr int kWrRows = 8, kWrQR = kWrRows + 2, kWrQC = 18;
constexpr int kWrQRowB = kWrQC * 64, kWrQPlaneB = kWrQR * kWrQRowB;          // 1152, 11520

__device__ __forceinline__ uint4 tr_operand2(const unsigned char* base, int rowstride, int lane) {
  typedef __attribute__((address_space(3))) i16x4 lds_i16x4;
  const unsigned char* p = base + ((8 * (lane >> 5) + ((lane & 15) >> 2)) * rowstride) + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4*)p);
  const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4*)(p + 4 * rowstride));
  uint4 r;
  r.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
  r.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
  r.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
  r.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
  return r;
}

template <bool NARROW>
__global__ void __launch_bounds__(256)
wgrad_ring16_kernel(const bf16_t* __restrict__ P, const bf16_t* __restrict__ Q, float* __restrict__ ws, WgRingGeom g) {
  constexpr int NKD = NARROW ? 3 : 1;
  constexpr int NACC = NARROW ? 7 : 9;
  constexpr int PVB = NARROW ? 64 : 256;                 // bytes of one staged P voxel line (32 / 128 channels)
  constexpr int QB = NKD * kWrQPlaneB;
  extern __shared__ __align__(16) unsigned char wr[];
  unsigned char* qr = wr;                                // [NKD][10 rows][18 voxels][32 ch]
  unsigned char* pr = wr + QB;                           // [8 rows][16 voxels][PVB]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lk = lane >> 5;
  // wide: 1-D grid in the XCD-aware order [chunk][channel tiles][kd] -- the workgroups that walk the SAME voxel chunk (every one
  // re-reads its P lines, every third its Q plane) are neighbours on one XCD, so the chunk comes from HBM once and from that L2 after
  int chunk = blockIdx.x, cqt = 0, cpg = 0, kdz = 0;
  if (!NARROW && g.xcd_order) {
    const unsigned n = gridDim.x, L = blockIdx.x;
    const unsigned xcd = L & 7, q = n >> 3, r = n & 7;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    unsigned Lp = base + (L >> 3);
    kdz = (int)(Lp % 3u); Lp /= 3u;
    cqt = (int)(Lp % (unsigned)g.ncqt); Lp /= (unsigned)g.ncqt;
    const unsigned ncpg = (unsigned)((g.Cp + 127) / 128);
    cpg = (int)(Lp % ncpg);
    chunk = (int)(Lp / ncpg);
  } else if (!NARROW) {
    cqt = (int)blockIdx.y % g.ncqt; cpg = (int)blockIdx.y / g.ncqt; kdz = (int)blockIdx.z;
  }
  const int cq0 = cqt * 32, cp0 = cpg * 128;
  const int ntap = NARROW ? (wave < 3 ? 7 : 6) : 9;

  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

  const int t_begin = chunk * g.tpc, t_end = min(g.ntiles, t_begin + g.tpc);
  for (int t = t_begin; t < t_end; ++t) {
    int tt = t;
    const int wt = tt % g.ntw; tt /= g.ntw;
    const int ht = tt % g.nth; tt /= g.nth;
    const int d = tt % g.D;
    const int b = tt / g.D;
    const int h0 = ht * kWrRows, w0 = wt * 16;
    __syncthreads();                                     // the previous tile's operands have been read
    // ---- Q: NKD planes x 10 rows x 18 voxels x 4 items of 8 channels
    {
      constexpr int items = NKD * kWrQR * kWrQC * 4;
      for (int e = wave; e * 64 < items; e += 4) {
        const int L = e * 64 + lane;
        if (L < items) {
          const int pl = L / (kWrQR * kWrQC * 4), rem = L % (kWrQR * kWrQC * 4);
          const int row = rem / (kWrQC * 4), it = rem % (kWrQC * 4);
          const int u = it >> 2, c = cq0 + (it & 3) * 8;
          const int ds = d - 1 + (NARROW ? pl : kdz), hs = h0 - 1 + row, wsrc = w0 - 1 + u;
          const bool ok = ds >= 0 && ds < g.D && hs >= 0 && hs < g.H && wsrc >= 0 && wsrc < g.W && c < g.Cq;
          const void* src = ok ? static_cast<const void*>(Q + ((((long)b * g.D + ds) * g.H + hs) * g.W + wsrc) * (long)g.Cq + c)
                               : static_cast<const void*>(&kZero16);
          __builtin_amdgcn_global_load_lds(src, qr + e * 1024, 16, 0, 0);
        }
      }
    }
    // ---- P: 8 rows x 16 voxels x (PVB / 16) items
    {
      constexpr int ipv = PVB / 16, items = kWrRows * 16 * ipv;
      for (int e = wave; e * 64 < items; e += 4) {
        const int L = e * 64 + lane;
        const int row = L / (16 * ipv), rem = L % (16 * ipv);
        const int u = rem / ipv, c = cp0 + (rem % ipv) * 8;
        const int hs = h0 + row;
        const bool ok = hs < g.H && c < g.Cp;
        const void* src = ok ? static_cast<const void*>(P + ((((long)b * g.D + d) * g.H + hs) * g.W + (w0 + u)) * (long)g.Cp + c)
                             : static_cast<const void*>(&kZero16);
        __builtin_amdgcn_global_load_lds(src, pr + e * 1024, 16, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int r = 0; r < kWrRows; ++r) {
      const uint4 bop = tr_operand2(pr + r * (16 * PVB) + (NARROW ? 0 : wave * 64), PVB, lane);
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if (i < ntap) {
          int kd, kh, kw;
          if (NARROW) { const int tp = wave + 4 * i; kd = tp / 9; kh = (tp / 3) % 3; kw = tp % 3; }
          else { kd = 0; kh = i / 3; kw = i % 3; }
          const uint4 aop = tr_operand2(qr + kd * kWrQPlaneB + (r + kh) * kWrQRowB + kw * 64, 64, lane);
          acc[i] = mfma16(aop, bop, acc[i]);
        }
      }
    }
  }

  // partial slab of this chunk: ws[chunk][tap][cq][cp]
  const int cpt = NARROW ? 0 : cpg * 4 + wave;
  const int pc = cpt * 32 + li;
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    if (i >= ntap) continue;
    const int tap = NARROW ? wave + 4 * i : kdz * 9 + i;
    float* dst = ws + (((size_t)chunk * 27 + tap) * g.Cq) * g.Cp;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = cq0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      if (row < g.Cq && pc < g.Cp) dst[(size_t)row * g.Cp + pc] = acc[i][r];
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
bool basic_ok(const ssbev_conv_dims* d) {
  if (!d || d->B <= 0 || d->Cin <= 0 || d->Cout <= 0) return false;
  if (d->kd <= 0 || d->kh <= 0 || d->kw <= 0 || d->kd > 8 || d->kh > 8 || d->kw > 8) return false;
  return true;
}

bool tap16_applicable(const ssbev_conv_dims* d, int mode) {
  static const bool enabled = !(ssbev_tune("SSBEV_TAP16") && atoi(ssbev_tune("SSBEV_TAP16")) == 0);       // A/B hook
  if (!enabled && d->tile_hint != 9) return false;
  if (d->transposed || d->kd != 3 || d->kh != 3 || d->kw != 3 || d->sd != 1 || d->sh != 1 || d->sw != 1) return false;
  if (d->pd != 1 || d->ph != 1 || d->pw != 1 || d->dd != 1 || d->dh != 1 || d->dw != 1) return false;
  if (d->Di != d->Do || d->Hi != d->Ho || d->Wi != d->Wo || d->tile_hint == 8) return false;
  if (d->Cin > 32 || d->Cout > 32 || d->Cin % 8 || d->Cout % 8) return false;
  const int K = mode == 0 ? d->Cin : d->Cout;
  if (K < 16 && d->tile_hint != 9) return false;
  if ((long)d->Ho * d->Wo * K >= (1L << 30)) return false;
  return d->tile_hint == 9 || (long)d->B * d->Do * d->Ho * ((d->Wo + kT16Wseg - 1) / kT16Wseg) >= 1024L * 16;
}

int wide16_ntl(const ssbev_conv_dims* d, int mode) {        // 0 = not applicable, else 32-channel column tiles per wave
  static const bool enabled = !(ssbev_tune("SSBEV_WIDE16") && atoi(ssbev_tune("SSBEV_WIDE16")) == 0);      // A/B hook
  if (!enabled && d->tile_hint != 7) return 0;
  if (d->transposed || d->kd != 3 || d->kh != 3 || d->kw != 3 || d->sd != 1 || d->sh != 1 || d->sw != 1) return 0;
  if (d->pd != 1 || d->ph != 1 || d->pw != 1 || d->dd != 1 || d->dh != 1 || d->dw != 1) return 0;
  if (d->Di != d->Do || d->Hi != d->Ho || d->Wi != d->Wo || d->tile_hint == 8) return 0;
  const int K = mode == 0 ? d->Cin : d->Cout, N = mode == 0 ? d->Cout : d->Cin;
  if (K % 32 != 0 || (N % 64 != 0 && N % 96 != 0) || K < 64 || d->Wo % 16 != 0) return 0;
  if ((long)d->Ho * d->Wo * K >= (1L << 29)) return 0;
  // worth it when the tiles fill the chip (512 resident workgroups); tile_hint 7 forces it on small problems (tests)
  const int ntl = N % 128 == 0 ? 4 : (N % 96 == 0 ? 3 : 2);      // (six tiles of 32: 256 VGPRs + spills; 192 = 2 x 96)
  const long tiles = (long)d->B * d->Do * ((d->Ho + kW16Rows - 1) / kW16Rows) * (d->Wo / 16) * (N / (32 * ntl));
  if (d->tile_hint != 7 && tiles < 512) return 0;
  return ntl;
}

template <int NTL, typename YT>
int launch_wide16(const bf16_t* x, const bf16_t* wp, const float* bias, YT* y, const ssbev_conv_dims* d, int mode, hipStream_t st) {
  Wide16Geom g;
  g.B = d->B; g.D = d->Do; g.H = d->Ho; g.W = d->Wo;
  g.K = mode == 0 ? d->Cin : d->Cout;
  g.N = mode == 0 ? d->Cout : d->Cin;
  g.KP = pad16(g.K) / 16; g.NPad = pad32(g.N);
  g.nth = (g.H + kW16Rows - 1) / kW16Rows; g.ntw = g.W / 16; g.ntn = g.N / (32 * NTL);
  g.relu = mode == 0 ? d->relu : 0;
  g.accumulate = d->accumulate;
  g.mirror = mode == 1 ? 1 : 0;
  auto kern = conv_wide16_kernel<NTL, YT>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kW16RingB) != hipSuccess)
    return SSBEV_ELAUNCH;
  const long blocks = (long)g.B * g.D * g.nth * g.ntw * g.ntn;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), kW16RingB, st, x, wp, mode == 0 ? bias : nullptr, y, g);
  return ssbev_launch_status();
}

Geom16 make_geom(const ssbev_conv_dims* d, int mode) {
  Geom16 g;
  g.B = d->B;
  if (mode == 0) {
    g.Cin = d->Cin; g.Cout = d->Cout;
    g.Di = d->Di; g.Hi = d->Hi; g.Wi = d->Wi; g.Do = d->Do; g.Ho = d->Ho; g.Wo = d->Wo;
    g.form = d->transposed ? 1 : 0; g.relu = d->relu;
  } else {     // data gradient: roles swapped, the gradient of a conv gathers like a transposed conv and vice versa
    g.Cin = d->Cout; g.Cout = d->Cin;
    g.Di = d->Do; g.Hi = d->Ho; g.Wi = d->Wo; g.Do = d->Di; g.Ho = d->Hi; g.Wo = d->Wi;
    g.form = d->transposed ? 0 : 1; g.relu = 0;
  }
  g.KP = pad16(g.Cin) / 16; g.CoutPad = pad32(g.Cout);
  g.kd = d->kd; g.kh = d->kh; g.kw = d->kw; g.sd = d->sd; g.sh = d->sh; g.sw = d->sw;
  g.pd = d->pd; g.ph = d->ph; g.pw = d->pw; g.dd = d->dd; g.dh = d->dh; g.dw = d->dw;
  g.accumulate = d->accumulate;
  return g;
}

long gather_blocks(const Geom16& g, int MT, int NT) {
  long M = (long)g.B * g.Do * g.Ho * g.Wo;
  long classes = 1;
  if (g.form == 1) {
    classes = (long)g.sd * g.sh * g.sw;
    M = (long)g.B * ((g.Do + g.sd - 1) / g.sd) * ((g.Ho + g.sh - 1) / g.sh) * ((g.Wo + g.sw - 1) / g.sw);
  }
  return ((M + 4 * MT * 32 - 1) / (4 * MT * 32)) * ((g.Cout + NT * 32 - 1) / (NT * 32)) * classes;
}

template <int MT, int NT, typename YT>
int launch_gather16(const bf16_t* x, const bf16_t* wp, const float* bias, YT* y, const Geom16& g, hipStream_t st) {
  long Mtot = (long)g.B * g.Do * g.Ho * g.Wo;
  int classes = 1;
  if (g.form == 1) {
    classes = g.sd * g.sh * g.sw;
    Mtot = (long)g.B * ((g.Do + g.sd - 1) / g.sd) * ((g.Ho + g.sh - 1) / g.sh) * ((g.Wo + g.sw - 1) / g.sw);
  }
  dim3 grid(cdiv(Mtot, 4 * MT * 32), cdiv(g.Cout, NT * 32), classes), block(256);
  hipLaunchKernelGGL((conv_gather16_kernel<MT, NT, YT>), grid, block, 0, st, x, wp, bias, y, g);
  return ssbev_launch_status();
}

// ------------------------------------------------------------------------------------------------ LDS-staged implicit GEMM
// conv_igemm16_kernel (round 5): conv_igemm_kernel (conv_mfma.hip) for bf16 tensors -- the strided / transposed / dilated /
// pointwise layers that conv_gather16_kernel served at 4-10 % of the bf16 matrix peak (95-250 TF/s: every lane fetching its
// voxel's 16 bytes from L1/L2 for each MFMA, one MFMA of work per load).  A workgroup owns a BM x BN tile; a stage = one tap x
// BKC source channels: the BM gathered rows (BKC bf16 = 128 or 64 bytes each) and the BKC x BN weight block go global -> LDS by
// global_load_lds_dwordx4, double buffered; both fragments are ONE ds_read_b128 per v_mfma_f32_32x32x16_bf16 (pack16_kernel's
// [tap][p][lk][n][8] layout is the B fragment layout).  Row table, tap masks, parity classes as in the fp32 kernel.
template <int WN, int MW, int WGN, int BKC, typename YT>
__global__ void __launch_bounds__(256, 2)
conv_igemm16_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp, const float* __restrict__ bias,
                    YT* __restrict__ y, Geom16 g, int mblocks, int nblocks) {
  constexpr int WGM = 4 / WGN;
  constexpr int BM = 32 * MW * WGM, BN = 32 * WN * WGN;
  constexpr int SPR = BKC / 8;                                    // 16-byte units per gathered row
  constexpr int KPS = BKC / 16;                                   // MFMA k-steps (16 channels) per stage
  constexpr int AU = BM * SPR, BU = KPS * 2 * BN, SU = AU + BU;   // 16-byte units per stage
  constexpr int AI = AU / 64, BI = BU / 64;                       // 1 KiB LDS-DMA instructions per stage
  constexpr int AE = (AI + 3) / 4, BE = (BI + 3) / 4;
  static_assert(AU % 64 == 0 && BU % 64 == 0, "slabs are whole 1 KiB pieces");
  extern __shared__ __align__(16) uint4 lds16[];                  // [2][A slab | B slab] | row table
  long* t_src = reinterpret_cast<long*>(lds16 + 2 * SU);
  long* t_dst = t_src + BM;
  unsigned* t_msk = reinterpret_cast<unsigned*>(t_dst + BM);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lk = lane >> 5;
  const int wm = wave / WGN, wn = wave % WGN;
  auto swz = [](int row) { return SPR == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3); };

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
  int mb, nb, bt;
  {
    const unsigned n = gridDim.x, L = blockIdx.x;
    const unsigned xcd = L & 7, q = n >> 3, r = n & 7;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    unsigned Lp = base + (L >> 3);
    nb = (int)(Lp % (unsigned)nblocks); Lp /= (unsigned)nblocks;
    mb = (int)(Lp % (unsigned)mblocks);
    bt = (int)(Lp / (unsigned)mblocks);
  }
  const long m0 = (long)mb * BM;
  const int n0 = nb * BN;
  if (m0 >= Mtot) return;
  if (g.form == 0) {                   // batched plain products: the batch element is part of the 1-D XCD-aware order, so that the
    x += (long)bt * g.bsx; y += (long)bt * g.bsy; wp += (long)bt * g.bsw;   // tiles of one element meet in one L2 (bt = 0 for a convolution)
  }

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
    t_src[tid] = ((((long)ob * g.Di + bd) * g.Hi + bh) * g.Wi + bw) * (long)g.Cin;
    long dst;
    if (g.form == 0) dst = (((long)ob * g.Do + od) * g.Ho + oh) * g.Wo + ow;
    else dst = (((long)ob * g.Do + (od * g.sd + par_d)) * g.Ho + (oh * g.sh + par_h)) * g.Wo + (ow * g.sw + par_w);
    t_dst[tid] = ok ? dst * (long)g.Cout : -1;
  }
  __syncthreads();

  const bf16_t* ap[AE]; unsigned am[AE];
#pragma unroll
  for (int e = 0; e < AE; ++e) {
    const int item = (wave + 4 * e) * 64 + lane, row = item / SPR, slot = item % SPR;
    const bool on = wave + 4 * e < AI;
    ap[e] = x + (on ? t_src[row] : 0) + ((slot ^ swz(row)) << 3);
    am[e] = on ? t_msk[row] : 0u;
  }
  // B slab: KPS * 2 rows (p, lk) of BN columns x 8 bf16, contiguous per row in the packed weights; piece j covers units [64 j, 64 j + 64)
  long boff[BE];
#pragma unroll
  for (int e = 0; e < BE; ++e) {
    const int j = wave + 4 * e, u = j * 64 + lane;
    const int r = u / BN, col = u % BN;
    boff[e] = n0 + col < g.CoutPad ? ((long)r * g.CoutPad + n0 + col) * 8 : -1;      // columns past the padded weights: zeros
  }
  const int cq_n = g.Cin / BKC;
  const int nst = ntaps * cq_n;
  int i_ia = 0, i_ib = 0, i_ic = 0, i_cq = 0;
  auto issue = [&](int buf) {
    const int tap = ((kd0 + i_ia * kds) * g.kh + (kh0 + i_ib * khs)) * g.kw + (kw0 + i_ic * kws);
    const long toff = (((long)i_ia * step_d * g.Hi + (long)i_ib * step_h) * g.Wi + (long)i_ic * step_w) * g.Cin + i_cq * BKC;
    const unsigned need = (1u << i_ia) | (1u << (8 + i_ib)) | (1u << (16 + i_ic));
#pragma unroll
    for (int e = 0; e < AE; ++e) {
      if (AI % 4 != 0 && wave + 4 * e >= AI) break;
      const void* asrc = ((am[e] & need) == need) ? static_cast<const void*>(ap[e] + toff) : static_cast<const void*>(&kZero16);
      __builtin_amdgcn_global_load_lds(asrc, lds16 + buf * SU + (wave + 4 * e) * 64, 16, 0, 0);
    }
    const bf16_t* wb = wp + ((size_t)tap * g.KP + (size_t)i_cq * KPS) * 2 * g.CoutPad * 8;
#pragma unroll
    for (int e = 0; e < BE; ++e) {
      if (BI % 4 != 0 && wave + 4 * e >= BI) break;
      const void* bsrc = boff[e] >= 0 ? static_cast<const void*>(wb + boff[e]) : static_cast<const void*>(&kZero16);
      __builtin_amdgcn_global_load_lds(bsrc, lds16 + buf * SU + AU + (wave + 4 * e) * 64, 16, 0, 0);
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

  if (nst > 0) issue(0);
  for (int st = 0; st < nst; ++st) {
    const int buf = st & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (st + 1 < nst) issue(buf ^ 1);
    const uint4* as = lds16 + buf * SU;
    const uint4* bs = as + AU;
    uint4 ac[MW], bc[WN], an[MW], bn[WN];
    auto fetch = [&](int ks, uint4 (&a)[MW], uint4 (&bf)[WN]) {
#pragma unroll
      for (int mt = 0; mt < MW; ++mt) {
        const int row = (wm * MW + mt) * 32 + li;
        a[mt] = as[row * SPR + ((2 * ks + lk) ^ swz(row))];
      }
#pragma unroll
      for (int nt = 0; nt < WN; ++nt) bf[nt] = bs[(2 * ks + lk) * BN + (wn * WN + nt) * 32 + li];
    };
    fetch(0, ac, bc);
#pragma unroll
    for (int ks = 0; ks < KPS; ++ks) {
      if (ks + 1 < KPS) fetch(ks + 1, an, bn);
#pragma unroll
      for (int mt = 0; mt < MW; ++mt)
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) acc[mt][nt] = mfma16(ac[mt], bc[nt], acc[mt][nt]);
#pragma unroll
      for (int mt = 0; mt < MW; ++mt) ac[mt] = an[mt];
#pragma unroll
      for (int nt = 0; nt < WN; ++nt) bc[nt] = bn[nt];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

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
            oldv[r][nt] = (rowbase[mt][r0 + r] >= 0 && co < g.Cout) ? ld1(y + rowbase[mt][r0 + r] + co) : 0.0f;
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
          st1(y + rowbase[mt][r] + co, v);
        }
      }
  }
}

// SSBEV_IGEMM16=0 keeps conv_gather16_kernel everywhere.  Source channels a multiple of 32, >= 64 destination channels, enough rows.
bool conv_igemm16_applicable(const Geom16& g, int hint) {
  const char* env = ssbev_env("SSBEV_IGEMM16");
  if ((env && atoi(env) == 0) || hint) return false;
  const char* mc = ssbev_tune("SSBEV_IGEMM16_MIN_COUT");             // A/B hook: 32 = also the layers with 32 destination channels
  const int min_cout = mc ? atoi(mc) : 64;
  if (g.Cin % 32 != 0 || g.Cout < min_cout || g.Cout % 8 != 0) return false;
  if (g.kd > 8 || g.kh > 8 || g.kw > 8) return false;
  return (long)g.B * g.Do * g.Ho * g.Wo >= 2048;
}

template <int WN, int MW, int WGN, int BKC, typename YT>
int launch_igemm16_t(const bf16_t* x, const bf16_t* wp, const float* bias, YT* y, const Geom16& g, hipStream_t st, int nbatch = 1) {
  constexpr int BM = 32 * MW * (4 / WGN), BN = 32 * WN * WGN;
  long Mtot = (long)g.B * g.Do * g.Ho * g.Wo;
  int classes = nbatch;
  if (g.form == 1) {
    classes = g.sd * g.sh * g.sw;
    Mtot = (long)g.B * ((g.Do + g.sd - 1) / g.sd) * ((g.Ho + g.sh - 1) / g.sh) * ((g.Wo + g.sw - 1) / g.sw);
  }
  const int mblocks = (int)((Mtot + BM - 1) / BM), nblocks = (g.Cout + BN - 1) / BN;
  const size_t lds = (size_t)2 * (BM * (BKC / 8) + (BKC / 16) * 2 * BN) * 16 + (size_t)BM * (2 * sizeof(long) + sizeof(unsigned));
  auto kern = conv_igemm16_kernel<WN, MW, WGN, BKC, YT>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return SSBEV_ELAUNCH;
  if (g.form == 0)
    hipLaunchKernelGGL(kern, dim3((unsigned)(mblocks * nblocks * classes)), dim3(256), lds, st, x, wp, bias, y, g, mblocks, nblocks);
  else
    hipLaunchKernelGGL(kern, dim3((unsigned)(mblocks * nblocks), 1, classes), dim3(256), lds, st, x, wp, bias, y, g, mblocks, nblocks);
  return ssbev_launch_status();
}

template <typename YT>
int launch_igemm16(const bf16_t* x, const bf16_t* wp, const float* bias, YT* y, const Geom16& g, hipStream_t st, int nbatch = 1) {
  long Mtot = (long)g.B * g.Do * g.Ho * g.Wo;
  long classes = nbatch;
  if (g.form == 1) {
    classes = (long)g.sd * g.sh * g.sw;
    Mtot = (long)g.B * ((g.Do + g.sd - 1) / g.sd) * ((g.Ho + g.sh - 1) / g.sh) * ((g.Wo + g.sw - 1) / g.sw);
  }
  const int force = ssbev_tune("SSBEV_IGEMM_TILE") ? atoi(ssbev_tune("SSBEV_IGEMM_TILE")) : 0;             // probing: bm * 1000 + bn
  const bool wide = g.CoutPad % 128 == 0;
  const long need = g.form == 1 ? 1024 : 512;
  const int cand[3][2] = {{128, wide ? 128 : 64}, {64, wide ? 128 : 64}, {64, 64}};
  int bm = 64, bn = 64;
  for (int i = 0; i < 3; ++i) {
    const long blocks = ((Mtot + cand[i][0] - 1) / cand[i][0]) * ((g.Cout + cand[i][1] - 1) / cand[i][1]) * classes;
    if (blocks >= need) { bm = cand[i][0]; bn = cand[i][1]; break; }
  }
  if (g.CoutPad == 32) { bm = 256; bn = 32; }          // 32 destination channels: four waves stacked along M, 64 rows each
  if (force) { bm = force / 1000; bn = force % 1000; }
  static const int bkc_force = ssbev_tune("SSBEV_IGEMM16_BKC") ? atoi(ssbev_tune("SSBEV_IGEMM16_BKC")) : 0;    // probing: 32 = 32-channel stages
  const bool k64 = g.Cin % 64 == 0 && bkc_force != 32;
#define SSBEV_IG16(WN_, MW_) (k64 ? launch_igemm16_t<WN_, MW_, 2, 64, YT>(x, wp, bias, y, g, st, nbatch) \
                                  : launch_igemm16_t<WN_, MW_, 2, 32, YT>(x, wp, bias, y, g, st, nbatch))
  if (bm == 128 && bn == 128) return SSBEV_IG16(2, 2);
  if (bm == 128 && bn == 64) return SSBEV_IG16(1, 2);
  if (bm == 64 && bn == 128) return SSBEV_IG16(2, 1);
  if (bm == 64 && bn == 64) return SSBEV_IG16(1, 1);
  if (bm == 256 && bn == 32)
    return k64 ? launch_igemm16_t<1, 2, 1, 64, YT>(x, wp, bias, y, g, st, nbatch) : launch_igemm16_t<1, 2, 1, 32, YT>(x, wp, bias, y, g, st, nbatch);
  if (bm == 128 && bn == 32)
    return k64 ? launch_igemm16_t<1, 1, 1, 64, YT>(x, wp, bias, y, g, st, nbatch) : launch_igemm16_t<1, 1, 1, 32, YT>(x, wp, bias, y, g, st, nbatch);
#undef SSBEV_IG16
  return SSBEV_EINVAL;
}

template <typename YT>
int dispatch_gather16(const bf16_t* x, const bf16_t* wp, const float* bias, YT* y, const Geom16& g, int hint, hipStream_t st) {
  if (conv_igemm16_applicable(g, hint)) return launch_igemm16(x, wp, bias, y, g, st);
  int mt, nt;
  if (hint >= 10) { mt = hint / 10; nt = hint % 10; }
  else {
    // the fp32 kernel's measured choices (dispatch_gather, conv_mfma.hip): wide column tiles when Cout allows and the grid
    // still fills the chip, two row tiles unless that leaves too few workgroups
    nt = (g.Cout % 192 == 0 && gather_blocks(g, 2, 6) >= 256) ? 6 : (g.Cout % 128 == 0) ? 4 : (g.Cout > 32 ? 2 : 1);
    if (nt == 4 && gather_blocks(g, 2, 4) < 256) nt = 2;
    mt = gather_blocks(g, 2, nt) >= 160 ? 2 : 1;
    if (g.form == 1 && g.sd * g.sh * g.sw > 1) { mt = 1; nt = nt > 2 ? 2 : nt; }     // parity classes: short loops, small tiles
  }
  switch (mt * 10 + nt) {
    case 11: return launch_gather16<1, 1, YT>(x, wp, bias, y, g, st);
    case 12: return launch_gather16<1, 2, YT>(x, wp, bias, y, g, st);
    case 14: return launch_gather16<1, 4, YT>(x, wp, bias, y, g, st);
    case 21: return launch_gather16<2, 1, YT>(x, wp, bias, y, g, st);
    case 22: return launch_gather16<2, 2, YT>(x, wp, bias, y, g, st);
    case 24: return launch_gather16<2, 4, YT>(x, wp, bias, y, g, st);
    case 26: return launch_gather16<2, 6, YT>(x, wp, bias, y, g, st);
    default: return SSBEV_EINVAL;
  }
}

template <typename YT>
int launch_tap16(const bf16_t* x, const float* wp, const float* bias, YT* y, const ssbev_conv_dims* d, int mode, hipStream_t st) {
  Tap16Geom g;
  g.B = d->B; g.D = d->Do; g.H = d->Ho; g.W = d->Wo;
  g.K = mode == 0 ? d->Cin : d->Cout;
  g.N = mode == 0 ? d->Cout : d->Cin;
  g.nseg = (g.W + kT16Wseg - 1) / kT16Wseg;
  g.NG = g.B * g.D * g.H;
  g.relu = mode == 0 ? d->relu : 0;
  g.has_bias = (mode == 0 && bias) ? 1 : 0;
  g.accumulate = d->accumulate;
  // three workgroups per CU (42.5 KB of LDS each): whole rounds of 768 workgroups, >= 24 rows each
  long nranges = 768 / g.nseg;
  for (long rounds = 8; rounds >= 1; --rounds) {
    const long nr = (768 * rounds) / g.nseg;
    if (nr >= 1 && (g.NG + nr - 1) / nr >= 24) { nranges = nr; break; }
  }
  if (const char* e = ssbev_tune("SSBEV_TAP16_RANGES")) { const long v = atol(e); if (v > 0) nranges = v; }   // tuning hook
  if (nranges < 1) nranges = 1;
  g.gpc = (int)((g.NG + nranges - 1) / nranges);
  nranges = (g.NG + g.gpc - 1) / g.gpc;
  auto kern = conv_tap16_kernel<YT>;
  hipLaunchKernelGGL(kern, dim3((unsigned)(nranges * g.nseg)), dim3(256), kT16LdsBytes, st, x, reinterpret_cast<const uint4*>(wp),
                     bias, y, g);
  return ssbev_launch_status();
}

struct WgCfg16 { int MQ, MP, TH, TW; };

WgCfg16 wg_cfg(int Cp, int Cq, int kh, int kw) {
  if (kh * kw == 1) return (Cp <= 32 && Cq <= 32) ? WgCfg16{1, 1, 1, 1} : WgCfg16{2, 2, 1, 1};
  if (Cp <= 32 && Cq <= 32) return {1, 1, 3, 3};
  return {2, 2, 1, 3};
}

Wg16Geom make_wg_geom(const ssbev_conv_dims* d) {
  Wg16Geom g;
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
  const WgCfg16 c = wg_cfg(g.Cp, g.Cq, g.kh, g.kw);
  const long tiles = (long)cdiv(g.Cp, 32 * c.MP) * cdiv(g.Cq, 32 * c.MQ) * g.kd * cdiv(g.kh, c.TH) * cdiv(g.kw, c.TW);
  // ~4096 wave-tasks in total (2 waves per SIMD x 2 rounds), chunks of at least 256 voxels, partial slabs bounded to 256 MB
  long want = std::max(1L, 4096 / std::max(1L, tiles));
  if (const char* e = ssbev_tune("SSBEV_WG16_WAVES")) { const long v = atol(e); if (v > 0) want = std::max(1L, v / std::max(1L, tiles)); }
  const long slab = (long)g.kd * g.kh * g.kw * g.Cp * g.Cq * 4;
  want = std::min(want, std::max(1L, (256L << 20) / slab));
  long chunk = (Mtot + want - 1) / want;
  if (chunk < 256) chunk = 256;
  chunk = (chunk + 15) & ~15L;
  g.chunk = (int)chunk;
  g.nchunks = (int)((Mtot + chunk - 1) / chunk);
  static const int xcd_order = ssbev_tune("SSBEV_WGRAD16_XCD") ? atoi(ssbev_tune("SSBEV_WGRAD16_XCD")) : 1;
  g.xcd_order = xcd_order;
  return g;
}

template <int MQ, int MP, int TH, int TW>
int launch_wg16(const bf16_t* P, const bf16_t* Q, float* ws, const Wg16Geom& g, hipStream_t st) {
  const int ytiles = cdiv(g.Cq, 32 * MQ) * cdiv(g.Cp, 32 * MP);
  dim3 grid((unsigned)((long)cdiv(g.nchunks, 4) * ytiles * (g.kd * cdiv(g.kh, TH) * cdiv(g.kw, TW))));
  const size_t lds = (size_t)4 * (MP + MQ * TH * TW) * 1024;
  hipLaunchKernelGGL((wgrad16_kernel<MQ, MP, TH, TW>), grid, dim3(256), lds, st, P, Q, ws, g);
  return ssbev_launch_status();
}

bool wg_ring_applicable(const ssbev_conv_dims* d) {
  static const bool enabled = !(ssbev_tune("SSBEV_WGRING16") && atoi(ssbev_tune("SSBEV_WGRING16")) == 0);    // A/B hook
  if (!enabled && d->tile_hint != 7 && d->tile_hint != 9) return false;
  if (d->transposed || d->kd != 3 || d->kh != 3 || d->kw != 3 || d->sd != 1 || d->sh != 1 || d->sw != 1) return false;
  if (d->pd != 1 || d->ph != 1 || d->pw != 1 || d->dd != 1 || d->dh != 1 || d->dw != 1) return false;
  if (d->Di != d->Do || d->Hi != d->Ho || d->Wi != d->Wo || d->tile_hint == 8) return false;
  if (d->Wo % 16 != 0 || d->Cin % 8 != 0 || d->Cout % 8 != 0 || d->Cin < 16 || d->Cout < 16) return false;
  const long tiles = (long)d->B * d->Do * ((d->Ho + kWrRows - 1) / kWrRows) * (d->Wo / 16);
  return d->tile_hint == 7 || d->tile_hint == 9 || tiles >= 2048;
}

WgRingGeom make_wr_geom(const ssbev_conv_dims* d) {
  WgRingGeom g;
  g.B = d->B; g.D = d->Do; g.H = d->Ho; g.W = d->Wo; g.Cp = d->Cout; g.Cq = d->Cin;
  g.nth = (g.H + kWrRows - 1) / kWrRows; g.ntw = g.W / 16;
  g.ntiles = g.B * g.D * g.nth * g.ntw;
  const bool narrow = g.Cq <= 32 && g.Cp <= 32;
  g.ncqt = narrow ? 1 : (g.Cq + 31) / 32;
  g.xcd_order = 0;
  const long types = narrow ? 1 : (long)g.ncqt * ((g.Cp + 127) / 128) * 3;
  // ~1024 workgroups in all (two resident per CU, two rounds), at least 4 tiles per chunk, partial slabs bounded to 256 MB
  long want = std::max(1L, 1024 / types);
  if (const char* e = ssbev_tune("SSBEV_WGRING16_WGS")) { const long v = atol(e); if (v > 0) want = std::max(1L, v / types); }
  const long slab = 27L * g.Cq * g.Cp * 4;
  want = std::min(want, std::max(1L, (256L << 20) / slab));
  long tpc = (g.ntiles + want - 1) / want;
  if (tpc < 4) tpc = std::min<long>(4, g.ntiles);
  g.tpc = (int)tpc;
  g.nchunks = (int)((g.ntiles + tpc - 1) / tpc);
  return g;
}

int launch_wg_ring(const bf16_t* P, const bf16_t* Q, float* ws, WgRingGeom g, hipStream_t st) {
  const bool narrow = g.Cq <= 32 && g.Cp <= 32;
  if (narrow) {
    const size_t lds = 3 * kWrQPlaneB + kWrRows * 16 * 64;
    hipLaunchKernelGGL(wgrad_ring16_kernel<true>, dim3(g.nchunks), dim3(256), lds, st, P, Q, ws, g);
  } else {
    const size_t lds = kWrQPlaneB + kWrRows * 16 * 256;
    static const int xcd_order = ssbev_tune("SSBEV_WGRAD_RING_XCD") ? atoi(ssbev_tune("SSBEV_WGRAD_RING_XCD")) : 1;
    g.xcd_order = xcd_order;
    if (xcd_order)
      hipLaunchKernelGGL(wgrad_ring16_kernel<false>, dim3(g.nchunks * g.ncqt * ((g.Cp + 127) / 128) * 3), dim3(256), lds, st, P, Q, ws, g);
    else
      hipLaunchKernelGGL(wgrad_ring16_kernel<false>, dim3(g.nchunks, g.ncqt * ((g.Cp + 127) / 128), 3), dim3(256), lds, st, P, Q, ws, g);
  }
  return ssbev_launch_status();
}

}  // namespace

namespace ssbev_bf16 {

bool storage_mode(const ssbev_conv_dims* d) { return d && (d->precision == 2 || d->precision == 3); }

bool dims_ok(const ssbev_conv_dims* d, int mode) {
  if (!basic_ok(d) || !storage_mode(d)) return false;
  if (d->Cin % 8 != 0) return false;                        // 16-byte voxel-line pieces of the bf16 source tensor
  if (mode != 0 && d->Cout % 8 != 0) return false;          // ... and the gradient tensor is a source in modes 1 / 2
  if (mode == 2 && d->precision != 2) return false;
  if ((long)d->B * d->Do * d->Ho * d->Wo >= (1L << 31) || (long)d->B * d->Di * d->Hi * d->Wi >= (1L << 31)) return false;
  return true;
}

int kernel_class(const ssbev_conv_dims* d, int mode) {
  if (mode == 2) return wg_ring_applicable(d) ? 20 : 18;
  if (wide16_ntl(d, mode)) return 19;
  if (tap16_applicable(d, mode)) return 17;
  return conv_igemm16_applicable(make_geom(d, mode), d->tile_hint >= 10 ? d->tile_hint : 0) ? 21 : 16;
}

size_t packed_elems(const ssbev_conv_dims* d) {
  if (!basic_ok(d)) return 0;
  const size_t taps = (size_t)d->kd * d->kh * d->kw;
  const size_t a = (size_t)pad16(d->Cin) * pad32(d->Cout), b = (size_t)pad16(d->Cout) * pad32(d->Cin);
  const size_t generic_bf16 = taps * (a > b ? a : b);
  const size_t bf = std::max(generic_bf16, (size_t)kT16PackedU4 * 8);
  return (bf + 1) / 2 + 8;                                  // bf16 elements -> floats
}

int pack(const float* w_src, float* w_packed, const ssbev_conv_dims* d, int mode, hipStream_t st) {
  if (!dims_ok(d, mode) || !w_src || !w_packed || (mode != 0 && mode != 1)) return SSBEV_EINVAL;
  bf16_t* dst = reinterpret_cast<bf16_t*>(w_packed);
  if (tap16_applicable(d, mode)) {
    hipLaunchKernelGGL(pack_tap16_kernel, dim3(cdiv(kT16PackedU4 * 8, 256)), dim3(256), 0, st, w_src, dst, d->Cout, d->Cin, mode);
    return ssbev_launch_status();
  }
  const int taps = d->kd * d->kh * d->kw;
  const int K = mode == 0 ? d->Cin : d->Cout, N = mode == 0 ? d->Cout : d->Cin;
  // (the wide kernel takes the generic layout too: its data gradient swaps the channel roles here and mirrors the taps itself)
  const int layout = (mode == 0) == (d->transposed == 0) ? 0 : 1;
  const int KP = pad16(K) / 16, NPad = pad32(N);
  const long total = (long)taps * KP * 2 * NPad * 8;
  hipLaunchKernelGGL(pack16_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w_src, dst, K, N, KP, NPad, taps, layout, total);
  return ssbev_launch_status();
}

static int run(const void* src, const float* wp, const float* bias, void* dst, const ssbev_conv_dims* d, int mode, hipStream_t st) {
  const bf16_t* x = static_cast<const bf16_t*>(src);
  const bool y16 = d->precision == 2;
  if (tap16_applicable(d, mode))
    return y16 ? launch_tap16(x, wp, bias, static_cast<bf16_t*>(dst), d, mode, st)
               : launch_tap16(x, wp, bias, static_cast<float*>(dst), d, mode, st);
  const bf16_t* w16 = reinterpret_cast<const bf16_t*>(wp);
  if (const int ntl = wide16_ntl(d, mode)) {
#define SSBEV_W16(NTL_) (y16 ? launch_wide16<NTL_>(x, w16, bias, static_cast<bf16_t*>(dst), d, mode, st) \
                             : launch_wide16<NTL_>(x, w16, bias, static_cast<float*>(dst), d, mode, st))
    return ntl == 4 ? SSBEV_W16(4) : (ntl == 3 ? SSBEV_W16(3) : SSBEV_W16(2));
#undef SSBEV_W16
  }
  const Geom16 g = make_geom(d, mode);
  const int hint = d->tile_hint >= 10 ? d->tile_hint : 0;
  return y16 ? dispatch_gather16(x, w16, bias, static_cast<bf16_t*>(dst), g, hint, st)
             : dispatch_gather16(x, w16, bias, static_cast<float*>(dst), g, hint, st);
}

int forward(const void* x, const float* wp, const float* bias, void* y, const ssbev_conv_dims* d, hipStream_t st) {
  if (!dims_ok(d, 0) || !x || !wp || !y) return SSBEV_EINVAL;
  return run(x, wp, bias, y, d, 0, st);
}

int backward_data(const void* gy, const float* wp, void* gx, const ssbev_conv_dims* d, hipStream_t st) {
  if (!dims_ok(d, 1) || !gy || !wp || !gx) return SSBEV_EINVAL;
  return run(gy, wp, nullptr, gx, d, 1, st);
}

size_t wgrad_workspace(const ssbev_conv_dims* d) {
  if (!dims_ok(d, 2)) return 0;
  if (wg_ring_applicable(d)) {
    const WgRingGeom r = make_wr_geom(d);
    return (size_t)r.nchunks * 27 * r.Cq * r.Cp * sizeof(float);
  }
  const Wg16Geom g = make_wg_geom(d);
  return (size_t)g.nchunks * d->kd * d->kh * d->kw * g.Cp * g.Cq * sizeof(float);
}

int backward_weight(const void* x, const void* gy, float* gw, const ssbev_conv_dims* d, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!dims_ok(d, 2) || !x || !gy || !gw || !ws) return SSBEV_EINVAL;
  if (ws_bytes < wgrad_workspace(d)) return SSBEV_EWORKSPACE;
  float* partial = static_cast<float*>(ws);
  if (wg_ring_applicable(d)) {
    const WgRingGeom r = make_wr_geom(d);
    const int rc = launch_wg_ring(static_cast<const bf16_t*>(gy), static_cast<const bf16_t*>(x), partial, r, st);
    if (rc != SSBEV_OK) return rc;
    ssbev_detail::wgrad_reduce(partial, gw, r.nchunks, 27, r.Cq, r.Cp, st);
    return ssbev_launch_status();
  }
  const Wg16Geom g = make_wg_geom(d);
  const bf16_t* P = static_cast<const bf16_t*>(d->transposed ? x : gy);
  const bf16_t* Q = static_cast<const bf16_t*>(d->transposed ? gy : x);
  const WgCfg16 c = wg_cfg(g.Cp, g.Cq, g.kh, g.kw);
  int rc;
  if (c.TH == 3) rc = launch_wg16<1, 1, 3, 3>(P, Q, partial, g, st);
  else if (c.TW == 3) rc = launch_wg16<2, 2, 1, 3>(P, Q, partial, g, st);
  else if (c.MQ == 2) rc = launch_wg16<2, 2, 1, 1>(P, Q, partial, g, st);
  else rc = launch_wg16<1, 1, 1, 1>(P, Q, partial, g, st);
  if (rc != SSBEV_OK) return rc;
  ssbev_detail::wgrad_reduce(partial, gw, g.nchunks, g.kd * g.kh * g.kw, g.Cq, g.Cp, st);
  return ssbev_launch_status();
}

// ---- batched plain products on conv_igemm16_kernel (the Winograd frequency products of the bf16 storage mode) ----
__global__ void pack_gemm16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int K, int N, int KP, int NPad, long total) {
  // dst [batch][p][lk][n][t] = B[batch][k = 16 p + 8 lk + t][n]  (the [tap][p][lk][n][8] layout of pack16_kernel, batch for tap)
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  long r = i;
  const int t = (int)(r & 7); r >>= 3;
  const int n = (int)(r % NPad); r /= NPad;
  const int lk = (int)(r & 1); r >>= 1;
  const int p = (int)(r % KP);
  const long bt = r / KP;
  const int k = 16 * p + 8 * lk + t;
  dst[i] = f2bf((k < K && n < N) ? src[(bt * K + k) * N + n] : 0.0f);
}

bool gemm16_ok(const ssbev_gemm16_dims* d) {
  return d && d->M > 0 && d->N > 0 && d->K > 0 && d->batch > 0 && d->K % 32 == 0 && d->N % 8 == 0 && d->batch < 65536 &&
         (long)d->M < (1L << 31);
}

size_t gemm16_packed_elems(const ssbev_gemm16_dims* d) {
  return gemm16_ok(d) ? (size_t)d->batch * (pad16(d->K) / 16) * 2 * pad32(d->N) * 8 : 0;
}

int gemm16_pack(const float* B, bf16_t* packed, const ssbev_gemm16_dims* d, hipStream_t st) {
  if (!gemm16_ok(d) || !B || !packed) return SSBEV_EINVAL;
  const long total = (long)gemm16_packed_elems(d);
  hipLaunchKernelGGL(pack_gemm16_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, B, packed, d->K, d->N, pad16(d->K) / 16, pad32(d->N), total);
  return ssbev_launch_status();
}

int gemm16_nn(const bf16_t* A, const bf16_t* packed, void* Cm, const ssbev_gemm16_dims* d, hipStream_t st) {
  if (!gemm16_ok(d) || !A || !packed || !Cm) return SSBEV_EINVAL;
  Geom16 g;
  g.B = 1; g.Cin = d->K; g.Cout = d->N; g.KP = pad16(d->K) / 16; g.CoutPad = pad32(d->N);
  g.Di = g.Hi = 1; g.Wi = d->M; g.Do = g.Ho = 1; g.Wo = d->M;
  g.kd = g.kh = g.kw = 1; g.sd = g.sh = g.sw = 1; g.pd = g.ph = g.pw = 0; g.dd = g.dh = g.dw = 1;
  g.form = 0; g.relu = 0; g.accumulate = 0;
  g.bsx = (long)d->M * d->K; g.bsy = (long)d->M * d->N; g.bsw = (long)g.KP * 2 * g.CoutPad * 8;
  return d->out_fp32 ? launch_igemm16(A, packed, nullptr, static_cast<float*>(Cm), g, st, d->batch)
                     : launch_igemm16(A, packed, nullptr, static_cast<bf16_t*>(Cm), g, st, d->batch);
}


// ---- batched TN products C[b][k][n] = sum_m A[b][m][k] * B[b][m][n]: the Winograd WEIGHT-GRADIENT frequency products ----
// Both operands are row-major over the reduced axis m (the tile axis of the transformed activations / output gradients), so the
// MFMA operands (8 consecutive m of one channel per lane) are k-strided in memory: the stage goes global -> LDS as lane-linear
// 1 KiB [16 m][32 channels] pieces (global_load_lds_dwordx4, the wgrad16_kernel tile) and every operand is one pair of
// ds_read_b64_tr_b16 -- the hardware transposes.  Workgroup = 128 (k) x 128 (n) output tile, 2 x 2 waves of 64 x 64; stage = 32
// rows of m (two MFMA k-steps), double buffered (2 x 16 KiB).  Waves 0/1 stage the two m-halves of A, waves 2/3 those of B.
// blockIdx.y = slice of the m axis (fixed-order partial sums through the workspace when the output tiles alone do not fill the
// chip), blockIdx.z = batch element.
template <int NBUF>
__global__ void __launch_bounds__(256, 2)
gemm16_tn_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Bm, float* __restrict__ Cm, int M, int K, int N,
                 int chunk, int ktiles, int ntiles, int nslices, long split_stride) {
  extern __shared__ __align__(16) uint4 tl16[];                    // [2][A: 8 pieces | B: 8 pieces] of 64 uint4
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lk = lane >> 5;
  // 1-D grid, XCD-aware: workgroup L runs on XCD L % 8, so XCD x takes the x-th CONTIGUOUS eighth of the logical order
  // [batch][slice][tile] -- the tiles of one batch element (which share its A / B panels) meet in one L2.
  int kt, nt, slice, bt;
  {
    const unsigned n = gridDim.x, L = blockIdx.x;
    const unsigned xcd = L & 7, q = n >> 3, r = n & 7;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    unsigned Lp = base + (L >> 3);
    const unsigned tiles = (unsigned)(ktiles * ntiles);
    const unsigned t = Lp % tiles; Lp /= tiles;
    slice = (int)(Lp % (unsigned)nslices);
    bt = (int)(Lp / (unsigned)nslices);
    kt = (int)(t / (unsigned)ntiles); nt = (int)(t % (unsigned)ntiles);
  }
  const long m_begin = (long)slice * chunk;
  const long m_end = min((long)M, m_begin + chunk);
  A += (size_t)bt * M * K;
  Bm += (size_t)bt * M * N;
  // staging role
  const int is_b = wave >> 1, mh = wave & 1;
  const bf16_t* src = is_b ? Bm : A;
  const int ld = is_b ? N : K;
  const int c0 = (is_b ? nt : kt) * 128 + (lane & 3) * 8;
  const int jrow = mh * 16 + (lane >> 2);
  auto issue = [&](int buf, long m0) {
    const long m = m0 + jrow;
    const bool mok = m < m_end;
    const bf16_t* rowp = src + (size_t)(mok ? m : 0) * ld;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ch = c0 + e * 32;
      const void* s = (mok && ch < ld) ? static_cast<const void*>(rowp + ch) : static_cast<const void*>(&kZero16);
      __builtin_amdgcn_global_load_lds(s, tl16 + buf * 1024 + is_b * 512 + (mh * 4 + e) * 64, 16, 0, 0);
    }
  };
  const int wk = wave >> 1, wn = wave & 1;
  const int trbase = (8 * lk + ((lane & 15) >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int nst = (int)((m_end - m_begin + 31) / 32);
  // NBUF stages in LDS, NBUF - 1 in flight: a stage is 8 MFMAs per wave (256 cycles), far less than one L2 / HBM round trip
#pragma unroll
  for (int p = 0; p < NBUF - 1; ++p)
    if (p < nst) issue(p, m_begin + (long)p * 32);
  for (int st = 0; st < nst; ++st) {
    const int buf = st % NBUF;
    const int ahead = min(nst - 1 - st, NBUF - 2);                 // younger stages that may stay in flight (4 loads each)
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (st + NBUF - 1 < nst) issue((st + NBUF - 1) % NBUF, m_begin + (long)(st + NBUF - 1) * 32);
    const unsigned char* base = reinterpret_cast<const unsigned char*>(tl16 + buf * 1024);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      uint4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = tr_operand(base + (h * 4 + 2 * wk + i) * 1024, trbase);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = tr_operand(base + 8192 + (h * 4 + 2 * wn + j) * 1024, trbase);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  float* dst = Cm + (size_t)slice * split_stride + (size_t)bt * K * N;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = nt * 128 + wn * 64 + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = kt * 128 + wk * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row < K && col < N) dst[(size_t)row * N + col] = acc[i][j][r];
      }
    }
}

typedef float g16_f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256)
gemm16_sum_kernel(const float* __restrict__ ws, float* __restrict__ out, int splits, long total4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const g16_f4* p = reinterpret_cast<const g16_f4*>(ws) + i;
  g16_f4 s = p[0];
  for (int c = 1; c < splits; ++c) s += p[(size_t)c * total4];          // fixed order: deterministic
  reinterpret_cast<g16_f4*>(out)[i] = s;
}

bool gemm16_tn_ok(const ssbev_gemm16_dims* d) {
  return d && d->M > 0 && d->N > 0 && d->K > 0 && d->batch > 0 && d->K % 8 == 0 && d->N % 8 == 0 && d->batch < 65536 &&
         (long)d->M < (1L << 31);
}

// slices of the m axis: enough workgroups for two per CU, never thinner than 256 rows
static int gemm16_tn_splits(const ssbev_gemm16_dims* d) {
  static const int forced = ssbev_tune("SSBEV_GEMM16_TN_SPLITS") ? atoi(ssbev_tune("SSBEV_GEMM16_TN_SPLITS")) : 0;
  const long tiles = (long)cdiv(d->K, 128) * cdiv(d->N, 128) * d->batch;
  long s = forced > 0 ? forced : cdiv(512, tiles);
  s = std::min<long>(s, std::max<long>(1, d->M / 256));
  return (int)std::max<long>(1, std::min<long>(s, 64));
}

size_t gemm16_tn_workspace(const ssbev_gemm16_dims* d) {
  if (!gemm16_tn_ok(d)) return 0;
  const int s = gemm16_tn_splits(d);
  return s > 1 ? (size_t)s * d->batch * d->K * d->N : 0;
}

int gemm16_tn(const bf16_t* A, const bf16_t* Bm, float* Cm, const ssbev_gemm16_dims* d, float* ws, size_t ws_elems, hipStream_t st) {
  if (!gemm16_tn_ok(d) || !A || !Bm || !Cm) return SSBEV_EINVAL;
  const int splits = gemm16_tn_splits(d);
  const size_t total = (size_t)d->batch * d->K * d->N;
  if (splits > 1 && (!ws || ws_elems < (size_t)splits * total)) return SSBEV_EINVAL;
  int chunk = cdiv(cdiv(d->M, splits), 32) * 32;
  const int ntiles = cdiv(d->N, 128);
  const int ktiles = cdiv(d->K, 128), nsl = cdiv(d->M, chunk);
  const long blocks = (long)ktiles * ntiles * nsl * d->batch;
  if (blocks >= (1L << 31)) return SSBEV_EINVAL;
  dim3 grid((unsigned)blocks);
  static const int nbuf = ssbev_tune("SSBEV_GEMM16_TN_NBUF") ? atoi(ssbev_tune("SSBEV_GEMM16_TN_NBUF")) : 2;
  float* out = nsl > 1 ? ws : Cm;
  if (nbuf == 4)
    hipLaunchKernelGGL(gemm16_tn_kernel<4>, grid, dim3(256), 4 * 16384, st, A, Bm, out, d->M, d->K, d->N, chunk, ktiles, ntiles, nsl, (long)total);
  else if (nbuf == 3)
    hipLaunchKernelGGL(gemm16_tn_kernel<3>, grid, dim3(256), 3 * 16384, st, A, Bm, out, d->M, d->K, d->N, chunk, ktiles, ntiles, nsl, (long)total);
  else
    hipLaunchKernelGGL(gemm16_tn_kernel<2>, grid, dim3(256), 2 * 16384, st, A, Bm, out, d->M, d->K, d->N, chunk, ktiles, ntiles, nsl, (long)total);
  int rc = ssbev_launch_status();
  if (rc != SSBEV_OK || nsl == 1) return rc;
  hipLaunchKernelGGL(gemm16_sum_kernel, dim3((unsigned)cdiv((long)(total / 4), 256)), dim3(256), 0, st, ws, Cm, nsl, (long)(total / 4));
  return ssbev_launch_status();
}

}  // namespace ssbev_bf16

extern "C" {

// Batched plain products C[b] = A[b] x B[b] with bf16 operands and fp32 accumulation on conv_igemm16_kernel: the Winograd
// frequency products of the bf16 storage mode (functional._WinoConv: 16 x [1920 x 640 x 640], 64 x [4096 x 256 x 256], ...),
// which round 4 sent to rocBLAS through torch.bmm.
size_t ssbev_gemm16_packed_elems(const ssbev_gemm16_dims* d) { return ssbev_bf16::gemm16_packed_elems(d); }

int ssbev_gemm16_pack(const float* B, uint16_t* packed, const ssbev_gemm16_dims* d, ssbev_stream_t stream) {
  return ssbev_bf16::gemm16_pack(B, packed, d, as_stream(stream));
}

int ssbev_gemm16_nn(const uint16_t* A, const uint16_t* packed, void* C, const ssbev_gemm16_dims* d, ssbev_stream_t stream) {
  return ssbev_bf16::gemm16_nn(A, packed, C, d, as_stream(stream));
}

// C[b] = A[b]^T x B[b] over the row axis (A [batch][M][K], B [batch][M][N] bf16 -> C [batch][K][N] fp32): the Winograd weight-
// gradient frequency products of the bf16 storage mode on gemm16_tn_kernel (round 4: torch.bmm(out_dtype=fp32) = rocBLAS).
size_t ssbev_gemm16_tn_workspace(const ssbev_gemm16_dims* d) { return ssbev_bf16::gemm16_tn_workspace(d); }

int ssbev_gemm16_tn(const uint16_t* A, const uint16_t* B, float* C, const ssbev_gemm16_dims* d, float* workspace,
                    size_t workspace_elems, ssbev_stream_t stream) {
  return ssbev_bf16::gemm16_tn(A, B, C, d, workspace, workspace_elems, as_stream(stream));
}

}  // extern "C"
