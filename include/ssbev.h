/* ssbev.h -- C ABI of libssbev_hip.so: MI355X (gfx950) kernels for the StereoScene hot path.
 *
 * Drop-in boundary.  Every entry point is what a maintainer of the
 * reference would bind (ctypes stub in INTEGRATION.md) in place of the third-party CUDA op or
 * the torch op sequence cited next to it.  Citations are relative to the reference checkout:
 *   VT  = projects/mmdet3d_plugin/occupancy/image2bev/ViewTransformerLSSVoxel.py
 *   BD  = projects/mmdet3d_plugin/occupancy/image2bev/ViewTransformerLSSBEVDepth.py
 *   ATT = projects/mmdet3d_plugin/occupancy/image2bev/attention.py
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - all data pointers are DEVICE pointers owned by the caller; the library never allocates or frees
 *     device memory, never synchronises and keeps no state that changes a result; work is enqueued
 *     on `stream` (a hipStream_t).  The one thing it remembers is host-side memoisation of launch
 *     PLANS (chunking searches of a few 10k iterations keyed by the problem shape: a mutex-guarded
 *     std::map in csrc/conv_mfma.hip, plan_wgrad_lds): same inputs -> same plan -> same bits, with
 *     or without the cache.
 *   - return 0 on success, <0 on error (no exceptions cross the boundary):
 *       SSBEV_EINVAL      bad dims / null pointer / unsupported configuration
 *       SSBEV_EWORKSPACE  workspace smaller than ssbev_*_workspace() says
 *       SSBEV_ELAUNCH     hipGetLastError() != hipSuccess after a launch
 *   - re-entrant and thread-safe (autograd backward threads call in concurrently).
 *   - "channels-last" below means the channel index is the fastest-moving one.
 */
#ifndef SSBEV_H
#define SSBEV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSBEV_OK 0
#define SSBEV_EINVAL (-1)
#define SSBEV_EWORKSPACE (-2)
#define SSBEV_ELAUNCH (-3)

typedef void* ssbev_stream_t; /* hipStream_t */

/* library / device identification */
int ssbev_version(void);                 /* 10000*major + 100*minor + patch */
const char* ssbev_build_arch(void);      /* "gfx950" */
/* The library reads its SSBEV_* environment switches once per process (first use) and answers from a table afterwards: no
 * getenv on the launch path.  A host that changes a switch later (tests do) calls this to have it read again; not to be called
 * while other threads are inside the library. */
void ssbev_env_refresh(void);

/* ------------------------------------------------------------------------------------------
 * Frustum -> voxel scatter  (replaces VT:432-476 `voxel_pooling` + mmdet3d.ops.bev_pool, VT:473)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int B;            /* batch                                                        */
  int P;            /* points per batch element (= N*D*fH*fW)                        */
  int C;            /* channels                                                      */
  int nx, ny, nz;   /* voxel grid of the LSS volume (e.g. 128,128,16)                */
  float origin[3];  /* fp32( bx - dx/2 ), computed by the caller in fp32 (VT:441)    */
  float dx[3];      /* voxel size                                                    */
} ssbev_pool_dims;

/* VT:441-451.  geom[B*P,3] fp32 -> vox[B*P] int32: linear voxel ((b*nx+ix)*ny+iy)*nz+iz, or -1
 * if the point is dropped.  idx3 (int32 [B*P,3], may be NULL) receives (ix,iy,iz) =
 * trunc_toward_zero((geom - origin)/dx) saturated to int32.  Bit-exact w.r.t. the fp32
 * subtract / IEEE divide / truncate sequence of the reference. */
int ssbev_voxel_index(const float* geom, int32_t* vox, int32_t* idx3, const ssbev_pool_dims* d,
                      ssbev_stream_t stream);

/* mmdet3d.ops.bev_pool's coords[n,4] = (ix,iy,iz,b) (int32) -> vox[n] (same linearisation). */
int ssbev_coords_to_vox(const int32_t* coords, int n, int32_t* vox, const ssbev_pool_dims* d,
                        ssbev_stream_t stream);

/* Frustum points -> ego frame: the per-point part of get_geometry (ViewTransformerLSSBEVDepth.py:123-156) in one kernel,
 *   p = frustum[d,h,w,:] - t0[b,n];  p = m1[b,n] p;  p = (p.x p.z, p.y p.z, p.z);  p -= t2[b,n] (if given);
 *   p = m2[b,n] p + tr[b,n];  p = m3[b] p (+ t3[b] if given)
 * with m1 = inv(post_rots), t0 = post_trans, m2 = rots inv(intrins), t2 = intrins[:3,3] (4-column intrinsics), tr = trans,
 * m3 / t3 = the BEV augmentation; all row-major fp32.  Separately rounded multiplies and adds in the reference's order: the
 * points (and the voxel indices of ssbev_voxel_index) are bit-identical to the tensor expression.  geom[B,N,D,H,W,3]. */
typedef struct { int B, N, D, H, W; } ssbev_geom_dims;
int ssbev_frustum_geometry(const float* frustum, const float* m1, const float* t0, const float* m2, const float* t2,
                           const float* tr, const float* m3, const float* t3, float* geom, const ssbev_geom_dims* d,
                           ssbev_stream_t stream);

/* CSR build: starts[NV+1], order[n] with NV = B*nx*ny*nz.  Points of voxel v are
 * order[starts[v] .. starts[v+1]) in ASCENDING point index (the canonical summation order,
 * = stable argsort by rank in the upstream op).  Entries with vox<0 (or >= NV) are skipped: order[] holds starts[NV] ids,
 * its tail is left untouched.  Own two-level counting sort (csrc/voxel_pool.hip, "CSR build"): a stable partition by the
 * high digit of the voxel id, then one wavefront per 2^lo-voxel bucket that counts, scans (= starts) and places; five
 * launches for NV <= 2^22, no device-wide library sort.  SSBEV_POOL_MAX_DIGIT_BITS (2..11) narrows the digit (tests). */
size_t ssbev_pool_prepare_workspace(int n_points, const ssbev_pool_dims* d);
/* ... 2: additionally long_list[ssbev_pool_long_list_elems(n)] (nullable): long_list[0] = number of voxels with more than 32
 * points, long_list[1 ..] their ids in no particular order -- the work list of ssbev_lift_splat_fwd2's long-list waves. */
size_t ssbev_pool_long_list_elems(int n_points);
int ssbev_pool_prepare2(const int32_t* vox, int n_points, int32_t* starts, int32_t* order, int32_t* long_list,
                        const ssbev_pool_dims* d, void* ws, size_t ws_bytes, ssbev_stream_t stream);
int ssbev_pool_prepare(const int32_t* vox, int n_points, int32_t* starts, int32_t* order,
                       const ssbev_pool_dims* d, void* ws, size_t ws_bytes, ssbev_stream_t stream);

/* Drop-in for mmdet3d.ops.bev_pool forward/backward (VT:473).  feats[n,C] fp32 ->
 * out[B,nx,ny,nz,C] (channels-last; the reference's [B,C,nz,nx,ny].permute(0,1,3,4,2) is a
 * stride view of it).  Sequential fp32 sums in canonical order -> bit-reproducible. */
int ssbev_bev_pool_fwd(const float* feats, const int32_t* starts, const int32_t* order, float* out,
                       const ssbev_pool_dims* d, ssbev_stream_t stream);
int ssbev_bev_pool_bwd(const float* grad_out, const int32_t* vox, int n_points, float* grad_feats,
                       const ssbev_pool_dims* d, ssbev_stream_t stream);

/* Fused Lift (VT:517-519) + Splat (VT:523): the [B,N,D,H,W,C] product is never materialised.
 * depth[B*P] fp32 (= depth_prob [B*N,D,fH,fW] contiguous), feat[B*N*HW, C] channels-last,
 * point p of batch b uses feature row  b*(P/D) + (p/(D*HW))*HW + p%HW.
 * out[B,nx,ny,nz,C].  Each product is rounded to fp32 before the sequential add, exactly as the
 * reference's materialised `volume` would be. */
typedef struct { int N, D, HW; } ssbev_lift_dims;
int ssbev_lift_splat_fwd(const float* depth, const float* feat, const int32_t* starts,
                         const int32_t* order, float* out, const ssbev_pool_dims* d,
                         const ssbev_lift_dims* l, ssbev_stream_t stream);
/* ... 2: with the long-voxel list of ssbev_pool_prepare2 (nullable = ssbev_lift_splat_fwd): the lists of more than 32 points are
 * summed by dedicated whole-wave workgroups, the short ones four voxels per wave; same sums in the same order. */
int ssbev_lift_splat_fwd2(const float* depth, const float* feat, const int32_t* starts, const int32_t* order,
                          const int32_t* long_list, float* out, const ssbev_pool_dims* d, const ssbev_lift_dims* l,
                          ssbev_stream_t stream);
int ssbev_lift_splat_bwd(const float* grad_out, const float* depth, const float* feat,
                         const int32_t* vox, float* grad_depth, float* grad_feat,
                         const ssbev_pool_dims* d, const ssbev_lift_dims* l, ssbev_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Stereo plane-sweep cost volume: group-wise correlation fused with the disparity->depth
 * resample (replaces VT:104-114 `build_gwc_volume` + VT:128-156 `warp`).
 * left/right [B,H,W,C] channels-last fp32, calib[B] fp32 (= fx * baseline),
 * vol [B,D,H,W,G] channels-last (logical [B,G,D,H,W]).  align_corners selects the grid_sample
 * convention of VT:151-154 (0 = as-trained under torch 1.10.1, 1 = torch >= 1.3 literal).
 * ------------------------------------------------------------------------------------------ */
typedef struct { int B, C, G, D, H, W; float down; int align_corners; } ssbev_gwc_dims;
int ssbev_gwc_warp_fwd(const float* left, const float* right, const float* calib, float* vol,
                       const ssbev_gwc_dims* d, ssbev_stream_t stream);
int ssbev_gwc_warp_bwd(const float* grad_vol, const float* left, const float* right,
                       const float* calib, float* grad_left, float* grad_right,
                       const ssbev_gwc_dims* d, ssbev_stream_t stream);
/* Same gradients from ONE read of grad_vol (both views in one launch; plane chunks write partial rows into the
 * caller-owned workspace and are summed in chunk order: deterministic, no atomics).  Falls back to the two-launch
 * kernels above for shapes it does not cover (G % 4 != 0, 64 % G != 0, rows wider than its register plan). */
size_t ssbev_gwc_warp_bwd_workspace(const ssbev_gwc_dims* d);
int ssbev_gwc_warp_bwd_fused(const float* grad_vol, const float* left, const float* right,
                             const float* calib, float* grad_left, float* grad_right,
                             const ssbev_gwc_dims* d, void* workspace, size_t ws_bytes, ssbev_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Dense N-d convolution family on MFMA (implicit GEMM, no im2col), channels-last fp32.
 * Replaces the ATen/cuDNN conv3d / conv_transpose3d / conv2d calls behind VT:66-88,167-187,
 * VT:239-241, ATT:94-110, resnet3d.py:18-32, second_fpn_3d.py:53-69, occhead.py:100-107.
 *   x   [B, Di, Hi, Wi, Cin]     (2-D convs: Di = 1, kd = 1)
 *   y   [B, Do, Ho, Wo, Cout]
 *   w   packed by ssbev_conv_pack_weight (layout private to the library)
 * transposed = 0: y[o] = sum_k w[k] x[o*stride - pad + k*dil]
 * transposed = 1: y[o] = sum_k w[k] x[(o + pad - k)/stride]  (taps where divisible)
 * The MFMA used is v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int B, Cin, Cout;
  int Di, Hi, Wi;
  int Do, Ho, Wo;
  int kd, kh, kw;
  int sd, sh, sw;
  int pd, ph, pw;
  int dd, dh, dw;      /* dilation */
  int transposed;
  int relu;            /* fused epilogue: y = max(y,0) after bias                        */
  int accumulate;      /* y += result (used for residual / split accumulations)          */
  int tile_hint;       /* 0 = library heuristic; MT*100+NT*10+QU forces a register tiling  */
  int precision;       /* 0 = fp32 MFMA (exact fp32 products, the parity contract);
                        * 1 = forward / data gradient round their operands to bf16 in registers (nearest even) and use
                        *     v_mfma_f32_32x32x16_bf16 with fp32 accumulation (BASELINE configs[3]); tensors in memory
                        *     stay fp32, the weight gradient kernels stay on the fp32 MFMA;
                        * 2 = bf16 STORAGE (round 4; the configs[3] path proper, csrc/conv_bf16.hip): the source tensor (x in
                        *     ssbev_conv_fwd_bf16, gy in ssbev_conv_bwd_data_bf16; both in ssbev_conv_bwd_weight_bf16) AND the
                        *     result (y / gx) are bf16 channels-last tensors (uint16_t bit patterns), the packed
                        *     weights hold bf16 operands, accumulation is fp32, gw is fp32.  Needs Cin % 8 == 0 (and
                        *     Cout % 8 == 0 wherever the Cout-channel tensor is a source); bias stays fp32.  Round 5: modes 2 / 3
                        *     are served by the typed *_bf16 entry points ONLY; ssbev_conv_fwd / _bwd_data / _bwd_weight
                        *     answer SSBEV_EINVAL for them (and the *_bf16 ones for precision 0 / 1): a precision value that
                        *     does not match the tensors cannot be passed silently any more;
                        * 3 = as 2 with an fp32 RESULT (the layer in front of an fp32 island, e.g. the logits conv)
                        *     ssbev_conv_kernel_class answers, for these modes: 16 generic gather (conv_gather16_kernel), 17
                        *     LDS-ring tap kernel (<= 32 channels, conv_tap16_kernel), 19 LDS-ring implicit GEMM for the wide
                        *     stride-1 3x3x3 layers (conv_wide16_kernel); mode 2: 18 generic weight gradient (wgrad16_kernel),
                        *     20 LDS-ring weight gradient (wgrad_ring16_kernel)                                       */
} ssbev_conv_dims;

/* weight packing: src is the torch layout  conv: [Cout,Cin,kd,kh,kw]  deconv: [Cin,Cout,kd,kh,kw]
 * mode 0: forward operand;  mode 1: operand of the data-gradient (channels swapped, taps flipped) */
size_t ssbev_conv_packed_weight_elems(const ssbev_conv_dims* d);
/* which kernel family ssbev_conv_fwd (mode 0) / ssbev_conv_bwd_data (mode 1) dispatches this problem to -- for FLOP
 * accounting in profilers (bench.py): 0 = generic gather kernels, 1 = conv_tap_kernel, 2 = conv_taph_kernel (Winograd
 * F(2,3) along h inside the kernel: executes 2/3 of the operator's multiply-adds), 3 = conv_thin_kernel, 4 =
 * conv_thinin_kernel (1 / 2 / 4 -> 32 channels: the K-role tensor is taken UNPADDED, Cin % 4 is not required),
 * 5 = a 32 -> 1 / 2 / 4 channel problem for which the two-pass ssbev_conv_thin_* entry points below are available
 * (ssbev_conv_fwd / _bwd_data themselves run such a problem on class 3 or 0); mode 2 asks about ssbev_conv_bwd_weight:
 * 6 = wgrad_thinside_kernel (32 <-> 1 / 2 / 4 channels; x and gy are taken UNPADDED), 0 = the other weight-gradient
 * kernels; 7 / 8 = the stride-2 "down" / "up" gathers on conv_tap2_kernel / conv_tap2up_kernel; 9 = conv_tapdh_kernel
 * (round 4: Winograd F(2,3) along d AND h inside the kernel, even D and H: executes 4/9 of the operator's multiply-adds;
 * SSBEV_TAPDH=0 or tile_hint 4 keeps class 2); < 0 = error.  The weight gradient of a class-9 problem runs on
 * wgrad_tapdh_kernel (same transform domain; SSBEV_WGRAD_DH=0 or tile_hint 4 / 5 / 6 / 7 keep wgrad_lds_kernel);
 * 10 = conv_pw32_kernel (1x1x1 stride-1 layers with <= 32 channels on both sides: an HBM stream through per-wave LDS tiles;
 * SSBEV_PW32=0 or tile_hint 8 keep the generic gather kernel); 11 = conv_igemm_kernel (round 5: LDS-staged implicit GEMM for the
 * strided / transposed / dilated layers whose gather source has a multiple of 32 channels and whose destination has >= 64;
 * SSBEV_IGEMM=0 or tile_hint >= 10 keep class 0); bf16 storage: 21 = conv_igemm16_kernel (SSBEV_IGEMM16=0 keeps 16). */
int ssbev_conv_kernel_class(const ssbev_conv_dims* d, int mode);

/* 32 -> 1 / 2 / 4 channel 3x3x3 stride-1 "same" layers (the 32 -> 1 classifiers of the cost-volume stack,
 * ViewTransformerLSSVoxel.py:185-187, 239-241; mode 1: the data gradient of a 1 / 2 / 4 -> 32 layer) as one pass over the
 * wide tensor on the matrix pipe + one pass over 9 N partial planes in a caller-owned workspace
 * (ssbev_conv_thin_workspace bytes; 0 = not applicable).  Weights: torch layout -> ssbev_conv_thin_pack
 * (ssbev_conv_thin_packed_elems floats).  Tensors channels-last as for ssbev_conv_fwd; mode 0 applies bias / ReLU. */
size_t ssbev_conv_thin_workspace(const ssbev_conv_dims* d, int mode);
size_t ssbev_conv_thin_packed_elems(const ssbev_conv_dims* d, int mode);
int ssbev_conv_thin_pack(const float* w_src, float* w_packed, const ssbev_conv_dims* d, int mode, ssbev_stream_t stream);
int ssbev_conv_thin_run(const float* x, const float* w_packed, const float* bias, float* y, const ssbev_conv_dims* d,
                        int mode, void* workspace, size_t ws_bytes, ssbev_stream_t stream);
int ssbev_conv_pack_weight(const float* w_src, float* w_packed, const ssbev_conv_dims* d, int mode,
                           ssbev_stream_t stream);
int ssbev_conv_fwd(const float* x, const float* w_packed, const float* bias, float* y,
                   const ssbev_conv_dims* d, ssbev_stream_t stream);
/* bf16 storage (precision 2: y / gx are uint16_t bf16 tensors; 3: fp32 results, forward / data gradient only); pack the
 * weights with ssbev_conv_pack_weight under the same dims; workspace of the weight gradient from
 * ssbev_conv_bwd_weight_workspace under the same dims */
int ssbev_conv_fwd_bf16(const uint16_t* x, const float* w_packed, const float* bias, void* y,
                        const ssbev_conv_dims* d, ssbev_stream_t stream);
int ssbev_conv_bwd_data_bf16(const uint16_t* gy, const float* w_packed_t, void* gx,
                             const ssbev_conv_dims* d, ssbev_stream_t stream);
int ssbev_conv_bwd_weight_bf16(const uint16_t* x, const uint16_t* gy, float* gw, const ssbev_conv_dims* d,
                               void* ws, size_t ws_bytes, ssbev_stream_t stream);
/* data gradient: gx from gy with weights packed in mode 1 (d describes the FORWARD problem) */
int ssbev_conv_bwd_data(const float* gy, const float* w_packed_t, float* gx,
                        const ssbev_conv_dims* d, ssbev_stream_t stream);
/* weight gradient in the torch layout of w_src; ws from ssbev_conv_bwd_weight_workspace */
size_t ssbev_conv_bwd_weight_workspace(const ssbev_conv_dims* d);
int ssbev_conv_bwd_weight(const float* x, const float* gy, float* gw, const ssbev_conv_dims* d,
                          void* ws, size_t ws_bytes, ssbev_stream_t stream);

/* Batched plain products of the bf16 storage mode (round 5): C[b][m][n] = sum_k A[b][m][k] B[b][k][n], A and (out_fp32 = 0) C
 * bf16 bit patterns, fp32 accumulation on v_mfma_f32_32x32x16_bf16 -- the Winograd frequency products (functional._WinoConv:
 * torch.bmm / rocBLAS in round 4).  B is fp32 [batch][K][N] (the transformed weights) and is packed once per call by
 * ssbev_gemm16_pack into ssbev_gemm16_packed_elems 16-bit elements (layout private to the library).  K % 32 == 0, N % 8 == 0. */
typedef struct { int M, N, K, batch; int out_fp32; } ssbev_gemm16_dims;
size_t ssbev_gemm16_packed_elems(const ssbev_gemm16_dims* d);
int ssbev_gemm16_pack(const float* B, uint16_t* packed, const ssbev_gemm16_dims* d, ssbev_stream_t stream);
int ssbev_gemm16_nn(const uint16_t* A, const uint16_t* packed, void* C, const ssbev_gemm16_dims* d, ssbev_stream_t stream);
/* C[b] = A[b]^T x B[b] over the row axis: A [batch][M][K], B [batch][M][N] (bf16, row-major) -> C [batch][K][N] fp32 -- the weight-
 * gradient frequency products (reference: the F.conv2d / F.conv3d weight gradients of the same layers).  K % 8 == 0, N % 8 == 0.
 * workspace: ssbev_gemm16_tn_workspace(d) floats, caller-owned (0 when the output tiles alone fill the chip; fixed-order sums). */
size_t ssbev_gemm16_tn_workspace(const ssbev_gemm16_dims* d);
int ssbev_gemm16_tn(const uint16_t* A, const uint16_t* B, float* C, const ssbev_gemm16_dims* d, float* workspace,
                    size_t workspace_elems, ssbev_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * GroupNorm / BatchNorm(train) over channels-last volumes with fused residual add + ReLU.
 * Replaces the ATen group_norm / batch_norm calls behind build_norm_layer (VT:31,45,69,83,86;
 * ATT:96,111; resnet3d.py:42-45; second_fpn_3d.py:68; occhead.py:104).
 *   x, y, residual [B, S, C] channels-last fp32; group g = channels [g*C/G, (g+1)*C/G)
 *   y = relu?( (x - mean[b,g]) * rstd[b,g] * gamma[c] + beta[c] + residual? )
 *   stats_given = 1: mean/rstd are inputs (eval-mode BatchNorm), otherwise outputs.
 * BatchNorm3d in training mode = the same call with G = C on the tensor viewed as [1, B*S, C].
 *   pre_act = 1: the normalised quantity is gelu(x) (exact erf form) -- the Conv3d -> GELU -> GroupNorm triples of CA3D
 *   (attention.py:94-111) without the two elementwise passes of the activation; backward returns the gradient w.r.t. x.
 * ------------------------------------------------------------------------------------------ */
/* ld_y / ld_gy (0 = dense, C floats): row stride of y in _fwd and of gy in _bwd when the operator writes / reads a channel slice
 * of a wider channels-last tensor -- the branches of a concatenation (SECONDFPN3D, second_fpn3d.py:113-116; ASPP) normalise
 * straight into their slice of the concatenated tensor and read their slice of its gradient, no torch.cat / slice copies.
 * Pass y / gy already offset to the first channel of the slice; multiples of 4, >= C.  A strided y/gy of a B > 1 per-sample
 * GroupNorm is addressed as [(b * S + s) * ld + c]. */
/* io_dtype (round 4, BASELINE configs[3]): 0 = the activation tensors x / residual / y (and gy / gx / gresidual) are fp32, 1 = they
 * hold bf16 bit patterns (pass the pointers cast to float*; ld_y / ld_gy then count bf16 elements).  Statistics, gamma / beta and their
 * gradients, and all arithmetic stay fp32.  ABI note: ld_y / ld_gy (round 3) and io_dtype (round 4) were appended to the struct; a
 * binding built against an older, shorter struct must be rebuilt -- the library reads all fields. */
typedef struct { int B, C, G; int64_t S; float eps; int relu; int stats_given; int pre_act; int64_t ld_y, ld_gy; int io_dtype; } ssbev_norm_dims;
size_t ssbev_groupnorm_workspace(const ssbev_norm_dims* d);
int ssbev_groupnorm_fwd(const float* x, const float* gamma, const float* beta, const float* residual,
                        float* y, float* mean, float* rstd, const ssbev_norm_dims* d, void* ws,
                        size_t ws_bytes, ssbev_stream_t stream);
/* gx (and gresidual = relu-masked gy, may be NULL), ggamma[C], gbeta[C]; y only needed when relu=1 */
int ssbev_groupnorm_bwd(const float* gy, const float* x, const float* y, const float* gamma,
                        const float* mean, const float* rstd, float* gx, float* gresidual,
                        float* ggamma, float* gbeta, const ssbev_norm_dims* d, void* ws,
                        size_t ws_bytes, ssbev_stream_t stream);
/* Variants that carry the fused ReLU's sign pattern as a bit mask instead of re-reading y in backward (2 of the 8 tensor
 * passes of a normalisation with ReLU): relu_mask has ssbev_groupnorm_mask_words(d) 64-bit words; with T = float4 index of
 * an element quadruple, word (T / 64) * 4 + k, bit T % 64 = [component k of y is > 0].  Written by _fwd_mask when
 * d->relu = 1 (ignored otherwise), read by _bwd_mask. */
size_t ssbev_groupnorm_mask_words(const ssbev_norm_dims* d);
int ssbev_groupnorm_fwd_mask(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                             float* mean, float* rstd, uint64_t* relu_mask, const ssbev_norm_dims* d, void* ws,
                             size_t ws_bytes, ssbev_stream_t stream);
int ssbev_groupnorm_bwd_mask(const float* gy, const float* x, const uint64_t* relu_mask, const float* gamma,
                             const float* mean, const float* rstd, float* gx, float* gresidual, float* ggamma, float* gbeta,
                             const ssbev_norm_dims* d, void* ws, size_t ws_bytes, ssbev_stream_t stream);

/* BatchNorm running statistics inside the statistics finalize (round 5).
 *   running_mean / running_var (BatchNorm, G == C with the batch folded into S; NULL = none): the momentum update
 *                 r = (1 - momentum) r + momentum {mean, var * n / (n - 1)} of nn.BatchNorm in training mode, done by the
 *                 finalize kernel (ssbev_bn_update_running as a launch of its own otherwise); n = samples per channel.
 * _fwd_ext / _bwd_ext are ssbev_groupnorm_fwd_mask / _bwd_mask (relu_mask may be NULL when relu = 0; _bwd_ext takes y OR
 * relu_mask for a fused ReLU) with `ext` (may be NULL).
 * (Round 5 also carried a `sync` word here for a finalize-in-the-statistics-tail variant; measured slower, removed in round 6.) */
typedef struct { float* running_mean; float* running_var; float momentum; int64_t n; } ssbev_norm_ext;
int ssbev_groupnorm_fwd_ext(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                            float* mean, float* rstd, uint64_t* relu_mask, const ssbev_norm_dims* d,
                            const ssbev_norm_ext* ext, void* ws, size_t ws_bytes, ssbev_stream_t stream);
int ssbev_groupnorm_bwd_ext(const float* gy, const float* x, const float* y, const uint64_t* relu_mask, const float* gamma,
                            const float* mean, const float* rstd, float* gx, float* gresidual, float* ggamma, float* gbeta,
                            const ssbev_norm_dims* d, const ssbev_norm_ext* ext, void* ws, size_t ws_bytes,
                            ssbev_stream_t stream);

/* Two normalisations and their sum in one operator: y = relu?( N_a(xa) + N_b(xb) ), the tail of every hourglass level of
 * the cost-volume aggregation (ViewTransformerLSSVoxel.py:92-95: relu(BatchNorm3d(deconv) + GN(redir conv))).  Each side is a
 * per-sample GroupNorm (x_batch = 0: statistics [B][G]) or a normalisation over the batch (x_batch = 1: statistics [G], BatchNorm
 * when G == C).  The apply pass reads the two raw tensors once and writes the sum once; backward reads (gy, mask, xa, xb) twice
 * instead of eleven tensor passes for the two separate operators.  C % 4 == 0, C <= 1024; relu_mask as in ssbev_groupnorm_*_mask
 * (required when relu = 1).  Statistics are outputs of _fwd and inputs of _bwd. */
typedef struct { int B, C, Ga, Gb; int64_t S; float eps_a, eps_b; int relu; int a_batch, b_batch; int io_dtype; /* as in ssbev_norm_dims */ } ssbev_norm2_dims;
size_t ssbev_groupnorm2_workspace(const ssbev_norm2_dims* d);
int ssbev_groupnorm2_fwd(const float* xa, const float* gamma_a, const float* beta_a, float* mean_a, float* rstd_a,
                         const float* xb, const float* gamma_b, const float* beta_b, float* mean_b, float* rstd_b, float* y,
                         uint64_t* relu_mask, const ssbev_norm2_dims* d, void* ws, size_t ws_bytes, ssbev_stream_t stream);
int ssbev_groupnorm2_bwd(const float* gy, const uint64_t* relu_mask, const float* xa, const float* gamma_a, const float* mean_a,
                         const float* rstd_a, const float* xb, const float* gamma_b, const float* mean_b, const float* rstd_b,
                         float* gxa, float* gxb, float* ggamma_a, float* gbeta_a, float* ggamma_b, float* gbeta_b,
                         const ssbev_norm2_dims* d, void* ws, size_t ws_bytes, ssbev_stream_t stream);

/* as ssbev_norm_ext for the two-norm operator (running statistics per side, n = B * S) */
typedef struct { float* running_mean_a; float* running_var_a; float momentum_a;
                 float* running_mean_b; float* running_var_b; float momentum_b; } ssbev_norm2_ext;
int ssbev_groupnorm2_fwd_ext(const float* xa, const float* gamma_a, const float* beta_a, float* mean_a, float* rstd_a,
                             const float* xb, const float* gamma_b, const float* beta_b, float* mean_b, float* rstd_b, float* y,
                             uint64_t* relu_mask, const ssbev_norm2_dims* d, const ssbev_norm2_ext* ext, void* ws, size_t ws_bytes,
                             ssbev_stream_t stream);
int ssbev_groupnorm2_bwd_ext(const float* gy, const uint64_t* relu_mask, const float* xa, const float* gamma_a, const float* mean_a,
                             const float* rstd_a, const float* xb, const float* gamma_b, const float* mean_b, const float* rstd_b,
                             float* gxa, float* gxb, float* ggamma_a, float* gbeta_a, float* ggamma_b, float* gbeta_b,
                             const ssbev_norm2_dims* d, const ssbev_norm2_ext* ext, void* ws, size_t ws_bytes,
                             ssbev_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Trilinear x2 upsample (align_corners=False) of channels-last volumes, forward and gather-form
 * backward.  Replaces F.interpolate(..., mode='trilinear') at occhead.py:293-294 and
 * bevdepth_occupancy.py:293 (logits [B,20,128,128,16] -> [B,20,256,256,32]).
 *   x [B, D, H, W, C] -> y [B, 2D, 2H, 2W, C];  C % 4 == 0.
 * ------------------------------------------------------------------------------------------ */
typedef struct { int B, D, H, W, C; } ssbev_upsample_dims;
int ssbev_trilinear2x_fwd(const float* x, float* y, const ssbev_upsample_dims* d, ssbev_stream_t stream);
int ssbev_trilinear2x_bwd(const float* gy, float* gx, const ssbev_upsample_dims* d, ssbev_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Optimiser step over one flat fp32 parameter buffer: global gradient L2 norm + fused AdamW with
 * clip-by-global-norm (the reference recipe, stereoscene.py:203-209: AdamW lr 1e-4 wd 0.01,
 * grad_clip max_norm 5).  torch.optim.AdamW semantics (decoupled decay, bias correction).
 * ------------------------------------------------------------------------------------------ */
typedef struct { float lr, beta1, beta2, eps, weight_decay, max_grad_norm; int step; } ssbev_adamw_cfg;
size_t ssbev_grad_norm_workspace(void);
int ssbev_grad_norm(const float* g, int64_t n, float* norm_out, void* ws, size_t ws_bytes,
                    ssbev_stream_t stream);
/* grad_norm: device pointer to the norm (may be NULL or max_grad_norm <= 0 to disable clipping) */
int ssbev_adamw_step(float* p, const float* g, float* m, float* v, int64_t n,
                     const ssbev_adamw_cfg* c, const float* grad_norm, ssbev_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused occupancy-head epilogue: x2 trilinear upsample + softmax + every reduction the SemanticKITTI
 * losses (CE_ssc / sem_scal / geo_scal, utils/semkitti.py:67-149) and the train-time metric
 * (occhead.py:345-359) need, in one pass over the coarse logits; nothing of the fine grid is stored.
 *   logits [B, D, H, W, 20] channels-last;  label [B, 2D, 2H, 2W] uint8 (255 = ignore);  class_weight[20]
 *   sums[ssbev_occ_loss_num_sums()] (double): ce_num, ce_den, M, sum_p[20], nom[20], cnt[20], conf[20][20]
 *   backward: coef[41] = dL/d{ce_num, sum_p[20], nom[20]}  ->  grad_logits [B, D, H, W, 20]
 * ------------------------------------------------------------------------------------------ */
typedef struct { int B, D, H, W, C; } ssbev_occloss_dims;
int ssbev_occ_loss_num_sums(void);
size_t ssbev_occ_loss_workspace(const ssbev_occloss_dims* d);
int ssbev_occ_loss_fwd(const float* logits, const uint8_t* label, const float* class_weight, double* sums,
                       const ssbev_occloss_dims* d, void* ws, size_t ws_bytes, ssbev_stream_t stream);
/* The elementwise shell of a BRI attention block (attention.py:45-86) around its six products, operands [B, D, T] with the token
 * axis contiguous; every scalar parameter is a DEVICE pointer to one float (the 1x1x1 single-channel convolutions' weight / bias,
 * gamma).  Forward: Q = q wq + bq, K = kv wk + bk, Vc = (kv wv + bv) conf[b, t]  (pre);  y = gamma out + kv  (post).
 * Backward: post_bwd writes gout = gy gamma, per-chunk partial column sums delta_part[chunks][B*T] of gout * out (their sum over
 * the chunk axis is the softmax-backward row term) and partial sums of gy * out (-> d gamma); pre_bwd writes gq, gkv (gres = the
 * residual branch's gradient, added in), gconf_part[chunks][B*T] and part[..][6] = partial sums for d(wq, bq, wk, bk, wv, bv).
 * chunks = ssbev_bri_shell_chunks(); part arrays have chunks * ceil(B*T / 256) rows (doubles), summed by the caller. */
int ssbev_bri_shell_chunks(void);
int ssbev_bri_shell_pre_fwd(const float* q, const float* kv, const float* conf, const float* wq, const float* bq,
                            const float* wk, const float* bk, const float* wv, const float* bv, float* Q, float* K, float* Vc,
                            int B, int D, int T, ssbev_stream_t stream);
int ssbev_bri_shell_post_fwd(const float* out, const float* kv, const float* gamma, float* y, int B, int D, int T,
                             ssbev_stream_t stream);
int ssbev_bri_shell_post_bwd(const float* gy, const float* out, const float* gamma, float* gout, float* delta_part, double* part,
                             int B, int D, int T, ssbev_stream_t stream);
int ssbev_bri_shell_pre_bwd(const float* gQ, const float* gK, const float* gVc, const float* q, const float* kv, const float* conf,
                            const float* gres, const float* wq, const float* wk, const float* wv, const float* bv, float* gq,
                            float* gkv, float* gconf_part, double* part, int B, int D, int T, ssbev_stream_t stream);

/* The scalar algebra behind the sums in one launch: out5 (float) = w_ce * CE, w_sem * sem_scal, w_geo * geo_scal (occhead.py:
 * 291-361, semkitti.py:67-149), completion IoU, mean IoU over classes 1..19; jac[3][41] (double) = the Jacobian of the three
 * weighted losses w.r.t. (ce_num, sum_p[20], nom[20]) -- backward is jac^T times the three incoming scalars, which is the
 * `coef` of ssbev_occ_loss_bwd. */
int ssbev_occ_loss_tail(const double* sums, float w_ce, float w_sem, float w_geo, float* out5, double* jac,
                        ssbev_stream_t stream);
size_t ssbev_occ_loss_bwd_workspace(const ssbev_occloss_dims* d);
int ssbev_occ_loss_bwd(const float* logits, const uint8_t* label, const float* class_weight,
                       const float* coef, float* grad_logits, const ssbev_occloss_dims* d, void* ws,
                       size_t ws_bytes, ssbev_stream_t stream);

/* Running statistics of a training-mode BatchNorm from the (mean, rstd) ssbev_groupnorm_fwd returned with G == C over the
 * batch: running_mean <- (1-m) running_mean + m mean; running_var <- (1-m) running_var + m var n/(n-1)
 * (torch.nn.BatchNorm3d as built at ViewTransformerLSSVoxel.py:83-88; n = elements per channel). */
int ssbev_bn_update_running(const float* mean, const float* rstd, float* running_mean, float* running_var, int C,
                            float momentum, float eps, int64_t n, ssbev_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Softmax along a strided axis: x = [outer][C][inner] (inner contiguous), y = softmax over C.
 * Replaces F.softmax(dim=1) on the depth-major matching distribution (ViewTransformerLSSVoxel.py:255-259)
 * and F.softmax(q, dim=2) of the BRI confidence (attention.py:66-68); bwd: gx = y * (gy - sum_c y * gy).
 * ------------------------------------------------------------------------------------------ */
int ssbev_softmax_axis_fwd(const float* x, float* y, int64_t outer, int C, int64_t inner, ssbev_stream_t stream);
int ssbev_softmax_axis_bwd(const float* y, const float* gy, float* gx, int64_t outer, int C, int64_t inner,
                           ssbev_stream_t stream);
/* Softmax along the innermost axis of `rows` rows of C contiguous floats (C % 4 == 0, C <= 8192): the BRI attention
 * matrix softmax(energy, -1) of attention.py:66-68 (T = 7680).  One read + one write of the matrix forward, two reads +
 * one write backward (gx = y * (gy - sum_c y * gy)); y may alias x and gx may alias gy. */
int ssbev_softmax_rows_fwd(const float* x, float* y, int64_t rows, int C, ssbev_stream_t stream);
int ssbev_softmax_rows_bwd(const float* y, const float* gy, float* gx, int64_t rows, int C, ssbev_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Deformable convolution v1, sampling stages (mmcv DeformConv2dPack as built by DepthNet, bevdepth.py:490-498;
 * mmcv/ops/csrc deform_conv: deformable_im2col / deformable_col2im / deformable_col2im_coord).  stride 1,
 * deform_groups 1.  x [B,H,W,C], offset [B,H,W,2*k*k] (channel 2t = dy, 2t+1 = dx of tap t),
 * cols / gcols [G][B*H*W][k*k*(C/G)]: slab g is a channels-last [B, k*k*C/G, H, W] tensor, so the grouped
 * contraction is ssbev_conv_fwd with a 1x1 kernel per group.  col2im overwrites gx and goffset; both are run-to-run identical
 * (round 6: the x gradient is a gather over the samples sorted by target pixel with ssbev_pool_prepare, summed in ascending
 * (pixel, tap, corner) order -- mmcv's col2im scatters with atomics).  ws: ssbev_dcn_col2im_workspace bytes.
 * ------------------------------------------------------------------------------------------ */
typedef struct { int B, C, H, W, G, k, pad, dil; } ssbev_dcn_dims;
int ssbev_dcn_im2col(const float* x, const float* offset, float* cols, const ssbev_dcn_dims* d, ssbev_stream_t stream);
size_t ssbev_dcn_col2im_workspace(const ssbev_dcn_dims* d);
int ssbev_dcn_col2im(const float* x, const float* offset, const float* gcols, float* gx, float* goffset,
                     const ssbev_dcn_dims* d, void* ws, size_t ws_bytes, ssbev_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Data side: CreateDepthFromLiDAR (datasets/pipelines/occ_to_depth.py:216-303), the producer of the
 * gt_depths slot of img_inputs.  points [N][3] fp32 (lidar frame), cam = 34 floats (host or device memory):
 * inv(rots) 3x3 row major | trans 3 | intrins 4x4 row major | post_rots[:2,:2] | post_trans[:2].
 * Writes uvd [N][3] (augmented pixel u, v and depth of every point), valid [N] (inside the image, depth > 0), the
 * depth map [H][W] = depth of the NEAREST valid point per pixel (pixel = round-half-even(u, v)), 0 where none, and --
 * when labels [N] and seg are given -- seg [H][W] = label of that nearest point.  Same operation order as the
 * reference's torch expressions; ties on depth resolve to the lowest point index.  ws: ssbev_lidar_depth_workspace bytes.
 * The 34 camera floats are copied synchronously (one implicit stream sync: this is a data-loading operator).
 * ------------------------------------------------------------------------------------------ */
/* Image loading (loading_semkitti.py:101-131, 176-232: `img.resize(resize_dims)`, `img.crop(crop)`, flip, mmcv
 * `imnormalize`, HWC -> CHW).  ssbev_resize_pil_u8 is byte-exact with Pillow's antialiased resize of 8-bit images: the
 * caller passes libImaging's fixed-point coefficient tables (kk [out][ksize] int32 with 22 fractional bits, bounds
 * [out][2] = first source index, tap count) for the horizontal and the vertical pass; tmp [Hs][Wd][C] is the 8-bit
 * intermediate.  ssbev_crop_normalize_u8: dst [3][h][w] = (src[y0+y][x0+x'][c'] - mean[c]) * stdinv[c], x' mirrored when
 * flip, c' = 2-c when swap_rb; pixels outside the source read 0 (PIL crop semantics); mean / stdinv are HOST pointers. */
int ssbev_resize_pil_u8(const uint8_t* src, int Hs, int Ws, int C, const int32_t* kk_h, const int32_t* bounds_h, int ksize_h,
                        const int32_t* kk_v, const int32_t* bounds_v, int ksize_v, uint8_t* tmp, uint8_t* dst, int Hd, int Wd,
                        ssbev_stream_t stream);
int ssbev_crop_normalize_u8(const uint8_t* src, int Hs, int Ws, float* dst, int x0, int y0, int w, int h, int flip,
                            const float* mean, const float* stdinv, int swap_rb, ssbev_stream_t stream);
/* Depth BCE loss (ViewTransformerLSSVoxel.py:349-416: get_downsampled_gt_depth + get_depth_loss): gt_depths [BN, fH*ds, fW*ds]
 * (0 = no LiDAR return), depth_pred [BN, D, fH, fW] (a probability distribution over the D bins per pixel).  out2[0] = weight *
 * sum of the binary cross entropies over the pixels that have a return / max(their number, 1), out2[1] = that divisor.  The
 * bin of a pixel is (min over its ds x ds block - c0) / dd in fp32, truncated, with c0 = d0 - dd / 2 computed by the caller as
 * the reference does (in double, rounded to fp32 once).  The workspace keeps the per-pixel labels for _bwd, which writes
 * grad_pred = grad_loss[0] * d out2[0] / d depth_pred. */
size_t ssbev_depth_bce_workspace(int BN, int fH, int fW);
int ssbev_depth_bce_fwd(const float* gt_depths, const float* depth_pred, float* out2, int BN, int D, int fH, int fW, int ds,
                        float c0, float dd, float weight, void* ws, size_t ws_bytes, ssbev_stream_t stream);
int ssbev_depth_bce_bwd(const float* depth_pred, const float* grad_loss, const float* out2, float* grad_pred, int BN, int D, int fH,
                        int fW, int ds, float c0, float dd, float weight, const void* ws, ssbev_stream_t stream);

size_t ssbev_lidar_depth_workspace(int H, int W);
int ssbev_lidar_depth_map(const float* points, int n_points, const float* cam, const float* labels, float* uvd,
                          unsigned char* valid, float* depth, float* seg, int H, int W, void* ws, size_t ws_bytes,
                          ssbev_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Image branch (the step before a1/a5): operators of CustomEfficientNet (backbones/efficientnet.py:112-229,
 * 275-519: InvertedResidual = expand 1x1 -> depthwise k x k -> SE -> linear 1x1) that are not dense contractions.
 * Channels-last fp32, C % 4 == 0.
 *   dwconv2d: depthwise k x k (k = 3 | 5, stride 1 | 2), y[b,ho,wo,c] = sum_ij x[b, ho*s - pad_t + i, wo*s - pad_l + j, c]
 *             * w[i*k + j][c]; out-of-range taps read zero.  pad_t / pad_l and Ho / Wo are explicit, so mmcv's
 *             Conv2dAdaptivePadding ("same": Ho = ceil(Hi/s), odd padding row/column at the bottom/right) is
 *             pad_t = pad_total_h / 2, pad_l = pad_total_w / 2.  w / gw layout: [k*k][C] (torch [C,1,k,k] transposed).
 *   swish:    y = x * sigmoid(x) (mmcv Swish) and its gradient.
 *   chan_sum: out[b][c] = scale * sum_s a[b][s][c] * (bmul ? bmul[b][s][c] : 1) -- AdaptiveAvgPool2d(1) of mmdet's
 *             SELayer (scale = 1/S) and the gate gradient of its rescale.
 *   chan_scale: y[b][s][c] = x[b][s][c] * gate[b][c]; x = NULL: y = gate[b][c] * scale (gradient of the pool).
 * ------------------------------------------------------------------------------------------ */
typedef struct { int B, C, Hi, Wi, Ho, Wo, k, stride, pad_t, pad_l; } ssbev_dw_dims;
int ssbev_dwconv2d_fwd(const float* x, const float* w, float* y, const ssbev_dw_dims* d, ssbev_stream_t stream);
int ssbev_dwconv2d_bwd_data(const float* gy, const float* w, float* gx, const ssbev_dw_dims* d, ssbev_stream_t stream);
size_t ssbev_dwconv2d_bwd_weight_workspace(const ssbev_dw_dims* d);                      /* floats */
int ssbev_dwconv2d_bwd_weight(const float* x, const float* gy, float* gw, const ssbev_dw_dims* d, float* ws,
                              size_t ws_elems, ssbev_stream_t stream);
int ssbev_swish_fwd(const float* x, float* y, int64_t n, ssbev_stream_t stream);        /* n % 4 == 0 */
int ssbev_swish_bwd(const float* x, const float* gy, float* gx, int64_t n, ssbev_stream_t stream);
size_t ssbev_chan_sum_workspace(int B, int64_t S, int C);                                 /* floats */
int ssbev_chan_sum(const float* a, const float* bmul, float* out, int B, int64_t S, int C, float scale, float* ws,
                   size_t ws_elems, ssbev_stream_t stream);
int ssbev_chan_scale(const float* x, const float* gate, float* y, int B, int64_t S, int C, float scale,
                     ssbev_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Winograd F(2x2x2, 3x3x3) transforms for stride-1 3x3x3 "same" convolutions (even D, H, W; channels-last).
 * T = B * D/2 * H/2 * W/2 tiles, 64 frequencies.  The element-wise stage between them is 64 plain GEMMs
 * [T x Cin] x [Cin x Cout] (forward / data gradient) or [Cin x T] x [T x Cout] (weight gradient).
 *   input_transform:   x  [B,D,H,W,C] -> V [64][T][C]   (B^T d B; also used on gy for the data gradient)
 *   output_transform:  M  [64][T][C]  -> y [B,D,H,W,C]  (A^T M A)
 *   output_adjoint:    gy [B,D,H,W,C] -> Z [64][T][C]   (A gy A^T, weight gradient)
 * Replaces nn.Conv3d at resnet3d.py:18-32, second_fpn_3d.py:53-69 (3x3x3 convs), occhead.py:100-107.
 * ------------------------------------------------------------------------------------------ */
typedef struct { int B, D, H, W, C; } ssbev_wino_dims;
int ssbev_wino_input_transform(const float* x, float* V, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino_output_transform(const float* M, float* y, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino_output_adjoint(const float* gy, float* Z, const ssbev_wino_dims* d, ssbev_stream_t stream);
/* 2-D variant F(2x2, 3x3) for the 3x3 conv2d layers of DepthNet (ViewTransformerLSSBEVDepth.py:461-504): 16 frequencies,
 * T = B * D * H/2 * W/2 (D is a batch axis), buffers [16][T][C]. */
int ssbev_wino2d_input_transform(const float* x, float* V, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino2d_output_transform(const float* M, float* y, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino2d_output_adjoint(const float* gy, float* Z, const ssbev_wino_dims* d, ssbev_stream_t stream);
/* bf16 variants (BASELINE configs[3], "bf16 mixed precision, MFMA 3D-conv path"): identical transforms, but the
 * transformed-domain tensor (V, M, Z) is stored as bf16 (uint16_t bit patterns, round-to-nearest-even) so that the
 * frequency GEMMs run on the bf16 matrix pipe with fp32 accumulation and the streaming passes move half the bytes.
 * Activations, gradients and weights on the caller's side stay fp32. */
int ssbev_wino_input_transform_bf16(const float* x, uint16_t* V, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino_output_transform_bf16(const uint16_t* M, float* y, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino_output_adjoint_bf16(const float* gy, uint16_t* Z, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino2d_input_transform_bf16(const float* x, uint16_t* V, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino2d_output_transform_bf16(const uint16_t* M, float* y, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino2d_output_adjoint_bf16(const float* gy, uint16_t* Z, const ssbev_wino_dims* d, ssbev_stream_t stream);
/* `_bf16a` (round 4, bf16 STORAGE mode): the activation side (x, gy, y) is a bf16 channels-last tensor too -- every pass of the
 * F(2,3) pipeline then moves 16-bit elements only. */
int ssbev_wino_input_transform_bf16a(const uint16_t* x, uint16_t* V, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino_output_transform_bf16a(const uint16_t* M, uint16_t* y, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino_output_adjoint_bf16a(const uint16_t* gy, uint16_t* Z, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino2d_input_transform_bf16a(const uint16_t* x, uint16_t* V, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino2d_output_transform_bf16a(const uint16_t* M, uint16_t* y, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino2d_output_adjoint_bf16a(const uint16_t* gy, uint16_t* Z, const ssbev_wino_dims* d, ssbev_stream_t stream);
/* F(2x4x4, 3x3x3) / F(4x4, 3x3): 4-wide tiles along h and w (Lavin & Gray's F(4,3)), 2-deep along d.  144 (2-D: 36)
 * frequencies, xi = (a*6 + e)*6 + f; T = B * D/2 * H/4 * W/4 (2-D: B * D * H/4 * W/4); needs H % 4 == W % 4 == 0
 * (3-D: D % 2 == 0).  Transformed domain 4.5x (2.25x) the activation instead of 8x (4x), GEMM stage 6x (4x) fewer
 * multiply-adds than the direct convolution.  Same roles as the ssbev_wino_* functions above; `_bf16` = bf16 storage of
 * the transformed-domain tensor. */
int ssbev_wino43_input_transform(const float* x, float* V, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino43_output_transform(const float* M, float* y, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino43_output_adjoint(const float* gy, float* Z, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino43_2d_input_transform(const float* x, float* V, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino43_2d_output_transform(const float* M, float* y, const ssbev_wino_dims* d, ssbev_stream_t stream);
/* y += transform(M): the data gradient of a second consumer of an activation is added to the first consumer's in the output
 * transform itself (functional.fork / GradSlot) instead of by a separate elementwise pass */
int ssbev_wino43_2d_output_transform_acc(const float* M, float* y, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino2d_output_transform_acc(const float* M, float* y, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino43_2d_output_adjoint(const float* gy, float* Z, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino43_input_transform_bf16(const float* x, uint16_t* V, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino43_output_transform_bf16(const uint16_t* M, float* y, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino43_output_adjoint_bf16(const float* gy, uint16_t* Z, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino43_2d_input_transform_bf16(const float* x, uint16_t* V, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino43_2d_output_transform_bf16(const uint16_t* M, float* y, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino43_2d_output_adjoint_bf16(const float* gy, uint16_t* Z, const ssbev_wino_dims* d, ssbev_stream_t stream);
/* F(4x4x4, 3x3x3): F(4,3) along d as well (D % 4 == 0): 216 frequencies per 64 outputs, T = B * D/4 * H/4 * W/4, 8x fewer
 * multiply-adds, 3.375x transformed domain.  Weight functions below: ndim = 4 selects this variant (3 = F(2x4x4), 2 = 2-D). */
int ssbev_wino444_input_transform(const float* x, float* V, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino444_output_transform(const float* M, float* y, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino444_output_adjoint(const float* gy, float* Z, const ssbev_wino_dims* d, ssbev_stream_t stream);
int ssbev_wino43_weight_transform(const float* w, float* U, int Cout, int Cin, int ndim, int mode, ssbev_stream_t stream);
int ssbev_wino43_weight_grad(const float* gU, float* gw, int Cout, int Cin, int ndim, ssbev_stream_t stream);
/* Weight side: U = G w G^T (mode 0: U [NF][Cin][Cout] from torch-layout w [Cout][Cin][taps]; mode 1: the data-gradient
 * operand [NF][Cout][Cin] from the mirrored taps), and gw = G^T gU G for the weight gradient.  ndim = 3 (27 taps, NF = 64)
 * or 2 (9 taps, NF = 16). */
int ssbev_wino_weight_transform(const float* w, float* U, int Cout, int Cin, int ndim, int mode, ssbev_stream_t stream);
int ssbev_wino_weight_grad(const float* gU, float* gw, int Cout, int Cin, int ndim, ssbev_stream_t stream);
/* Depth-fused frequency GEMM of the 3-D path: P = ssbev_wino2d_input_transform(x) [16][B*D*Thw][K] -> Mo [16][B*D*Thw][N]
 * (-> ssbev_wino2d_output_transform); the depth axis of F(2,3) is applied in registers (d->C = K input channels of this
 * product; mode 0 forward: K = Cin, N = Cout; mode 1 data gradient: K = Cout, N = Cin). */
size_t ssbev_wino_dgemm_packed_elems(int Cout, int Cin);
int ssbev_wino_dgemm_pack(const float* w, float* Wp, int Cout, int Cin, int mode, ssbev_stream_t stream);
int ssbev_wino_dgemm(const float* P, const float* Wp, float* Mo, const ssbev_wino_dims* d, int N, ssbev_stream_t stream);
/* Batched frequency GEMM of the plain 3-D pipeline: Cm[xi][T][N] = A[xi][T][K] x U[xi][K][N] for the 64 frequencies, A
 * streamed through LDS once, U in the ssbev_wino_dgemm_pack layout. */
int ssbev_wino_bgemm(const float* A, const float* Wp, float* Cm, int64_t T, int K, int N, ssbev_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Plain fp32 GEMMs on the matrix cores (csrc/gemm.hip): the layers that are matrix products in the channels-last layout
 * -- kernel == stride transposed convolutions of SECONDFPN3D (second_fpn_3d.py:50-69), wide pointwise convolutions
 * (ASPP 3200 -> 640), the DCN group contractions (BD:490-498), the six products of a BRI block (attention.py:72-81), the
 * batched frequency products of the 2-D Winograd layers.  Replaces the rocBLAS calls behind torch.mm / torch.bmm.
 *   nn: C[b][m][n] = sum_k A[b][m][k] B[b][k][n]      nt: ... W[b][n][k]      tn: C[b][k][n] = sum_r A[b][r][k] B[b][r][n]
 * Leading dimensions / batch strides in floats; K, N, lda, ldb multiples of 4.  d2s_*: depth-to-space index map of a k == s
 * transposed convolution (d2s_kd = 0: off): row m = coarse voxel (bb, d, h, w), wide index tap * Co + co <-> fine voxel
 * (d kd + a, h kh + b, w kw + c), channel co.  nn scatters C, nt gathers A, tn gathers B through it.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int M, N, K, batch;
  int64_t lda, ldb, ldc, sa, sb, sc;
  int relu;
  int d2s_D, d2s_H, d2s_W, d2s_kd, d2s_kh, d2s_kw, d2s_Co;
  const int64_t* d2s_rowoff;      /* device table of M row offsets (ssbev_gemm_d2s_rowoff), required when d2s_kd > 0 */
  /* ssbev_gemm_tn only, both or neither: C[b][k][n] = ep_mul[b][k][n] * ((A^T B)[b][k][n] - ep_rowsub[b][k]), ep_mul laid out
   * like C.  With A = grad_out, B = V conf, ep_mul = att, ep_rowsub[i] = <grad_out[:, i], out[:, i]> this is the softmax
   * backward gE = att (.) (gatt - rowsum(gatt (.) att)) of the BRI attention (attention.py:63-81) inside the product that
   * forms gatt: neither gatt nor a separate softmax-backward pass over the T x T matrices exists. */
  const float* ep_mul;
  const float* ep_rowsub;
} ssbev_gemm_dims;
int ssbev_gemm_d2s_rowoff(int64_t* rowoff, int M, int D, int H, int W, int kd, int kh, int kw, int Co, ssbev_stream_t stream);
size_t ssbev_gemm_nn_workspace(const ssbev_gemm_dims* d);      /* split-K partials (0 when the output tiles fill the chip) */
size_t ssbev_gemm_nt_workspace(const ssbev_gemm_dims* d);
int ssbev_gemm_nn(const float* A, const float* B, const float* bias, float* C, const ssbev_gemm_dims* d, void* workspace,
                  size_t ws_bytes, ssbev_stream_t stream);
int ssbev_gemm_nt(const float* A, const float* W, const float* bias, float* C, const ssbev_gemm_dims* d, void* workspace,
                  size_t ws_bytes, ssbev_stream_t stream);
size_t ssbev_gemm_tn_workspace(const ssbev_gemm_dims* d);
int ssbev_gemm_tn(const float* A, const float* B, float* C, const ssbev_gemm_dims* d, void* workspace, size_t ws_bytes,
                  ssbev_stream_t stream);

/* Depth-fused Winograd contraction (csrc/winograd_fused.hip): F(4,3) x F(4,3) over (h, w) in memory (P / Mo / Z are
 * [36][B*D*(H/4)*(W/4)][C], the 2-D transforms above with D as a batch axis), F(2,3) along d in registers around the MFMAs.
 * The default realisation of the wide stride-1 3x3x3 layers (resnet3d.py:18-32, second_fpn_3d.py:53-69, occhead.py:100-107):
 * forward / data gradient = ssbev_wino43_2d_input_transform -> ssbev_wino43_df_gemm -> ssbev_wino43_2d_output_transform,
 * weight gradient = ssbev_wino43_2d_output_adjoint(gy) + ssbev_wino43_df_wgrad(P saved by the forward).
 * d = (B, D, H, W, C = K).  Supported when H % 4 == W % 4 == 0, D even, K % 32 == 0. */
int ssbev_wino43_df_supported(const ssbev_wino_dims* d, int N);
/* template instance of the contraction kernel a problem runs on: MT * 10 + NW of wino_df_kernel<MT, NW> (0: unsupported dims) --
 * lets a profiler attribute launches to kernel symbols (bench.py picks its roofline kernel by summed time per SYMBOL) */
int ssbev_wino43_df_instance(const ssbev_wino_dims* d, int N);
size_t ssbev_wino43_df_packed_elems(int Cout, int Cin);
int ssbev_wino43_df_pack(const float* w, float* Wp, int Cout, int Cin, int mode, ssbev_stream_t stream);
int ssbev_wino43_df_gemm(const float* P, const float* Wp, float* Mo, const ssbev_wino_dims* d, int N, ssbev_stream_t stream);
size_t ssbev_wino43_df_wgrad_workspace(const ssbev_wino_dims* d, int N);
int ssbev_wino43_df_wgrad(const float* P, const float* Z, float* gw, const ssbev_wino_dims* d, int N, void* workspace,
                          size_t ws_bytes, ssbev_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SSBEV_H */
