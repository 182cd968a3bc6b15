// bf16-STORAGE convolution family (BASELINE configs[3]; csrc/conv_bf16.hip): activations and activation gradients live in HBM
// as bf16 channels-last tensors, weights are packed to bf16 MFMA operands on the fly, accumulation is fp32
// (v_mfma_f32_32x32x16_bf16), weight gradients come out in fp32.  Selected by ssbev_conv_dims.precision:
//     0  fp32 storage, fp32 MFMA                     (the parity path)
//     1  fp32 storage, operands rounded to bf16      (rounds 1-3 "bf16 mode")
//     2  bf16 source tensor (x / gy) AND bf16 result (y / gx);  weight gradient: x and gy bf16, gw fp32
//     3  bf16 source tensor, fp32 result             (the layer in front of an fp32 island)
#pragma once
#include "common.h"

namespace ssbev_bf16 {

bool storage_mode(const ssbev_conv_dims* d);                       // precision 2 or 3
bool dims_ok(const ssbev_conv_dims* d, int mode);                  // channel multiples the kernels need (mode 0 fwd, 1 dgrad, 2 wgrad)
int kernel_class(const ssbev_conv_dims* d, int mode);              // 16 / 17 / 19 forward + data gradient kernels, 18 / 20 weight gradient (ssbev.h)
size_t packed_elems(const ssbev_conv_dims* d);                     // in floats (the buffer holds bf16 operands)
int pack(const float* w_src, float* w_packed, const ssbev_conv_dims* d, int mode, hipStream_t st);
int forward(const void* x, const float* wp, const float* bias, void* y, const ssbev_conv_dims* d, hipStream_t st);
int backward_data(const void* gy, const float* wp, void* gx, const ssbev_conv_dims* d, hipStream_t st);
size_t wgrad_workspace(const ssbev_conv_dims* d);
int backward_weight(const void* x, const void* gy, float* gw, const ssbev_conv_dims* d, void* ws, size_t ws_bytes, hipStream_t st);

}  // namespace ssbev_bf16
