// Internal interface of conv_thin_mfma.hip (thin-side 3x3x3 layers of the cost-volume stack); called from the
// ssbev_conv_* dispatch in conv_mfma.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "ssbev.h"

namespace ssbev_thin {

// K-role channels in {1, 2, 4}, 32 output-role channels (mode 0: forward, mode 1: data gradient)
bool thinin_applicable(const ssbev_conv_dims* d, int mode);
int thinin_pack(const float* w, float* wp, const ssbev_conv_dims* d, int mode, hipStream_t st);
int thinin_launch(const float* x, const float* wp, const float* bias, float* y, const ssbev_conv_dims* d, int mode,
                  hipStream_t st);

// 32 K-role channels, N-role channels in {1, 2, 4}: two-pass kernels with a caller-owned workspace
bool thinout_applicable(const ssbev_conv_dims* d, int mode);
size_t thinout_workspace(const ssbev_conv_dims* d, int mode);
size_t thinout_packed_elems(const ssbev_conv_dims* d, int mode);
int thinout_pack(const float* w, float* wp, const ssbev_conv_dims* d, int mode, hipStream_t st);
int thinout_launch(const float* x, const float* wp, const float* bias, float* y, const ssbev_conv_dims* d, int mode,
                   void* ws, size_t ws_bytes, hipStream_t st);

// weight gradient of both kinds (32 <-> 1 / 2 / 4 channels, unpadded tensors)
bool wgrad_applicable(const ssbev_conv_dims* d);
size_t wgrad_workspace(const ssbev_conv_dims* d);
int wgrad_launch(const float* x, const float* gy, float* gw, const ssbev_conv_dims* d, void* ws, size_t ws_bytes, hipStream_t st);

}  // namespace ssbev_thin
