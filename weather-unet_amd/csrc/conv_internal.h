// internal: production conv3x3 path (conv3x3_mfma_v2.hip), dispatched from wu_conv3x3_fwd
#pragma once
#include <hip/hip_runtime.h>

// ldy: pixel stride of the output (32-bit byte offsets inside one output image, like ldx for the input)
bool conv_v2_eligible(int H, int W, int ldx, int ldy, int Cin, int Cout, int stride, int dtype, bool masked);
// pool != NULL (act must be ReLU, no gate): the epilogue also writes the 2x2 max-pool of y
int conv_v2_launch(const void* x, int ldx, const void* w, const float* bias, void* y, int ldy,
                   const void* egate, int ldegate, int egate_act, int N, int H, int W, int Cin, int Cout, int act, hipStream_t s,
                   void* pool = nullptr, int ldpool = 0,
                   // ReLU gate as bits (wu_conv3x3_fwd_bits): written by a forward with act == RELU / read by a data-gradient pass
                   void* gate_bits_out = nullptr, const void* egate_bits = nullptr,
                   // with pool != NULL and gate_bits_out: also the pool's arg-max bits (same word layout): the forward + pool + bits instance
                   void* sel_bits_out = nullptr,
                   // head_out != NULL (act ReLU, Cout == 64, Cin < 256): the 64 -> 3 pointwise head + tanh from the same epilogue (wu_conv3x3_relu_head_fwd); y may be NULL
                   const float* head_w = nullptr, const float* head_b = nullptr, float* head_out = nullptr);

// Data gradient of a stride-2 conv as FOUR sparse-tap stride-1 convs, one per parity class (p, q) of the input site (conv3x3_mfma.hip):
// site (2i + p, 2j + q) receives forward taps kh in {1} (p = 0) or {0, 2} (p = 1) only, so the classes take 1 / 2 / 2 / 4 of the nine
// tap products -- no zero-stuffed copy of dY, no multiplication by structural zeros.  dy: (N, Cout, Ho, Wo) pre-gated; w_dgrad: the
// rotated 9-slab pack [tap'][Cin][Cout]; dx: (N, Cin, H, W), optionally gated by act'(egate) in the epilogue.
int conv_s2_dgrad_parity_launch(const void* dy, int lddy, const void* w_dgrad, void* dx, int lddx, const void* egate, int ldegate, int egate_act,
                                int N, int H, int W, int Cin, int Cout, int dtype, hipStream_t s);

// 3 x 3 conv, stride 1 or 2 (bf16, unmasked; Cin % 64 == 0, Cout % 128 == 0), on the persistent LDS-DMA GEMM pipeline of the pointwise convs with
// gathered activation rows (resnet.hip, conv1x1_pw3_kernel<.., CONV>).  0 = launched, 1 = not applicable (fall back to conv3x3_mfma_kernel).
int conv3x3_gather_launch(const void* x, int ldx, const void* w_packed, const float* bias, void* y, int ldy, const void* egate, int ldegate, int egate_act,
                          int N, int H, int W, int Cin, int Cout, int stride, int act, hipStream_t s);
// ... and the data gradient of a stride-2 conv (bf16): the four parity classes above as one launch of that pipeline, outputs scattered to their sites.
int conv_s2_dgrad_gather_launch(const void* dy, int lddy, const void* w_dgrad, void* dx, int lddx, const void* egate, int ldegate, int egate_act,
                                int N, int H, int W, int Cin, int Cout, hipStream_t s);

// 3 x 3 conv, stride 1, bf16, on images at most 16 pixels wide (conv3x3_small.hip: 128-pixel x 64-cout tiles, K split between wave pairs, images stacked
// in a tile): the estimator's layer3 / layer4 convs and their data gradient.  0 = launched, 1 = not applicable (fall back to conv3x3_mfma_kernel).
int conv_small_launch(const void* x, int ldx, const void* w_packed, const float* bias, void* y, int ldy, const void* egate, int ldegate, int egate_act,
                      int N, int H, int W, int Cin, int Cout, int act, hipStream_t s,
                      // w_chunked: weights in chunk-major order (wu_conv3x3_small_fwd); mode 0 = launch where the dispatch rule prefers this kernel,
                      // 1 = only answer whether it would, 2 = launch whatever the rule says
                      int w_chunked = 0, int mode = 0);
