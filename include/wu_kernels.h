/*
 * wu_kernels.h -- C ABI of libwu_kernels.so, the MI355X (gfx950) kernels behind the
 * conditional-U-Net hot path of Sota0726/weather-Unet.
 *
 * The reference has no native code: every entry point below replaces a torch.nn op that the
 * reference's Python modules call (file:line cited per function).  The reference-side binding
 * is the ctypes stub shown in INTEGRATION.md (weather-unet_amd/wu/_lib.py is that stub).
 *
 * Conventions
 *   - All pointers are DEVICE pointers unless said otherwise.  The library never allocates,
 *     frees or synchronises: buffers and workspaces are owned by the caller and only borrowed
 *     for the duration of the launch; every call only enqueues work on `stream`
 *     (a hipStream_t passed as void*), so calls are graph-capturable and re-entrant.
 *   - Activations are NHWC ("channels-last"): element (n,h,w,c) of a tensor with pixel stride
 *     `ld` (in elements, >= C, multiple of 8) lives at ((n*H + h)*W + w)*ld + c.  A channel
 *     slice of a wider buffer is passed as (ptr + c0, ld = total channels); that is how the
 *     skip-concat (cunet.py:62,69,76) is done without a copy.
 *   - `dtype`: WU_F32 (0) = fp32 storage, exact-fp32 MFMA;  WU_BF16 (1) = bf16 storage,
 *     bf16 MFMA with fp32 accumulation.  Parameters, their gradients, statistics are fp32.
 *   - Network input / output images are NCHW fp32 (the reference's layout).
 *   - Return value: 0 on success, a hipError_t (>0) from the launch, or < 0 for a rejected
 *     argument (shape / alignment); wu_last_error() gives a host string for the last failure
 *     on the calling thread.
 */
#ifndef WU_KERNELS_H
#define WU_KERNELS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WU_F32 0
#define WU_BF16 1

#define WU_ACT_NONE 0
#define WU_ACT_RELU 1   /* nets.py:21,23 nn.ReLU */
#define WU_ACT_LEAKY 2  /* nets.py:32 nn.LeakyReLU(0.2) */

const char* wu_last_error(void);
int wu_version(void);
/* Compute units of the current HIP device (256 on MI355X): the persistent conv / weight-gradient grids are sized from it. */
int wu_cu_count(void);
/* Kernel-variant switches for in-process A/B benchmarking -- DIAGNOSTIC / TEST-ONLY: process-global mutable state, not part of
 * the re-entrant launcher contract.  A production caller never touches it: the defaults ARE the production
 * choices and every variant computes identical results; set it only from a single-threaded benchmark harness, between launches.
 *   0: conv LDS-DMA path (0 off = generic template, 1 auto wave shape (default), 2 always 4 waves, 3 always 8 waves)
 *   1: persistent tile loop on/off      2: LDS-DMA wgrad (0 off, 1 = 8 waves (default), 2 = 4 waves)
 *   3: stride-1 bf16 convs on images at most 16 pixels wide on the small-image kernel (128-pixel tiles, in-workgroup K split, stacked images; default 1 = where the generic template's 256-pixel tiles
 *      would under-fill the chip or lie mostly outside the image, 2 = always, 0 = never)
 *   4: wgrad DMA issue spread over K-steps (default 1)   5: first conv (0 matrix cores in bf16 (default), 1 rows kernel, 2 VALU kernel)
 *   6: static priority for the younger wave half (default 1)   7: grid-strided tile assignment (default 1)
 *   8: AdaIN-upsample backward (0 = 16-tap gather, 1 = default: bf16 with keep bits streams through an LDS ring, everything else marches
 *      with the columns per thread chosen by size; 2 / 3 = marching with one / two columns, 5 = marching also where the ring kernel would run)
 *   9: marching AdaIN-upsample forward (default 1)
 *   10: persistent-grid size override in compute units (0 = the device's count; experiments on CU-masked streams)
 *   11: Cin = 64 / one-cout-tile convs keep both weight chunks resident in LDS across a workgroup's tiles (default 1)
 *   12: pointwise-GEMM pixel tile (0 = 64 rows (default), 1 = 256 rows, 2 = 128 rows)
 *   13: image-layout 3 -> 3 conv (SNDisc's first layer): 1 = LDS-tiled kernels (default), 0 = one thread per pixel
 *   14: stride-2 data gradient: 1 = four parity-class sparse-tap convs in one launch (default), 0 = zero-stuffed dY + stride-1 conv
 *   15: persistent LDS-DMA GEMM pipeline (round 4: pointwise convs with 128 x 128 / 128 x 64 tiles; stride-2 3x3 convs and the parity classes of
 *       their data gradient with gathered rows).  Bits 0-2 = ring depth D (0 = off: the register-staged kernels everywhere), bit 3 = eight waves
 *       per workgroup (else four), bit 4 = 64-cout tiles also for the 3x3 forms, bits 5.. = the least number of tiles that takes this path.
 *       Default 2 + 8 + (128 << 5): D = 2, eight waves, two workgroups per CU, from 128 tiles. */
int wu_set_option(int key, int value);
/* Diagnostic: device buffer of 256*8*8 uint64 receiving per-wave phase cycle sums of the persistent conv / wgrad kernels
 * (DMA wait, compute, whole-kernel s_memtime and s_memrealtime deltas -> in-kernel clock, barrier, epilogue, tiles, chunks);
 * NULL (default) disables stamping. */
int wu_set_debug_buffer(void* p);
/* Stream ordering inside one device without a system-scope fence (host plumbing of the fused backward, wu/unet_graph.py: the
 * weight-gradient hand-offs between its two streams).  wu_stream_order_after(waiter, producer, event): work enqueued on `waiter` after
 * the call runs after everything enqueued on `producer` before it; `event` (wu_event_create: no timing, hipEventDisableSystemFence) may
 * be reused for every call.  Nothing in the reference corresponds to this (autograd there runs on one stream). */
int wu_event_create(void** event_out);
int wu_event_destroy(void* event);
int wu_stream_order_after(void* waiter_stream, void* producer_stream, void* event);

/* Experiment support (DIAGNOSTIC): a HIP stream confined to the compute units whose bits are set in mask[0..words) (hipExtStreamCreateWithCUMask),
 * for measuring CU-partitioned overlap of the data-gradient and weight-gradient kernels (scratch/ab_cumask.py); unused by the product path. */
int wu_stream_create_cu_mask(const unsigned* mask, int words, void** stream_out);
int wu_stream_destroy(void* stream);

/* ---- weights -------------------------------------------------------------------------------
 * Repack one 3x3 conv weight (nets.py:20,22,28-31; OIHW fp32, the state-dict layout) into the two
 * MFMA operand layouts: w_fwd[tap][Cout][Cin] and w_dgrad[tap'][Cin][Cout] with tap' the
 * 180-degree-rotated tap (the data-gradient is a correlation with the flipped, transposed
 * filter).  `inv_sigma` (device scalar, may be NULL) multiplies every weight: the spectral-norm
 * division W/sigma of torch.nn.utils.spectral_norm (nets.py:27-31).  Either output may be NULL. */
int wu_pack_conv3x3(const float* w_oihw, void* w_fwd, void* w_dgrad, int Cout, int Cin,
                    const float* inv_sigma, int dtype, void* stream);
/* wu_pack_conv3x3 (no spectral-norm factor) for n <= 16 weights in one launch: entry i packs w_oihw[i] (Cout[i] x Cin[i] x 3 x 3)
 * into w_fwd[i] and w_dgrad[i].  The pointer / size arrays live in HOST memory and are consumed before the call returns. */
int wu_pack_conv3x3_multi(int n, const float* const* w_oihw, void* const* w_fwd, void* const* w_dgrad,
                          const int* Cout, const int* Cin, int dtype, void* stream);

/* Spectral normalisation of a conv weight (torch.nn.utils.spectral_norm around the convs of nets.py:28-31):
 * one power iteration (power_iter != 0: v <- normalize(W^T u), u <- normalize(W v), buffers updated in place),
 * sigma = u . (W v); sigma_out[0] = sigma, sigma_out[1] = 1/sigma; w_eff (may be NULL) = W / sigma.  W is the OIHW
 * fp32 weight viewed as rows = Cout, cols = Cin*9.  Backward: dw = g/sigma - (<g,w>/sigma^2) u v^T with u, v the
 * buffers the forward used (constants of the graph, as in torch).  scratch: wu_spectral_norm_scratch_floats(). */
size_t wu_spectral_norm_scratch_floats(int rows, int cols);
int wu_spectral_norm_fwd(const float* w, int rows, int cols, float* u, float* v, int power_iter, float eps,
                         float* sigma_out, float* w_eff, float* scratch, void* stream);
int wu_spectral_norm_bwd(const float* g, const float* w, const float* u, const float* v, const float* sigma,
                         float* dw, int rows, int cols, float* scratch, void* stream);
/* The same for n <= 16 weights per call -- all ten SN layers of SNDisc (disc.py:11-24: eight convs, `l`, `embed`) in 5 launches forward
 * and 2 backward instead of 5 n / 2 n.  Arrays of n entries in HOST memory (read during the call), entry i as in the single-weight
 * call; results are bit-identical to n single calls.  u_save / v_save (the arrays and their entries may be NULL): copies of the u, v
 * that define sigma_out[i], for the backward pass (the buffers advance on the next forward). */
int wu_spectral_norm_fwd_multi(int n, const float* const* w, const int* rows, const int* cols, float* const* u, float* const* v,
                               int power_iter, float eps, float* const* sigma_out, float* const* w_eff, float* const* scratch,
                               float* const* u_save, float* const* v_save, void* stream);
int wu_spectral_norm_bwd_multi(int n, const float* const* g, const float* const* w, const float* const* u, const float* const* v,
                               const float* const* sigma, float* const* dw, const int* rows, const int* cols, float* const* scratch,
                               void* stream);

/* ---- conv3x3, pad 1, MFMA implicit GEMM -----------------------------------------------------
 * y = act(conv3x3(x, w) + bias)   replaces nn.Conv2d(cin,cout,3,padding=1[,stride=2]) + ReLU /
 * LeakyReLU of nets.py:18-33.  x: (N,H,W,Cin) ld=ldx; y: (N,Ho,Wo,Cout) ld=ldy with
 * Ho = (H-1)/stride + 1.  Cin % (64/sizeof(T)) == 0, Cout % 64 == 0.  bias may be NULL.
 * If `mask` != NULL the staged input is gated by the activation derivative of `mask`
 * (same geometry as x, ld=ldmask):  x * (mask > 0 ? 1 : slope(mask_act)) -- this is how the
 * data-gradient pass (called with w_dgrad, Cin<->Cout swapped, stride 1) fuses the ReLU backward
 * of autograd (t_cls_train.py:272,307) into its input gather.
 * If `egate` != NULL the OUTPUT is multiplied by act'(egate) in the epilogue (egate: output geometry,
 * ld=ldegate): used by the data-gradient pass to hand the upstream layer a gradient that is already gated
 * by that layer's own activation (egate = this conv's forward input, which is that layer's output). */
int wu_conv3x3_fwd(const void* x, int ldx, const void* w_packed, const float* bias, void* y, int ldy,
                   int N, int H, int W, int Cin, int Cout, int stride, int act,
                   const void* mask, int ldmask, int mask_act,
                   const void* egate, int ldegate, int egate_act, int dtype, void* stream);

/* "Gate bits": the ReLU gate of an activation tensor y (N,H,W,C; C % 64 == 0) as one bit per element instead of the tensor,
 *     uint32 bits[N*H*W][C/64][2];   bit (8k + i) of bits[p][ct][hf]  =  y[p][64 ct + 16 k + 8 hf + i] > 0   (k < 4, i < 8)
 * -- the 32 channels one lane of the conv epilogue holds, so a gated epilogue reads one dword per lane and row where the gate
 * tensor costs four 16-byte loads (the gated data-gradient convs of the 64-channel 256x256 layers are HBM-bound: 805 -> 554 MB).
 * wu_conv3x3_fwd_bits is wu_conv3x3_fwd (stride 1, bf16) with either
 *   gate_bits_out != NULL: act must be WU_ACT_RELU; the bits of the output are written next to y (forward of a block's first conv), or
 *   egate_bits   != NULL: act = NONE, bias = NULL; the output is multiplied by the gate the bits encode (data-gradient pass of the
 *                          block's second conv; egate_bits has the OUTPUT geometry, Cout channels).
 * Only the LDS-DMA kernel has this epilogue: wu_conv3x3_gate_bits_supported(...) != 0 must hold for the shape (else use the
 * gate-tensor form of wu_conv3x3_fwd).  wu_gate_bits_bytes: size of a bits buffer. */
int wu_conv3x3_gate_bits_supported(int H, int W, int ldx, int ldy, int Cin, int Cout, int dtype);
size_t wu_gate_bits_bytes(int N, int H, int W, int C);
int wu_conv3x3_fwd_bits(const void* x, int ldx, const void* w_packed, const float* bias, void* y, int ldy,
                        void* gate_bits_out, const void* egate_bits, int N, int H, int W, int Cin, int Cout,
                        int act, int dtype, void* stream);

/* y = ReLU(conv3x3(x) + bias) AND pool = max_pool2d(y, 2) (floor: [N][H/2][W/2][Cout], pixel stride ldpool) in one call
 * (cunet.py:45-46, 49-50, 53-54).  On the bf16 LDS-DMA path the conv epilogue writes the pooled tensor itself; otherwise the
 * conv is followed by wu_maxpool2_fwd.  Same argument rules as wu_conv3x3_fwd (stride 1); H and W even. */
int wu_conv3x3_relu_pool_fwd(const void* x, int ldx, const void* w_packed, const float* bias, void* y, int ldy,
                             void* pool, int ldpool, int N, int H, int W, int Cin, int Cout, int dtype, void* stream);
/* The same with TWO bits per element of y from the epilogue (round 4; LDS-DMA path: ask wu_conv3x3_gate_bits_supported): the ReLU gate
 * (gate-bit layout, above) and, in the same layout, "this element is the FIRST maximum of its 2x2 window" (scan order (0,0), (0,1), (1,0),
 * (1,1): torch.nn.MaxPool2d's rule).  wu_maxpool2_bwd_bits computes MaxPool2d's backward fused with the skip-gradient sum and the ReLU
 * gate (cunet.py:46,49,52 in backward) from those bits instead of the activation tensor. */
int wu_conv3x3_relu_pool_bits_fwd(const void* x, int ldx, const void* w_packed, const float* bias, void* y, int ldy,
                                  void* pool, int ldpool, void* gate_bits_out, void* sel_bits_out,
                                  int N, int H, int W, int Cin, int Cout, int dtype, void* stream);
/* 3x3 conv, stride 1, bf16, on images at most 16 pixels wide with CHUNK-MAJOR weights (round 4; the frozen estimator's layer3 / layer4 convs and their
 * data gradient, classifier.py:106): w_chunked[Cin / 32][9][Cout][32] = the pack of wu_pack_conv3x3 ([9][Cout][Cin]) viewed as [9][Cout][Cin / 32][32] and
 * permuted (2, 0, 1, 3), so that a chunk's tap slab is 4 KiB contiguous (the small-image kernel is bound by the bytes it pulls out of L2).  y = act(conv(x) +
 * bias) [* act'(egate)] exactly as wu_conv3x3_fwd computes it on the small-image kernel (option 3).  wu_conv3x3_small_supported: 1 = the shape runs there
 * and the dispatch rule prefers it to the generic template; otherwise call wu_conv3x3_fwd with the ordinary pack. */
int wu_conv3x3_small_supported(int N, int H, int W, int ldx, int ldy, int ldegate, int Cin, int Cout);
int wu_conv3x3_small_fwd(const void* x, int ldx, const void* w_chunked, const float* bias, void* y, int ldy,
                         const void* egate, int ldegate, int egate_act, int N, int H, int W, int Cin, int Cout, int act, void* stream);

/* The last decoder conv AND the network's head in one launch (round 4; cunet.py:78-82: dconv_up1[2] + ReLU, then tanh(conv_last(y))):
 * y = ReLU(conv3x3(x) + bias) as above and out_nchw[N][3][H][W] (fp32) = tanh(head_w[3][64] . y + head_bias), the head computed on the matrix
 * cores from the conv epilogue's packed registers (head weights split into three bf16 terms: fp32-exact products of the STORED
 * bf16 y).  y == NULL: the 64-channel tensor is not written at all (inference; ldy ignored).  bf16 LDS-DMA path only, Cout == 64,
 * 64 <= Cin < 256, Cin % 32 == 0: ask wu_conv3x3_relu_head_supported (1 = yes); otherwise call wu_conv3x3_fwd + wu_conv1x1_tanh_fwd. */
int wu_conv3x3_relu_head_supported(int H, int W, int ldx, int ldy, int Cin, int Cout, int dtype);
int wu_conv3x3_relu_head_fwd(const void* x, int ldx, const void* w_packed, const float* bias, void* y, int ldy,
                             const float* head_w, const float* head_bias, float* out_nchw,
                             int N, int H, int W, int Cin, int Cout, int dtype, void* stream);
int wu_maxpool2_bwd_bits(const unsigned* gate_bits, const unsigned* sel_bits, const void* dy, int lddy, const void* dskip, int lddskip,
                         void* dx, int lddx, int N, int H, int W, int C, int dtype, void* stream);

/* Weight + bias gradient of the conv above: dw_oihw[Cout][Cin][3][3] (+)= sum_pixels dy (x) x,
 * dbias[Cout] (+)= sum dy, with dy gated by act'(y) when `y` != NULL.  `workspace` must hold
 * wu_conv3x3_wgrad_workspace() bytes (deterministic split-K slabs).  accumulate != 0 adds into
 * dw/dbias (autograd's .grad accumulation), else overwrites. */
size_t wu_conv3x3_wgrad_workspace(int N, int H, int W, int Cin, int Cout, int stride, int dtype);
int wu_conv3x3_wgrad(const void* x, int ldx, const void* dy, int lddy, const void* y, int ldy_, int act,
                     float* dw_oihw, float* dbias, void* workspace, size_t workspace_bytes,
                     int N, int H, int W, int Cin, int Cout, int stride, int accumulate,
                     int dtype, void* stream);

/* Data gradient of the stride-2 conv (nets.py:30-31): dx (N,H,W,Cin) from dy (N,Ho,Wo,Cout),
 * w_dgrad as packed above; dy gated by act'(y) when y != NULL.  `workspace` holds
 * wu_conv3x3_s2_dgrad_workspace() bytes (the zero-upsampled gradient); egate as in wu_conv3x3_fwd. */
size_t wu_conv3x3_s2_dgrad_workspace(int N, int H, int W, int Cout, int dtype);
int wu_conv3x3_s2_dgrad(const void* dy, int lddy, const void* y, int ldy_, int act, const void* w_dgrad,
                        void* dx, int lddx, void* workspace, size_t workspace_bytes,
                        const void* egate, int ldegate, int egate_act,
                        int N, int H, int W, int Cin, int Cout, int dtype, void* stream);

/* Activation backward as one streaming pass: out = g * act'(y) (ReLU: y > 0; LeakyReLU: y > 0 ? 1 : 0.2),
 * all three NHWC with their own pixel strides; `out` may alias `g`.  Lets both gradient GEMMs of a conv
 * consume a pre-gated gradient (y == NULL / mask == NULL paths), which is what the LDS-DMA wgrad needs. */
int wu_act_gate(const void* g, int ldg, const void* y, int ldy, void* out, int ldo,
                int N, int H, int W, int C, int act, int dtype, void* stream);

/* ---- thin layers (HBM-bound, no MFMA) -------------------------------------------------------
 * First conv of dconv_down1 / of the discriminator: Cin = 3 read straight from the NCHW fp32
 * image (cunet.py:45 via nets.py:20; disc.py:28 via nets.py:28-31).
 * out_nchw == 0: y is NHWC `dtype` with ld=ldy (Cout % 8 == 0);  out_nchw != 0: y is NCHW fp32
 * (the 3->3 SN conv of disc.conv1[0]).  w is OIHW fp32 (Cout,3,3,3) scaled by *inv_sigma if given. */
int wu_conv3x3_c3_fwd(const float* x_nchw, const float* w_oihw, const float* bias, const float* inv_sigma,
                      void* y, int ldy, int out_nchw, int N, int H, int W, int Cout, int stride, int act,
                      int dtype, void* stream);
/* wu_conv3x3_c3_fwd with NHWC output, Cout = 64, ReLU, that also writes the gate bits of its output (see "gate bits" above;
 * the matrix-core form only: wu_conv3x3_c3_gate_bits_supported(...) != 0). */
int wu_conv3x3_c3_gate_bits_supported(int N, int H, int W, int Cout, int stride, const float* bias, int dtype);
int wu_conv3x3_c3_fwd_bits(const float* x_nchw, const float* w_oihw, const float* bias, const float* inv_sigma,
                           void* y, int ldy, void* gate_bits_out, int N, int H, int W, int Cout, int stride, int dtype, void* stream);
/* its weight/bias gradient (the image needs no data gradient in the generator); dy NHWC or NCHW fp32 */
int wu_conv3x3_c3_wgrad(const float* x_nchw, const void* dy, int lddy, int dy_nchw, const void* y, int ldy_,
                        int act, float* dw_oihw, float* dbias, void* workspace, size_t workspace_bytes,
                        int N, int H, int W, int Cout, int stride, int accumulate, int dtype, void* stream);
/* Bytes of caller-owned scratch that make the two thin-layer weight gradients (wu_conv3x3_c3_wgrad on its bf16 64-channel
 * path, wu_conv1x1_tanh_bwd) bitwise reproducible: per-workgroup partial sums land there and are folded in a fixed order.
 * Passing workspace = NULL (or a smaller size) selects fp32 atomics across workgroups instead (last-bit run-to-run noise). */
size_t wu_thin_workspace_bytes(void);
/* data gradient wrt the NCHW fp32 image (needed when D is differentiated wrt G's output,
 * t_cls_train.py:243,272): dx_nchw (N,3,H,W) fp32. */
int wu_conv3x3_c3_dgrad(const void* dy, int lddy, int dy_nchw, const void* y, int ldy_, int act,
                        const float* w_oihw, const float* inv_sigma, float* dx_nchw, int N, int H, int W,
                        int Cout, int stride, int accumulate, int dtype, void* stream);

/* conv_last + Tanh (cunet.py:39-40,80-82): out_nchw (N,3,H,W) fp32 = tanh(W x + b),
 * x NHWC (N,H,W,Cin) ld=ldx, w (3,Cin) fp32. */
int wu_conv1x1_tanh_fwd(const void* x, int ldx, const float* w, const float* bias, float* out_nchw,
                        int N, int H, int W, int Cin, int dtype, void* stream);
/* backward: g = dout*(1-out^2); dx = W^T g (NHWC ld=lddx) [* act'(x) if x_gate_act]; dw (+)= g x^T; dbias (+)= sum g. */
int wu_conv1x1_tanh_bwd(const float* dout_nchw, const float* out_nchw, const void* x, int ldx, const float* w,
                        void* dx, int lddx, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                        int N, int H, int W, int Cin, int accumulate, int x_gate_act, int dtype, void* stream);

/* ---- glue ------------------------------------------------------------------------------------
 * nn.MaxPool2d(2) (cunet.py:27; calls :46,49,52).  x (N,H,W,C) -> y (N,H/2,W/2,C). */
int wu_maxpool2_fwd(const void* x, int ldx, void* y, int ldy, int N, int H, int W, int C, int dtype, void* stream);
/* dx = route(dy to the first arg-max of each window, PyTorch's tie rule) [+ dskip]:
 * `dskip` (may be NULL, ld=lddskip) is the gradient arriving over the skip connection of the same
 * tensor (cunet.py:62,69,76), summed here instead of by a separate autograd add.  gate_act != 0: the
 * result is additionally multiplied by act'(x) (x is the pooled conv's activation output). */
int wu_maxpool2_bwd(const void* x, int ldx, const void* dy, int lddy, const void* dskip, int lddskip,
                    void* dx, int lddx, int N, int H, int W, int C, int gate_act, int dtype, void* stream);

/* AdaIN style statistics (utils.py:41-48): y_ = l1(y).view(N, C, 4) with l1 = Linear(nc, 4C) (w: [4C][nc] fp32, b: [4C] or NULL);
 * y_mean[n][c] = mean_k y_[n][c][k], y_std[n][c] = sqrt(unbiased var_k + eps).  `y4` (may be NULL; N*C*4 floats, 16-B aligned)
 * receives y_ for the backward pass.  One launch instead of ~8 stock kernels; nc <= 32. */
int wu_adain_style_fwd(const float* y, const float* w, const float* b, float eps, float* y_std, float* y_mean, float* y4,
                       int N, int C, int nc, void* stream);
/* Gradients of the above wrt l1.weight (dw [4C][nc]) and l1.bias (db [4C], may be NULL) from d_std, d_mean [N][C]
 * (autograd of utils.py:41-48); fixed summation order over n.  accumulate != 0 adds into dw / db. */
int wu_adain_style_bwd(const float* d_std, const float* d_mean, const float* y, const float* y4, const float* y_std,
                       const float* y_mean, float* dw, float* db, int N, int C, int nc, int accumulate, void* stream);

/* The same two calls for up to 4 AdaIN layers that share the conditioning input y -- the three decoder levels of a U-Net pass (cunet.py:59,66,73) --
 * in ONE launch each (round 4).  Host arrays of `levels` entries: w / b / y_std / y_mean / y4 / dw / db pointers (b[i], y4[i], db[i] may be NULL), eps, C.
 * Per level the arithmetic is the single call's: results are bit-identical. */
int wu_adain_style_fwd_multi(int levels, const float* y, const float* const* w, const float* const* b, const float* eps,
                             float* const* y_std, float* const* y_mean, float* const* y4, int N, const int* C, int nc, void* stream);
int wu_adain_style_bwd_multi(int levels, const float* const* d_std, const float* const* d_mean, const float* y, const float* const* y4,
                             const float* const* y_std, const float* const* y_mean, float* const* dw, float* const* db,
                             int N, const int* C, int nc, int accumulate, void* stream);

/* AdaIN instance statistics (utils.py:34-39,47): per (n,c) over H*W: stats[n][c] = {mean, rstd}
 * with rstd = 1/sqrt(unbiased_var + eps).  `scratch` holds N*C*2*WU_MAX_SPLITS floats (per-split partial
 * sums, folded in fixed order: results are bitwise reproducible). */
#define WU_MAX_SPLITS 16
int wu_adain_stats(const void* x, int ldx, float* stats, float* scratch, int N, int H, int W, int C,
                   float eps, int dtype, void* stream);

/* Fused AdaIN-apply (utils.py:49-50) -> bilinear x2 align_corners=True (cunet.py:26,60,67,74) ->
 * Dropout(p) (cunet.py:28,61,68,75) written into channels [0,C) of the concat buffer `y`
 * (N,2H,2W,*) ld=ldy; the skip tensor already lives in channels [C, ...) (torch.cat, cunet.py:62).
 * y_std / y_mean: (N,C) fp32 style statistics of utils.py:46,48.  p_drop == 0 -> eval mode.
 * Dropout keep-mask = counter RNG(seed, element index); keep scale 1/(1-p).  `mask_bits` (may be NULL):
 * N*2H*2W*(C / elements-per-16-B) bytes receiving one keep-bit per element, so the backward pass reads the
 * mask instead of re-hashing (pass the same pointer, or NULL to regenerate from the seed).
 * `seed_dev` (may be NULL): device-resident uint64 added to `seed` inside the kernel -- a captured hipGraph freezes the
 * `seed` argument, the counter it points to can be bumped between replays (the reference's inference loops run with
 * Dropout ACTIVE: inference/inf_transfer_c.py:88-96 never calls .eval()).
 * mask_is_input != 0: `mask_bits` is READ instead of drawn (an externally supplied keep-mask, e.g. one captured from the
 * reference's own nn.Dropout, for parity tests and replays of a recorded step). */
int wu_adain_upcat_fwd(const void* x, int ldx, const float* stats, const float* y_std, const float* y_mean,
                       void* y, int ldy, int N, int H, int W, int C, float p_drop, uint64_t seed,
                       const uint64_t* seed_dev, uint8_t* mask_bits, int mask_is_input, int dtype, void* stream);
/* Backward of the above.  dy: gradient of the concat buffer channels [0,C) (N,2H,2W) ld=lddy.
 * Produces dx (N,H,W,C) ld=lddx and d_y_std, d_y_mean (N,C) fp32.  `gtmp` (N*H*W*C elements of `dtype`) and
 * `sums` (N*C*2*(1+WU_MAX_SPLITS) floats) are caller-provided scratch.  x_gate_act != 0: dx is additionally multiplied by
 * act'(x) (x is a conv activation output; the producer conv then receives a pre-gated gradient). */
int wu_adain_upcat_bwd(const void* dy, int lddy, const void* x, int ldx, const float* stats, const float* y_std,
                       void* dx, int lddx, float* d_y_std, float* d_y_mean, void* gtmp, float* sums,
                       int N, int H, int W, int C, float p_drop, uint64_t seed, const uint8_t* mask_bits,
                       int x_gate_act, int dtype, void* stream);
/* The keep-mask wu_adain_upcat_fwd draws for (seed, p): mask[n][c][h2][w2] (NCHW uint8), for tests. */
int wu_dropout_mask(uint8_t* mask_nchw, int N, int H2, int W2, int C, float p_drop, uint64_t seed, void* stream);

/* Mean absolute error between two fp32 tensors of n elements (reference ops.py:22-24 l1_loss = F.l1_loss): *loss = mean|a - b|
 * and, when grad != NULL, grad[i] = sign(a[i] - b[i]) / n (the gradient wrt a for an upstream gradient of 1), in one pass.
 * scratch: wu_l1_mean_scratch_floats() floats of per-workgroup partial sums, folded in index order (deterministic). */
size_t wu_l1_mean_scratch_floats(void);
int wu_l1_mean(const float* a, const float* b, float* grad, float* scratch, float* loss, long long n, void* stream);

/* Discriminator head (disc.py:32): feat[n][c] = sum_{h,w} x[n,h,w,c]  (fp32), and its backward
 * dx[n,h,w,c] = dfeat[n][c]. */
int wu_sumpool_fwd(const void* x, int ldx, float* feat, int N, int H, int W, int C, int dtype, void* stream);
int wu_sumpool_bwd(const float* dfeat, void* dx, int lddx, int N, int H, int W, int C, int dtype, void* stream);

/* ---- frozen ResNet-101 estimator in the GAN loop -------------------------------------------------------------------------
 * The reference runs torchvision.models.resnet101 (classifier.py:106-112, estimator.py:143-151) four times per iteration and
 * differentiates it once wrt its input (t_cls_train.py:237,247-250,297,424); it is frozen and in eval mode, so every BatchNorm
 * is a per-channel affine that the caller folds into the conv weight / bias.  The 3x3 convs of the Bottleneck blocks use
 * wu_conv3x3_fwd / wu_conv3x3_s2_dgrad above; these are the remaining ops.
 *
 * 1x1 conv as a GEMM on the matrix cores.  One GEMM row per point (n, hc, wc) of a COARSE grid N x Hc x Wc:
 *     in  = x[n, hc*in_stride, wc*in_stride, :]      (x: N x Hin x Win x Cin,  pixel stride ldx)
 *     out = y[n, hc*out_stride, wc*out_stride, :]    (y: N x Hout x Wout x Cout, pixel stride ldy)
 *     out = act(w . in + bias + residual) * act'(egate)          w: [Cout][Cin] in `dtype`, bias fp32 (may be NULL)
 * in_stride = 2 is the downsample conv (nn.Conv2d(cin, cout, 1, stride=2)); out_stride = 2 is its data gradient (called with
 * the transposed weight): the other pixels of each 2x2 output block receive act(residual) * act'(egate), or zero.
 * `residual` / `egate` (may be NULL) have y's geometry (pixel strides ldres / ldegate).  Cin % (128 / sizeof(T)) == 0,
 * Cout % 64 == 0. */
int wu_conv1x1_fwd(const void* x, int ldx, const void* w, const float* bias, const void* residual, int ldres,
                   void* y, int ldy, int N, int Hc, int Wc, int in_stride, int Hin, int Win,
                   int out_stride, int Hout, int Wout, int Cin, int Cout, int act,
                   const void* egate, int ldegate, int egate_act, int dtype, void* stream);
/* Two chained 1x1 convs in ONE launch (bf16): y1 = act1(wa . x + bias_a + res) * act'(gate1), y2 = act2(wb . y1 + bias_b) * act'(gate2),
 * both stored.  x: [M][K1] (pixel stride ldx); wa = the [C1][K1] weight and wb = the [C2][C1] weight in MFMA-FRAGMENT ORDER
 * [cout / 32][k / 16][k half (2)][cout row (32)][8 elements] (a wave's 16-byte A fragments of one (cout block, K step) are 1 KiB
 * contiguous: weights are streamed straight into registers); res / gate1 with y1's geometry, gate2 with y2's; any of bias_a, bias_b,
 * res, gate1, gate2 may be NULL.  Forward: torchvision Bottleneck conv3 + bn3 + residual + ReLU of one block followed
 * by conv1 + bn1 + ReLU of the next (classifier.py:106-112 / estimator.py:143-151 build resnet101); backward: conv1^T of a block (+ the
 * identity-path gradient, gated by the block boundary's ReLU) followed by conv3^T of the previous block (gated by its 3x3 conv's ReLU).
 * Bit-identical to two wu_conv1x1_fwd calls.  Shapes: wu_conv1x1_chain_supported(K1, C1, C2, dtype) != 0. */
int wu_conv1x1_chain_supported(int K1, int C1, int C2, int dtype);
int wu_conv1x1_chain(const void* x, int ldx, const void* wa, const float* bias_a, const void* res, int ldres, int act1,
                     const void* gate1, int ldg1, int gate1_act, void* y1, int ldy1,
                     const void* wb, const float* bias_b, int act2, const void* gate2, int ldg2, int gate2_act, void* y2, int ldy2,
                     long long M, int K1, int C1, int C2, int dtype, void* stream);
/* Stem: nn.Conv2d(3, 64, 7, stride=2, padding=3) + folded BN + ReLU from the NCHW fp32 image to NHWC `dtype`
 * (N, Ho, Wo, 64), Ho = (H-1)/2 + 1; w: OIHW fp32 [64][3][7][7], bias [64] (may be NULL). */
int wu_stem7x7_fwd(const float* x_nchw, const float* w_oihw, const float* bias, void* y, int ldy,
                   int N, int H, int W, int act, int dtype, void* stream);
/* its data gradient wrt the image (g_loss flows through estimator(fake_out) into the generator, t_cls_train.py:247-250,272):
 * dx_nchw (N,3,H,W) fp32 (+)= conv_transpose(dy); dy (N,Ho,Wo,64) already gated by the stem's ReLU. */
int wu_stem7x7_dgrad(const void* dy, int lddy, const float* w_oihw, float* dx_nchw, int N, int H, int W,
                     int accumulate, int dtype, void* stream);
/* nn.MaxPool2d(kernel_size=3, stride=2, padding=1): x (N,H,W,C) -> y (N,Ho,Wo,C); `argmax` (may be NULL; N*Ho*Wo*C bytes)
 * receives the window-local index 0..8 of the FIRST maximum (PyTorch's tie rule) for the backward pass. */
int wu_maxpool3s2_fwd(const void* x, int ldx, void* y, int ldy, uint8_t* argmax, int N, int H, int W, int C,
                      int dtype, void* stream);
/* dx (N,H,W,C) = sum over the (overlapping) windows whose arg-max is this pixel of dy; gate_act != 0: * act'(x). */
int wu_maxpool3s2_bwd(const void* dy, int lddy, const uint8_t* argmax, const void* x, int ldx, void* dx, int lddx,
                      int N, int H, int W, int C, int gate_act, int dtype, void* stream);

/* ---- input pipeline (t_cls_train.py:81-108: the torchvision / PIL transforms every training image goes through) ----------
 * A batch of decoded RGB images lives in one uint8 buffer `src`; image n starts at byte geo[n].src_off, is src_h x src_w pixels
 * with a row stride of src_ld pixels.  `geo` is an array of N records of wu_image_geo_bytes() (= 72) bytes:
 *     int64 src_off; int32 src_h, src_w, src_ld, crop_top, crop_left, crop_h, crop_w, flip, rot[6], do_rot, pad;
 * The crop window is resized to S x S with Pillow's two-pass fixed-point bilinear resample (transforms.Resize: the window is the
 * whole image; RandomResizedCrop: the drawn box); rot[] are the 16.16 fixed-point coefficients of Image.rotate(angle, NEAREST)
 * (libImaging/Geometry.c affine_fixed) -- for the S x S image when rot_first == 0 (Resize -> RandomRotation, :96-97), for the
 * source image when rot_first != 0 (RandomRotation -> RandomResizedCrop, :83-84); flip = RandomHorizontalFlip.  Results are
 * bit-identical to Pillow's.  dst_u8 (N,S,S,3) and / or dst_nchw (N,3,S,S fp32, ToTensor + Normalize(0.5, 0.5)) may be NULL.
 * ksize >= 2 * ceil(max(1, crop / S)) + 1 over the batch; `workspace` holds wu_image_workspace_bytes(N, S, ksize) bytes. */
size_t wu_image_geo_bytes(void);
size_t wu_image_workspace_bytes(int N, int S, int ksize);
int wu_image_geometry(const uint8_t* src, const void* geo, void* workspace, size_t workspace_bytes,
                      uint8_t* dst_u8, float* dst_nchw, int N, int S, int ksize, int rot_first, void* stream);
/* transforms.ColorJitter(brightness, contrast, saturation, hue=0) in place on the (N,S,S,3) uint8 batch: for image n the ops
 * order[n][0..2] (0 brightness, 1 contrast, 2 saturation, -1 none) with factors[n][op], each Image.blend(degenerate, image, f)
 * with Pillow's arithmetic; then (dst_nchw != NULL) ToTensor + Normalize(0.5, 0.5) into (N,3,S,S) fp32. */
int wu_image_color_jitter(uint8_t* img_u8, const float* factors, const int* order, float* dst_nchw, int N, int S, void* stream);

/* layout helpers for the module boundary: NHWC `dtype` <-> NCHW fp32 (feature maps returned by
 * SNDisc.forward, disc.py:38; gradients flowing back into them). */
int wu_nhwc_to_nchw_f32(const void* x, int ldx, float* y_nchw, int N, int H, int W, int C, int dtype, void* stream);
int wu_nchw_f32_to_nhwc(const float* x_nchw, void* y, int ldy, int N, int H, int W, int C, int dtype, void* stream);

/* ---- instrumentation (bench.py roofline leg) -------------------------------------------------
 * While enabled, the launchers of the kernel families in `family_mask` (bit f = family f) bracket each
 * launch with hipEvents on the launch stream (events come from a pool created by wu_prof_begin; nothing
 * is allocated per launch).  wu_prof_query synchronises on a family's events and returns its number of
 * launches, summed duration, and summed ALGORITHMIC flops / bytes (DESIGN.md gives the per-launch
 * formulas).  One kernel (template instantiation) per family, so the totals match one rocprofv3 row. */
#define WU_FAM_CONV_FWD 1    /* conv3x3_mfma_kernel<T,1,false>: forward convs */
#define WU_FAM_WGRAD 2       /* conv3x3_wgrad_kernel<T,1,P> */
#define WU_FAM_CONV_DGRAD 3  /* conv3x3_mfma_kernel<T,1,true>: gated data-gradient pass */
#define WU_FAM_CONV_S2 4     /* conv3x3_mfma_kernel<T,2,false>: discriminator stride-2 forward */
#define WU_FAM_WGRAD_S2 5    /* conv3x3_wgrad_kernel<T,2,P> */
#define WU_FAM_CONV1X1 6     /* conv1x1_mfma_kernel<T>: pointwise convs of the estimator */
int wu_prof_begin(unsigned family_mask, int max_launches);
int wu_prof_query(int family, int* launches, double* total_ms, double* total_flops, double* total_bytes);
int wu_prof_end(void);

#ifdef __cplusplus
}
#endif
#endif
