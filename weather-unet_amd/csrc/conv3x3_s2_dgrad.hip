// Data gradient of the stride-2 SN conv of the discriminator (nets.py:30-31; autograd call sites
// t_cls_train.py:272,307).  dX = conv3x3_s1(zero-upsampled gated dY, rotated/transposed filter): the
// gated dY is scattered onto the even sites of an (H, W) grid (one streaming pass) and the stride-1
// MFMA kernel does the rest.  3/4 of that GEMM's K is structurally zero; the stride-2 convs are ~2 % of
// a GAN step's FLOPs, so the dedicated 4-parity-class kernel is left for a later round.
#include "wu_common.h"
#include "conv_internal.h"

namespace {
template <typename T>
__global__ void upsample_zero_gate_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ y, int ldy, int act,
                                          T* __restrict__ up, int N, int H, int W, int Ho, int Wo, int C) {
    constexpr int E = ElemTraits<T>::kPer16B;
    const int cpp = C / E;
    const long long total = (long long)N * H * W * cpp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cpp);
        long long p = i / cpp;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H);
        const int n = (int)(p / H);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (!(h & 1) && !(w & 1) && (h >> 1) < Ho && (w >> 1) < Wo) {
            const size_t o = ((size_t)(n * Ho + (h >> 1)) * Wo + (w >> 1));
            v = *(const uint4*)(dy + o * lddy + ch * E);
            if (y) v = gate16<T>(v, *(const uint4*)(y + o * ldy + ch * E), act);
        }
        *(uint4*)(up + (size_t)(i / cpp) * C + ch * E) = v;
    }
}
}  // namespace

extern "C" size_t wu_conv3x3_s2_dgrad_workspace(int N, int H, int W, int Cout, int dtype) {
    return (size_t)N * H * W * Cout * (dtype == WU_BF16 ? 2 : 4);
}

extern "C" int wu_conv3x3_s2_dgrad(const void* dy, int lddy, const void* y, int ldy_, int act, const void* w_dgrad,
                                   void* dx, int lddx, void* workspace, size_t workspace_bytes,
                                   const void* egate, int ldegate, int egate_act,
                                   int N, int H, int W, int Cin, int Cout, int dtype, void* stream) {
    const int esz = dtype == WU_BF16 ? 2 : 4;
    WU_REQUIRE(Cout % (16 / esz) == 0 && ((uintptr_t)dy % 16) == 0 && (lddy * esz) % 16 == 0, "conv3x3_s2_dgrad: alignment");
    hipStream_t s = (hipStream_t)stream;
    // Round 4: a pre-gated dY (the autograd nodes gate it first) goes through four sparse-tap convs, one per parity class of the input
    // site (conv_internal.h): no zero-stuffed copy, 9 instead of 36 tap products.  A dY that still needs its gate keeps the scatter path.
    if (!y && g_wu_opt[WU_OPT_S2_DGRAD_PARITY] && Cin % 64 == 0 && Cout % (64 / esz) == 0 && ((uintptr_t)dx % 16) == 0 && (lddx * esz) % 16 == 0 &&
        ((uintptr_t)w_dgrad % 16) == 0 && (size_t)H * W * (size_t)lddx < (1ull << 31)) {
        if (egate) WU_REQUIRE(((uintptr_t)egate % 16) == 0 && (ldegate * esz) % 16 == 0 && ldegate >= Cin, "conv3x3_s2_dgrad: bad egate");
        wu_prof_pre(WU_FAM_CONV_S2, s);
        // bf16: the classes on the persistent LDS-DMA GEMM with gathered rows (resnet.hip, option 15); else the register-staged tap-list kernel
        int rc = 1;
        if (dtype == WU_BF16 && (g_wu_opt[WU_OPT_PW3] & 7))
            rc = conv_s2_dgrad_gather_launch(dy, lddy, w_dgrad, dx, lddx, egate, ldegate, egate_act, N, H, W, Cin, Cout, s);
        if (rc != 0) rc = conv_s2_dgrad_parity_launch(dy, lddy, w_dgrad, dx, lddx, egate, ldegate, egate_act, N, H, W, Cin, Cout, dtype, s);
        if (rc == 0) {
            const double pix = (double)N * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1);
            wu_prof_post(WU_FAM_CONV_S2, s, 2.0 * pix * Cout * 9.0 * Cin, (pix * Cout + (double)N * H * W * Cin + 9.0 * Cin * Cout) * esz);
            WU_LAUNCH_CHECK("conv3x3_s2_dgrad(parity)");
            return 0;
        }
    }
    WU_REQUIRE(workspace && ((uintptr_t)workspace % 16) == 0 && workspace_bytes >= wu_conv3x3_s2_dgrad_workspace(N, H, W, Cout, dtype),
               "conv3x3_s2_dgrad: workspace too small");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long total = (long long)N * H * W * (Cout / (16 / esz));
    long long g = (total + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (dtype == WU_BF16)
        hipLaunchKernelGGL(upsample_zero_gate_kernel<bf16_t>, dim3((int)g), dim3(256), 0, s, (const bf16_t*)dy, lddy, (const bf16_t*)y, ldy_, act, (bf16_t*)workspace, N, H, W, Ho, Wo, Cout);
    else
        hipLaunchKernelGGL(upsample_zero_gate_kernel<float>, dim3((int)g), dim3(256), 0, s, (const float*)dy, lddy, (const float*)y, ldy_, act, (float*)workspace, N, H, W, Ho, Wo, Cout);
    WU_LAUNCH_CHECK("conv3x3_s2_dgrad(upsample)");
    // stride-1 correlation of the upsampled gradient with the rotated filter: channels swap roles
    return wu_conv3x3_fwd(workspace, Cout, w_dgrad, nullptr, dx, lddx, N, H, W, /*Cin=*/Cout, /*Cout=*/Cin, 1, WU_ACT_NONE,
                          nullptr, 0, 0, egate, ldegate, egate_act, dtype, stream);
}
