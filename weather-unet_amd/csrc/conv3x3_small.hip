// conv3x3 (pad 1, stride 1, bf16) on SMALL images (W <= 16): the 3x3 convs of the estimator's layer3 / layer4 Bottlenecks (classifier.py:106,
// estimator.py:143: ResNet-101, 256 -> 256 @16x16 and 512 -> 512 @8x8 for 256x256 inputs) forward and, on the rotated pack, their data gradient.
//
// Round 4.  The generic template (conv3x3_mfma.hip) gives a workgroup 256 output pixels x 64 couts.  On these layers that is ONE 16x16 image per
// tile -- 128 workgroups for a batch of 32, half the chip idle, 24.5 us for 9.7 GFLOP -- or, at 8x8, a 32-row tile of which 24 rows lie outside
// the image (3/4 of the matrix work multiplies padding).  This kernel:
//   * 128 pixels x 64 couts per workgroup: twice the workgroups (256 at B = 32 for layer3);
//   * K split inside the workgroup: waves (pg, kpart) = (pixel half, k-step of the 32-channel chunk); each wave keeps the 2 x 2 block of 32 x 32
//     accumulators of a 64-pixel x 64-cout tile (4 fragment reads per 4 MFMAs, as in the generic kernel -- a 4-wave 128-pixel tile without the split
//     would be 3 reads per 2) and the two k-halves are summed once, through LDS, in fp32, in a fixed order (kpart 0 + kpart 1);
//   * images narrower than the tile are STACKED: a tile of TH rows holds TH / H whole images, each with its own zero halo rows in the LDS image
//     (output row r of the tile reads halo row r + 2 (r / H) + kh), so an 8x8 layer fills its tiles.
// Staging by LDS-DMA (buffer_load ... lds through two buffer descriptors, as in conv3x3_mfma_v2.hip) into a ring of D chunk buffers: chunk c + D - 1 is
// requested before the MFMAs of chunk c, one barrier per chunk, no staging registers, the waits counted by the kernel (vmcnt retires in issue order:
// `vmcnt(pieces per chunk)` leaves the youngest chunk in flight).  A 36-MFMA chunk (0.55 us) does not cover an L2 round trip, two of them do: the first,
// register-staged version of this kernel ran 256 -> 256 @16x16 B = 32 in 18.6 us (generic template: 24.1).
#include <type_traits>

#include "wu_common.h"
#include "conv_internal.h"

namespace {

constexpr int kTP = 128;        // output pixels per workgroup
constexpr int kBN = 64;         // output channels per workgroup
constexpr int kCB = 64;         // bytes per pixel and chunk in LDS (32 bf16 channels)

struct SmArgs {
    const bf16_t* x; const bf16_t* w; const float* bias; bf16_t* y; const bf16_t* egate;
    int ldx, ldy, ldegate, egate_act, act;
    int N, H, W, Cin, Cout;
    int tw_log2, ipt;            // tile width 1 << tw_log2; images stacked per tile (1: the tile lies inside one image)
    int tiles_x, tiles_y, cout_tiles;
    int halo_w, halo_h, halo_pix;
    // weights: 0 = the library's pack [tap][Cout][Cin] (a chunk's slab = 64-byte pieces of 2 Cin-byte rows: every 128-byte line is fetched for half its
    // bytes, and its other half one chunk later from L2 again); 1 = chunk-major [Cin / 32][tap][Cout][32] (wu_conv3x3_small_fwd: a tap's 64 rows of a
    // chunk are 4 KiB contiguous).  The kernel is bound by the bytes it pulls out of L2 (all 256 workgroups stream one of four weight slabs): -25 %
    int w_chunked;
};

__device__ __forceinline__ int swz_off(int row, int slot) { return row * kCB + ((slot ^ ((row >> 2) & 3)) << 4); }
__device__ __forceinline__ void mma(f32x16_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}

template <int NHK, int D>
__global__ __launch_bounds__(256, 1) void conv3x3_small_kernel(const SmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HALO_BYTES = NHK * 4096;             // every wave issues NHK halo pieces (pad pieces land zeros): compile-time piece counts
    constexpr int CH = HALO_BYTES + 9 * kBN * kCB;     // one chunk buffer
    constexpr int P = NHK + 9;                         // DMA pieces per wave and chunk

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int pg = wave >> 1, kpart = wave & 1;        // pixel half of the tile; which 16-channel k-step of every chunk this wave multiplies

    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int ct = bid % a.cout_tiles; bid /= a.cout_tiles;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int ng = bid / a.tiles_y;                    // image (ipt == 1) or group of ipt stacked images
    const int TW = 1 << a.tw_log2;
    const int oh0 = ty * (kTP >> a.tw_log2), ow0 = tx * TW, co0 = ct * kBN;
    const int rows_valid = a.ipt > 1 ? a.ipt * a.H : (kTP >> a.tw_log2);

    // ---- per-lane DMA byte offsets: LDS 16-B slot i = piece * 64 + lane; this wave issues pieces 4 k + wave ----
    // halo: pixel p = i >> 2, LDS slot i & 3 holds channel slot (i & 3) ^ ((p >> 2) & 3) (the bank swizzle is applied to the SOURCE offset: the LDS image
    // of a DMA is lane-linear); zero padding / pixels outside the image / pad pieces are lanes pushed out of the descriptor's range
    unsigned hoff[NHK];
#pragma unroll
    for (int k = 0; k < NHK; ++k) {
        const int i = (4 * k + wave) * 64 + lane;
        const int p = i >> 2, slot = (i & 3) ^ ((p >> 2) & 3);
        hoff[k] = kWuOOB;
        if (p < a.halo_pix) {
            const int hy = p / a.halo_w, hx = p - hy * a.halo_w;
            int n = ng, ih = oh0 - 1 + hy;
            if (a.ipt > 1) {
                const int j = hy / (a.H + 2);
                n = ng * a.ipt + j;
                ih = hy - j * (a.H + 2) - 1;
            }
            const int iw = ow0 - 1 + hx;
            if (n < a.N && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) hoff[k] = (unsigned)((((n * a.H + ih) * a.W + iw) * a.ldx + slot * 8) * 2);
        }
    }
    // weights: piece 4 j + wave = rows 16 wave + (lane >> 2) of tap j's [64][32-channel] slab: one per-lane offset, the tap travels in the scalar offset
    const int wco = 16 * wave + (lane >> 2);
    const unsigned wsl = (unsigned)(((lane & 3) ^ ((wco >> 2) & 3)) * 8);
    const unsigned woff = a.w_chunked ? (unsigned)(((co0 + wco) * 32 + wsl) * 2) : (unsigned)((((size_t)(co0 + wco)) * a.Cin + wsl) * 2);
    const unsigned wtap_bytes = (unsigned)__builtin_amdgcn_readfirstlane(a.w_chunked ? a.Cout * 64 : (int)((size_t)a.Cout * a.Cin * 2));
    const unsigned wchunk_bytes = (unsigned)__builtin_amdgcn_readfirstlane(a.w_chunked ? 9 * a.Cout * 64 : 64);      // from one 32-channel chunk to the next
    const wu_rsrc_t rs_x = wu_make_rsrc(a.x, (unsigned)(((size_t)a.N * a.H * a.W * a.ldx) * 2));
    const wu_rsrc_t rs_w = wu_make_rsrc(a.w, (unsigned)((size_t)9 * a.Cout * a.Cin * 2));
    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    auto issue_chunk = [&](int c0, int b) __attribute__((always_inline)) {
        const unsigned lds = smem_base + (unsigned)b * CH + (unsigned)wave * 1024;
#pragma unroll
        for (int k = 0; k < NHK; ++k) wu_dma16b(hoff[k], rs_x, (unsigned)c0 * 2, __builtin_amdgcn_readfirstlane(lds + k * 4096));
#pragma unroll
        for (int j = 0; j < 9; ++j) wu_dma16b(woff, rs_w, (unsigned)(c0 >> 5) * wchunk_bytes + (unsigned)j * wtap_bytes, __builtin_amdgcn_readfirstlane(lds + HALO_BYTES + j * 4096));
    };

    // ---- per-lane fragment addresses (k-step kpart: 16-byte slots 2 kpart + lh) ----
    int apix[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int r = 64 * pg + 32 * mi + l31;
        int ry = r >> a.tw_log2;
        const int rx = r & (TW - 1);
        ry = min(ry, rows_valid - 1);                  // rows past the stacked images: results discarded, reads kept inside the halo image
        apix[mi] = (ry + (a.ipt > 1 ? 2 * (ry / a.H) : 0)) * a.halo_w + rx;
    }
    int boff[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) boff[ni] = HALO_BYTES + (swz_off(32 * ni + l31, lh) ^ (kpart << 5));

    f32x16_t acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

    const int nchunks = a.Cin / 32;
#pragma unroll
    for (int d = 0; d < D - 1; ++d)
        if (d < nchunks) issue_chunk(d * 32, d);
    int b = 0;                                         // ring slot of chunk c
    for (int c = 0; c < nchunks; ++c) {
        // chunk c has landed once at most the younger chunks' pieces are outstanding: those of chunks c + 1 .. min(c + D - 2, nchunks - 1)
        const int younger = min(D - 2, nchunks - 1 - c);
        if (D == 3 && younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // ... for every wave's pieces; and everyone is done with the slot of chunk c - 1
        if (c + D - 1 < nchunks) issue_chunk((c + D - 1) * 32, (b + D - 1) % D);
        const char* buf = smem + b * CH;
        // nine taps, this wave's k-step; the fragments of tap t + 1 are requested before the MFMAs of tap t
        uint4 af[2][2], bf[2][2];
        auto load_tap = [&](int t, uint4 (&fa)[2], uint4 (&fb)[2]) __attribute__((always_inline)) {
            const int kh = t / 3, kw = t - 3 * kh;
            const int sh = kh * a.halo_w + kw;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) fa[mi] = *(const uint4*)(buf + (swz_off(apix[mi] + sh, lh) ^ (kpart << 5)));
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) fb[ni] = *(const uint4*)(buf + t * (kBN * kCB) + boff[ni]);
        };
        load_tap(0, af[0], bf[0]);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t + 1 < 9) load_tap(t + 1, af[(t + 1) & 1], bf[(t + 1) & 1]);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) mma(acc[mi][ni], af[t & 1][mi], bf[t & 1][ni]);
        }
        b = b + 1 == D ? 0 : b + 1;
    }
    __syncthreads();                                   // the ring becomes the reduction / epilogue scratch

    // ---- the two k-halves of a pixel half: kpart 1 parks its accumulators in LDS ([pg][register][lane]: conflict-free dwords), kpart 0 adds ----
    float* red = (float*)smem;
    if (kpart == 1) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int i = 0; i < 16; ++i) red[(pg * 64 + (mi * 2 + ni) * 16 + i) * 64 + lane] = acc[mi][ni][i];
    }
    __syncthreads();
    if (kpart == 0) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mi][ni][i] += red[(pg * 64 + (mi * 2 + ni) * 16 + i) * 64 + lane];
    }
    __syncthreads();

    // ---- epilogue: bias + activation in fp32, transpose through LDS ([128 pixels][64 couts] bf16, 144-byte rows), 16-B coalesced stores ----
    constexpr int kRow = kBN * 2 + 16;
    if (kpart == 0) {
        float bv[2] = {0.f, 0.f};
        if (a.bias) { bv[0] = a.bias[co0 + l31]; bv[1] = a.bias[co0 + 32 + l31]; }
        auto epi_write = [&](auto act_tag) __attribute__((always_inline)) {
            constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int row = 64 * pg + 32 * mi + (i & 3) + 8 * (i >> 2) + 4 * lh;
                        ElemTraits<bf16_t>::store((bf16_t*)(smem + row * kRow) + 32 * ni + l31, act_apply(acc[mi][ni][i] + bv[ni], ACT));
                    }
        };
        if (a.act == WU_ACT_RELU) epi_write(std::integral_constant<int, WU_ACT_RELU>{});
        else if (a.act == WU_ACT_LEAKY) epi_write(std::integral_constant<int, WU_ACT_LEAKY>{});
        else epi_write(std::integral_constant<int, WU_ACT_NONE>{});
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kTP * 8 / 256; ++k) {
        const int q = tid + 256 * k;
        const int r = q >> 3, s = q & 7;
        const int ry = r >> a.tw_log2, ow = ow0 + (r & (TW - 1));
        int n = ng, oh = oh0 + ry;
        if (a.ipt > 1) { const int j = ry / a.H; n = ng * a.ipt + j; oh = ry - j * a.H; }
        if (ry < rows_valid && n < a.N && oh < a.H && ow < a.W) {
            uint4 v = *(const uint4*)(smem + r * kRow + s * 16);
            const size_t pix = ((size_t)n * a.H + oh) * a.W + ow;
            if (a.egate) v = gate16<bf16_t>(v, *(const uint4*)(a.egate + pix * a.ldegate + co0 + s * 8), a.egate_act);
            *(uint4*)(a.y + pix * a.ldy + co0 + s * 8) = v;
        }
    }
}

}  // namespace

// 0 = launched; 1 = not this kernel's shape (the caller falls back to the generic template)
int conv_small_launch(const void* x, int ldx, const void* w_packed, const float* bias, void* y, int ldy, const void* egate, int ldegate, int egate_act,
                      int N, int H, int W, int Cin, int Cout, int act, hipStream_t s, int w_chunked, int mode) {
    // mode: 0 = launch if the shape allows and the dispatch rule below prefers this kernel; 1 = only answer (0 = would launch); 2 = launch whatever the rule says
    if (W > 16 || Cin % 32 != 0 || Cout % kBN != 0) return 1;
    const int ldmax = ldx > ldy ? (ldx > ldegate ? ldx : ldegate) : (ldy > ldegate ? ldy : ldegate);
    if ((unsigned long long)N * H * W * (unsigned long long)ldmax >= (1ull << 31)) return 1;
    SmArgs a;
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w_packed; a.bias = bias; a.y = (bf16_t*)y; a.egate = (const bf16_t*)egate;
    a.ldx = ldx; a.ldy = ldy; a.ldegate = ldegate; a.egate_act = egate_act; a.act = act;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.w_chunked = w_chunked;
    a.tw_log2 = W > 8 ? 4 : (W > 4 ? 3 : 2);
    const int TW = 1 << a.tw_log2, TH = kTP >> a.tw_log2;
    a.ipt = H < TH ? TH / H : 1;
    if (a.ipt < 1) a.ipt = 1;
    while (a.ipt > 1 && a.ipt * (H + 2) * (TW + 2) > 320) --a.ipt;       // the stacked halo image must fit 5 pieces per wave (tiny images: fewer per tile)
    a.tiles_x = cdiv(W, TW);
    a.tiles_y = a.ipt > 1 ? 1 : cdiv(H, TH);
    a.cout_tiles = Cout / kBN;
    a.halo_w = TW + 2;
    a.halo_h = a.ipt > 1 ? a.ipt * (H + 2) : TH + 2;
    a.halo_pix = a.halo_w * a.halo_h;
    const int nhk = cdiv(cdiv(a.halo_pix, 16), 4);       // 1-KiB halo pieces (16 pixels each) per wave
    if (nhk > 5) return 1;
    if (((size_t)N * H * W * ldx) * 2 >= (1ull << 31) || (size_t)9 * Cout * Cin * 2 >= (1ull << 31)) return 1;      // 32-bit DMA byte offsets below kWuOOB
    const long long groups = a.ipt > 1 ? cdiv(N, a.ipt) : N;
    const long long grid = groups * a.tiles_x * a.tiles_y * a.cout_tiles;
    if (grid >= (1ll << 31)) return 1;
    // Where the generic template's 256-pixel tiles fill the chip AND lie inside the image (W > 8: a 16-wide tile is 16 rows), it stages the weight slab
    // once per 256 pixels instead of once per 128 and wins (256 -> 256 @16x16: B = 64 27.3 vs 31.6 us, B = 128 42.6 vs 60.1; B = 32 24.1 vs 16.2)
    if (W > 8 && H >= 16 && g_wu_opt[WU_OPT_CONV_SMALL] != 2 && mode != 2) {
        const long long generic_grid = (long long)N * cdiv(H, 16) * a.cout_tiles;
        if (generic_grid * 4 >= (long long)wu_num_cus() * 3) return 1;
    }
    if (mode == 1) return 0;
    const int depth = nhk <= 4 ? 3 : 2;                  // three chunk buffers where they fit the CU's 160 KiB
    const size_t lds = (size_t)depth * (nhk * 4096 + 9 * kBN * kCB);     // >= the 32-KiB reduction image and the 18-KiB epilogue image
    static thread_local bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv3x3_small_kernel<3, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv3x3_small_kernel<4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv3x3_small_kernel<5, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (nhk <= 3) hipLaunchKernelGGL((conv3x3_small_kernel<3, 3>), dim3((int)grid), dim3(256), lds + (3 - nhk) * 3 * 4096, s, a);
    else if (nhk == 4) hipLaunchKernelGGL((conv3x3_small_kernel<4, 3>), dim3((int)grid), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((conv3x3_small_kernel<5, 2>), dim3((int)grid), dim3(256), lds, s, a);
    return 0;
}

// The estimator's 3x3 convs on small images with CHUNK-MAJOR weights (round 4): w_chunked = [Cin / 32][9 taps][Cout][32] bf16, i.e. the library's pack
// [9][Cout][Cin] viewed as [9][Cout][Cin / 32][32] and permuted (2, 0, 1, 3) -- for frozen weights (classifier.py:106, eval mode) a one-off
// re-arrangement at plan time.  Same arithmetic as wu_conv3x3_fwd on the small-image kernel; callers ask wu_conv3x3_small_supported first (1 = this
// shape runs on the small-image kernel AND the dispatch rule prefers it to the generic template).
extern "C" int wu_conv3x3_small_supported(int N, int H, int W, int ldx, int ldy, int ldegate, int Cin, int Cout) {
    if (!g_wu_opt[WU_OPT_CONV_SMALL]) return 0;
    return conv_small_launch(nullptr, ldx, nullptr, nullptr, nullptr, ldy, nullptr, ldegate, 0, N, H, W, Cin, Cout, 0, nullptr, 1, 1) == 0 ? 1 : 0;
}

extern "C" int wu_conv3x3_small_fwd(const void* x, int ldx, const void* w_chunked, const float* bias, void* y, int ldy,
                                    const void* egate, int ldegate, int egate_act, int N, int H, int W, int Cin, int Cout, int act, void* stream) {
    WU_REQUIRE(N > 0 && H > 0 && W > 0 && x && y && w_chunked, "conv3x3_small_fwd: bad args");
    WU_REQUIRE(Cin % 32 == 0 && Cout % 64 == 0 && ldx >= Cin && ldy >= Cout && (ldx * 2) % 16 == 0 && (ldy * 2) % 16 == 0, "conv3x3_small_fwd: bad channels / ld");
    WU_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)w_chunked % 16) == 0, "conv3x3_small_fwd: pointers must be 16-B aligned");
    if (egate) WU_REQUIRE(((uintptr_t)egate % 16) == 0 && (ldegate * 2) % 16 == 0 && ldegate >= Cout, "conv3x3_small_fwd: bad egate");
    hipStream_t s = (hipStream_t)stream;
    wu_prof_pre(WU_FAM_CONV_FWD, s);
    const int rc = conv_small_launch(x, ldx, w_chunked, bias, y, ldy, egate, egate ? ldegate : 0, egate_act, N, H, W, Cin, Cout, act, s, 1, 2);
    WU_REQUIRE(rc == 0, "conv3x3_small_fwd: shape outside the small-image kernel (ask wu_conv3x3_small_supported)");
    const double pix = (double)N * H * W;
    wu_prof_post(WU_FAM_CONV_FWD, s, 2.0 * pix * Cout * 9.0 * Cin, (pix * (Cin + Cout) + 9.0 * Cin * Cout) * 2);
    WU_LAUNCH_CHECK("conv3x3 (small images, chunk-major weights)");
    return 0;
}
