// conv3x3 (pad 1, stride 1|2) as an im2col-free implicit GEMM on the CDNA4 matrix cores.
//
// Replaces nn.Conv2d(cin, cout, 3, padding=1[, stride=2]) + ReLU / LeakyReLU of the reference's
// r_double_conv / sn_double_conv (nets.py:18-33); with the rotated/transposed weight pack and a gated
// input it is also the data-gradient pass of the same convs.
//
// GEMM view: M = output pixels, N = Cout, K = 9 * Cin.  One 256-thread workgroup (4 waves, one per
// SIMD) owns a TH x TW tile of 256 output pixels x 64 output channels; each wave owns 64 pixels x 64
// channels as 2x2 accumulators of v_mfma_f32_32x32x16_bf16 (bf16) or v_mfma_f32_32x32x2_f32 (fp32).
// Per 64-byte channel chunk (32 bf16 / 16 fp32 channels) the input HALO tile ((TH+2) x (TW+2) pixels)
// and the [9][64][chunk] weight slab are staged once into LDS and re-used by all nine taps: the tap
// shift is an LDS address offset, so no im2col tensor ever exists.  Global -> register -> LDS staging
// of chunk k+1 is issued before the MFMAs of chunk k (loads in flight under the matrix work); with
// <= 64 KiB of LDS two workgroups share a CU and cover each other's barriers.
//
// LDS images: pixel-major, 64 B per pixel (one chunk), the four 16-B slots of a pixel XOR-swizzled
// with (pixel >> 2) & 3 so a ds_read_b128 lane group touches 16 distinct slots (conflict-free for
// TW = 32).  Same for the weight rows ([tap][cout] rows of 64 B).
#include <type_traits>

#include "wu_common.h"
#include "conv_internal.h"

#define WU_REP9(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8)

namespace {

constexpr int kTilePix = 256;   // output pixels per workgroup
constexpr int kBN = 64;         // output channels per workgroup
constexpr int kChunkBytes = 64; // channel chunk per pixel in LDS

struct ConvArgs {
    const void* x;
    const void* mask;
    const void* w;
    const float* bias;
    void* y;
    const void* egate;
    int ldx, ldmask, ldy, ldegate, egate_act;
    int N, H, W, Ho, Wo, Cin, Cout;
    int act, mask_act;
    int tw_log2;      // tile width  = 1 << tw_log2  (output pixels)
    int tiles_x, tiles_y, cout_tiles;
    int halo_w, halo_h, halo_pix;
    // SPARSE instances (round 4: one parity class of the stride-2 data gradient): a LIST of taps instead of the full 3 x 3 -- tap t reads
    // the halo at (tap_dy, tap_dx) (0..2, as kh / kw) and multiplies by slab tap_w[t] of the 9-slab weight pack -- and output pixel
    // (oh, ow) of the Ho x Wo grid lands at (oh * osy + ooy, ow * osx + oox) of an OH x OW image (skipped when outside it)
    // Up to four such (tap list, output offset) CLASSES share one launch: the class is the slowest digit of the block index.
    int nclasses, cls_ntaps[4], cls_tap_dy[4][4], cls_tap_dx[4][4], cls_tap_w[4][4], cls_ooy[4], cls_oox[4];
    int osy, osx, OH, OW;
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static __device__ __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    // exact-fp32 MFMA: lane half h of a 16-B chunk supplies k = 4h+j to the j-th 32x32x2 step; A and B use
    // the same k permutation, so the sum over the 8 channels of the chunk pair is complete.
    static __device__ __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
};

__device__ __forceinline__ int swz_off(int row, int slot) { return row * kChunkBytes + ((slot ^ ((row >> 2) & 3)) << 4); }

template <typename T, int STRIDE, bool MASKED, int NHALO, bool SPARSE = false>
__global__ __launch_bounds__(256, STRIDE == 1 ? 2 : 1) void conv3x3_mfma_kernel(const ConvArgs a) {
    static_assert(!SPARSE || STRIDE == 1, "tap lists are a stride-1 form");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kChunkElems = kChunkBytes / (int)sizeof(T);
    char* halo_lds = smem;
    char* w_lds = smem + a.halo_pix * kChunkBytes;  // halo_pix*64 is a multiple of 16

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;

    // ---- which tile ----
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    // (SPARSE) which tap list / output offset this workgroup computes: the FASTEST digit -- the classes of one tile read the same input
    // tile (L2 hits) and every XCD's contiguous range of logical ids holds the same mix of 4-, 2-, 2- and 1-tap workgroups
    const int cls = SPARSE ? bid % a.nclasses : 0;
    if (SPARSE) bid /= a.nclasses;
    const int ct = bid % a.cout_tiles; bid /= a.cout_tiles;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y; bid /= a.tiles_y;
    const int n = bid;
    const int ntaps = SPARSE ? a.cls_ntaps[cls] : 9;
    const int TW = 1 << a.tw_log2, TH = kTilePix >> a.tw_log2;
    const int oh0 = ty * TH, ow0 = tx * TW;        // output tile origin
    const int ih0 = oh0 * STRIDE - 1, iw0 = ow0 * STRIDE - 1;  // input halo origin
    const int co0 = ct * kBN;

    const T* xin = (const T*)a.x + (size_t)n * a.H * a.W * a.ldx;
    const T* xmask = MASKED ? (const T*)a.mask + (size_t)n * a.H * a.W * a.ldmask : nullptr;

    // ---- per-thread staging descriptors: item = (halo pixel, 16-B slot) ----
    int hoff[NHALO];   // element offset of the pixel inside image n (-1: zero fill; -2: no such item)
    int moff[MASKED ? NHALO : 1];
#pragma unroll
    for (int k = 0; k < NHALO; ++k) {
        const int item = tid + 256 * k;
        const int p = item >> 2, slot = item & 3;
        hoff[k] = -2;
        if (MASKED) moff[k] = -1;
        if (p < a.halo_pix) {
            const int hy = p / a.halo_w, hx = p - hy * a.halo_w;
            const int ih = ih0 + hy, iw = iw0 + hx;
            hoff[k] = -1;
            if (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) {
                hoff[k] = (ih * a.W + iw) * a.ldx + slot * (16 / (int)sizeof(T));
                if (MASKED) moff[k] = (ih * a.W + iw) * a.ldmask + slot * (16 / (int)sizeof(T));
            }
        }
    }
    // weights: thread -> (cout row tid>>2, slot tid&3) for each of the 9 taps
    const int wrow = tid >> 2, wslot = tid & 3;
    const T* wsrc = (const T*)a.w + (size_t)(co0 + wrow) * a.Cin + wslot * (16 / (int)sizeof(T));
    const size_t wtap_stride = (size_t)a.Cout * a.Cin;
    const int wdst = swz_off(wrow, wslot);

    // ---- per-lane fragment addresses ----
    // A: wave owns tile pixels [64*wave, 64*wave+64): two 32-row groups mi; row r -> (ty_, tx_) in the tile
    int apix[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int r = 64 * wave + 32 * mi + l31;
        const int ry = r >> a.tw_log2, rx = r & (TW - 1);
        apix[mi] = (ry * STRIDE) * a.halo_w + rx * STRIDE;
    }
    int boff[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) boff[ni] = swz_off(32 * ni + l31, lh);

    f32x16_t acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

    uint4 hreg[NHALO];
    uint4 wr0, wr1, wr2, wr3, wr4, wr5, wr6, wr7, wr8;   // named (not an array): keeps them out of scratch
    uint4 mreg[MASKED ? NHALO : 1];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);

    auto load_chunk = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NHALO; ++k) {
            hreg[k] = zero4;
            if (hoff[k] >= 0) hreg[k] = *(const uint4*)(xin + hoff[k] + c0);
            if (MASKED) {
                mreg[k] = zero4;
                if (moff[k] >= 0) mreg[k] = *(const uint4*)(xmask + moff[k] + c0);
            }
        }
        if constexpr (SPARSE) {                    // 1, 2 or 4 taps: only their slabs travel
            wr0 = *(const uint4*)(wsrc + a.cls_tap_w[cls][0] * wtap_stride + c0);
            if (ntaps > 1) wr1 = *(const uint4*)(wsrc + a.cls_tap_w[cls][1] * wtap_stride + c0);
            if (ntaps > 2) {
                wr2 = *(const uint4*)(wsrc + a.cls_tap_w[cls][2] * wtap_stride + c0);
                wr3 = *(const uint4*)(wsrc + a.cls_tap_w[cls][3] * wtap_stride + c0);
            }
        } else {
#define WU_LOADW(t) wr##t = *(const uint4*)(wsrc + (t) * wtap_stride + c0);
            WU_REP9(WU_LOADW)
#undef WU_LOADW
        }
    };
    auto store_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NHALO; ++k) {
            if (hoff[k] != -2) {                 // the LDS slot follows from the item index (kept in no register)
                uint4 v = hreg[k];
                if (MASKED) v = gate16<T>(v, mreg[k], a.mask_act);
                const int item = tid + 256 * k;
                *(uint4*)(halo_lds + swz_off(item >> 2, item & 3)) = v;
            }
        }
#define WU_STOREW(t) *(uint4*)(w_lds + (t) * (kBN * kChunkBytes) + wdst) = wr##t;
        if constexpr (SPARSE) {
            WU_STOREW(0)
            if (ntaps > 1) { WU_STOREW(1) }
            if (ntaps > 2) { WU_STOREW(2) WU_STOREW(3) }
        } else {
            WU_REP9(WU_STOREW)
        }
#undef WU_STOREW
    };

    const int nchunks = a.Cin / kChunkElems;
    // Both strides: chunk k+1 travels global -> registers under the MFMAs of chunk k, then registers -> LDS between two barriers.
    // (Round 4: the stride-2 tiles -- a 4x larger halo, 17-19 items per thread -- used to be staged global -> LDS synchronously in
    //  front of every chunk's MFMAs, with one workgroup per CU: nothing overlapped, 146-168 TFLOP/s.)
    load_chunk(0);
    store_chunk();
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks) load_chunk((c + 1) * kChunkElems);
        auto tap_mma = [&](int hoff_tap, int slab) __attribute__((always_inline)) {
            int aoff[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) aoff[mi] = swz_off(apix[mi] + hoff_tap, lh);
            const char* wt = w_lds + slab * (kBN * kChunkBytes);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 af[2], bf[2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) af[mi] = *(const uint4*)(halo_lds + (aoff[mi] ^ (ks << 5)));
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) bf[ni] = *(const uint4*)(wt + (boff[ni] ^ (ks << 5)));
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) Mma<T>::run(acc[mi][ni], af[mi], bf[ni]);
            }
        };
        if constexpr (SPARSE) {
#pragma unroll 1
            for (int t = 0; t < ntaps; ++t) tap_mma(a.cls_tap_dy[cls][t] * a.halo_w + a.cls_tap_dx[cls][t], t);
        } else {
#pragma unroll 1
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) tap_mma(kh * a.halo_w + kw, kh * 3 + kw);
        }
        __syncthreads();
        if (c + 1 < nchunks) {
            store_chunk();
            __syncthreads();
        }
    }

    // ---- epilogue: bias + activation in fp32, transpose through LDS, 16-B coalesced stores ----
    // LDS image [256 pixels][64 cout] of T, row stride 64*sizeof(T) + 16 B pad
    constexpr int kRow = kBN * (int)sizeof(T) + 16;
    float bv[2] = {0.f, 0.f};
    if (a.bias) {
        bv[0] = a.bias[co0 + l31];
        bv[1] = a.bias[co0 + 32 + l31];
    }
    // the activation is selected once (a per-element runtime switch costs a scalar branch per value)
    auto epi_write = [&](auto act_tag) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = 64 * wave + 32 * mi + (i & 3) + 8 * (i >> 2) + 4 * lh;
                    const float v = act_apply(acc[mi][ni][i] + bv[ni], ACT);
                    ElemTraits<T>::store((T*)(smem + row * kRow) + 32 * ni + l31, v);
                }
    };
    if (a.act == WU_ACT_RELU) epi_write(std::integral_constant<int, WU_ACT_RELU>{});
    else if (a.act == WU_ACT_LEAKY) epi_write(std::integral_constant<int, WU_ACT_LEAKY>{});
    else epi_write(std::integral_constant<int, WU_ACT_NONE>{});
    __syncthreads();
    constexpr int kSlotsPerRow = kBN * (int)sizeof(T) / 16;  // 8 (bf16) / 16 (fp32)
    const int OHW = SPARSE ? a.OH * a.OW : a.Ho * a.Wo;
    T* yout = (T*)a.y + (size_t)n * OHW * a.ldy + co0;
#pragma unroll
    for (int k = 0; k < kTilePix * kSlotsPerRow / 256; ++k) {
        const int q = tid + 256 * k;
        const int r = q / kSlotsPerRow, s = q % kSlotsPerRow;
        const int oh = oh0 + (r >> a.tw_log2), ow = ow0 + (r & (TW - 1));
        // where the pixel lands: itself, or (SPARSE) its site in the strided output image
        const int ph = SPARSE ? oh * a.osy + a.cls_ooy[cls] : oh, pw = SPARSE ? ow * a.osx + a.cls_oox[cls] : ow;
        const int PW = SPARSE ? a.OW : a.Wo;
        if (oh < a.Ho && ow < a.Wo && (!SPARSE || (ph < a.OH && pw < a.OW))) {
            uint4 v = *(const uint4*)(smem + r * kRow + s * 16);
            if (a.egate) {
                const T* eg = (const T*)a.egate + ((size_t)n * OHW + (size_t)(ph * PW + pw)) * a.ldegate + co0 + s * (16 / (int)sizeof(T));
                v = gate16<T>(v, *(const uint4*)eg, a.egate_act);
            }
            *(uint4*)(yout + (size_t)(ph * PW + pw) * a.ldy + s * (16 / (int)sizeof(T))) = v;
        }
    }
}

template <typename T, int STRIDE, bool MASKED, bool SPARSE = false>
int launch_conv(const ConvArgs& a, size_t lds_bytes, int grid, hipStream_t s) {
    constexpr int NH = STRIDE == 1 ? 7 : 19;  // staging items per thread: halo pixels x 4 slots / 256 (stride 2: up to 129 x 9 halo pixels)
    auto kern = conv3x3_mfma_kernel<T, STRIDE, MASKED, NH, SPARSE>;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, s, a);
    return 0;
}

}  // namespace

int conv_s2_dgrad_parity_launch(const void* dy, int lddy, const void* w_dgrad, void* dx, int lddx, const void* egate, int ldegate, int egate_act,
                                int N, int H, int W, int Cin, int Cout, int dtype, hipStream_t s) {
    const int esz = dtype == WU_BF16 ? 2 : 4;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    ConvArgs a;
    a.x = dy; a.mask = nullptr; a.w = w_dgrad; a.bias = nullptr; a.y = dx; a.egate = egate;
    a.ldx = lddy; a.ldmask = 0; a.ldy = lddx; a.ldegate = ldegate; a.egate_act = egate_act;
    a.N = N; a.H = Ho; a.W = Wo; a.Ho = Ho; a.Wo = Wo;
    a.Cin = Cout;                       // the GEMM's K: the forward conv's output channels
    a.Cout = Cin;                       // ... and its N: the forward conv's input channels
    a.act = WU_ACT_NONE; a.mask_act = WU_ACT_NONE;
    int twl = 5;
    while (twl > 2 && (1 << (twl - 1)) >= Wo) --twl;
    a.tw_log2 = twl;
    const int TW = 1 << twl, TH = kTilePix >> twl;
    a.tiles_x = cdiv(Wo, TW); a.tiles_y = cdiv(Ho, TH); a.cout_tiles = Cin / kBN;
    a.halo_w = TW + 2; a.halo_h = TH + 2; a.halo_pix = a.halo_w * a.halo_h;
    if (a.halo_pix > 7 * 256 / 4) return -1;
    size_t lds = (size_t)a.halo_pix * kChunkBytes + 9 * kBN * kChunkBytes;
    const size_t epi = (size_t)kTilePix * (kBN * esz + 16);
    if (epi > lds) lds = epi;
    const long long grid = (long long)N * a.tiles_x * a.tiles_y * a.cout_tiles;
    if (grid >= (1ll << 31)) return -1;
    a.osy = a.osx = 2; a.OH = H; a.OW = W;
    a.nclasses = 4;
    for (int c = 0; c < 4; ++c) {
        const int p = c < 2 ? 1 : 0, q = (c == 0 || c == 2) ? 1 : 0;
        // forward taps reaching input parity p: kh = 1 from output row i (halo row 1); kh = 0 from row i + 1 (halo row 2), kh = 2 from row i
        const int nk_y = p ? 2 : 1, nk_x = q ? 2 : 1;
        const int khs[2] = {p ? 0 : 1, 2}, dys[2] = {p ? 2 : 1, 1};
        const int kws[2] = {q ? 0 : 1, 2}, dxs[2] = {q ? 2 : 1, 1};
        a.cls_ntaps[c] = nk_y * nk_x;
        for (int iy = 0; iy < nk_y; ++iy)
            for (int ix = 0; ix < nk_x; ++ix) {
                const int t = iy * nk_x + ix;
                a.cls_tap_dy[c][t] = dys[iy]; a.cls_tap_dx[c][t] = dxs[ix];
                a.cls_tap_w[c][t] = 8 - (khs[iy] * 3 + kws[ix]);       // the rotated pack keeps forward tap (kh, kw) in slab 8 - (3 kh + kw)
            }
        a.cls_ooy[c] = p; a.cls_oox[c] = q;
    }
    if (4 * grid >= (1ll << 31)) return -1;
    if (dtype == WU_BF16) launch_conv<bf16_t, 1, false, true>(a, lds, (int)(4 * grid), s);
    else launch_conv<float, 1, false, true>(a, lds, (int)(4 * grid), s);
    return 0;
}

extern "C" int wu_conv3x3_fwd(const void* x, int ldx, const void* w_packed, const float* bias, void* y, int ldy,
                              int N, int H, int W, int Cin, int Cout, int stride, int act,
                              const void* mask, int ldmask, int mask_act,
                              const void* egate, int ldegate, int egate_act, int dtype, void* stream) {
    const int esz = dtype == WU_BF16 ? 2 : 4;
    const int chunk = kChunkBytes / esz;
    WU_REQUIRE(dtype == WU_F32 || dtype == WU_BF16, "conv3x3_fwd: bad dtype %d", dtype);
    WU_REQUIRE(stride == 1 || stride == 2, "conv3x3_fwd: stride %d", stride);
    WU_REQUIRE(N > 0 && H > 0 && W > 0, "conv3x3_fwd: empty shape");
    WU_REQUIRE(Cin % chunk == 0 && Cin > 0, "conv3x3_fwd: Cin=%d must be a multiple of %d", Cin, chunk);
    WU_REQUIRE(Cout % kBN == 0 && Cout > 0, "conv3x3_fwd: Cout=%d must be a multiple of %d", Cout, kBN);
    WU_REQUIRE(ldx >= Cin && ldy >= Cout && (ldx * esz) % 16 == 0 && (ldy * esz) % 16 == 0, "conv3x3_fwd: bad ld (%d,%d)", ldx, ldy);
    WU_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)w_packed % 16) == 0, "conv3x3_fwd: pointers must be 16-B aligned");
    WU_REQUIRE((size_t)H * W * (size_t)(ldx > ldmask ? ldx : ldmask) < (1ull << 31), "conv3x3_fwd: image too large for 32-bit offsets");
    if (mask) WU_REQUIRE(stride == 1 && ldmask >= Cin && (ldmask * esz) % 16 == 0 && ((uintptr_t)mask % 16) == 0, "conv3x3_fwd: bad mask");

    if (egate) WU_REQUIRE(((uintptr_t)egate % 16) == 0 && (ldegate * esz) % 16 == 0 && ldegate >= Cout, "conv3x3_fwd: bad egate");
    ConvArgs a;
    a.x = x; a.mask = mask; a.w = w_packed; a.bias = bias; a.y = y; a.egate = egate;
    a.ldx = ldx; a.ldmask = ldmask; a.ldy = ldy; a.ldegate = ldegate; a.egate_act = egate_act;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1;
    a.act = act; a.mask_act = mask_act;
    int twl = 5;
    while (twl > 2 && (1 << (twl - 1)) >= a.Wo) --twl;   // narrow images: TW = 16 / 8 / 4
    a.tw_log2 = twl;
    const int TW = 1 << twl, TH = kTilePix >> twl;
    a.tiles_x = cdiv(a.Wo, TW); a.tiles_y = cdiv(a.Ho, TH); a.cout_tiles = Cout / kBN;
    a.halo_w = (TW - 1) * stride + 3; a.halo_h = (TH - 1) * stride + 3;
    a.halo_pix = a.halo_w * a.halo_h;
    WU_REQUIRE(a.halo_pix <= (stride == 1 ? 7 : 19) * 256 / 4, "conv3x3_fwd: halo %d exceeds staging capacity", a.halo_pix);
    size_t lds = (size_t)a.halo_pix * kChunkBytes + 9 * kBN * kChunkBytes;
    const size_t epi = (size_t)kTilePix * (kBN * esz + 16);
    if (epi > lds) lds = epi;
    const long long grid = (long long)N * a.tiles_x * a.tiles_y * a.cout_tiles;
    WU_REQUIRE(grid < (1ll << 31), "conv3x3_fwd: grid too large");
    hipStream_t s = (hipStream_t)stream;
    const bool m = mask != nullptr;
    const int fam = stride == 2 ? WU_FAM_CONV_S2 : (m ? WU_FAM_CONV_DGRAD : WU_FAM_CONV_FWD);
    if (g_wu_opt[WU_OPT_CONV_V2] && conv_v2_eligible(H, W, ldx, ldy, Cin, Cout, stride, dtype, m) && (egate == nullptr || (act == WU_ACT_NONE && bias == nullptr))) {
        wu_prof_pre(fam, s);
        const int rc = conv_v2_launch(x, ldx, w_packed, bias, y, ldy, egate, ldegate, egate_act, N, H, W, Cin, Cout, act, s);
        WU_REQUIRE(rc == 0, "conv3x3_fwd: grid too large");
        wu_prof_post(fam, s, 2.0 * N * H * W * (double)Cout * 9.0 * Cin,
                     ((double)N * H * W * (Cin + Cout) + 9.0 * Cin * Cout) * esz);
        WU_LAUNCH_CHECK("conv3x3_mfma_v2");
        return 0;
    }
    wu_prof_pre(fam, s);
    // Round 4: stride 2 goes to the gathered-row form of the persistent LDS-DMA GEMM (option 15; resnet.hip) where its shape allows
    // (stride 1 on small images was measured too: 256 -> 256 @16x16 26.6 us against 24.1 on the register-staged kernel -- nine L2 reads of every row
    //  against a halo in LDS -- and 512 -> 512 @8x8 48 against 68 only from batch 64: not taken)
    // Round 4 (later): small images on their own kernel -- on a 16x16 image the 256-pixel tile of the template below is the whole image (128 workgroups
    // at B = 32), on 8x8 three quarters of it lie outside
    if (dtype == WU_BF16 && !m && stride == 1 && g_wu_opt[WU_OPT_CONV_SMALL] &&
        conv_small_launch(x, ldx, w_packed, bias, y, ldy, egate, ldegate, egate_act, N, H, W, Cin, Cout, act, s) == 0) {
        const double pix = (double)N * H * W;
        wu_prof_post(fam, s, 2.0 * pix * Cout * 9.0 * Cin, (pix * (Cin + Cout) + 9.0 * Cin * Cout) * esz);
        WU_LAUNCH_CHECK("conv3x3 (small images)");
        return 0;
    }
    if (dtype == WU_BF16 && !m && stride == 2 && (g_wu_opt[WU_OPT_PW3] & 7) &&
        conv3x3_gather_launch(x, ldx, w_packed, bias, y, ldy, egate, ldegate, egate_act, N, H, W, Cin, Cout, stride, act, s) == 0) {
        const double pix = (double)N * a.Ho * a.Wo;
        wu_prof_post(fam, s, 2.0 * pix * Cout * 9.0 * Cin, ((double)N * H * W * Cin + pix * Cout) * esz + 9.0 * Cin * Cout * esz);
        WU_LAUNCH_CHECK("conv3x3 (gathered rows)");
        return 0;
    }
    if (dtype == WU_BF16) {
        if (stride == 1) { if (m) launch_conv<bf16_t, 1, true>(a, lds, (int)grid, s); else launch_conv<bf16_t, 1, false>(a, lds, (int)grid, s); }
        else launch_conv<bf16_t, 2, false>(a, lds, (int)grid, s);
    } else {
        if (stride == 1) { if (m) launch_conv<float, 1, true>(a, lds, (int)grid, s); else launch_conv<float, 1, false>(a, lds, (int)grid, s); }
        else launch_conv<float, 2, false>(a, lds, (int)grid, s);
    }
    const double pix = (double)N * a.Ho * a.Wo;
    wu_prof_post(fam, s, 2.0 * pix * Cout * 9.0 * Cin,
                 ((double)N * H * W * Cin * (m ? 2 : 1) + pix * Cout) * esz + 9.0 * Cin * Cout * esz);
    WU_LAUNCH_CHECK("conv3x3_mfma");
    return 0;
}

// ReLU gates carried as bits between a forward conv and the data-gradient conv of the same block (wu_kernels.h, "gate bits").
// Only the LDS-DMA kernel (conv3x3_mfma_v2) has this epilogue: callers ask wu_conv3x3_gate_bits_supported first.
extern "C" int wu_conv3x3_gate_bits_supported(int H, int W, int ldx, int ldy, int Cin, int Cout, int dtype) {
    return (g_wu_opt[WU_OPT_CONV_V2] && conv_v2_eligible(H, W, ldx, ldy, Cin, Cout, 1, dtype, false)) ? 1 : 0;
}

extern "C" size_t wu_gate_bits_bytes(int N, int H, int W, int C) { return (size_t)N * H * W * (size_t)(C / 64) * 2 * sizeof(unsigned); }

extern "C" int wu_conv3x3_fwd_bits(const void* x, int ldx, const void* w_packed, const float* bias, void* y, int ldy,
                                   void* gate_bits_out, const void* egate_bits, int N, int H, int W, int Cin, int Cout,
                                   int act, int dtype, void* stream) {
    WU_REQUIRE(N > 0 && H > 0 && W > 0 && x && y && w_packed, "conv3x3_fwd_bits: bad args");
    WU_REQUIRE(wu_conv3x3_gate_bits_supported(H, W, ldx, ldy, Cin, Cout, dtype), "conv3x3_fwd_bits: shape/dtype outside the LDS-DMA conv (ask wu_conv3x3_gate_bits_supported)");
    WU_REQUIRE(ldx >= Cin && ldy >= Cout && (ldx * 2) % 16 == 0 && (ldy * 2) % 16 == 0, "conv3x3_fwd_bits: bad ld (%d,%d)", ldx, ldy);
    WU_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)w_packed % 16) == 0, "conv3x3_fwd_bits: pointers must be 16-B aligned");
    WU_REQUIRE(!(gate_bits_out && egate_bits), "conv3x3_fwd_bits: a pass either writes gate bits (forward) or reads them (data gradient)");
    WU_REQUIRE((unsigned long long)N * H * W * (unsigned long long)(Cout / 64) * 2 < (1ull << 32), "conv3x3_fwd_bits: gate-word index must fit 32 bits");
    if (gate_bits_out) WU_REQUIRE(act == WU_ACT_RELU && ((uintptr_t)gate_bits_out % 4) == 0, "conv3x3_fwd_bits: gate bits are written with act == RELU");
    if (egate_bits) WU_REQUIRE(act == WU_ACT_NONE && bias == nullptr && ((uintptr_t)egate_bits % 4) == 0, "conv3x3_fwd_bits: a gated pass has no bias / activation");
    hipStream_t s = (hipStream_t)stream;
    wu_prof_pre(WU_FAM_CONV_FWD, s);
    const int rc = conv_v2_launch(x, ldx, w_packed, bias, y, ldy, nullptr, 0, 0, N, H, W, Cin, Cout, act, s, nullptr, 0, gate_bits_out, egate_bits);
    WU_REQUIRE(rc == 0, "conv3x3_fwd_bits: grid too large");
    // a gate word is 1/16 of the bf16 tensor it stands for
    wu_prof_post(WU_FAM_CONV_FWD, s, 2.0 * N * H * W * (double)Cout * 9.0 * Cin,
                 ((double)N * H * W * (Cin + Cout) + 9.0 * Cin * Cout) * 2 + ((gate_bits_out || egate_bits) ? (double)N * H * W * Cout / 8.0 : 0.0));
    WU_LAUNCH_CHECK("conv3x3_mfma_v2 (gate bits)");
    return 0;
}

// conv3x3 + bias + ReLU with its 2x2 max-pool from the same epilogue (cunet.py:45-46, 49-50, 53-54: every encoder block's
// second conv feeds both the skip tensor y and max_pool2d(y)).  On the LDS-DMA path the pooled tensor is written by the conv
// kernel itself (the stand-alone pool kernel re-reads all of y: 470 MB per B=32 step); otherwise conv, then the pool kernel.
extern "C" int wu_conv3x3_relu_pool_fwd(const void* x, int ldx, const void* w_packed, const float* bias, void* y, int ldy,
                                        void* pool, int ldpool, int N, int H, int W, int Cin, int Cout, int dtype, void* stream) {
    const int esz = dtype == WU_BF16 ? 2 : 4;
    WU_REQUIRE(pool && ldpool >= Cout && (ldpool * esz) % 16 == 0 && ((uintptr_t)pool % 16) == 0, "conv3x3_relu_pool_fwd: bad pool output");
    WU_REQUIRE(H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, "conv3x3_relu_pool_fwd: H=%d W=%d must be even (as for wu_maxpool2_fwd)", H, W);
    hipStream_t s = (hipStream_t)stream;
    const bool fused = g_wu_opt[WU_OPT_CONV_V2] && dtype == WU_BF16 && x && y && w_packed && N > 0 && Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 64 == 0 &&
                       ldx >= Cin && ldy >= Cout && (ldx * esz) % 16 == 0 && (ldy * esz) % 16 == 0 &&
                       ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)w_packed % 16) == 0 &&
                       conv_v2_eligible(H, W, ldx, ldy, Cin, Cout, 1, dtype, false);
    if (fused) {
        wu_prof_pre(WU_FAM_CONV_FWD, s);
        const int rc = conv_v2_launch(x, ldx, w_packed, bias, y, ldy, nullptr, 0, 0, N, H, W, Cin, Cout, WU_ACT_RELU, s, pool, ldpool);
        WU_REQUIRE(rc == 0, "conv3x3_relu_pool_fwd: grid too large");
        wu_prof_post(WU_FAM_CONV_FWD, s, 2.0 * N * H * W * (double)Cout * 9.0 * Cin, ((double)N * H * W * (Cin + Cout) + 9.0 * Cin * Cout) * esz);
        WU_LAUNCH_CHECK("conv3x3_mfma_v2 (+pool)");
        return 0;
    }
    const int rc = wu_conv3x3_fwd(x, ldx, w_packed, bias, y, ldy, N, H, W, Cin, Cout, 1, WU_ACT_RELU, nullptr, 0, 0, nullptr, 0, 0, dtype, stream);
    if (rc) return rc;
    return wu_maxpool2_fwd(y, ldy, pool, ldpool, N, H, W, Cout, dtype, stream);
}

// ... and, from the same epilogue, TWO bits per element of y (round 4): the ReLU gate (y > 0) and "this element is its 2x2 window's first
// maximum" -- both in the gate-bit word layout of wu_conv3x3_fwd_bits (wu_kernels.h).  wu_maxpool2_bwd_bits then routes the pooled gradient,
// adds the skip gradient and applies the gate from 2 bits per element instead of re-reading y (470 MB per B = 32 step over the three
// encoder levels).  LDS-DMA path only: ask wu_conv3x3_gate_bits_supported first.
extern "C" int wu_conv3x3_relu_pool_bits_fwd(const void* x, int ldx, const void* w_packed, const float* bias, void* y, int ldy,
                                             void* pool, int ldpool, void* gate_bits_out, void* sel_bits_out,
                                             int N, int H, int W, int Cin, int Cout, int dtype, void* stream) {
    WU_REQUIRE(N > 0 && H > 0 && W > 0 && x && y && w_packed && pool && gate_bits_out && sel_bits_out, "conv3x3_relu_pool_bits_fwd: bad args");
    WU_REQUIRE(wu_conv3x3_gate_bits_supported(H, W, ldx, ldy, Cin, Cout, dtype), "conv3x3_relu_pool_bits_fwd: shape/dtype outside the LDS-DMA conv (ask wu_conv3x3_gate_bits_supported)");
    WU_REQUIRE(H % 2 == 0 && W % 2 == 0, "conv3x3_relu_pool_bits_fwd: H=%d W=%d must be even", H, W);
    WU_REQUIRE(ldx >= Cin && ldy >= Cout && ldpool >= Cout && (ldx * 2) % 16 == 0 && (ldy * 2) % 16 == 0 && (ldpool * 2) % 16 == 0, "conv3x3_relu_pool_bits_fwd: bad ld");
    WU_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)w_packed % 16) == 0 && ((uintptr_t)pool % 16) == 0 &&
               ((uintptr_t)gate_bits_out % 4) == 0 && ((uintptr_t)sel_bits_out % 4) == 0, "conv3x3_relu_pool_bits_fwd: alignment");
    hipStream_t s = (hipStream_t)stream;
    wu_prof_pre(WU_FAM_CONV_FWD, s);
    const int rc = conv_v2_launch(x, ldx, w_packed, bias, y, ldy, nullptr, 0, 0, N, H, W, Cin, Cout, WU_ACT_RELU, s, pool, ldpool, gate_bits_out, nullptr, sel_bits_out);
    WU_REQUIRE(rc == 0, "conv3x3_relu_pool_bits_fwd: grid too large");
    wu_prof_post(WU_FAM_CONV_FWD, s, 2.0 * N * H * W * (double)Cout * 9.0 * Cin, ((double)N * H * W * (Cin + Cout) + 9.0 * Cin * Cout) * 2 + (double)N * H * W * Cout / 4.0);
    WU_LAUNCH_CHECK("conv3x3_mfma_v2 (+pool +bits)");
    return 0;
}

// conv3x3 + bias + ReLU of the last decoder conv AND the network's head, tanh(conv1x1(y) + b) (cunet.py:78-82), in one launch (round 4): the
// head is 8 extra MFMAs per wave and tile on the packed registers of the conv's epilogue, so the stand-alone head kernel's re-read of y
// (268 MB at B = 32 256x256) disappears; with y == NULL (a forward nobody differentiates) the 64-channel tensor is never written either.
// LDS-DMA path only (bf16, Cout == 64, Cin in [64, 256)): ask wu_conv3x3_relu_head_supported first -- there is no fallback in here.
extern "C" int wu_conv3x3_relu_head_supported(int H, int W, int ldx, int ldy, int Cin, int Cout, int dtype) {
    return (g_wu_opt[WU_OPT_CONV_V2] && Cout == 64 && Cin < 256 && conv_v2_eligible(H, W, ldx, ldy, Cin, Cout, 1, dtype, false)) ? 1 : 0;
}

extern "C" int wu_conv3x3_relu_head_fwd(const void* x, int ldx, const void* w_packed, const float* bias, void* y, int ldy,
                                        const float* head_w, const float* head_bias, float* out_nchw,
                                        int N, int H, int W, int Cin, int Cout, int dtype, void* stream) {
    WU_REQUIRE(N > 0 && H > 0 && W > 0 && x && w_packed && bias && head_w && head_bias && out_nchw, "conv3x3_relu_head_fwd: bad args");
    WU_REQUIRE(wu_conv3x3_relu_head_supported(H, W, ldx, y ? ldy : 64, Cin, Cout, dtype), "conv3x3_relu_head_fwd: shape/dtype outside the fused head (ask wu_conv3x3_relu_head_supported)");
    WU_REQUIRE(ldx >= Cin && (ldx * 2) % 16 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w_packed % 16) == 0, "conv3x3_relu_head_fwd: input alignment / ld");
    if (y) WU_REQUIRE(ldy >= Cout && (ldy * 2) % 16 == 0 && ((uintptr_t)y % 16) == 0, "conv3x3_relu_head_fwd: output alignment / ld");
    WU_REQUIRE(((uintptr_t)head_w % 16) == 0 && ((uintptr_t)out_nchw % 4) == 0, "conv3x3_relu_head_fwd: head weights must be 16-B aligned");
    WU_REQUIRE((unsigned long long)N * 3 * H * W < (1ull << 32), "conv3x3_relu_head_fwd: image index must fit 32 bits");
    hipStream_t s = (hipStream_t)stream;
    wu_prof_pre(WU_FAM_CONV_FWD, s);
    const int rc = conv_v2_launch(x, ldx, w_packed, bias, y, y ? ldy : 64, nullptr, 0, 0, N, H, W, Cin, Cout, WU_ACT_RELU, s, nullptr, 0, nullptr, nullptr, nullptr,
                                  head_w, head_bias, out_nchw);
    WU_REQUIRE(rc == 0, "conv3x3_relu_head_fwd: launch refused (%d)", rc);
    wu_prof_post(WU_FAM_CONV_FWD, s, 2.0 * N * H * W * (double)Cout * (9.0 * Cin + 3.0),
                 ((double)N * H * W * (Cin + (y ? Cout : 0)) + 9.0 * Cin * Cout) * 2 + (double)N * H * W * 12.0);
    WU_LAUNCH_CHECK("conv3x3_mfma_v2 (+head)");
    return 0;
}
