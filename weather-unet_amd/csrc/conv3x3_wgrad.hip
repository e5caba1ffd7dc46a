// Weight gradient of conv3x3 (pad 1, stride 1|2) on the matrix cores -- autograd's conv wgrad for the
// convs of nets.py:18-33 (t_cls_train.py:272,307 call .backward()).
//
//   dW[tap][co][ci] = sum over output pixels  dY'[pix][co] * X[pix shifted by tap][ci],   dY' = dY * act'(Y)
//
// GEMM view per tap: M = Cout, N = Cin, K = output pixels.  A workgroup (4 waves, 2x2) owns a 64(co) x
// 64(ci) block for ALL nine taps -- 9 accumulators of 32x32 per wave (144 fp32 registers per lane) -- and
// walks a strided list of pixel tiles (split-K).  Per tile the dY' tile and the X halo tile are staged
// once in LDS (pixel-major, as they lie in HBM) and re-used by the nine taps: one dY fragment + nine
// shifted X fragments per K-step.  The reduction index (pixel) is the slow index of both NHWC operands,
// so bf16 fragments are fetched with the CDNA4 transposing LDS read ds_read_b64_tr_b16 (4 pixels x 16
// channels -> column-major), no register shuffles; fp32 fragments are plain ds_read_b32.
// The 64-B halves of each 128-B pixel row are swapped on odd pixel pairs ((x>>1)&1) so the four pixel
// rows of a transposed read fall in four disjoint bank ranges.
// Partial blocks go to per-split slabs; a second kernel sums the slabs in fixed order (deterministic),
// transposes to OIHW and optionally accumulates into .grad.  dbias rides along (ci-block 0 only).
#include <type_traits>
#include <utility>

#include "wu_common.h"
#include "wgrad_internal.h"

namespace {

// compile-time loop f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>) (as in conv3x3_wgrad_v2.hip: a 72-step `#pragma unroll` body stays rolled)
template <typename F, int... I> __device__ __forceinline__ void wgrad_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void wgrad_static_for(F&& f) { wgrad_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

struct WgradArgs {
    const void* x; const void* dy; const void* y;
    float* slab;       // [9][Cout][splits][Cin]: the partials of one output row are contiguous for the reducer
    float* bslab;      // [splits][Cout]
    int ldx, lddy, ldy;
    int N, H, W, Ho, Wo, Cin, Cout;
    int act;
    int tw_log2, tiles_x, tiles_y, ntiles;   // pixel tiles (over all images)
    int splits, co_blocks, ci_blocks;
    int halo_w, halo_h, halo_pix;
};

template <typename T> struct WTraits;
template <> struct WTraits<bf16_t> { static constexpr int kRowBytes = 128; static constexpr int kStepPix = 16; };
template <> struct WTraits<float> { static constexpr int kRowBytes = 256; static constexpr int kStepPix = 2; };

// LDS byte offset of (pixel row `row`, whose x-coordinate inside its image row is `xc`, channel byte `cb`)
template <typename T> __device__ __forceinline__ int lds_off(int row, int xc, int cb) {
    if (sizeof(T) == 2) return row * 128 + (cb ^ (((xc >> 1) & 1) << 6));
    return row * 256 + cb;
}

template <typename T, int STRIDE, int P>
__global__ __launch_bounds__(256, STRIDE == 1 ? 2 : 1) void conv3x3_wgrad_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RB = WTraits<T>::kRowBytes;
    constexpr int SLOTS = RB / 16;            // 16-B slots per pixel row (64 channels)
    constexpr int E = 16 / (int)sizeof(T);
    char* dy_lds = smem;                      // [P][64 co]
    char* x_lds = smem + P * RB;              // [halo_pix][64 ci]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;  // co / ci quadrant
    int bid = blockIdx.x;
    const int split = bid % a.splits; bid /= a.splits;
    const int cib = bid % a.ci_blocks;
    const int cob = bid / a.ci_blocks;
    const int TW = 1 << a.tw_log2, TH = P >> a.tw_log2;

    f32x16_t acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float bsum = 0.f;

    // Stride 2 (round 4): the next tile's operands travel global -> REGISTERS under this tile's MFMAs and are written to LDS between two
    // barriers.  (The X halo of a stride-2 tile is 4x the stride-1 one -- 75 KiB + 16 KiB of dY, one workgroup per CU -- and used to be
    // staged synchronously in front of every tile's MFMAs: nothing overlapped, 117-190 TFLOP/s.)  Tile-invariant item descriptors:
    // X item k of this thread = halo pixel (hy, hx), 16-byte slot sx; dY item k = tile pixel (ry, rx), slot sd.
    // staging items per thread: halo pixels x slots / 256 -- at most 585 x 8 (bf16, P = 128) / 325 x 16 (fp32, P = 64); the host checks
    constexpr int NX = STRIDE == 2 ? (sizeof(T) == 2 ? 19 : 21) : 1, ND = STRIDE == 2 ? (P * SLOTS + 255) / 256 : 1;
    // Round 4, second pass: every per-item quantity that does not depend on the tile is computed ONCE -- the byte offset of the item from the tile's origin
    // (through a per-image buffer descriptor: rows below the image fall off its end and load zeros), its LDS address, its halo column and border flags.  A tile then
    // costs three VALU instructions + one buffer load per item and one LDS write, no masks: the first version decoded, clamped and 64-bit-addressed every item of
    // every tile (~3 000 instructions per tile on the ONE wave a SIMD holds: the kernel was issue-bound, SQ active 54 %, profiles/r04_s2_wgrad.txt).
    unsigned x_voff[NX], x_meta[NX], d_voff[ND], y_voff[ND], d_meta[ND];       // meta = LDS byte offset | column << 17 | flags << 27 (1 top row, 2 left column, 4 valid)
    uint4 xr[NX], dr[ND], yr[ND];
    constexpr unsigned kOOB = 0x80000000u;
    if constexpr (STRIDE == 2) {
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int item = tid + 256 * k, p_ = item / SLOTS, s_ = item % SLOTS;
            const int hy = p_ / a.halo_w, hx = p_ - hy * a.halo_w;
            const bool ok = p_ < a.halo_pix;
            x_voff[k] = ok ? (unsigned)(((hy * a.W + hx) * a.ldx + s_ * E) * (int)sizeof(T)) : kOOB;
            x_meta[k] = (unsigned)(ok ? lds_off<T>(p_, hx, s_ * 16) : 0) | (unsigned)hx << 17 | ((hy == 0 ? 1u : 0u) | (hx == 0 ? 2u : 0u) | (ok ? 4u : 0u)) << 27;
        }
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            const int item = tid + 256 * k, r_ = item / SLOTS, s_ = item % SLOTS;
            const int ry = r_ >> a.tw_log2, rx = r_ & (TW - 1);
            const bool ok = item < P * SLOTS;
            d_voff[k] = ok ? (unsigned)(((ry * a.Wo + rx) * a.lddy + s_ * E) * (int)sizeof(T)) : kOOB;
            y_voff[k] = ok ? (unsigned)(((ry * a.Wo + rx) * a.ldy + s_ * E) * (int)sizeof(T)) : kOOB;
            d_meta[k] = (unsigned)(ok ? lds_off<T>(r_, r_, s_ * 16) : 0) | (unsigned)rx << 17 | (ok ? 4u : 0u) << 27;
        }
    }
    auto tile_origin = [&](int tile, int& n, int& oh0, int& ow0) __attribute__((always_inline)) {
        int tt = tile;
        const int tx = tt % a.tiles_x; tt /= a.tiles_x;
        const int ty = tt % a.tiles_y;
        n = tt / a.tiles_y;
        oh0 = ty * TH; ow0 = tx * TW;
    };
    auto u4 = [](auto v) __attribute__((always_inline)) { return __builtin_bit_cast(uint4, v); };
    auto prefetch = [&](int tile) __attribute__((always_inline)) {       // STRIDE == 2: one buffer load per item; padding = out-of-range offsets (zeros)
        int n, oh0, ow0;
        tile_origin(tile, n, oh0, ow0);
        n = __builtin_amdgcn_readfirstlane(n); oh0 = __builtin_amdgcn_readfirstlane(oh0); ow0 = __builtin_amdgcn_readfirstlane(ow0);
        // X: descriptor base one row + one pixel BEFORE the image so that halo offsets are non-negative (for n = 0 it points before the tensor: the lanes that
        // would touch those bytes are the flagged top-row / left-column ones); a pixel of row H or below lies past num_records
        const T* xb = (const T*)a.x + ((long long)n * a.H * a.W - (a.W + 1)) * a.ldx + cib * 64;
        const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (int)(((size_t)a.H * a.W + a.W + 1) * a.ldx * sizeof(T)), 0x00020000);
        const int sox = (oh0 * STRIDE * a.W + ow0 * STRIDE) * a.ldx * (int)sizeof(T);
        const unsigned border = (oh0 == 0 ? 1u : 0u) | (ow0 == 0 ? 2u : 0u);
        const unsigned limx = (unsigned)(a.W - (ow0 * STRIDE - 1));               // halo columns >= limx lie right of the image
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const bool bad = ((x_meta[k] >> 27) & border) != 0 || ((x_meta[k] >> 17) & 0x3ffu) >= limx;
            xr[k] = u4(__builtin_amdgcn_raw_buffer_load_b128(rx_, (int)(bad ? kOOB : x_voff[k]), sox, 0));
        }
        const T* dyb = (const T*)a.dy + (size_t)n * a.Ho * a.Wo * a.lddy + cob * 64;
        const __amdgpu_buffer_rsrc_t rd_ = __builtin_amdgcn_make_buffer_rsrc((void*)dyb, 0, (int)((size_t)a.Ho * a.Wo * a.lddy * sizeof(T)), 0x00020000);
        const int sod = (oh0 * a.Wo + ow0) * a.lddy * (int)sizeof(T);
        const unsigned limd = (unsigned)(a.Wo - ow0);
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            const bool bad = ((d_meta[k] >> 17) & 0x3ffu) >= limd;
            dr[k] = u4(__builtin_amdgcn_raw_buffer_load_b128(rd_, (int)(bad ? kOOB : d_voff[k]), sod, 0));
        }
        if (a.y) {
            const T* yb = (const T*)a.y + (size_t)n * a.Ho * a.Wo * a.ldy + cob * 64;
            const __amdgpu_buffer_rsrc_t ry_ = __builtin_amdgcn_make_buffer_rsrc((void*)yb, 0, (int)((size_t)a.Ho * a.Wo * a.ldy * sizeof(T)), 0x00020000);
            const int soy = (oh0 * a.Wo + ow0) * a.ldy * (int)sizeof(T);
#pragma unroll
            for (int k = 0; k < ND; ++k) {
                const bool bad = ((d_meta[k] >> 17) & 0x3ffu) >= limd;
                yr[k] = u4(__builtin_amdgcn_raw_buffer_load_b128(ry_, (int)(bad ? kOOB : y_voff[k]), soy, 0));
            }
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {                // registers -> LDS (padding arrived as zeros)
#pragma unroll
        for (int k = 0; k < NX; ++k)
            if ((x_meta[k] >> 29) & 1u) *(uint4*)(x_lds + (x_meta[k] & 0x1ffffu)) = xr[k];
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            uint4 w_ = dr[k];
            if (a.y) w_ = gate16<T>(w_, yr[k], a.act);
            if ((d_meta[k] >> 29) & 1u) *(uint4*)(dy_lds + (d_meta[k] & 0x1ffffu)) = w_;
        }
    };
    if constexpr (STRIDE == 2) {
        if (split < a.ntiles) prefetch(split);
    }
    // Stride 2, bf16 (round 4): the fragment addresses of a K-step depend on the lane and the tile SHAPE only -- tile pixel r = 16 ks + 8 h + q (+ 4) sits at halo pixel
    // (2 ry, 2 rx), and a tap adds (kh halo_w + kw) rows; the 64-byte swizzle bit of halo column 2 rx + kw is rx & 1 for kw = 0, 1 and its complement for kw = 2.  The
    // runtime tile shape made the compiler recompute all of it per K-step and tap (~180 VALU instructions beside 9 MFMAs on the one wave of a SIMD).
    constexpr int KSN = (sizeof(T) == 2 && STRIDE == 2) ? P / 16 : 1;
    int fa_off[KSN][2], fb_off[KSN][2][2], tap_off[9];
    if constexpr (sizeof(T) == 2 && STRIDE == 2) {
        const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3, h = g >> 1;
        const int acb = (32 * wm + 16 * (g & 1) + 4 * pp) * 2, bcb = (32 * wn + 16 * (g & 1) + 4 * pp) * 2;
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = 16 * ks + 8 * h + q + 4 * u;
                const int ry = r >> a.tw_log2, rx = r & (TW - 1);
                const int hp = ry * STRIDE * a.halo_w + rx * STRIDE;
                fa_off[ks][u] = lds_off<T>(r, r, acb);
                fb_off[ks][u][0] = lds_off<T>(hp, rx * STRIDE, bcb);
                fb_off[ks][u][1] = lds_off<T>(hp, rx * STRIDE + 2, bcb);
            }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) tap_off[tap] = __builtin_amdgcn_readfirstlane(((tap / 3) * a.halo_w + tap % 3) * 128);
    }

    for (int tile = split; tile < a.ntiles; tile += a.splits) {
        if constexpr (STRIDE == 2) {
            commit();
            __syncthreads();
            if (tile + a.splits < a.ntiles) prefetch(tile + a.splits);
        } else {
        int tt = tile;
        const int tx = tt % a.tiles_x; tt /= a.tiles_x;
        const int ty = tt % a.tiles_y;
        const int n = tt / a.tiles_y;
        const int oh0 = ty * TH, ow0 = tx * TW;
        const int ih0 = oh0 * STRIDE - 1, iw0 = ow0 * STRIDE - 1;

        // ---- stage dY' tile (gated by act'(Y)) ----
        {
            const T* dyb = (const T*)a.dy + (size_t)n * a.Ho * a.Wo * a.lddy + cob * 64;
            const T* yb = a.y ? (const T*)a.y + (size_t)n * a.Ho * a.Wo * a.ldy + cob * 64 : nullptr;
            // loads in batches of 4, UNCONDITIONAL from clamped coordinates and zeroed afterwards: a conditional load in a rolled
            // loop is one full memory round trip per item
            static_assert((P * SLOTS) % 1024 == 0 || P * SLOTS < 1024, "dY staging batch");
            for (int item0 = tid; item0 < P * SLOTS; item0 += 1024) {
                uint4 v[4], yv[4];
                bool okv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int item = min(item0 + 256 * u, P * SLOTS - 1);
                    const int r = item / SLOTS, s = item % SLOTS;
                    const int rx = r & (TW - 1), oh = oh0 + (r >> a.tw_log2), ow = ow0 + rx;
                    okv[u] = oh < a.Ho && ow < a.Wo;
                    const size_t o = (size_t)(min(oh, a.Ho - 1) * a.Wo + min(ow, a.Wo - 1));
                    v[u] = *(const uint4*)(dyb + o * a.lddy + s * E);
                    if (yb) yv[u] = *(const uint4*)(yb + o * a.ldy + s * E);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int item = item0 + 256 * u;
                    if (item < P * SLOTS) {
                        const int r = item / SLOTS, s = item % SLOTS;
                        uint4 w = v[u];
                        if (yb) w = gate16<T>(w, yv[u], a.act);
                        *(uint4*)(dy_lds + lds_off<T>(r, r, s * 16)) = okv[u] ? w : make_uint4(0, 0, 0, 0);
                    }
                }
            }
        }
        // ---- stage X halo tile ----
        {
            const T* xb = (const T*)a.x + (size_t)n * a.H * a.W * a.ldx + cib * 64;
            const int nitems = a.halo_pix * SLOTS;
            for (int item0 = tid; item0 < nitems; item0 += 1024) {
                uint4 v[4];
                bool okv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int item = min(item0 + 256 * u, nitems - 1);
                    const int p = item / SLOTS, s = item % SLOTS;
                    const int hy = p / a.halo_w, hx = p - hy * a.halo_w;
                    const int ih = ih0 + hy, iw = iw0 + hx;
                    okv[u] = ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
                    v[u] = *(const uint4*)(xb + (size_t)(min(max(ih, 0), a.H - 1) * a.W + min(max(iw, 0), a.W - 1)) * a.ldx + s * E);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int item = item0 + 256 * u;
                    if (item < nitems) {
                        const int p = item / SLOTS, s = item % SLOTS;
                        const int hy = p / a.halo_w, hx = p - hy * a.halo_w;
                        const uint32_t m = okv[u] ? 0xffffffffu : 0u;      // lane mask, not a 128-bit select (lowered through scratch)
                        *(uint4*)(x_lds + lds_off<T>(p, hx, s * 16)) = make_uint4(v[u].x & m, v[u].y & m, v[u].z & m, v[u].w & m);
                    }
                }
            }
        }
        __syncthreads();
        }

        // ---- bias gradient: column sums of the dY' tile (ci-block 0 only) ----
        if (cib == 0 && a.bslab) {
            const int co = tid & 63, part = tid >> 6;
            for (int r = part; r < P; r += 4) bsum += ElemTraits<T>::load((const T*)(dy_lds + lds_off<T>(r, r, co * (int)sizeof(T))));
        }

        // ---- MFMA over the tile's pixels ----
        if constexpr (sizeof(T) == 2 && STRIDE == 2) {
            // tile-invariant fragment addresses (tables built once per workgroup); the 72 (K-step, tap) MFMAs of a tile software-pipelined by hand: the X fragment of
            // step s + LA and the dY fragment of the next K-step are requested ahead of the MFMA of step s (one wave per SIMD: nothing else covers an LDS round trip --
            // the compiler's own order was two reads / wait / MFMA, 54 exposed waits per tile); sched_barrier pins the order, the compiler counts the lgkmcnt waits
            constexpr int KS = P / 16, NS = KS * 9, LA = 5;
            auto load_a = [&](int ks) __attribute__((always_inline)) {
                const s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(dy_lds + fa_off[ks][0]));
                const s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(dy_lds + fa_off[ks][1]));
                return make_uint4(((const uint32_t*)&a0)[0], ((const uint32_t*)&a0)[1], ((const uint32_t*)&a1)[0], ((const uint32_t*)&a1)[1]);
            };
            auto load_b = [&](int st) __attribute__((always_inline)) {
                const int ks = st / 9, tap = st % 9, v = tap % 3 == 2 ? 1 : 0;
                const s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(x_lds + fb_off[ks][0][v] + tap_off[tap]));
                const s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(x_lds + fb_off[ks][1][v] + tap_off[tap]));
                return make_uint4(((const uint32_t*)&b0)[0], ((const uint32_t*)&b0)[1], ((const uint32_t*)&b1)[0], ((const uint32_t*)&b1)[1]);
            };
            uint4 afr[2], bfr[LA + 1];
            afr[0] = load_a(0);
#pragma unroll
            for (int j = 0; j < LA; ++j) bfr[j] = load_b(j);
            wgrad_static_for<NS>([&](auto st_tag) __attribute__((always_inline)) {
                constexpr int st = decltype(st_tag)::value, ks = st / 9, tap = st % 9;
                if constexpr (st + LA < NS) bfr[(st + LA) % (LA + 1)] = load_b(st + LA);
                if constexpr (tap == 3 && ks + 1 < KS) afr[(ks + 1) & 1] = load_a(ks + 1);
                __builtin_amdgcn_sched_barrier(0);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, afr[ks & 1]), __builtin_bit_cast(bf16x8_t, bfr[st % (LA + 1)]), acc[tap], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        } else if constexpr (sizeof(T) == 2) {
            const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
            const int h = g >> 1;
            const int acb = (32 * wm + 16 * (g & 1) + 4 * pp) * 2;   // channel byte of this lane's A address
            const int bcb = (32 * wn + 16 * (g & 1) + 4 * pp) * 2;
#pragma unroll 2
            for (int ks = 0; ks < P / 16; ++ks) {
                // this lane addresses tile pixel r0 (first read) and r0 + 4 (second read)
                const int r0 = 16 * ks + 8 * h + q;
                s16x4_t a0, a1;
                {
                    const int o0 = lds_off<T>(r0, r0, acb), o1 = lds_off<T>(r0 + 4, r0 + 4, acb);
                    a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(dy_lds + o0));
                    a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(dy_lds + o1));
                }
                const uint4 af = make_uint4(((const uint32_t*)&a0)[0], ((const uint32_t*)&a0)[1], ((const uint32_t*)&a1)[0], ((const uint32_t*)&a1)[1]);
                int hp[2], hxx[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int r = r0 + 4 * u;
                    const int ry = r >> a.tw_log2, rx = r & (TW - 1);
                    hp[u] = ry * STRIDE * a.halo_w + rx * STRIDE;
                    hxx[u] = rx * STRIDE;
                }
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int kh = tap / 3, kw = tap % 3;
                    const int o0 = lds_off<T>(hp[0] + kh * a.halo_w + kw, hxx[0] + kw, bcb);
                    const int o1 = lds_off<T>(hp[1] + kh * a.halo_w + kw, hxx[1] + kw, bcb);
                    const s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(x_lds + o0));
                    const s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(x_lds + o1));
                    const uint4 bf = make_uint4(((const uint32_t*)&b0)[0], ((const uint32_t*)&b0)[1], ((const uint32_t*)&b1)[0], ((const uint32_t*)&b1)[1]);
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af), __builtin_bit_cast(bf16x8_t, bf), acc[tap], 0, 0, 0);
                }
            }
        } else {
            const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll 2
            for (int ks = 0; ks < P / 2; ++ks) {
                const int r = 2 * ks + lh;
                const float av = *(const float*)(dy_lds + lds_off<T>(r, r, (32 * wm + l31) * 4));
                const int ry = r >> a.tw_log2, rx = r & (TW - 1);
                const int hp = ry * STRIDE * a.halo_w + rx * STRIDE;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int kh = tap / 3, kw = tap % 3;
                    const float bv = *(const float*)(x_lds + lds_off<T>(hp + kh * a.halo_w + kw, 0, (32 * wn + l31) * 4));
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tap], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- write the partial block: slab[tap][co][split][ci] ----
    {
        const int l31 = lane & 31, lh = lane >> 5;
        float* sl = a.slab + (size_t)split * a.Cin;
        const size_t rs = (size_t)a.splits * a.Cin;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = cob * 64 + 32 * wm + (i & 3) + 8 * (i >> 2) + 4 * lh;
                const int ci = cib * 64 + 32 * wn + l31;
                sl[((size_t)tap * a.Cout + co) * rs + ci] = acc[tap][i];
            }
    }
    if (cib == 0 && a.bslab) {
        float* red = (float*)smem;     // all waves are past the last tile's barrier
        red[tid] = bsum;
        __syncthreads();
        if (tid < 64) a.bslab[(size_t)split * a.Cout + cob * 64 + tid] = red[tid] + red[tid + 64] + red[tid + 128] + red[tid + 192];
    }
}

// Fixed-order reduction of the split-K partials slab[tap][co][split][ci] (+ bias partials [split][Cout]) into OIHW fp32.
// One WAVE per (co, tap) row: its splits x Cin partials are ONE contiguous run, read as float4 by TX (a power of two <= Cin/4, 64) lanes x
// G = 64/TX split groups (group g takes splits g, g+G, ...); the G partial sums are combined with xor-shuffles in a fixed tree ->
// deterministic for a given (splits, Cin).  No LDS and <= 32 registers, on purpose: the reducer follows every weight-gradient
// launch on the side stream, and a workgroup that needs LDS or a 64-register wave cannot be placed on a CU whose 160 KiB / 480 of
// 512 registers per SIMD lane are held by a persistent conv workgroup -- it then waits for a whole conv KERNEL to retire (measured:
// 90 us per reducer in the step against 12 us stand-alone).
// Round 4: WPR waves share a (co, tap) row, each taking a ci range (Cin / WPR channels, whole float4s).  With one wave per row the
// 256-split rows of the Cin = 64 layers were 64 dependent 16-byte loads per lane from 576 waves: 2.3 MB in flight chip-wide, 1.9 TB/s,
// 20 us for 37.7 MB.  WPR = 4 there: 16 loads per lane from 2304 waves.  Still no LDS, still <= 32 registers, still one fixed order per
// (splits, Cin): a lane adds its splits g, g + G, ... in that order and the G lane groups are combined by the same xor tree.
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(32))) void wgrad_reduce_kernel(
    const float* __restrict__ slab, const float* __restrict__ bslab, float* __restrict__ dw, float* __restrict__ dbias,
    int splits, int Cout, int Cin, int accumulate, int wpr) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));    // provably wave-uniform: scalar row base
    const int row = wid / wpr, part = wid - row * wpr;            // (co, tap) rows, tap fastest: row = co * 9 + t
    if (row >= Cout * 9) return;
    const int co = row / 9, t = row - co * 9;
    const int c4n = (Cin >> 2) / wpr, c4lo = part * c4n;          // this wave's float4 columns [c4lo, c4lo + c4n)
    int TX = 64;                                                  // lanes across ci (float4 each): the largest power of two <= min(c4n, 64)
    while (TX > c4n) TX >>= 1;
    const int G = 64 / TX;
    const int tx = lane & (TX - 1), g = lane / TX;
    // wave-uniform row base (scalar registers) + a 32-bit per-lane element offset: the loads take the saddr form, no 64-bit VALU adds
    const float* base = slab + ((size_t)t * Cout + co) * splits * Cin;
    for (int c4b = 0; c4b < c4n; c4b += TX) {                     // wave-uniform trip count (the shuffles below need every lane)
        const int c4 = c4b + tx;
        const bool valid = c4 < c4n;
        unsigned off = (unsigned)(g * Cin + (c4lo + (valid ? c4 : 0)) * 4);
        const unsigned step = (unsigned)(G * Cin);
        float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);           // one chain: registers are what places this kernel beside a conv workgroup
        int k = g;
        for (; k + G < splits; k += 2 * G, off += 2 * step) {     // two loads in flight per lane
            const float4 v0 = *(const float4*)(base + off);
            const float4 v1 = *(const float4*)(base + off + step);
            sacc.x += v0.x; sacc.y += v0.y; sacc.z += v0.z; sacc.w += v0.w;
            sacc.x += v1.x; sacc.y += v1.y; sacc.z += v1.z; sacc.w += v1.w;
        }
        if (k < splits) {
            const float4 v0 = *(const float4*)(base + off);
            sacc.x += v0.x; sacc.y += v0.y; sacc.z += v0.z; sacc.w += v0.w;
        }
        for (int m = TX; m < 64; m <<= 1) {                       // combine the G groups (lanes tx + j * TX): fixed xor tree
            sacc.x += __shfl_xor(sacc.x, m); sacc.y += __shfl_xor(sacc.y, m);
            sacc.z += __shfl_xor(sacc.z, m); sacc.w += __shfl_xor(sacc.w, m);
        }
        if (g == 0 && valid) {
            float* o = dw + ((size_t)co * Cin + (c4lo + c4) * 4) * 9 + t;
            if (accumulate) { o[0] += sacc.x; o[9] += sacc.y; o[18] += sacc.z; o[27] += sacc.w; }
            else { o[0] = sacc.x; o[9] = sacc.y; o[18] = sacc.z; o[27] = sacc.w; }
        }
    }
    if (dbias && t == 0 && part == 0) {      // bias partials [split][Cout]: lanes stride the splits, xor-shuffle tree
        float b = 0.f;
        for (int k = lane; k < splits; k += 64) b += bslab[(size_t)k * Cout + co];
        for (int m = 1; m < 64; m <<= 1) b += __shfl_xor(b, m);
        if (lane == 0) dbias[co] = accumulate ? dbias[co] + b : b;
    }
}

void launch_wgrad_reduce(const float* slab, const float* bslab, float* dw, float* dbias, int splits, int Cout, int Cin, int accumulate, hipStream_t s) {
    // waves per row: ~16 loads per lane (splits x Cin/4 float4 over 64 lanes), a power of two that divides Cin / 4 into whole float4 ranges
    int wpr = 1;
    while (wpr < 8 && (long long)splits * (Cin >> 2) > 64LL * 16 * wpr && ((Cin >> 2) % (2 * wpr)) == 0) wpr <<= 1;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(Cout * 9 * wpr, 4)), dim3(256), 0, s, slab, bslab, dw, dbias, splits, Cout, Cin, accumulate, wpr);
}

struct WPlan { int P, twl, tiles_x, tiles_y, ntiles, splits, halo_w, halo_h; size_t lds, ws; };

WPlan wgrad_plan(int N, int H, int W, int Cin, int Cout, int stride, int dtype) {
    WPlan p;
    const bool bf = dtype == WU_BF16;
    p.P = bf ? (stride == 1 ? 256 : 128) : (stride == 1 ? 128 : 64);
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    int twl = 5;
    while (twl > 2 && (1 << (twl - 1)) >= Wo) --twl;
    p.twl = twl;
    const int TW = 1 << twl, TH = p.P >> twl;
    p.tiles_x = cdiv(Wo, TW); p.tiles_y = cdiv(Ho, TH);
    p.ntiles = N * p.tiles_x * p.tiles_y;
    const int blocks = (Cout / 64) * (Cin / 64);
    int splits = cdiv(512, blocks);
    if (splits > p.ntiles) splits = p.ntiles;
    if (splits < 1) splits = 1;
    p.splits = splits;
    p.halo_w = (TW - 1) * stride + 3; p.halo_h = (TH - 1) * stride + 3;
    const int rb = bf ? 128 : 256;
    p.lds = (size_t)(p.P + p.halo_w * p.halo_h) * rb;
    p.ws = ((size_t)splits * 9 * Cout * Cin + (size_t)splits * Cout) * sizeof(float);
    return p;
}

template <typename T, int STRIDE, int P>
void launch_wgrad(const WgradArgs& a, size_t lds, int grid, hipStream_t s) {
    auto kern = conv3x3_wgrad_kernel<T, STRIDE, P>;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
}

}  // namespace

extern "C" size_t wu_conv3x3_wgrad_workspace(int N, int H, int W, int Cin, int Cout, int stride, int dtype) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin % 64 || Cout % 64 || (stride != 1 && stride != 2)) return 0;
    size_t ws = wgrad_plan(N, H, W, Cin, Cout, stride, dtype).ws;
    if (wgrad_v2_eligible(H, W, Cin, Cout, stride, dtype, false)) {
        const size_t w2 = wgrad_v2_plan(N, H, W, Cin, Cout).ws;
        if (w2 > ws) ws = w2;
    }
    return ws;
}

extern "C" int wu_conv3x3_wgrad(const void* x, int ldx, const void* dy, int lddy, const void* y, int ldy_, int act,
                                float* dw_oihw, float* dbias, void* workspace, size_t workspace_bytes,
                                int N, int H, int W, int Cin, int Cout, int stride, int accumulate,
                                int dtype, void* stream) {
    const int esz = dtype == WU_BF16 ? 2 : 4;
    WU_REQUIRE(dtype == WU_F32 || dtype == WU_BF16, "conv3x3_wgrad: bad dtype");
    WU_REQUIRE(stride == 1 || stride == 2, "conv3x3_wgrad: stride");
    WU_REQUIRE(N > 0 && H > 0 && W > 0 && Cin % 64 == 0 && Cout % 64 == 0 && Cin > 0 && Cout > 0, "conv3x3_wgrad: Cin=%d Cout=%d must be multiples of 64", Cin, Cout);
    WU_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0 && (ldx * esz) % 16 == 0 && (lddy * esz) % 16 == 0 && ldx >= Cin && lddy >= Cout, "conv3x3_wgrad: alignment");
    if (y) WU_REQUIRE(((uintptr_t)y % 16) == 0 && (ldy_ * esz) % 16 == 0 && ldy_ >= Cout, "conv3x3_wgrad: y alignment");
    hipStream_t s = (hipStream_t)stream;
    const int fam = stride == 2 ? WU_FAM_WGRAD_S2 : WU_FAM_WGRAD;
    const double flops = 2.0 * N * ((H - 1) / stride + 1) * ((W - 1) / stride + 1) * 9.0 * Cin * Cout;
    const double bytes = ((double)N * H * W * Cin + (double)N * ((H - 1) / stride + 1) * ((W - 1) / stride + 1) * Cout * (y ? 2 : 1)) * esz + 9.0 * Cin * Cout * 4;
    if (g_wu_opt[WU_OPT_WGRAD_V2] && wgrad_v2_eligible(H, W, Cin, Cout, stride, dtype, y != nullptr)) {
        const WgradV2Plan p2 = wgrad_v2_plan(N, H, W, Cin, Cout);
        WU_REQUIRE(workspace && workspace_bytes >= p2.ws && ((uintptr_t)workspace % 16) == 0, "conv3x3_wgrad: workspace too small (%zu < %zu)", workspace_bytes, p2.ws);
        WU_REQUIRE(((size_t)H * W + W + 2) * (size_t)(ldx > lddy ? ldx : lddy) * 2 < (1ull << 31), "conv3x3_wgrad: image too large for 32-bit offsets");
        float* slab = (float*)workspace;
        float* bslab = dbias ? slab + (size_t)p2.splits * 9 * Cout * Cin : nullptr;
        wu_prof_pre(fam, s);
        wgrad_v2_launch(x, ldx, dy, lddy, slab, bslab, N, H, W, Cin, Cout, p2, s);
        wu_prof_post(fam, s, flops, bytes);
        launch_wgrad_reduce(slab, bslab, dw_oihw, dbias, p2.splits, Cout, Cin, accumulate, s);
        WU_LAUNCH_CHECK("conv3x3_wgrad_v2");
        return 0;
    }
    const WPlan p = wgrad_plan(N, H, W, Cin, Cout, stride, dtype);
    WU_REQUIRE(workspace && workspace_bytes >= p.ws && ((uintptr_t)workspace % 16) == 0, "conv3x3_wgrad: workspace too small (%zu < %zu)", workspace_bytes, p.ws);
    WU_REQUIRE(p.lds <= 160 * 1024, "conv3x3_wgrad: LDS %zu", p.lds);
    if (stride == 2) WU_REQUIRE(p.halo_w * p.halo_h * (dtype == WU_BF16 ? 8 : 16) <= (dtype == WU_BF16 ? 19 : 21) * 256, "conv3x3_wgrad: halo %d x %d exceeds the stride-2 staging registers", p.halo_w, p.halo_h);
    WgradArgs a;
    a.x = x; a.dy = dy; a.y = y;
    a.slab = (float*)workspace;
    a.bslab = dbias ? a.slab + (size_t)p.splits * 9 * Cout * Cin : nullptr;
    a.ldx = ldx; a.lddy = lddy; a.ldy = ldy_;
    a.N = N; a.H = H; a.W = W; a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1; a.Cin = Cin; a.Cout = Cout;
    a.act = act;
    a.tw_log2 = p.twl; a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.ntiles = p.ntiles;
    a.splits = p.splits; a.co_blocks = Cout / 64; a.ci_blocks = Cin / 64;
    a.halo_w = p.halo_w; a.halo_h = p.halo_h; a.halo_pix = p.halo_w * p.halo_h;
    const int grid = p.splits * a.co_blocks * a.ci_blocks;
    wu_prof_pre(fam, s);
    if (dtype == WU_BF16) {
        if (stride == 1) launch_wgrad<bf16_t, 1, 256>(a, p.lds, grid, s); else launch_wgrad<bf16_t, 2, 128>(a, p.lds, grid, s);
    } else {
        if (stride == 1) launch_wgrad<float, 1, 128>(a, p.lds, grid, s); else launch_wgrad<float, 2, 64>(a, p.lds, grid, s);
    }
    wu_prof_post(fam, s, flops, bytes);
    launch_wgrad_reduce(a.slab, a.bslab, dw_oihw, dbias, p.splits, Cout, Cin, accumulate, s);
    WU_LAUNCH_CHECK("conv3x3_wgrad");
    return 0;
}
