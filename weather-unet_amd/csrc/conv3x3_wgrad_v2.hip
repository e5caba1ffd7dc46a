// conv3x3 weight gradient, production path (bf16, stride 1, pre-gated dY, output width > 16).
//
// Same GEMM as conv3x3_wgrad.hip (dW[tap][co][ci] = sum_pix dY[pix][co] * X[pix + tap][ci]; autograd's conv
// wgrad for nets.py:18-24) re-designed around what limited that kernel on MI355X:
//
//  * 8 waves (two per SIMD, <= 256 registers each) share one 64(co) x 64(ci) x 9-tap block.  A wave owns the
//    32-wide ci fragment `wci`, BOTH co fragments for four of the taps (taps 0-3 or 5-8) plus the centre tap
//    for one co fragment: 9 accumulators of v_mfma_f32_32x32x16_bf16 (144 registers), and per 16-pixel K-step
//    2 dY fragments + 5 shifted X fragments feed 9 MFMAs (0.78 LDS fragment reads per MFMA instead of 1.1).
//    The two wave sets take the even / odd K-steps of a tile and are summed through LDS at the end;
//  * fragments come through ds_read_b64_tr_b16 (the reduction index, the pixel, is the slow NHWC index);
//  * staging is LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write pass; the DMA of tile t+1 is in
//    flight under the MFMAs of tile t (two LDS buffers, ONE barrier per tile).  Out-of-image halo pixels and
//    ragged tile edges are DMA'd from a 16-byte zero constant (the per-lane SOURCE address is free); the LDS
//    images are lane-linear, so the bank swizzle is applied to the source address (cdna guide rule 21);
//  * the tile is fixed at 8 x 32 pixels: every fragment address is lane_base + compile-time immediate, so the
//    fully unrolled K loop contains no address arithmetic.
#include <type_traits>

#include "wu_common.h"
#include "wgrad_internal.h"

namespace {

__device__ const uint4 g_zero16 = {0u, 0u, 0u, 0u};

struct C {
    static constexpr int P = 256, TH = 8;                // pixels per tile (8 rows x 32)
    static constexpr int DY_ROW = 128, DY_SLOTS = 8;     // 64 co x bf16
    static constexpr int HALO_W = 34, HALO_H = TH + 2, HALO_PIX = HALO_W * HALO_H;
    static constexpr int DY_BYTES = P * DY_ROW;          // 32 KiB
    static constexpr int DY_PIECES = DY_BYTES / 1024;    // 32 DMA pieces of 1 KiB
    static constexpr int X_PIECES = (HALO_PIX * 128 + 1023) / 1024;   // 43
    static constexpr int X_BYTES = X_PIECES * 1024;
    static constexpr int BUF = DY_BYTES + X_BYTES;       // 76800 B; two buffers = 150 KiB
    static constexpr int NW = 8;                         // waves per workgroup
    static constexpr int NDY = DY_PIECES / NW;           // 4 DMA pieces per wave
    static constexpr int NX = (X_PIECES + NW - 1) / NW;  // 6
};

struct W2Args {
    const bf16_t* x; const bf16_t* dy;
    float* slab; float* bslab;
    int ldx, lddy;
    int N, H, W, Cin, Cout;
    int tiles_x, tiles_y, ntiles, tiles_per_split, splits, co_blocks, ci_blocks;
    int dma_interleave;     // 1: spread the next tile's DMA issue over the first K-steps; 0: burst right after the barrier
};

// LDS-DMA from inline asm (see conv3x3_mfma_v2.hip): invisible to hipcc's waitcnt bookkeeping, so issuing it in the
// middle of the MFMA loop does not make the compiler drain it before the next fragment read; the kernel waits for
// it itself (dma_wait_all) ahead of the barrier that publishes the buffer.
__device__ __forceinline__ void dma16(const void* g, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ uint4 tr_frag(const char* p0) {
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)p0);
    const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(p0 + 4 * 128));
    return make_uint4(((const uint32_t*)&a)[0], ((const uint32_t*)&a)[1], ((const uint32_t*)&b)[0], ((const uint32_t*)&b)[1]);
}
__device__ __forceinline__ void mma(f32x16_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}

// tap owned by accumulator slot i (0..3) of tap-half HF; slot 4 is the centre tap (4)
template <int HF> __device__ __forceinline__ constexpr int tap_of(int i) { return i == 4 ? 4 : (HF ? 5 + i : i); }

template <int HF, typename Issue>
__device__ __forceinline__ void compute_tile(const char* lds, const int (&a_lane)[2], const int (&b_lane)[3], f32x16_t (&acc)[9], Issue&& issue) {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        // the next tile's DMA pieces are issued inside the first K-steps: their address arithmetic overlaps MFMAs
        issue(kk);
        // K-step ks = 2*kk + parity covers tile pixels 32*kk + 16*parity + [0,16): tile row kk (parity is in the lane base)
        const int a_off = 32 * kk * C::DY_ROW;
        const int b_off = kk * C::HALO_W * 128;
        uint4 af[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) af[m] = tr_frag(lds + a_lane[m] + a_off);
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int tap = tap_of<HF>(i), kh = tap / 3, kw = tap % 3;
            const uint4 bf = tr_frag(lds + b_lane[kw] + b_off + (kh * C::HALO_W + kw) * 128);
            if (i < 4) {
                mma(acc[2 * i], af[0], bf);
                mma(acc[2 * i + 1], af[1], bf);
            } else {
                mma(acc[8], af[HF], bf);
            }
        }
    }
}

__global__ __launch_bounds__(512, 2) void conv3x3_wgrad_v2_kernel(const W2Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform (DMA base, branches)
    const int wci = wave & 1, hf = (wave >> 1) & 1, par = wave >> 2;

    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int blocks = a.co_blocks * a.ci_blocks;
    const int blk = bid % blocks, split = bid / blocks;
    const int cib = blk % a.ci_blocks, cob = blk / a.ci_blocks;
    const int t_begin = split * a.tiles_per_split;
    const int t_end = min(a.ntiles, t_begin + a.tiles_per_split);

    // ---- tile-invariant DMA descriptors, one packed register per 16-B item: LDS slot i = piece*64 + lane
    //      -> (row << 8 | col | channel-slot << 16) of the pixel it is loaded from (-1: always zero) ----
    int dy_d[C::NDY];
#pragma unroll
    for (int j = 0; j < C::NDY; ++j) {
        const int i = (C::NW * j + wave) * 64 + lane;
        const int r = i / C::DY_SLOTS, sl = i % C::DY_SLOTS;
        const int s = sl ^ (((r >> 1) & 1) << 2);
        dy_d[j] = ((r >> 5) << 8) | (r & 31) | (s << 16);
    }
    int x_d[C::NX];
#pragma unroll
    for (int j = 0; j < C::NX; ++j) {
        const int i = (C::NW * j + wave) * 64 + lane;
        const int p = i >> 3, sl = i & 7;
        const int hy = p / C::HALO_W, hx = p - hy * C::HALO_W;
        const int s = sl ^ (((hx >> 1) & 1) << 2);
        x_d[j] = (p < C::HALO_PIX && C::NW * j + wave < C::X_PIECES) ? ((hy << 8) | hx | (s << 16)) : -1;
    }

    // DMA of one tile = NDY + NX pieces per wave; `set_fetch_tile` decodes the tile once, `issue_piece(j)` issues piece j
    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const char* dyb_f = nullptr;
    const char* xb_f = nullptr;
    int oh0_f = 0, ow0_f = 0;
    auto set_fetch_tile = [&](int tile) __attribute__((always_inline)) {
        int tt = tile;
        const int tx = tt % a.tiles_x; tt /= a.tiles_x;
        const int ty = tt % a.tiles_y;
        const int n = tt / a.tiles_y;
        oh0_f = ty * C::TH; ow0_f = tx * 32;
        dyb_f = (const char*)(a.dy + ((size_t)n * a.H * a.W + (size_t)oh0_f * a.W + ow0_f) * a.lddy + cob * 64);
        // x halo origin (oh0-1, ow0-1) may lie outside the image: keep it as a signed element offset
        const long long xo = ((long long)n * a.H * a.W + (long long)(oh0_f - 1) * a.W + (ow0_f - 1)) * a.ldx + cib * 64;
        xb_f = (const char*)a.x + xo * 2;
    };
    auto issue_piece = [&](int j, int buf) __attribute__((always_inline)) {
        const unsigned lds = smem_base + buf * C::BUF;
        if (j < C::NDY) {
            const int ty_ = (dy_d[j] >> 8) & 255, tx_ = dy_d[j] & 255, s_ = dy_d[j] >> 16;
            const int rel = ((ty_ * a.W + tx_) * a.lddy + s_ * 8) * 2;
            const void* g = (ty_ < a.H - oh0_f && tx_ < a.W - ow0_f) ? (const void*)(dyb_f + rel) : (const void*)&g_zero16;
            dma16(g, __builtin_amdgcn_readfirstlane(lds + (C::NW * j + wave) * 1024));
        } else {
            const int jj = j - C::NDY;
            if (C::NW * jj + wave < C::X_PIECES) {             // wave-uniform
                const int hy = (x_d[jj] >> 8) & 255, hx = x_d[jj] & 255, s_ = (x_d[jj] >> 16) & 15;
                const int ih = oh0_f - 1 + hy, iw = ow0_f - 1 + hx;
                const int rel = ((hy * a.W + hx) * a.ldx + s_ * 8) * 2;
                const bool ok = x_d[jj] >= 0 && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
                const void* g = ok ? (const void*)(xb_f + rel) : (const void*)&g_zero16;
                dma16(g, __builtin_amdgcn_readfirstlane(lds + C::DY_BYTES + (C::NW * jj + wave) * 1024));
            }
        }
    };

    // ---- per-lane fragment bases (everything else is an immediate) ----
    const int g16 = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3, h = g16 >> 1;
    const int r_lane = 16 * par + 8 * h + q;             // tile pixel within the 32-pixel row pair of a K-step
    int a_lane[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int cb = (32 * m + 16 * (g16 & 1) + 4 * pp) * 2;
        a_lane[m] = r_lane * C::DY_ROW + (cb ^ (((q >> 1) & 1) << 6));
    }
    int b_lane[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int cb = (32 * wci + 16 * (g16 & 1) + 4 * pp) * 2;
        b_lane[kw] = C::DY_BYTES + r_lane * 128 + (cb ^ ((((q + kw) >> 1) & 1) << 6));
    }

    f32x16_t acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float bsum = 0.f;
    const bool do_bias = (cib == 0) && (a.bslab != nullptr);

    // the tile loop is instantiated once per tap-half so each copy has a branch-free, fully unrolled body
    auto tile_loop = [&](auto hf_tag) __attribute__((always_inline)) {
        constexpr int HF = decltype(hf_tag)::value;
        if (t_begin < t_end) {
            set_fetch_tile(t_begin);
#pragma unroll
            for (int j = 0; j < C::NDY + C::NX; ++j) issue_piece(j, 0);
        }
        for (int tile = t_begin; tile < t_end; ++tile) {
            const int buf = (tile - t_begin) & 1;
            dma_wait_all();       // this wave's pieces of the tile have landed ...
            __syncthreads();      // ... and everyone else's; every wave is also done reading the other buffer
            const bool more = tile + 1 < t_end;
            if (more) set_fetch_tile(tile + 1);
            if (more && !a.dma_interleave) {
#pragma unroll
                for (int j = 0; j < C::NDY + C::NX; ++j) issue_piece(j, buf ^ 1);
            }
            const char* lds = smem + buf * C::BUF;
            if (do_bias) {
                const int co = tid & 63, part = tid >> 6;
                for (int r = part; r < C::P; r += 8)
                    bsum += bf16_to_f32(*(const bf16_t*)(lds + r * C::DY_ROW + ((co * 2) ^ (((r >> 1) & 1) << 6))));
            }
            compute_tile<HF>(lds, a_lane, b_lane, acc, [&](int kk) __attribute__((always_inline)) {
                // 10 pieces over the first three K-steps (4 + 3 + 3)
                if (more && a.dma_interleave && kk < 3) {
                    const int j0 = kk == 0 ? 0 : (kk == 1 ? 4 : 7), j1 = kk == 0 ? 4 : (kk == 1 ? 7 : 10);
#pragma unroll
                    for (int j = j0; j < j1; ++j) issue_piece(j, buf ^ 1);
                }
            });
        }
    };
    if (hf == 0) tile_loop(std::integral_constant<int, 0>{});
    else tile_loop(std::integral_constant<int, 1>{});
    __syncthreads();   // all fragment reads done: LDS is free for the reductions below

    // ---- add the odd-K-step wave set into the even one through LDS ----
    {
        float* red = (float*)smem + (size_t)(wave & 3) * 144 * 64;
        if (par == 1) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) red[(t * 16 + i) * 64 + lane] = acc[t][i];
        }
        __syncthreads();
        if (par == 0) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] += red[(t * 16 + i) * 64 + lane];
        }
        __syncthreads();
    }

    // ---- partial block -> slab[tap][co][split][ci] (the reducer reads one (tap, co) row's partials as a contiguous run) ----
    if (par == 0) {
        const int l31 = lane & 31, lh = lane >> 5;
        float* sl = a.slab + (size_t)split * a.Cin;
        const size_t rs = (size_t)a.splits * a.Cin;
        const int ci = cib * 64 + 32 * wci + l31;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int tap = t == 8 ? 4 : (hf ? 5 + (t >> 1) : (t >> 1));
            const int m = t == 8 ? hf : (t & 1);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = cob * 64 + 32 * m + (i & 3) + 8 * (i >> 2) + 4 * lh;
                sl[((size_t)tap * a.Cout + co) * rs + ci] = acc[t][i];
            }
        }
    }
    if (do_bias) {
        float* red = (float*)smem;
        red[tid] = bsum;
        __syncthreads();
        if (tid < 64) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += red[tid + 64 * k];
            a.bslab[(size_t)split * a.Cout + cob * 64 + tid] = s;
        }
    }
}

}  // namespace

bool wgrad_v2_eligible(int H, int W, int Cin, int Cout, int stride, int dtype, bool gated) {
    return dtype == WU_BF16 && stride == 1 && !gated && W > 16 && Cin % 64 == 0 && Cout % 64 == 0;
}

WgradV2Plan wgrad_v2_plan(int N, int H, int W, int Cin, int Cout) {
    WgradV2Plan p;
    p.mode = 0;
    p.tiles_x = cdiv(W, 32); p.tiles_y = cdiv(H, C::TH);
    p.ntiles = N * p.tiles_x * p.tiles_y;
    p.co_blocks = Cout / 64; p.ci_blocks = Cin / 64;
    const int blocks = p.co_blocks * p.ci_blocks;
    int splits = 256 / blocks;                      // one 512-thread workgroup per CU, one wave of workgroups
    if (splits < 1) splits = 1;
    if (splits > p.ntiles) splits = p.ntiles;
    p.tiles_per_split = cdiv(p.ntiles, splits);
    p.splits = cdiv(p.ntiles, p.tiles_per_split);
    p.ws = ((size_t)p.splits * 9 * Cout * Cin + (size_t)p.splits * Cout) * sizeof(float);
    return p;
}

int wgrad_v2_launch(const void* x, int ldx, const void* dy, int lddy, float* slab, float* bslab,
                    int N, int H, int W, int Cin, int Cout, const WgradV2Plan& p, hipStream_t s) {
    W2Args a;
    a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.slab = slab; a.bslab = bslab;
    a.ldx = ldx; a.lddy = lddy; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.ntiles = p.ntiles; a.tiles_per_split = p.tiles_per_split;
    a.splits = p.splits; a.co_blocks = p.co_blocks; a.ci_blocks = p.ci_blocks;
    a.dma_interleave = g_wu_opt[WU_OPT_WGRAD_DMA_INTERLEAVE];
    const int grid = p.splits * p.co_blocks * p.ci_blocks;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_v2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(conv3x3_wgrad_v2_kernel, dim3(grid), dim3(512), 2 * C::BUF, s, a);
    return 0;
}
