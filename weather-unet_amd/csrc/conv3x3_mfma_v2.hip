// conv3x3 (pad 1, stride 1) implicit GEMM, production path: bf16, output width > 16, ungated input.
// Forward convs of nets.py:18-24 and, on the rotated weight pack, their data-gradient pass.
//
// Same algorithm as conv3x3_mfma.hip (per 64-byte channel chunk: halo tile + [9][64][chunk] weight slab in LDS,
// tap shift = LDS address offset, no im2col), re-built the way the weight-gradient kernel was:
//   * 8 waves (2 per SIMD) share one 16x32-pixel x 64-Cout tile: the weight slab is staged once per 512 pixels;
//   * staging is LDS-DMA (global_load_lds_dwordx4) into TWO buffers: chunk k+1 streams in under the MFMAs of
//     chunk k, one barrier per chunk, no staging registers, no ds_write pass.  Out-of-image halo pixels come
//     from a 16-byte zero constant; the XOR bank swizzle is applied to the per-lane SOURCE address (the LDS image
//     of a DMA is lane-linear);
//   * the DMA issue of the next chunk is spread over the nine taps of the current one (its address VALU work
//     hides behind the MFMAs instead of sitting between the barrier and the first fragment read);
//   * tile width fixed at 32: fragment addresses are lane_base[kw][ks] + immediate, no VALU in the tap loop.
#include "wu_common.h"
#include "conv_internal.h"

namespace {

__device__ const uint4 g_zero16v2 = {0u, 0u, 0u, 0u};

struct K {
    static constexpr int TH = 16, TW = 32, P = TH * TW;          // 512 output pixels
    static constexpr int HALO_W = TW + 2, HALO_H = TH + 2, HALO_PIX = HALO_W * HALO_H;   // 34 x 18 = 612
    static constexpr int H_PIECES = (HALO_PIX * 64 + 1023) / 1024;   // 39
    static constexpr int H_BYTES = H_PIECES * 1024;
    static constexpr int W_PIECES = 9 * 64 * 64 / 1024;              // 36
    static constexpr int W_BYTES = W_PIECES * 1024;
    static constexpr int BUF = H_BYTES + W_BYTES;                    // 76800 B, two buffers = 150 KiB
    static constexpr int NW = 8;
    static constexpr int NH = (H_PIECES + NW - 1) / NW;              // 5 halo pieces per wave
    static constexpr int NWT = (W_PIECES + NW - 1) / NW;             // 5 weight pieces per wave
};

struct V2Args {
    const bf16_t* x; const bf16_t* w; const float* bias; bf16_t* y; const bf16_t* egate;
    int ldx, ldy, ldegate, egate_act;
    int N, H, W, Cin, Cout, act;
    int tiles_x, tiles_y, cout_tiles;
};

// LDS-DMA issued from inline asm: hipcc cannot see it, so it neither drains it (vmcnt(0)) before the next
// ds_read of the OTHER buffer nor counts it -- the kernel waits for it itself (dma_wait_all) before the barrier
// that publishes the buffer (cdna guide 5.7: M0 = wave-uniform LDS byte address, restored in the same statement).
__device__ __forceinline__ void dma16(const void* g, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void mma(f32x16_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}

__global__ __launch_bounds__(512, 2) void conv3x3_mfma_v2_kernel(const V2Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;

    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int ct = bid % a.cout_tiles; bid /= a.cout_tiles;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int n = bid / a.tiles_y;
    const int oh0 = ty * K::TH, ow0 = tx * K::TW, co0 = ct * 64;

    // ---- per-lane DMA sources (fixed for the whole kernel; only the channel chunk offset moves) ----
    // halo: LDS slot i = piece*64 + lane -> pixel p = i >> 2, LDS 16-B slot sl = i & 3 holds channel slot sl ^ swz(hx)
    const bf16_t* xin = a.x + (size_t)n * a.H * a.W * a.ldx;
    int hsrc[K::NH];                     // element offset inside image n, -1 = zero fill
#pragma unroll
    for (int j = 0; j < K::NH; ++j) {
        const int i = (K::NW * j + wave) * 64 + lane;
        const int p = i >> 2, sl = i & 3;
        const int hy = p / K::HALO_W, hx = p - hy * K::HALO_W;
        const int ih = oh0 - 1 + hy, iw = ow0 - 1 + hx;
        const int s = sl ^ ((hx >> 2) & 3);
        hsrc[j] = (p < K::HALO_PIX && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) ? (ih * a.W + iw) * a.ldx + s * 8 : -1;
    }
    // weights: LDS slot i -> row = i >> 2 = tap*64 + co, slot sl holds channel slot sl ^ swz(co)
    int wsrc[K::NWT];
#pragma unroll
    for (int j = 0; j < K::NWT; ++j) {
        const int i = (K::NW * j + wave) * 64 + lane;
        const int row = i >> 2, sl = i & 3;
        const int tap = row >> 6, co = row & 63;
        const int s = sl ^ ((co >> 2) & 3);
        wsrc[j] = (tap * a.Cout + co0 + co) * a.Cin + s * 8;     // < 9*512*768 elements: fits int32
    }

    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    auto issue_piece = [&](int j, int c0, int buf) __attribute__((always_inline)) {
        const unsigned lds = smem_base + buf * K::BUF;
        // j in [0, NH): halo piece; j in [NH, NH + NWT): weight piece (wave-uniform guards)
        if (j < K::NH) {
            if (K::NW * j + wave < K::H_PIECES) {
                const void* g = hsrc[j] >= 0 ? (const void*)(xin + hsrc[j] + c0) : (const void*)&g_zero16v2;
                dma16(g, __builtin_amdgcn_readfirstlane(lds + (K::NW * j + wave) * 1024));
            }
        } else {
            const int jj = j - K::NH;
            if (K::NW * jj + wave < K::W_PIECES)
                dma16((const void*)(a.w + wsrc[jj] + c0), __builtin_amdgcn_readfirstlane(lds + K::H_BYTES + (K::NW * jj + wave) * 1024));
        }
    };

    // ---- per-lane fragment bases ----
    // A: wave owns tile rows 2*wave + mi; lane row l31 = tx; 16-B slot = 2*ks + lh, swizzled with ((tx + kw) >> 2) & 3
    int a_lane[3][2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            a_lane[kw][ks] = ((2 * wave) * K::HALO_W + l31) * 64 + (((2 * ks + lh) ^ (((l31 + kw) >> 2) & 3)) << 4);
    // B: cout row 32*ni + l31 (swizzle depends on l31 only), slot 2*ks + lh
    int b_lane[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) b_lane[ks] = K::H_BYTES + l31 * 64 + (((2 * ks + lh) ^ ((l31 >> 2) & 3)) << 4);

    f32x16_t acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

    const int nchunks = a.Cin / 32;
#pragma unroll
    for (int j = 0; j < K::NH + K::NWT; ++j) issue_piece(j, 0, 0);
    for (int c = 0; c < nchunks; ++c) {
        const char* lds = smem + (c & 1) * K::BUF;
        const int nxt = (c + 1) & 1;
        const bool more = c + 1 < nchunks;
        const int c1 = (c + 1) * 32;
        dma_wait_all();      // this wave's pieces of chunk c have landed ...
        __syncthreads();     // ... and so have everyone else's; everyone is also done with the other buffer
        // 18 steps (tap, ks), software-pipelined by hand: the fragments of step s+1 are requested BEFORE the four
        // MFMAs of step s are issued, so one LDS round trip is always covered by matrix work of this wave.
        auto load_step = [&](int step, uint4 (&af)[2], uint4 (&bf)[2]) __attribute__((always_inline)) {
            const int tap = step >> 1, ks = step & 1, kh = tap / 3, kw = tap % 3;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                af[mi] = *(const uint4*)(lds + a_lane[kw][ks] + ((mi + kh) * K::HALO_W + kw) * 64);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                bf[ni] = *(const uint4*)(lds + b_lane[ks] + (tap * 64 + 32 * ni) * 64);
        };
        uint4 af[2][2], bf[2][2];
        load_step(0, af[0], bf[0]);
#pragma unroll
        for (int step = 0; step < 18; ++step) {
            const int cur = step & 1;
            if (step + 1 < 18) load_step(step + 1, af[cur ^ 1], bf[cur ^ 1]);
            // spread the next chunk's DMA issue over the steps (10 pieces over 18 steps)
            if (more && (step & 1) == 0) {
                issue_piece(step >> 1, c1, nxt);
                if (step == 16) issue_piece(9, c1, nxt);
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) mma(acc[mi][ni], af[cur][mi], bf[cur][ni]);
        }
    }
    __syncthreads();

    // ---- epilogue: bias + activation in fp32, transpose through LDS, 16-B coalesced stores ----
    constexpr int kRow = 64 * 2 + 16;
    float bv[2] = {0.f, 0.f};
    if (a.bias) {
        bv[0] = a.bias[co0 + l31];
        bv[1] = a.bias[co0 + 32 + l31];
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (2 * wave + mi) * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
                const float v = act_apply(acc[mi][ni][i] + bv[ni], a.act);
                *((bf16_t*)(smem + row * kRow) + 32 * ni + l31) = f32_to_bf16(v);
            }
    __syncthreads();
    bf16_t* yout = a.y + (size_t)n * a.H * a.W * a.ldy + co0;
#pragma unroll
    for (int k = 0; k < K::P * 8 / 512; ++k) {
        const int q = tid + 512 * k;
        const int r = q >> 3, s = q & 7;
        const int oh = oh0 + (r >> 5), ow = ow0 + (r & 31);
        if (oh < a.H && ow < a.W) {
            uint4 v = *(const uint4*)(smem + r * kRow + s * 16);
            if (a.egate)
                v = gate16<bf16_t>(v, *(const uint4*)(a.egate + ((size_t)n * a.H * a.W + (size_t)(oh * a.W + ow)) * a.ldegate + co0 + s * 8), a.egate_act);
            *(uint4*)(yout + (size_t)(oh * a.W + ow) * a.ldy + s * 8) = v;
        }
    }
}

}  // namespace

bool conv_v2_eligible(int H, int W, int Cin, int Cout, int stride, int dtype, bool masked) {
    return dtype == WU_BF16 && stride == 1 && !masked && W > 16 && Cin % 32 == 0 && Cout % 64 == 0 &&
           (size_t)9 * Cout * Cin < (1ull << 31);
}

int conv_v2_launch(const void* x, int ldx, const void* w, const float* bias, void* y, int ldy,
                   const void* egate, int ldegate, int egate_act, int N, int H, int W, int Cin, int Cout, int act, hipStream_t s) {
    V2Args a;
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.bias = bias; a.y = (bf16_t*)y; a.egate = (const bf16_t*)egate;
    a.ldx = ldx; a.ldy = ldy; a.ldegate = ldegate; a.egate_act = egate_act; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.act = act;
    a.tiles_x = cdiv(W, K::TW); a.tiles_y = cdiv(H, K::TH); a.cout_tiles = Cout / 64;
    const long long grid = (long long)N * a.tiles_x * a.tiles_y * a.cout_tiles;
    if (grid >= (1ll << 31)) return -1;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv3x3_mfma_v2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(conv3x3_mfma_v2_kernel, dim3((int)grid), dim3(512), 2 * K::BUF, s, a);
    return 0;
}
