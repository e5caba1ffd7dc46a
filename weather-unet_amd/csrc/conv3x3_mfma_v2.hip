// conv3x3 (pad 1, stride 1) implicit GEMM, production path: bf16, output width > 16, ungated input.
// Forward convs of nets.py:18-24 and, on the rotated weight pack, their data-gradient pass.
//
// Same algorithm as conv3x3_mfma.hip (per 64-byte channel chunk: halo tile + [9][64][chunk] weight slab in LDS,
// tap shift = LDS address offset, no im2col), re-built the way the weight-gradient kernel was:
//   * 8 waves (2 per SIMD) share one 16x32-pixel x 64-Cout tile: the weight slab is staged once per 512 pixels;
//   * staging is LDS-DMA (buffer_load_dwordx4 ... lds) into TWO buffers: chunk k+1 streams in under the MFMAs of
//     chunk k, one barrier per chunk, no staging registers, no ds_write pass.  A piece costs two instructions: the
//     per-lane byte offsets are tile-invariant registers, tile origin and chunk travel in the scalar offset, and
//     out-of-image halo pixels are lanes pushed out of the descriptor's range (they write zeros); the XOR bank
//     swizzle is applied to the per-lane SOURCE offset (the LDS image of a DMA is lane-linear);
//   * the DMA issue of the next chunk is spread over the nine taps of the current one (its address VALU work
//     hides behind the MFMAs instead of sitting between the barrier and the first fragment read);
//   * tile width fixed at 32: fragment addresses are lane_base[kw][ks] + immediate, no VALU in the tap loop.
#include <type_traits>

#include "wu_common.h"
#include "conv_internal.h"

#ifndef WU_CONV_STORE_AUX
#define WU_CONV_STORE_AUX 0
#endif

namespace {

struct K {
    static constexpr int TH = 16, TW = 32, P = TH * TW;          // 512 output pixels
    static constexpr int HALO_W = TW + 2, HALO_H = TH + 2, HALO_PIX = HALO_W * HALO_H;   // 34 x 18 = 612
    // 1-KiB DMA pieces: 39 carry halo pixels, 36 carry weights; both regions are padded to 40 so every wave issues the same
    // number of pieces with no guard (the pad pieces are out of range for every lane: zeros into unused LDS)
    static constexpr int H_PIECES = 40, H_BYTES = H_PIECES * 1024;
    static constexpr int W_REAL = 9 * 64 * 64 / 1024;                // 36
    static constexpr int W_PIECES = 40, W_BYTES = W_PIECES * 1024;
    static constexpr int BUF = H_BYTES + W_BYTES;                    // 80 KiB, two buffers = all 160 KiB of the CU's LDS
};
// NW waves share the tile: 8 (two per SIMD, 2 tile rows = 4 accumulators each) or 4 (one per SIMD, 4 rows = 8 accumulators)
template <int NW_> struct KW {
    static constexpr int NW = NW_;
    static constexpr int RPW = K::TH / NW;                               // tile rows per wave
    static constexpr int NH = K::H_PIECES / NW;                          // halo pieces per wave
    static constexpr int NWT = K::W_PIECES / NW;                         // weight pieces per wave
    static constexpr int NP = NH + NWT;
    static constexpr int NST = K::P * 8 / (NW * 64);                     // 16-B output stores per thread and tile
};

struct V2Args {
    const bf16_t* x; const bf16_t* w; const float* bias; bf16_t* y; const bf16_t* egate;
    bf16_t* pool;                // optional: 2x2 max-pool of y (ReLU outputs), written from the same epilogue
    // ReLU gate as BITS (include/wu_kernels.h, "gate bits"): uint32 [pixel][Cout/64][2]; bit 8k + i of word (pixel, ct, hf) is
    // y[pixel][64 ct + 16 k + 8 hf + i] > 0 -- exactly the 32 channels lane (pixel, hf) holds in the epilogue, so a gate is one
    // dword per lane and row instead of four 16-byte loads
    unsigned* gbits;             // optional output (forward, act == RELU)
    unsigned* sbits;             // optional output (forward + pool, GATED == 4): bit set = this element is its 2x2 window's FIRST maximum
    const unsigned* egbits;      // optional input (data gradient): replaces egate / egate_act = RELU
    int ldx, ldy, ldegate, egate_act, ldpool;
    int N, H, W, Cin, Cout, act;
    int tiles_x, tiles_y, cout_tiles, ntiles, prio_mode, strided;
    int w_resident;              // Cin == 64 and ONE cout tile: both weight chunks stay in LDS for the life of the workgroup
    // GATED == 5 (round 4, cunet.py:78-82): the network's 1x1 head + tanh from the epilogue of the last decoder conv.  head_w = conv_last.weight
    // [3][64] fp32, head_b [3], head_out the NCHW fp32 image; y may then be NULL (a forward nobody differentiates: the 64-channel tensor is never written)
    const float* head_w; const float* head_b; float* head_out;
    unsigned long long* dbg;     // diagnostic: per-workgroup phase cycle sums (NULL in production)
};

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef short s16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
// ReLU of a packed bf16 pair: as int16 a negative float is negative, a positive one positive -> v_pk_max_i16 with 0
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t d) {
    const s16x2_t v = __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, d), s16x2_t{0, 0});
    return __builtin_bit_cast(uint32_t, v);
}
// g * ReLU'(y) on packed bf16 pairs: keep a half iff y != 0 and y's sign bit is clear (y > 0)
__device__ __forceinline__ uint32_t relu_gate_bf16x2(uint32_t g, uint32_t y) {
    const u16x2_t nz = __builtin_elementwise_min(__builtin_bit_cast(u16x2_t, y), u16x2_t{1, 1});
    const u16x2_t m = u16x2_t{0, 0} - nz;                                              // 0xffff where y != 0
    const s16x2_t sg = __builtin_bit_cast(s16x2_t, y) >> s16x2_t{15, 15};               // 0xffff where y < 0
    return g & __builtin_bit_cast(uint32_t, m) & ~__builtin_bit_cast(uint32_t, sg);
}
// max of two packed pairs of NON-NEGATIVE bf16 (they order like unsigned 16-bit integers): v_pk_max_u16
__device__ __forceinline__ uint32_t max_u16x2(uint32_t a, uint32_t b) {
    const u16x2_t m = __builtin_elementwise_max(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, b));
    return __builtin_bit_cast(uint32_t, m);
}
// bit e = (element e of eight NON-NEGATIVE bf16 != 0).  v_pk_min_u16 with 1 turns a pair into 0/1 per half (inline asm: the compiler
// rewrites the equivalent C into a compare + select per element, ~30 instructions per group); three v_lshl_or gather the four
// dwords (low halves at bits 0,2,4,6, high halves at 16,18,20,22) and one shift-or folds the high halves onto bits 1,3,5,7.
__device__ __forceinline__ uint32_t nonzero_byte(const uint4& v) {
    uint32_t t0, t1, t2, t3;
    const uint32_t one = 0x00010001u;
    asm("v_pk_min_u16 %0, %1, %2" : "=&v"(t0) : "v"(v.x), "v"(one));
    asm("v_pk_min_u16 %0, %1, %2" : "=&v"(t1) : "v"(v.y), "v"(one));
    asm("v_pk_min_u16 %0, %1, %2" : "=&v"(t2) : "v"(v.z), "v"(one));
    asm("v_pk_min_u16 %0, %1, %2" : "=&v"(t3) : "v"(v.w), "v"(one));
    const uint32_t xx = ((t3 << 6) | t2 << 4) | ((t1 << 2) | t0);
    return (xx | (xx >> 15)) & 0xffu;
}
// AND mask for a packed bf16 pair from two adjacent gate bits: bit b -> low half, bit b + 1 -> high half (v_bfe_i32 x 2 + merge)
__device__ __forceinline__ uint32_t gate_mask2(int bits, int b) {
    const uint32_t lo = (uint32_t)((bits << (31 - b)) >> 31), hi = (uint32_t)((bits << (30 - b)) >> 31);
    return (lo & 0xffffu) | (hi << 16);
}
// tanh(s) = 1 - 2 / (exp(2 s) + 1) on the hardware exp2 / rcp units (the formula of thin.hip's stand-alone head kernel)
__device__ __forceinline__ float fast_tanh(float s) {
    const float e = __builtin_amdgcn_exp2f(s * 2.885390081777927f);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void mma(f32x16_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
// 16 couts x 16 pixels x 32 channels (a whole K chunk of one tap): the shape of the one-wave-per-SIMD instances (round 3)
__device__ __forceinline__ void mma16(f32x4_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
// accumulator storage of one (tile row, 32-pixel half... ) -- see M16 in the kernel
template <bool M16_, int RPW_> struct AccT { using type = f32x16_t[RPW_][2]; };
template <int RPW_> struct AccT<true, RPW_> { using type = f32x4_t[RPW_][2][4]; };

// GATED: 0 = forward; 1 / 2 = the data-gradient form (a gate in the epilogue, no bias) with the gate as a tensor / as bits;
// 3 = forward (ReLU) that also writes the gate bits of its output; 4 = forward (ReLU) + 2x2 max-pool that also writes, per element of
// its output, the gate bit AND the pool's arg-max bit (round 4: the max-pool backward then reads 2 bits instead of the tensor) -- separate instances so that
// the gate-tensor prefetch registers (32 / 64), the bias registers (32) and neither of them are allocated as each case needs;
// 5 = forward (ReLU, Cout == 64, 8 waves) + the 64 -> 3 pointwise head + tanh (cunet.py:80-82) on the matrix cores: the packed bf16 registers of the
// epilogue ARE the B operand of a 32x32x16 MFMA whose K block is their 16 channels (lane (pixel, half) holds channels 8 half .. 8 half + 7), the head's
// weights are four A fragments built once per persistent workgroup -- rows 0..2 the bf16 high parts, rows 8..10 and 16..18 the first and second bf16 residuals of the fp32 weights, so
// the products carry the weights exactly -- 8 extra MFMAs per wave and tile (144 in the K loop), three fp32 values per pixel leave as deferred dword stores
template <int NW, int GATED>
__global__ __launch_bounds__(NW * 64) void conv3x3_mfma_v2_kernel(const V2Args a) {
    using Q = KW<NW>;
    // M16 (the 4-wave, one-wave-per-SIMD instances): v_mfma_f32_16x16x32_bf16 instead of 32x32x16.  Same flops, same LDS fragment
    // traffic (12 16-byte reads per tap and wave); measured on the identical data flow, the 16x16x32 stream runs the Cin >= 256
    // layers 5.5-6.5 % faster (the chip holds a higher clock on it: MI355X_MICROARCH.md, DVFS item 7), while the two-waves-per-SIMD
    // instances gain nothing.  A lane then holds 4 consecutive couts of ONE pixel per accumulator (not 32 channels of a pixel
    // half-row), so the tile's results are brought into the 32x32 epilogue's register layout through the LDS buffer that has just
    // been computed from (bias + activation + bf16 packing happen before that hop; gates, pool, gate bits and the deferred
    // stores after it are shared with the 8-wave instances).
    constexpr bool M16 = NW == 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;

    // ---- persistent workgroup: a contiguous range of (n, ty, tx, cout-tile) items, cout-tile fastest ----
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    // contiguous: this workgroup walks tiles [t_begin, t_end) (its own consecutive cout tiles re-read one halo);
    // strided:    tiles wg, wg + grid, ...: at any moment the 32 workgroups of an XCD hold 32 CONSECUTIVE tile ids = a few pixel
    //             tiles x all their cout tiles, so the halo re-reads of the cout tiles are concurrent hits in that XCD's L2
    const int t_step = a.strided ? (int)gridDim.x : 1;
    const int t_begin = a.strided ? wg : (int)((long long)a.ntiles * wg / gridDim.x);
    const int t_end = a.strided ? a.ntiles : (int)((long long)a.ntiles * (wg + 1) / gridDim.x);
    if (t_begin >= t_end) return;
    // Tile coordinates (cout tile fastest, then tx, ty, n) are carried INCREMENTALLY: one decode by division per workgroup,
    // then the constant stride t_step is added digit by digit.  (Four div/mod pairs per tile were ~200 VALU instructions that
    // a one-wave-per-SIMD workgroup cannot hide: +2400 cycles on the last chunk of every tile.)
    struct Tc { int ct, tx, ty, n; };
    auto decode = [&](int tile) __attribute__((always_inline)) {
        Tc t;
        int tt = tile;
        t.ct = tt % a.cout_tiles; tt /= a.cout_tiles;
        t.tx = tt % a.tiles_x; tt /= a.tiles_x;
        t.ty = tt % a.tiles_y;
        t.n = tt / a.tiles_y;
        return t;
    };
    const Tc dstep = decode(t_step);              // t_step in the same mixed radix
    auto advance = [&](Tc t) __attribute__((always_inline)) {
        t.ct += dstep.ct; int cy = t.ct >= a.cout_tiles ? 1 : 0; t.ct -= cy * a.cout_tiles;
        t.tx += dstep.tx + cy; cy = t.tx >= a.tiles_x ? 1 : 0; t.tx -= cy * a.tiles_x;
        t.ty += dstep.ty + cy; cy = t.ty >= a.tiles_y ? 1 : 0; t.ty -= cy * a.tiles_y;
        t.n += dstep.n + cy;
        return t;
    };

    // ---- tile-invariant per-lane DMA byte offsets (wu_common.h, wu_dma16b): LDS slot i = piece*64 + lane ----
    // halo: pixel p = i >> 2 = (hy, hx) relative to the halo origin (oh0-1, ow0-1), LDS 16-B slot sl = i & 3 holds channel
    // slot sl ^ swz(hx).  The descriptor base is shifted back by one row + one pixel so the offsets are non-negative.
    unsigned hoff[Q::NH];
#pragma unroll
    for (int j = 0; j < Q::NH; ++j) {
        const int i = (NW * j + wave) * 64 + lane;
        const int p = i >> 2, sl = i & 3;
        const int hy = p / K::HALO_W, hx = p - hy * K::HALO_W;
        // bank swizzle of the 16-byte channel slot, matched to the fragment reads: 32x32x16 reads 32 pixels x 2 slots per
        // instruction, 16x16x32 reads 16 pixels x 4 slots (conflict-free iff {s(x), s(x+4)^1, s(x+8)^1, s(x+12)} are distinct)
        const int swz = M16 ? 2 * ((hx >> 2) & 1) : ((hx >> 2) & 3);
        hoff[j] = p < K::HALO_PIX ? (unsigned)(((hy * a.W + hx) * a.ldx + (sl ^ swz) * 8) * 2) : kWuOOB;
    }
    // weights: row = i >> 2 = tap*64 + co, slot sl holds channel slot sl ^ swz(co): byte offset relative to the cout tile
    unsigned woff[Q::NWT];
#pragma unroll
    for (int j = 0; j < Q::NWT; ++j) {
        const int i = (NW * j + wave) * 64 + lane;
        const int row = i >> 2, sl = i & 3;
        const int tap = row >> 6, co = row & 63;
        const int swz = M16 ? 2 * ((co >> 2) & 1) : ((co >> 2) & 3);
        woff[j] = NW * j + wave < K::W_REAL ? (unsigned)(((tap * a.Cout + co) * a.Cin + (sl ^ swz) * 8) * 2) : kWuOOB;
    }

    // descriptors of the tile whose chunks are currently being FETCHED (one tile ahead at tile boundaries): scalar state
    // (two buffer descriptors, the tile's byte offset) + the halo offsets with this tile's border lanes pushed out of range
    // (top row / left / right columns; rows past the bottom fall off the end of the per-image descriptor by themselves)
    const unsigned x_img_bytes = (unsigned)((((size_t)a.H * a.W + a.W) * a.ldx + a.Cin) * 2);
    unsigned hv[Q::NH];
    wu_rsrc_t rs_x = wu_make_rsrc(a.x, 0), rs_w = rs_x;
    unsigned so_tile = 0;
    auto set_fetch_tile = [&](const Tc& t) __attribute__((always_inline)) {
        const int ct = t.ct, n = t.n;
        const int oh0 = t.ty * K::TH, ow0 = t.tx * K::TW;
        // may point before the tensor for n = 0: never dereferenced (the lanes that would are out of range below)
        rs_x = wu_make_rsrc(a.x + ((long long)n * a.H * a.W - (a.W + 1)) * a.ldx, x_img_bytes);
        rs_w = wu_make_rsrc(a.w + (size_t)ct * 64 * a.Cin, (unsigned)(((size_t)9 * a.Cout - (size_t)ct * 64) * a.Cin * 2));
        so_tile = (unsigned)__builtin_amdgcn_readfirstlane((oh0 * a.W + ow0) * a.ldx * 2);
        const int ymin = __builtin_amdgcn_readfirstlane(oh0 == 0 ? 1 : 0), xmin = __builtin_amdgcn_readfirstlane(ow0 == 0 ? 1 : 0);
        const int xmax = __builtin_amdgcn_readfirstlane(a.W - ow0);
#pragma unroll
        for (int j = 0; j < Q::NH; ++j) {
            // (hy, hx) recomputed per tile from the lane's slot index rather than kept in a register per piece
            const int p_ = ((NW * j + wave) * 64 + lane) >> 2, hy = p_ / K::HALO_W, hx = p_ - hy * K::HALO_W;
            hv[j] = (hy >= ymin && hx >= xmin && hx <= xmax) ? hoff[j] : kWuOOB;
        }
    };

    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // one piece = s_mov m0 + buffer_load ... lds; c0 (the chunk's first channel) travels in the scalar offset
    // (kill = kWuOOB turns the piece into an all-lanes-out-of-range no-op that still lands zeros: branch-free "no more chunks")
    auto issue_piece = [&](int j, int c0, int buf, unsigned kill = 0u) __attribute__((always_inline)) {
        const unsigned lds = smem_base + buf * K::BUF;
        if (j < Q::NH) {
            wu_dma16b(hv[j], rs_x, (so_tile + (unsigned)c0 * 2) | kill, __builtin_amdgcn_readfirstlane(lds + (NW * j + wave) * 1024));
        } else {
            const int jj = j - Q::NH;
            wu_dma16b(woff[jj], rs_w, ((unsigned)c0 * 2) | kill, __builtin_amdgcn_readfirstlane(lds + K::H_BYTES + (NW * jj + wave) * 1024));
        }
    };

    // ---- per-lane fragment bases ----
    // A: wave owns tile rows RPW*wave + mi; lane row l31 = tx; 16-B slot = 2*ks + lh, swizzled with ((tx + kw) >> 2) & 3
    int a_lane[3][2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            a_lane[kw][ks] = ((Q::RPW * wave) * K::HALO_W + l31) * 64 + (((2 * ks + lh) ^ (((l31 + kw) >> 2) & 3)) << 4);
    // B: cout row 32*ni + l31 (swizzle depends on l31 only), slot 2*ks + lh
    int b_lane[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) b_lane[ks] = K::H_BYTES + l31 * 64 + (((2 * ks + lh) ^ ((l31 >> 2) & 3)) << 4);

    // M16 fragments: lane (l15 = lane & 15, q = lane >> 4) reads pixel x0 + l15 (or cout 16 cb + l15), 16-byte K slot q
    const int l15 = lane & 15, q16 = lane >> 4;
    int a16_lane[3], b16_lane = 0;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
        a16_lane[kw] = ((Q::RPW * wave) * K::HALO_W + l15 + kw) * 64 + ((q16 ^ (2 * (((l15 + kw) >> 2) & 1))) << 4);
    b16_lane = K::H_BYTES + l15 * 64 + ((q16 ^ (2 * ((l15 >> 2) & 1))) << 4);

    // head A fragments (GATED == 5): K block kb = channels 16 kb .. + 15, lane (m = l31, half lh) holds row m's channels 16 kb + 8 lh + i
    uint4 hA[4];
    float hbias[3] = {0.f, 0.f, 0.f};
    if constexpr (GATED == 5) {
        // rows 0..2: bf16(w); rows 8..10: bf16(w - hi); rows 16..18: bf16(w - hi - lo) -- three terms carry an fp32 weight exactly (24 mantissa bits)
        const int part = l31 >> 3, hrow = ((l31 & 7) < 3 && part < 3) ? (l31 & 7) : -1;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            uint32_t d[4] = {0u, 0u, 0u, 0u};
            if (hrow >= 0) {
                const float4 w0 = *(const float4*)(a.head_w + hrow * 64 + 16 * kb + 8 * lh);
                const float4 w1 = *(const float4*)(a.head_w + hrow * 64 + 16 * kb + 8 * lh + 4);
                const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t hi = pack_bf16x2(wv[2 * i], wv[2 * i + 1]);
                    const float r0 = wv[2 * i] - __builtin_bit_cast(float, hi << 16), r1 = wv[2 * i + 1] - __builtin_bit_cast(float, hi & 0xffff0000u);
                    const uint32_t lo = pack_bf16x2(r0, r1);
                    const float q0 = r0 - __builtin_bit_cast(float, lo << 16), q1 = r1 - __builtin_bit_cast(float, lo & 0xffff0000u);
                    d[i] = part == 0 ? hi : (part == 1 ? lo : pack_bf16x2(q0, q1));
                }
            }
            hA[kb] = make_uint4(d[0], d[1], d[2], d[3]);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) hbias[c] = a.head_b[c];
    }
    unsigned long long t_wait = 0, t_comp = 0, t_comp_rest = 0, t_epi_b1 = 0, t_epi_b2 = 0, t_epi_s = 0, t_mark = 0;
#define WU_STAMP(acc_var) do { if (a.dbg) { const unsigned long long t_ = __builtin_readcyclecounter(); acc_var += t_ - t_mark; t_mark = t_; } } while (0)
    unsigned long long t_k0 = 0, t_r0 = 0;
    if (a.dbg) { t_mark = __builtin_readcyclecounter(); t_k0 = t_mark; t_r0 = __builtin_amdgcn_s_memrealtime(); }
    // experiment (A/B switch): static priority for the later-dispatched half of the workgroup (cdna guide T5 static form)
    if (NW == 8 && a.prio_mode == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    const int nchunks = a.Cin / 32;
    int buf = 0;                                  // LDS buffer holding the chunk being computed
    bool stores_in_flight = false;                // true after a chunk 0 in which all NST deferred stores were issued
    bool bits_in_flight = false;                  // ... followed by the RPW gate-word stores
    // Deferred output stores.  A tile's epilogue only PACKS its results (bias, activation, bf16, gate: registers `ov`); the NST
    // 16-byte stores are issued one per K-step from inside the NEXT tile's first chunk, behind that chunk's DMA pieces, so the
    // store traffic runs under matrix work instead of in front of it (stores were ~4000 of a Cin = 64 tile's ~19000 cycles;
    // with the stores removed the forward convs ran 9 %, the data-gradient convs 16 % faster).  `ov` takes the place the bias /
    // gate prefetch registers have in the last chunk, so the register peak does not move.
    uint4 ov[Q::NST];
    const bf16_t* ov_img = a.y;                   // (uniform) image of the pending tile in y
    unsigned ov_off = 0;                          // this lane's byte offset of store 0 inside that image (< 2^31: host check)
    unsigned ov_ok = 0;                           // bit k: store k of this lane is inside the image
    bool ov_pending = false, ov_interior = false;
    unsigned ovb[Q::RPW];                         // the pending tile's gate words (forward with gate bits), stored after `ov`
    unsigned ovb_off = 0;                         // this lane's dword index of row 0's gate word
    bool ovb_pending = false;
    // GATED == 5: the tile's head outputs (tanh applied), [row][channel], lanes of half 0 = pixel l31; stored after `ov` like the gate words
    float ovh[Q::RPW][3];
    unsigned ovh_off = 0;                         // this lane's float index of (row 0, channel 0) in head_out
    bool ovh_pending = false;
    const bool has_y = a.y != nullptr;            // (uniform) GATED == 5 may run without the 64-channel output
    const unsigned plane = (unsigned)(a.H * a.W);
    auto store_ovh = [&](int k) __attribute__((always_inline)) {       // k = 3 * row + channel
        const int mi = k / 3, c = k % 3;
        if (lh == 0 && ((ov_ok >> (8 * (mi >> 1) + (mi & 1))) & 1u)) a.head_out[ovh_off + (unsigned)c * plane + (unsigned)mi * (unsigned)a.W] = ovh[mi][c];
    };
    auto store_ovb = [&](int mi) __attribute__((always_inline)) {
        if ((ov_ok >> (8 * (mi >> 1) + (mi & 1))) & 1u) a.gbits[ovb_off + (unsigned)mi * (unsigned)(a.W * 2 * a.cout_tiles)] = ovb[mi];
    };
    const unsigned ov_row = (unsigned)(a.W * a.ldy * 2);                 // bytes per output row
    auto store_ov = [&](int k) __attribute__((always_inline)) {
        // k = ((mp * 2 + ni) * 2 + gp) * 2 + r  ->  row 2 mp + r, channels + 32 ni + 16 gp
        const int r = k & 1, gp = (k >> 1) & 1, ni = (k >> 2) & 1, mp = k >> 3;
#if WU_CONV_STORE_AUX
        // experiment (scratch/ab_unet_lib.sh): the same 16-byte stores through a buffer descriptor with cache-policy bits (aux 16 = sc1, write-through: no dirty
        // lines left in the XCD L2s for the end-of-kernel write-back; 2 = nt)
        if ((ov_ok >> k) & 1u) {
            const unsigned long long u = (unsigned long long)(uintptr_t)ov_img;
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(u & 0xffffffffull)), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32));
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, ov[k]), rs,
                                                   (int)(ov_off + (unsigned)(2 * mp + r) * ov_row + (unsigned)(64 * ni + 32 * gp)), 0, WU_CONV_STORE_AUX);
        }
#else
        if ((ov_ok >> k) & 1u) *(uint4*)((char*)ov_img + (ov_off + (unsigned)(2 * mp + r) * ov_row + (unsigned)(64 * ni + 32 * gp))) = ov[k];
#endif
    };
    Tc cur = decode(t_begin), fetch = cur;
    set_fetch_tile(fetch);
#pragma unroll
    for (int j = 0; j < Q::NP; ++j) issue_piece(j, 0, 0);

    for (int tile = t_begin; tile < t_end; tile += t_step) {
        // accumulators are kept TRANSPOSED (rows = cout, cols = pixels: the weight fragment is the MFMA A operand):
        // a lane then owns 4 consecutive channels of one pixel per register quad -> 8-byte epilogue writes
        typename AccT<M16, Q::RPW>::type acc;            // 32x32: [row][cout half] x 16; M16: [row][pixel half][cout quarter] x 4
#pragma unroll
        for (int mi = 0; mi < Q::RPW; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                if constexpr (M16) {
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) acc[mi][ni][cb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;
                }
            }

        // bias of this tile's 64 channels, in the transposed-accumulator layout (4 consecutive channels per register quad)
        float4 bvq[2][4];
        float4 b16[4];                // M16: bias of couts 16 cb + 4 q .. + 3
        uint4 egv[Q::RPW][2][2];      // gate values (dgrad): prefetched in the last chunk
        unsigned egb[Q::RPW];         // gate bits (dgrad): one dword per row
        // One K chunk.  Chunk 0 is a separate instance of this code (FIRST): only it issues the previous tile's deferred stores, and
        // it is never the last chunk (Cin >= 64) -- so `ov` is dead before the loop over the later chunks, where the bias / gate
        // registers come alive (inside one loop the allocator has to keep all three sets at once: 47 spilled registers).
        // Fragment ring (three register sets at 8 waves, two at 4), at TILE scope: the MFMAs of a chunk's last step(s) are carried
        // across the chunk-top barrier (round 3).  Their fragments are in registers before the barrier (the LDS buffer can be handed
        // to the DMA), and they execute BEHIND the first fragment reads of the next chunk -- the pipeline no longer drains at the
        // end of every chunk and refills (one LDS round trip with the matrix pipe idle, for both waves of a SIMD at once) behind
        // every barrier.  Same MFMA order per accumulator: results are bit-identical.  Only a tile's last chunk runs all 18 steps.
        constexpr int RING = 3, CARRY = NW == 8 ? 2 : 1;     // carried: (tap, k-step) steps 16, 17 at 8 waves; tap 8 at 4 waves (M16)
        uint4 raf[RING][Q::RPW], rbf[RING][2];               // 32x32x16 fragments (8 waves)
        uint4 xr[RING][Q::RPW][2], wr[RING][4];              // 16x16x32 fragments (M16): [row][pixel half], [cout quarter]
        auto do_chunk = [&](const int c, auto first_tag, auto last_tag) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;
            constexpr bool LAST = decltype(last_tag)::value;      // the tile's last chunk: nothing is carried out of it
            static_assert(!(FIRST && LAST), "a tile has at least two chunks (Cin >= 64)");
            const char* lds = smem + buf * K::BUF;
            const int nxt = buf ^ 1;
            // what to fetch while computing this chunk: the next chunk of this tile, or chunk 0 of the next tile
            constexpr bool last = LAST;
            const bool need_w = !a.w_resident || (FIRST && tile == t_begin);     // (wave-uniform)
            const bool more = !last || tile + t_step < t_end;
            const int c1 = last ? 0 : (c + 1) * 32;
            // this wave's pieces of the current chunk must have landed.  Right after an interior tile's epilogue the 8
            // output stores are the YOUNGEST vector-memory ops and every DMA piece is older: vmcnt(8) retires the DMA
            // without draining the stores to HBM (vmcnt counts loads, stores and LDS-DMA together, in issue order).
            if (GATED == 5 && stores_in_flight && has_y) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Q::NST + 3 * Q::RPW) : "memory");
            else if (GATED == 5 && stores_in_flight) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * Q::RPW) : "memory");
            else if (stores_in_flight && bits_in_flight) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Q::NST + Q::RPW) : "memory");
            else if (stores_in_flight) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Q::NST) : "memory");
            else dma_wait_all();
            stores_in_flight = false;
            WU_STAMP(t_wait);
            __syncthreads();     // ... and so have everyone else's; everyone is also done with the other buffer
            WU_STAMP(t_epi_b2);  // (diagnostic) chunk-top barrier time is folded into the 'barrier2' slot
            if (last && (GATED == 0 || GATED == 3 || GATED == 4 || GATED == 5)) {      // requested in the LAST chunk: lands under its MFMAs, and its registers are free for `ov` before
                const int ct_ = cur.ct;
                if constexpr (M16) {
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        b16[cb] = a.bias ? *(const float4*)(a.bias + ct_ * 64 + 16 * cb + 4 * q16) : make_float4(0.f, 0.f, 0.f, 0.f);
                } else {
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            bvq[ni][g] = a.bias ? *(const float4*)(a.bias + ct_ * 64 + 32 * ni + 8 * g + 4 * lh) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            if (last && more) { fetch = advance(fetch); set_fetch_tile(fetch); }
            // the gate values of this tile's outputs are requested at the start of its LAST chunk: they land under the MFMAs
            // instead of stalling every store of the epilogue (out-of-image pixels are clamped, their stores are skipped)
            if (last && GATED == 1) {
                const int ct_ = cur.ct, tx_ = cur.tx, ty_ = cur.ty, n_ = cur.n;
#pragma unroll
                for (int mi = 0; mi < Q::RPW; ++mi) {
                    const int oh = min(ty_ * K::TH + Q::RPW * wave + mi, a.H - 1), ow = min(tx_ * K::TW + l31, a.W - 1);
                    const bf16_t* ep = a.egate + ((size_t)n_ * a.H * a.W + (size_t)(oh * a.W + ow)) * a.ldegate + ct_ * 64 + 8 * lh;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int gp = 0; gp < 2; ++gp) egv[mi][ni][gp] = *(const uint4*)(ep + 32 * ni + 16 * gp);
                }
            }
            if (last && GATED == 2) {
                const int ct_ = cur.ct, tx_ = cur.tx, ty_ = cur.ty, n_ = cur.n;
#pragma unroll
                for (int mi = 0; mi < Q::RPW; ++mi) {
                    const int oh = min(ty_ * K::TH + Q::RPW * wave + mi, a.H - 1), ow = min(tx_ * K::TW + l31, a.W - 1);
                    egb[mi] = a.egbits[(((size_t)n_ * a.H + oh) * a.W + ow) * (2 * a.cout_tiles) + 2 * ct_ + lh];
                }
            }
            // 18 steps (tap, ks), software-pipelined by hand: the fragments of step s+1 are requested BEFORE the four
            // MFMAs of step s are issued, so one LDS round trip is always covered by matrix work of this wave
            // (a 2-step look-ahead measured 2-3 % slower).
            auto load_step = [&](int step, uint4 (&af)[Q::RPW], uint4 (&bf)[2]) __attribute__((always_inline)) {
                const int tap = step >> 1, ks = step & 1, kh = tap / 3, kw = tap % 3;
#pragma unroll
                for (int mi = 0; mi < Q::RPW; ++mi)
                    af[mi] = *(const uint4*)(lds + a_lane[kw][ks] + ((mi + kh) * K::HALO_W + kw) * 64);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    bf[ni] = *(const uint4*)(lds + b_lane[ks] + (tap * 64 + 32 * ni) * 64);
            };
            if constexpr (NW == 8) {
                // fragments TWO steps ahead (ring of three register sets): whichever wave of a SIMD loses the arbitration runs
                // the tail of the chunk alone, and a lone wave's 4 MFMAs per step (128 cycles) do not cover an LDS round trip
                auto (&af) = raf;
                auto (&bf) = rbf;
                auto mma_step = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
                    for (int mi = 0; mi < Q::RPW; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) mma(acc[mi][ni], bf[slot][ni], af[slot][mi]);   // D^T = W * X^T
                };
                if constexpr (FIRST) {
                    load_step(0, af[0], bf[0]);
                    load_step(1, af[1], bf[1]);
                } else {
                    // carried in: steps 16 / 17 of the previous chunk sit in ring slots 1 / 2 (16 % 3, 17 % 3); slot 0 is free
                    load_step(0, af[0], bf[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    mma_step(1);
                    __builtin_amdgcn_sched_barrier(0);
                    load_step(1, af[1], bf[1]);
                    __builtin_amdgcn_sched_barrier(0);
                    mma_step(2);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int step = 0; step < 18; ++step) {
                    if (step + 2 < 18) load_step(step + 2, af[(step + 2) % 3], bf[(step + 2) % 3]);
                    // issue the next chunk's 10 DMA pieces at the START of this chunk (2 per step): they get ~3/4 of the chunk to
                    // land (LDS-DMA latency ~1.2 us)
                    // (resident weights: with two chunks per tile, chunk q always sits in buffer q, and with one cout tile every tile
                    //  multiplies by the same slab -- after the workgroup's first tile only the halo pieces are fetched: -4...-8 %.
                    //  The mirror image -- a pixel tile's halo chunks kept for its several cout tiles (64 -> 128 / 192 layers, contiguous
                    //  tile walk) -- measured 2-2.6 % SLOWER than the grid-strided walk that re-fetches them from L2: not built.)
                    if (more && 2 * step < Q::NP) {
                        if (need_w || 2 * step < Q::NH) issue_piece(2 * step, c1, nxt);
                        if (2 * step + 1 < Q::NP && (need_w || 2 * step + 1 < Q::NH)) issue_piece(2 * step + 1, c1, nxt);
                    }
                    // the previous tile's outputs, one store per step once this chunk's DMA pieces are out (they stay the
                    // youngest vector-memory ops: the next chunk-top wait is vmcnt(NST))
                    static_assert(2 * 5 >= Q::NP || NW != 8, "deferred stores must follow the last DMA piece");
                    if (FIRST && ov_pending && (GATED != 5 || has_y) && step >= 5 && step < 5 + Q::NST) store_ov(step - 5);
                    if (GATED == 3 && FIRST && ovb_pending && step >= 5 + Q::NST && step < 5 + Q::NST + Q::RPW) store_ovb(step - 5 - Q::NST);
                    // head outputs: two dword stores per step behind the tile's 16-byte stores (3 RPW = 6 of them at 8 waves: steps 13..15)
                    if (GATED == 5 && FIRST && ovh_pending && step >= 5 + Q::NST && 2 * (step - 5 - Q::NST) < 3 * Q::RPW) {
                        store_ovh(2 * (step - 5 - Q::NST));
                        if (2 * (step - 5 - Q::NST) + 1 < 3 * Q::RPW) store_ovh(2 * (step - 5 - Q::NST) + 1);
                    }
                    // pin the order: left alone, the scheduler sinks the fragment reads of the DMA-free steps (6..17) to just
                    // before their first use and waits lgkmcnt(0) in front of every MFMA
                    __builtin_amdgcn_sched_barrier(0);
                    if (LAST || step < 18 - CARRY) mma_step(step % 3);          // steps 16, 17: carried into the next chunk
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                // ONE wave per SIMD, 16x16x32: a step is a whole TAP (all 32 channels of the chunk): 12 fragments (4 weight quarters,
                // 4 rows x 2 pixel halves) feed 32 MFMAs.  Nothing else fills the matrix pipe while this wave issues loads, so the
                // stream is pinned instruction by instruction: after MFMA m of tap t comes fragment read m of tap t+1 (12 reads over
                // the first 12 of 32 MFMAs, in the order the next tap's MFMAs need them); DMA pieces and the previous tile's deferred
                // stores ride behind MFMAs 15 and 31.  Ring of three fragment sets (slot = tap % 3: tap 8, carried over the barrier
                // in slot 2, leaves slots 0 and 1 to the next chunk's first taps).
                constexpr int NF = 12, NM = 32;
                auto load_frag = [&](int tap, int f, uint4 (&xf)[Q::RPW][2], uint4 (&wf)[4]) __attribute__((always_inline)) {
                    const int kh = tap / 3, kw = tap % 3;
                    // arrival order = use order (MFMA m: cout quarter m >> 3, row (m & 7) >> 1, pixel half m & 1): W0, X00 .. X31, W1, W2, W3
                    if (f == 0 || f >= 9) {
                        const int cb = f == 0 ? 0 : f - 8;
                        wf[cb] = *(const uint4*)(lds + b16_lane + (tap * 64 + 16 * cb) * 64);
                    } else {
                        const int mi = (f - 1) >> 1, ph = (f - 1) & 1;
                        xf[mi][ph] = *(const uint4*)(lds + a16_lane[kw] + ((mi + kh) * K::HALO_W + 16 * ph) * 64);
                    }
                };
                auto mma_m = [&](int m, int slot) __attribute__((always_inline)) {
                    const int cb = m >> 3, mi = (m & 7) >> 1, ph = m & 1;
                    mma16(acc[mi][ph][cb], wr[slot][cb], xr[slot][mi][ph]);                 // D^T (couts x pixels) = W * X^T
                };
                const unsigned kill = more ? 0u : kWuOOB;
                if constexpr (FIRST) {
#pragma unroll
                    for (int f = 0; f < NF; ++f) load_frag(0, f, xr[0], wr[0]);
                } else {
                    // carried in: tap 8 of the previous chunk (ring slot 2); its MFMAs cover the fragment reads of tap 0
#pragma unroll
                    for (int m = 0; m < NM; ++m) {
                        mma_m(m, 2);
                        if (m < NF) load_frag(0, m, xr[0], wr[0]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int cur = tap % 3, nx = (tap + 1) % 3;
#pragma unroll
                    for (int m = 0; m < NM; ++m) {
                        if (LAST || tap < 9 - CARRY) mma_m(m, cur);                             // tap 8: carried into the next chunk
                        if (tap + 1 < 9 && m < NF) load_frag(tap + 1, m, xr[nx], wr[nx]);
                        // 20 DMA pieces, two behind MFMAs 15 and 31 of taps 0..4
                        if ((m == 15 || m == NM - 1) && 4 * tap < Q::NP) {
                            const int j0 = 4 * tap + (m == 15 ? 0 : 2);
                            if (j0 < Q::NP) issue_piece(j0, c1, nxt, kill);
                            if (j0 + 1 < Q::NP) issue_piece(j0 + 1, c1, nxt, kill);
                        }
                        // the previous tile's 16 outputs: two stores behind MFMAs 15 and 31 of taps 5..8 (after the DMA pieces)
                        static_assert(4 * 5 >= Q::NP || NW != 4, "deferred stores must follow the last DMA piece");
                        if (FIRST && (m == 15 || m == NM - 1) && ov_pending && tap >= 5) {
                            const int k0 = 4 * (tap - 5) + (m == 15 ? 0 : 2);
                            if (k0 + 1 < Q::NST) { store_ov(k0); store_ov(k0 + 1); }
                        }
                        if (GATED == 3 && FIRST && m == NM - 1 && ovb_pending && tap == 8) {
#pragma unroll
                            for (int mi = 0; mi < Q::RPW; ++mi) store_ovb(mi);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            buf = nxt;
            if (FIRST && ov_pending) { stores_in_flight = ov_interior; bits_in_flight = ovb_pending; ov_pending = false; ovb_pending = false; ovh_pending = false; }
            if (FIRST) WU_STAMP(t_comp); else WU_STAMP(t_comp_rest);
        };
        do_chunk(0, std::true_type{}, std::false_type{});
        for (int c = 1; c + 1 < nchunks; ++c) do_chunk(c, std::false_type{}, std::false_type{});
        do_chunk(nchunks - 1, std::false_type{}, std::true_type{});
        // M16: every wave is done with the fragment reads of the tile's last chunk -- its LDS buffer becomes the transpose scratch
        // of the epilogue (one extra workgroup barrier per tile; a tile of these instances is 8-24 chunks long)
        if constexpr (M16) __syncthreads();

        // ---- epilogue of this tile (its last chunk sat in buffer buf^1, now free; buffer `buf` is receiving the next
        //      tile's chunk 0): bias + activation in fp32, packed to bf16 in registers (`ov`); the stores follow later ----
        const int n = cur.n;
        const int oh0 = cur.ty * K::TH, ow0 = cur.tx * K::TW, co0 = cur.ct * 64;
        // Direct epilogue, no LDS and no workgroup barrier: bias + activation in fp32, bf16 packing, then one
        // v_permlane32_swap per dword pairs the two half-waves' 8-byte channel groups into 16 contiguous bytes per lane
        // (lanes 0-31: channels 8k..8k+7 of their pixel, lanes 32-63: 8k+8..8k+15) -> 16-byte stores straight from registers.
        // A wave that finishes its MFMAs early does this while its SIMD partner still computes.
        WU_STAMP(t_epi_b1);
        const size_t img_pix = (size_t)n * a.H * a.W;
        // Specialised per (activation, gate) pair -- selected once per tile -- so the body is straight packed arithmetic:
        // v_pk_add_f32 bias, v_cvt_pk_bf16_f32, ReLU as v_pk_max_i16 on the packed pair (ReLU commutes with the monotonic
        // rounding), the ReLU gate as packed 16-bit integer masks; no per-element compare / select chains.
        auto epi_store = [&](auto act_tag, auto eg_tag, auto pool_tag) __attribute__((always_inline)) {
            constexpr int ACT = decltype(act_tag)::value, EG = decltype(eg_tag)::value;
            constexpr bool POOL = decltype(pool_tag)::value;       // only with ACT == RELU (non-negative outputs)
            // M16: the 16x16 accumulators (4 couts of one pixel per lane) -> bias, activation, bf16 -> this wave's slice of the LDS
            // buffer the tile's last chunk was computed from, [row][pixel] rows of 64 channels at a 144-byte pitch (conflict-free for
            // the 8-byte writes and the 16-byte reads); read back below, 16 bytes per (pixel l31, half lh), exactly the register
            // contents the 32x32 path gets from v_permlane32_swap.  A wave only reads what it wrote itself.
            constexpr int SCR_PITCH = 144;
            char* const scr = smem + (buf ^ 1) * K::BUF + wave * (Q::RPW * 32 * SCR_PITCH);
            if constexpr (M16) {
                static_assert(NW * Q::RPW * 32 * SCR_PITCH <= K::BUF, "transpose scratch must fit the free LDS buffer");
#pragma unroll
                for (int mi = 0; mi < Q::RPW; ++mi)
#pragma unroll
                    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb) {
                            const float4 bv = (GATED == 1 || GATED == 2) ? make_float4(0.f, 0.f, 0.f, 0.f) : b16[cb];
                            f32x2_t v0 = f32x2_t{acc[mi][ph][cb][0], acc[mi][ph][cb][1]} + f32x2_t{bv.x, bv.y};
                            f32x2_t v1 = f32x2_t{acc[mi][ph][cb][2], acc[mi][ph][cb][3]} + f32x2_t{bv.z, bv.w};
                            if (ACT == WU_ACT_LEAKY) {      // max(v, 0.2 v)
                                const f32x2_t s0_ = v0 * 0.2f, s1_ = v1 * 0.2f;
                                v0 = f32x2_t{fmaxf(v0.x, s0_.x), fmaxf(v0.y, s0_.y)};
                                v1 = f32x2_t{fmaxf(v1.x, s1_.x), fmaxf(v1.y, s1_.y)};
                            }
                            uint2 o = make_uint2(pack_bf16x2(v0.x, v0.y), pack_bf16x2(v1.x, v1.y));
                            if (ACT == WU_ACT_RELU) { o.x = relu_bf16x2(o.x); o.y = relu_bf16x2(o.y); }
                            *(uint2*)(scr + (mi * 32 + 16 * ph + l15) * SCR_PITCH + (16 * cb + 4 * q16) * 2) = o;
                        }
            }
            unsigned gb[Q::RPW], sb[Q::RPW];
#pragma unroll
            for (int i = 0; i < Q::RPW; ++i) gb[i] = sb[i] = 0u;
            f32x16_t hacc[Q::RPW];                                   // GATED == 5: head products, rows = head channel (+ 8: residual part), columns = pixels
            if constexpr (GATED == 5) {
#pragma unroll
                for (int mi = 0; mi < Q::RPW; ++mi)
#pragma unroll
                    for (int i = 0; i < 16; ++i) hacc[mi][i] = 0.f;
            }
#pragma unroll
            for (int mp = 0; mp < Q::RPW / 2; ++mp) {                // the wave's rows in vertical pairs (even, odd)
                const int ohe = oh0 + Q::RPW * wave + 2 * mp, ow = ow0 + l31;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int g = 0; g < 4; g += 2) {
                        uint4 vr[2];
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            const int mi = 2 * mp + r;
                            uint4 v;
                            if constexpr (M16) {
                                v = *(const uint4*)(scr + (mi * 32 + l31) * SCR_PITCH + (32 * ni + 8 * g + 8 * lh) * 2);
                            } else {
                            uint32_t o[2][2];
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const float4 bv = (GATED == 1 || GATED == 2) ? make_float4(0.f, 0.f, 0.f, 0.f) : bvq[ni][g + h];
                                const int r0 = 4 * (g + h);
                                f32x2_t v0 = f32x2_t{acc[mi][ni][r0 + 0], acc[mi][ni][r0 + 1]} + f32x2_t{bv.x, bv.y};
                                f32x2_t v1 = f32x2_t{acc[mi][ni][r0 + 2], acc[mi][ni][r0 + 3]} + f32x2_t{bv.z, bv.w};
                                if (ACT == WU_ACT_LEAKY) {      // max(v, 0.2 v)
                                    const f32x2_t s0_ = v0 * 0.2f, s1_ = v1 * 0.2f;
                                    v0 = f32x2_t{fmaxf(v0.x, s0_.x), fmaxf(v0.y, s0_.y)};
                                    v1 = f32x2_t{fmaxf(v1.x, s1_.x), fmaxf(v1.y, s1_.y)};
                                }
                                o[h][0] = pack_bf16x2(v0.x, v0.y);
                                o[h][1] = pack_bf16x2(v1.x, v1.y);
                                if (ACT == WU_ACT_RELU) { o[h][0] = relu_bf16x2(o[h][0]); o[h][1] = relu_bf16x2(o[h][1]); }
                            }
                            // vdst = group g, src = group g+1 (cdna guide T21)
                            const auto s0 = __builtin_amdgcn_permlane32_swap(o[0][0], o[1][0], false, false);
                            const auto s1 = __builtin_amdgcn_permlane32_swap(o[0][1], o[1][1], false, false);
                            v = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                            }
                            if (GATED == 2) {
                                // byte k of the row's gate word: element e of this 16-byte group passes iff bit e is set
                                const int by = (int)(egb[mi] >> (8 * (2 * ni + (g >> 1))));
                                v.x &= gate_mask2(by, 0);
                                v.y &= gate_mask2(by, 2);
                                v.z &= gate_mask2(by, 4);
                                v.w &= gate_mask2(by, 6);
                            } else if (EG != WU_ACT_NONE) {
                                const uint4 yv = egv[mi][ni][g >> 1];
                                if (EG == WU_ACT_RELU) {
                                    v.x = relu_gate_bf16x2(v.x, yv.x); v.y = relu_gate_bf16x2(v.y, yv.y);
                                    v.z = relu_gate_bf16x2(v.z, yv.z); v.w = relu_gate_bf16x2(v.w, yv.w);
                                } else {
                                    v = gate16<bf16_t>(v, yv, WU_ACT_LEAKY);
                                }
                            }
                            ov[((mp * 2 + ni) * 2 + (g >> 1)) * 2 + r] = v;       // stored from inside the next tile's first chunk
                            vr[r] = v;
                            if constexpr (GATED == 5) mma(hacc[mi], hA[2 * ni + (g >> 1)], v);      // z[head channel][pixel] += Wh[:, 16 channels] * y[16 channels][pixel]
                            if ((GATED == 3 || GATED == 4) && ACT == WU_ACT_RELU)
                                gb[2 * mp + r] |= nonzero_byte(v) << (8 * (2 * ni + (g >> 1)));
                        }
                        if (POOL) {
                            // 2x2 max-pool (cunet.py:46,50,54) of the ReLU outputs: non-negative bf16 order like unsigned integers,
                            // so the window maximum is v_pk_max_u16 over the row pair and over the neighbouring lane (pixel ow ^ 1)
                            uint4 m = make_uint4(max_u16x2(vr[0].x, vr[1].x), max_u16x2(vr[0].y, vr[1].y), max_u16x2(vr[0].z, vr[1].z), max_u16x2(vr[0].w, vr[1].w));
                            m.x = max_u16x2(m.x, (uint32_t)__builtin_amdgcn_mov_dpp((int)m.x, 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
                            m.y = max_u16x2(m.y, (uint32_t)__builtin_amdgcn_mov_dpp((int)m.y, 0xB1, 0xF, 0xF, true));
                            m.z = max_u16x2(m.z, (uint32_t)__builtin_amdgcn_mov_dpp((int)m.z, 0xB1, 0xF, 0xF, true));
                            m.w = max_u16x2(m.w, (uint32_t)__builtin_amdgcn_mov_dpp((int)m.w, 0xB1, 0xF, 0xF, true));
                            if ((l31 & 1) == 0 && ohe + 1 < a.H && ow + 1 < a.W) {
                                const int Hp = a.H >> 1, Wp = a.W >> 1;
                                *(uint4*)(a.pool + (((size_t)n * Hp + (ohe >> 1)) * Wp + (ow >> 1)) * a.ldpool + co0 + 8 * lh + 32 * ni + 8 * g) = m;
                            }
                            if constexpr (GATED == 4) {
                                // arg-max bits: element e of (row r, this lane's column) is the window's FIRST maximum in scan order (0,0), (0,1),
                                // (1,0), (1,1) -- PyTorch's rule, what maxpool2_bwd_kernel re-derives from the tensor.  eq byte: bit e = (v == max).
                                const uint4 x0 = make_uint4(vr[0].x ^ m.x, vr[0].y ^ m.y, vr[0].z ^ m.z, vr[0].w ^ m.w);
                                const uint4 x1 = make_uint4(vr[1].x ^ m.x, vr[1].y ^ m.y, vr[1].z ^ m.z, vr[1].w ^ m.w);
                                const uint32_t eq0 = ~nonzero_byte(x0) & 0xffu, eq1 = ~nonzero_byte(x1) & 0xffu;
                                const uint32_t nb = (uint32_t)__builtin_amdgcn_mov_dpp((int)(eq0 | (eq1 << 8)), 0xB1, 0xF, 0xF, true);   // the other column
                                const uint32_t oddm = (l31 & 1) ? 0xffu : 0u;
                                const uint32_t nb0 = nb & 0xffu, nb1 = (nb >> 8) & 0xffu;
                                const uint32_t s0 = eq0 & ~(nb0 & oddm);                                   // (0,1) yields to (0,0)
                                const uint32_t s1 = eq1 & ~eq0 & ~nb0 & ~(nb1 & oddm);                     // row 1 yields to row 0; (1,1) also to (1,0)
                                sb[2 * mp] |= s0 << (8 * (2 * ni + (g >> 1)));
                                sb[2 * mp + 1] |= s1 << (8 * (2 * ni + (g >> 1)));
                            }
                        }
                    }
            }
            if constexpr (GATED == 4) {
                // stored right here, like the pooled tensor above (a pool instance pays the immediate-store wait at the next chunk top anyway)
#pragma unroll
                for (int mi = 0; mi < Q::RPW; ++mi) {
                    const int oh = oh0 + Q::RPW * wave + mi, ow = ow0 + l31;
                    if (oh < a.H && ow < a.W) {
                        const size_t wi = (((size_t)n * a.H + oh) * a.W + ow) * (2 * a.cout_tiles) + 2 * cur.ct + lh;
                        a.gbits[wi] = gb[mi];
                        a.sbits[wi] = sb[mi];
                    }
                }
            }
            if constexpr (GATED == 5) {                   // bias + tanh; parked like `ov` (lanes of half 0 hold rows 0..3, 8..11 and 16..19 of the product)
#pragma unroll
                for (int mi = 0; mi < Q::RPW; ++mi)
#pragma unroll
                    for (int c = 0; c < 3; ++c) ovh[mi][c] = fast_tanh((hacc[mi][c] + (hacc[mi][4 + c] + hacc[mi][8 + c])) + hbias[c]);
                ovh_off = (unsigned)(((size_t)n * 3 * a.H + oh0 + Q::RPW * wave) * a.W + ow0 + l31);
                ovh_pending = true;
            }
            if (GATED == 3 && ACT == WU_ACT_RELU) {       // parked like `ov`: issued from the next tile's first chunk
#pragma unroll
                for (int mi = 0; mi < Q::RPW; ++mi) ovb[mi] = gb[mi];
                ovb_off = (unsigned)((((size_t)n * a.H + oh0 + Q::RPW * wave) * a.W + ow0 + l31) * (2 * a.cout_tiles) + 2 * cur.ct + lh);
                ovb_pending = true;
            }
        };
        using A0 = std::integral_constant<int, WU_ACT_NONE>;
        using A1 = std::integral_constant<int, WU_ACT_RELU>;
        using A2 = std::integral_constant<int, WU_ACT_LEAKY>;
        using NoPool = std::false_type;
        if constexpr (GATED == 5) {        // forward + ReLU + pointwise head + tanh
            epi_store(A1{}, A0{}, NoPool{});
        } else if constexpr (GATED == 4) {        // forward + ReLU + pool + gate / arg-max bits
            epi_store(A1{}, A0{}, std::true_type{});
        } else if constexpr (GATED == 3) { // forward + gate bits: ReLU, no pool (conv_v2_launch)
            epi_store(A1{}, A0{}, NoPool{});
        } else if constexpr (GATED == 2) { // ReLU gate from bits
            epi_store(A0{}, A0{}, NoPool{});
        } else if constexpr (GATED == 1) { // host guarantees act == NONE and no bias with a gate (conv_v2_launch)
            if (a.egate_act == WU_ACT_RELU) epi_store(A0{}, A1{}, NoPool{});
            else epi_store(A0{}, A2{}, NoPool{});
        } else {
            if (a.act == WU_ACT_RELU) {
                if (a.pool) epi_store(A1{}, A0{}, std::true_type{});      // host guarantees act == RELU with a pool output
                else epi_store(A1{}, A0{}, NoPool{});
            }
            else if (a.act == WU_ACT_LEAKY) epi_store(A2{}, A0{}, NoPool{});
            else epi_store(A0{}, A0{}, NoPool{});
        }
        // where the deferred stores go: lane base = pixel (oh0 + RPW wave, ow0 + l31), channel co0 + 8 lh; bit k of ov_ok = store k
        // lies inside the image.  Interior tile: every lane will issue all NST stores (the counted vmcnt wait after the next
        // tile's first chunk relies on it).
        {
            const int ohw = oh0 + Q::RPW * wave, ow = ow0 + l31;
            ov_img = has_y ? a.y + img_pix * a.ldy : a.y;
            ov_off = (unsigned)(((ohw * a.W + ow) * a.ldy + co0 + 8 * lh) * 2);
            unsigned rows = 0;
#pragma unroll
            for (int rr = 0; rr < Q::RPW; ++rr) rows |= (ohw + rr < a.H && ow < a.W) ? (1u << rr) : 0u;
            ov_ok = 0;
#pragma unroll
            for (int k = 0; k < Q::NST; ++k) ov_ok |= ((rows >> (2 * (k >> 3) + (k & 1))) & 1u) << k;
            ov_interior = oh0 + K::TH <= a.H && ow0 + K::TW <= a.W;
            ov_pending = true;
        }
        WU_STAMP(t_epi_s);
        cur = advance(cur);
    }
    if (ov_pending) {                // the last tile's outputs
        if (GATED != 5 || has_y) {
#pragma unroll
            for (int k = 0; k < Q::NST; ++k) store_ov(k);
        }
        if (GATED == 5 && ovh_pending) {
#pragma unroll
            for (int k = 0; k < 3 * Q::RPW; ++k) store_ovh(k);
        }
        if (GATED == 3 && ovb_pending) {
#pragma unroll
            for (int mi = 0; mi < Q::RPW; ++mi) store_ovb(mi);
        }
    }
    if (NW == 4) dma_wait_all();     // the killed pieces of the last chunk still write LDS: drain before the LDS is released
    if (a.dbg && lane == 0) {
        unsigned long long* d = a.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
        d[0] = t_wait; d[1] = t_comp + t_comp_rest; d[2] = __builtin_readcyclecounter() - t_k0; d[3] = __builtin_amdgcn_s_memrealtime() - t_r0;   // in-kernel clock = d2 / d3 * 100 MHz
         d[4] = t_epi_b2; d[5] = t_epi_s; d[6] = (unsigned long long)((t_end - t_begin + t_step - 1) / t_step); d[7] = nchunks | (t_comp << 8);     // bits 8..: compute time of the tiles' FIRST chunks (diagnostic)
    }
#undef WU_STAMP
}

}  // namespace

bool conv_v2_eligible(int H, int W, int ldx, int ldy, int Cin, int Cout, int stride, int dtype, bool masked) {
    // 32-bit DMA byte offsets below kWuOOB: the weight pack and one (row + pixel padded) image
    return dtype == WU_BF16 && stride == 1 && !masked && W > 16 && W <= 4096 && Cin % 32 == 0 && Cin >= 64 && Cout % 64 == 0 &&
           (size_t)9 * Cout * Cin * 2 < (1ull << 31) && ((size_t)H * W + W + 2) * (size_t)ldx * 2 < (1ull << 31) &&
           ((size_t)H * W + W + 2) * (size_t)ldy * 2 < (1ull << 31);
}

int conv_v2_launch(const void* x, int ldx, const void* w, const float* bias, void* y, int ldy,
                   const void* egate, int ldegate, int egate_act, int N, int H, int W, int Cin, int Cout, int act, hipStream_t s,
                   void* pool, int ldpool, void* gate_bits_out, const void* egate_bits, void* sel_bits_out,
                   const float* head_w, const float* head_b, float* head_out) {
    V2Args a;
    a.head_w = head_w; a.head_b = head_b; a.head_out = head_out;
    a.pool = (bf16_t*)pool; a.ldpool = ldpool;
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.bias = bias; a.y = (bf16_t*)y; a.egate = (const bf16_t*)egate;
    a.ldx = ldx; a.ldy = ldy; a.ldegate = ldegate; a.egate_act = egate_act; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.act = act;
    a.tiles_x = cdiv(W, K::TW); a.tiles_y = cdiv(H, K::TH); a.cout_tiles = Cout / 64;
    a.dbg = (unsigned long long*)g_wu_dbg_ptr;
    a.prio_mode = g_wu_opt[WU_OPT_CONV_PRIO];
    a.strided = g_wu_opt[WU_OPT_CONV_STRIDED];
    a.w_resident = 0;
    const long long ntiles = (long long)N * a.tiles_x * a.tiles_y * a.cout_tiles;
    if (ntiles >= (1ll << 31)) return -1;
    a.ntiles = (int)ntiles;
    // persistent: one 8-wave workgroup per CU; each walks a contiguous tile range and prefetches across tiles
    const int cus = wu_num_cus();
    const long long grid = (ntiles < cus || !g_wu_opt[WU_OPT_CONV_PERSISTENT]) ? ntiles : cus;
    static thread_local bool attr_set = false;
    if (!attr_set) {
#define WU_V2_ATTR(NW_, G_) (void)hipFuncSetAttribute((const void*)conv3x3_mfma_v2_kernel<NW_, G_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
        WU_V2_ATTR(8, 5); WU_V2_ATTR(8, 0); WU_V2_ATTR(4, 0); WU_V2_ATTR(8, 1); WU_V2_ATTR(4, 1); WU_V2_ATTR(8, 2); WU_V2_ATTR(4, 2); WU_V2_ATTR(8, 3); WU_V2_ATTR(4, 3); WU_V2_ATTR(8, 4); WU_V2_ATTR(4, 4);
#undef WU_V2_ATTR
        attr_set = true;
    }
    // one wave per SIMD with 8 accumulators pays off once a tile has >= 8 chunks (fewer LDS reads per MFMA, no intra-SIMD
    // skew); with few chunks per tile its un-overlapped epilogue costs more than that.  option 0: 1 = auto, 2 = always 4, 3 = always 8
    const int mode = g_wu_opt[WU_OPT_CONV_V2];
    int gated = egate_bits ? 2 : ((egate != nullptr && egate_act != WU_ACT_NONE) ? 1 : 0);
    if (gated != 1) a.egate = nullptr;
    a.egbits = (const unsigned*)egate_bits;
    a.gbits = (gated == 0 && act == WU_ACT_RELU && !pool) ? (unsigned*)gate_bits_out : nullptr;
    a.sbits = nullptr;
    if (a.gbits) gated = 3;
    if (gated == 0 && act == WU_ACT_RELU && pool && gate_bits_out && sel_bits_out) {       // + pool + gate / arg-max bits
        a.gbits = (unsigned*)gate_bits_out; a.sbits = (unsigned*)sel_bits_out;
        gated = 4;
    }
    if (head_out) {                    // forward + ReLU + head: the 8-wave shape with one cout tile (checked by the caller)
        if (gated != 0 || act != WU_ACT_RELU || pool || Cout != 64 || Cin >= 256) return -2;
        gated = 5;
    }
    const bool nw4 = gated != 5 && (mode == 2 || (mode == 1 && Cin >= 256));
    a.w_resident = (!nw4 && Cin == 64 && a.cout_tiles == 1 && g_wu_opt[WU_OPT_CONV_W_RESIDENT]) ? 1 : 0;
#define WU_V2_GO(NW_, G_) hipLaunchKernelGGL((conv3x3_mfma_v2_kernel<NW_, G_>), dim3((int)grid), dim3(NW_ * 64), 2 * K::BUF, s, a)
    if (nw4) { if (gated == 4) WU_V2_GO(4, 4); else if (gated == 3) WU_V2_GO(4, 3); else if (gated == 2) WU_V2_GO(4, 2); else if (gated == 1) WU_V2_GO(4, 1); else WU_V2_GO(4, 0); }
    else { if (gated == 5) WU_V2_GO(8, 5); else if (gated == 4) WU_V2_GO(8, 4); else if (gated == 3) WU_V2_GO(8, 3); else if (gated == 2) WU_V2_GO(8, 2); else if (gated == 1) WU_V2_GO(8, 1); else WU_V2_GO(8, 0); }
#undef WU_V2_GO
    return 0;
}
