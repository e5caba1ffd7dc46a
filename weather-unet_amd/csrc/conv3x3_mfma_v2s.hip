// conv3x3 (pad 1, stride 1) implicit GEMM, bf16 -- the TWO-WORKGROUPS-PER-CU form of conv3x3_mfma_v2.hip.
//
// conv3x3_mfma_v2 keeps the matrix pipe fed from ONE workgroup per CU with two 80-KiB LDS buffers.  Its two wave shapes each
// lose something: 8 waves (two per SIMD, 4 accumulators each) pay a chunk-top barrier skew and 1.0 LDS fragment
// reads per MFMA; 4 waves (one per SIMD, 8 accumulators, 0.75 reads per MFMA) cannot fill the pipe alone (34.7 instead of 25.5
// cycles per MFMA measured for a lone wave) and nothing covers their epilogue -- which is why the 64-input-channel layers (two
// chunks per tile) sit at 0.75-0.95 PFLOP/s.
//
// Here a workgroup is the 4-wave / 8-accumulator shape with ONE 80-KiB buffer, so TWO workgroups share a CU: every SIMD holds
// two waves of the good shape from independent workgroups.  Nothing synchronises the two, so while one waits for its LDS-DMA
// (no double buffering inside a workgroup: the next chunk is fetched after the barrier that retires the current one), sits in
// a barrier or converts and stores its accumulators, the other one's MFMAs own the pipe.  Registers are cut to <= 256 per wave
// (bias and gate values are fetched inside the epilogue instead of being prefetched into 96 registers; the tile-invariant DMA
// offsets are recomputed per tile).  Same tiles, same LDS image, same swizzles, same results bit for bit as conv3x3_mfma_v2.
#include <type_traits>

#include "wu_common.h"
#include "conv_internal.h"

namespace {

struct KS {
    static constexpr int TH = 16, TW = 32;
    static constexpr int HALO_W = TW + 2, HALO_H = TH + 2, HALO_PIX = HALO_W * HALO_H;   // 34 x 18 = 612
    static constexpr int H_PIECES = 40, H_BYTES = H_PIECES * 1024;
    static constexpr int W_REAL = 36, W_PIECES = 40;
    static constexpr int BUF = H_BYTES + W_PIECES * 1024;            // 80 KiB: half of the CU's LDS
    static constexpr int NW = 4, RPW = 4;
    static constexpr int NH = H_PIECES / NW, NWT = W_PIECES / NW, NP = NH + NWT;          // 10 + 10 DMA pieces per wave and chunk
};

struct VSArgs {
    const bf16_t* x; const bf16_t* w; const float* bias; bf16_t* y; const bf16_t* egate; bf16_t* pool;
    int ldx, ldy, ldegate, egate_act, ldpool;
    int N, H, W, Cin, Cout, act;
    int tiles_x, tiles_y, cout_tiles, ntiles;
};

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef short s16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t s_relu_bf16x2(uint32_t d) {
    const s16x2_t v = __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, d), s16x2_t{0, 0});
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ uint32_t s_relu_gate_bf16x2(uint32_t g, uint32_t y) {
    const u16x2_t nz = __builtin_elementwise_min(__builtin_bit_cast(u16x2_t, y), u16x2_t{1, 1});
    const u16x2_t m = u16x2_t{0, 0} - nz;
    const s16x2_t sg = __builtin_bit_cast(s16x2_t, y) >> s16x2_t{15, 15};
    return g & __builtin_bit_cast(uint32_t, m) & ~__builtin_bit_cast(uint32_t, sg);
}
__device__ __forceinline__ uint32_t s_max_u16x2(uint32_t a, uint32_t b) {
    const u16x2_t m = __builtin_elementwise_max(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, b));
    return __builtin_bit_cast(uint32_t, m);
}
__device__ __forceinline__ void s_mma(f32x16_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}

__global__ __launch_bounds__(256, 2) void conv3x3_mfma_v2s_kernel(const VSArgs a) {
    using Q = KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;

    // persistent, grid-strided: at any moment the workgroups of an XCD hold consecutive tile ids (cout tile fastest)
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int t_step = (int)gridDim.x;
    if (wg >= a.ntiles) return;
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
    const Tc dstep = decode(t_step);
    auto advance = [&](Tc t) __attribute__((always_inline)) {
        t.ct += dstep.ct; int cy = t.ct >= a.cout_tiles ? 1 : 0; t.ct -= cy * a.cout_tiles;
        t.tx += dstep.tx + cy; cy = t.tx >= a.tiles_x ? 1 : 0; t.tx -= cy * a.tiles_x;
        t.ty += dstep.ty + cy; cy = t.ty >= a.tiles_y ? 1 : 0; t.ty -= cy * a.tiles_y;
        t.n += dstep.n + cy;
        return t;
    };

    // ---- DMA bookkeeping (LDS slot i = piece * 64 + lane, as in conv3x3_mfma_v2): per piece the halo position (hy, hx) and the
    //      16-byte slot are tile-invariant; the byte offset is rebuilt from them whenever the fetch tile changes ----
    int hpos[Q::NH];                       // (hy << 8) | hx, or -1 for the padding lanes
#pragma unroll
    for (int j = 0; j < Q::NH; ++j) {
        const int i = (Q::NW * j + wave) * 64 + lane;
        const int p = i >> 2;
        const int hy = p / Q::HALO_W, hx = p - hy * Q::HALO_W;
        hpos[j] = p < Q::HALO_PIX ? ((hy << 8) | hx) : -1;
    }
    const int sl = lane & 3;
    unsigned woff[Q::NWT];
#pragma unroll
    for (int j = 0; j < Q::NWT; ++j) {
        const int i = (Q::NW * j + wave) * 64 + lane;
        const int row = i >> 2;
        const int tap = row >> 6, co = row & 63;
        woff[j] = Q::NW * j + wave < Q::W_REAL ? (unsigned)(((tap * a.Cout + co) * a.Cin + (sl ^ ((co >> 2) & 3)) * 8) * 2) : kWuOOB;
    }
    const unsigned x_img_bytes = (unsigned)((((size_t)a.H * a.W + a.W) * a.ldx + a.Cin) * 2);
    unsigned hv[Q::NH];
    wu_rsrc_t rs_x = wu_make_rsrc(a.x, 0), rs_w = rs_x;
    unsigned so_tile = 0;
    auto set_fetch_tile = [&](const Tc& t) __attribute__((always_inline)) {
        const int oh0 = t.ty * Q::TH, ow0 = t.tx * Q::TW;
        rs_x = wu_make_rsrc(a.x + ((long long)t.n * a.H * a.W - (a.W + 1)) * a.ldx, x_img_bytes);
        rs_w = wu_make_rsrc(a.w + (size_t)t.ct * 64 * a.Cin, (unsigned)(((size_t)9 * a.Cout - (size_t)t.ct * 64) * a.Cin * 2));
        so_tile = (unsigned)__builtin_amdgcn_readfirstlane((oh0 * a.W + ow0) * a.ldx * 2);
        const int ymin = __builtin_amdgcn_readfirstlane(oh0 == 0 ? 1 : 0), xmin = __builtin_amdgcn_readfirstlane(ow0 == 0 ? 1 : 0);
        const int xmax = __builtin_amdgcn_readfirstlane(a.W - ow0);
#pragma unroll
        for (int j = 0; j < Q::NH; ++j) {
            const int hy = hpos[j] >> 8, hx = hpos[j] & 255;
            const bool ok = hpos[j] >= 0 && hy >= ymin && hx >= xmin && hx <= xmax;
            hv[j] = ok ? (unsigned)(((hy * a.W + hx) * a.ldx + (sl ^ ((hx >> 2) & 3)) * 8) * 2) : kWuOOB;
        }
    };
    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    auto issue_chunk = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < Q::NH; ++j)
            wu_dma16b(hv[j], rs_x, so_tile + (unsigned)c0 * 2, __builtin_amdgcn_readfirstlane(smem_base + (Q::NW * j + wave) * 1024));
#pragma unroll
        for (int j = 0; j < Q::NWT; ++j)
            wu_dma16b(woff[j], rs_w, (unsigned)c0 * 2, __builtin_amdgcn_readfirstlane(smem_base + Q::H_BYTES + (Q::NW * j + wave) * 1024));
    };

    // ---- per-lane fragment bases (identical to conv3x3_mfma_v2) ----
    int a_lane[3][2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            a_lane[kw][ks] = ((Q::RPW * wave) * Q::HALO_W + l31) * 64 + (((2 * ks + lh) ^ (((l31 + kw) >> 2) & 3)) << 4);
    int b_lane[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) b_lane[ks] = Q::H_BYTES + l31 * 64 + (((2 * ks + lh) ^ ((l31 >> 2) & 3)) << 4);

    const int nchunks = a.Cin / 32;
    Tc cur = decode(wg);
    set_fetch_tile(cur);
    issue_chunk(0);

    for (int tile = wg; tile < a.ntiles; tile += t_step) {
        f32x16_t acc[Q::RPW][2];
#pragma unroll
        for (int mi = 0; mi < Q::RPW; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

        for (int c = 0; c < nchunks; ++c) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces (and its epilogue stores) have retired
            __syncthreads();                                         // ... so have everyone else's: the chunk is in LDS
            // 18 (tap, k-step) steps, one basic block: after MFMA m of step s comes fragment read m of step s+1
            constexpr int NF = Q::RPW + 2, NM = 2 * Q::RPW;
            uint4 af[2][Q::RPW], bf[2][2];
            auto load_frag = [&](int step, int f, uint4 (&af_)[Q::RPW], uint4 (&bf_)[2]) __attribute__((always_inline)) {
                const int tap = step >> 1, ks = step & 1, kh = tap / 3, kw = tap % 3;
                if (f == 0 || f == 2) bf_[f >> 1] = *(const uint4*)(smem + b_lane[ks] + (tap * 64 + 32 * (f >> 1)) * 64);
                else {
                    const int mi = f == 1 ? 0 : f - 2;
                    af_[mi] = *(const uint4*)(smem + a_lane[kw][ks] + ((mi + kh) * Q::HALO_W + kw) * 64);
                }
            };
#pragma unroll
            for (int f = 0; f < NF; ++f) load_frag(0, f, af[0], bf[0]);
#pragma unroll
            for (int step = 0; step < 18; ++step) {
                const int cb = step & 1;
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    s_mma(acc[m >> 1][m & 1], bf[cb][m & 1], af[cb][m >> 1]);
                    if (step + 1 < 18 && m < NF) load_frag(step + 1, m, af[cb ^ 1], bf[cb ^ 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();                                         // every wave has read its last fragment: the buffer is free
            // fetch the next chunk of this tile: its latency is covered by the OTHER workgroup of this CU.  (Chunk 0 of the next
            // tile is requested AFTER the epilogue: vmcnt retires in order, so bias / gate loads issued behind 20 DMA pieces
            // would stall the epilogue until the whole chunk had landed.)
            if (c + 1 < nchunks) issue_chunk((c + 1) * 32);
        }

        // ---- direct register epilogue (as conv3x3_mfma_v2), bias / gate values fetched here ----
        const int n = cur.n;
        const int oh0 = cur.ty * Q::TH, ow0 = cur.tx * Q::TW, co0 = cur.ct * 64;
        const size_t img_pix = (size_t)n * a.H * a.W;
        auto epi_store = [&](auto act_tag, auto eg_tag, auto pool_tag) __attribute__((always_inline)) {
            constexpr int ACT = decltype(act_tag)::value, EG = decltype(eg_tag)::value;
            constexpr bool POOL = decltype(pool_tag)::value;
#pragma unroll
            for (int mp = 0; mp < Q::RPW / 2; ++mp) {
                const int ohe = oh0 + Q::RPW * wave + 2 * mp, ow = ow0 + l31;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int g = 0; g < 4; g += 2) {
                        float4 bv[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            bv[h] = a.bias ? *(const float4*)(a.bias + co0 + 32 * ni + 8 * (g + h) + 4 * lh) : make_float4(0.f, 0.f, 0.f, 0.f);
                        uint4 vr[2];
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            const int mi = 2 * mp + r;
                            const int oh = ohe + r;
                            uint4 yv = make_uint4(0, 0, 0, 0);
                            if (EG != WU_ACT_NONE) {
                                const int ohc = min(oh, a.H - 1), owc = min(ow, a.W - 1);
                                yv = *(const uint4*)(a.egate + (img_pix + (size_t)(ohc * a.W + owc)) * a.ldegate + co0 + 8 * lh + 32 * ni + 8 * g);
                            }
                            uint32_t o[2][2];
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const int r0 = 4 * (g + h);
                                f32x2_t v0 = f32x2_t{acc[mi][ni][r0 + 0], acc[mi][ni][r0 + 1]} + f32x2_t{bv[h].x, bv[h].y};
                                f32x2_t v1 = f32x2_t{acc[mi][ni][r0 + 2], acc[mi][ni][r0 + 3]} + f32x2_t{bv[h].z, bv[h].w};
                                if (ACT == WU_ACT_LEAKY) {
                                    const f32x2_t s0_ = v0 * 0.2f, s1_ = v1 * 0.2f;
                                    v0 = f32x2_t{fmaxf(v0.x, s0_.x), fmaxf(v0.y, s0_.y)};
                                    v1 = f32x2_t{fmaxf(v1.x, s1_.x), fmaxf(v1.y, s1_.y)};
                                }
                                o[h][0] = pack_bf16x2(v0.x, v0.y);
                                o[h][1] = pack_bf16x2(v1.x, v1.y);
                                if (ACT == WU_ACT_RELU) { o[h][0] = s_relu_bf16x2(o[h][0]); o[h][1] = s_relu_bf16x2(o[h][1]); }
                            }
                            const auto s0 = __builtin_amdgcn_permlane32_swap(o[0][0], o[1][0], false, false);
                            const auto s1 = __builtin_amdgcn_permlane32_swap(o[0][1], o[1][1], false, false);
                            uint4 v = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                            if (EG == WU_ACT_RELU) {
                                v.x = s_relu_gate_bf16x2(v.x, yv.x); v.y = s_relu_gate_bf16x2(v.y, yv.y);
                                v.z = s_relu_gate_bf16x2(v.z, yv.z); v.w = s_relu_gate_bf16x2(v.w, yv.w);
                            } else if (EG == WU_ACT_LEAKY) {
                                v = gate16<bf16_t>(v, yv, WU_ACT_LEAKY);
                            }
                            if (oh < a.H && ow < a.W)
                                *(uint4*)(a.y + (img_pix + (size_t)(oh * a.W + ow)) * a.ldy + co0 + 8 * lh + 32 * ni + 8 * g) = v;
                            vr[r] = v;
                        }
                        if (POOL) {
                            uint4 m = make_uint4(s_max_u16x2(vr[0].x, vr[1].x), s_max_u16x2(vr[0].y, vr[1].y), s_max_u16x2(vr[0].z, vr[1].z), s_max_u16x2(vr[0].w, vr[1].w));
                            m.x = s_max_u16x2(m.x, (uint32_t)__builtin_amdgcn_mov_dpp((int)m.x, 0xB1, 0xF, 0xF, true));
                            m.y = s_max_u16x2(m.y, (uint32_t)__builtin_amdgcn_mov_dpp((int)m.y, 0xB1, 0xF, 0xF, true));
                            m.z = s_max_u16x2(m.z, (uint32_t)__builtin_amdgcn_mov_dpp((int)m.z, 0xB1, 0xF, 0xF, true));
                            m.w = s_max_u16x2(m.w, (uint32_t)__builtin_amdgcn_mov_dpp((int)m.w, 0xB1, 0xF, 0xF, true));
                            if ((l31 & 1) == 0 && ohe + 1 < a.H && ow + 1 < a.W) {
                                const int Hp = a.H >> 1, Wp = a.W >> 1;
                                *(uint4*)(a.pool + (((size_t)n * Hp + (ohe >> 1)) * Wp + (ow >> 1)) * a.ldpool + co0 + 8 * lh + 32 * ni + 8 * g) = m;
                            }
                        }
                    }
            }
        };
        using A0 = std::integral_constant<int, WU_ACT_NONE>;
        using A1 = std::integral_constant<int, WU_ACT_RELU>;
        using A2 = std::integral_constant<int, WU_ACT_LEAKY>;
        using NoPool = std::false_type;
        if (a.egate) {
            if (a.egate_act == WU_ACT_RELU) epi_store(A0{}, A1{}, NoPool{});
            else if (a.egate_act == WU_ACT_LEAKY) epi_store(A0{}, A2{}, NoPool{});
            else epi_store(A0{}, A0{}, NoPool{});
        } else if (a.act == WU_ACT_RELU) {
            if (a.pool) epi_store(A1{}, A0{}, std::true_type{});
            else epi_store(A1{}, A0{}, NoPool{});
        }
        else if (a.act == WU_ACT_LEAKY) epi_store(A2{}, A0{}, NoPool{});
        else epi_store(A0{}, A0{}, NoPool{});
        cur = advance(cur);
        if (tile + t_step < a.ntiles) {
            set_fetch_tile(cur);
            issue_chunk(0);
        }
    }
}

}  // namespace

bool conv_use_v2s(int Cin) {
    const int m = g_wu_opt[WU_OPT_CONV_V2S];
    return m == 1 || (m == 2 && Cin <= 128) || (m == 3 && Cin <= 64);
}

int conv_v2s_launch(const void* x, int ldx, const void* w, const float* bias, void* y, int ldy,
                    const void* egate, int ldegate, int egate_act, int N, int H, int W, int Cin, int Cout, int act, hipStream_t s,
                    void* pool, int ldpool) {
    VSArgs a;
    a.pool = (bf16_t*)pool; a.ldpool = ldpool;
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.bias = bias; a.y = (bf16_t*)y; a.egate = (const bf16_t*)egate;
    a.ldx = ldx; a.ldy = ldy; a.ldegate = ldegate; a.egate_act = egate_act; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.act = act;
    a.tiles_x = cdiv(W, KS::TW); a.tiles_y = cdiv(H, KS::TH); a.cout_tiles = Cout / 64;
    const long long ntiles = (long long)N * a.tiles_x * a.tiles_y * a.cout_tiles;
    if (ntiles >= (1ll << 31)) return -1;
    a.ntiles = (int)ntiles;
    const int slots = 2 * wu_num_cus();                     // two resident workgroups per CU
    const long long grid = ntiles < slots ? ntiles : slots;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv3x3_mfma_v2s_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, KS::BUF);
        attr_set = true;
    }
    hipLaunchKernelGGL(conv3x3_mfma_v2s_kernel, dim3((int)grid), dim3(256), KS::BUF, s, a);
    return 0;
}
