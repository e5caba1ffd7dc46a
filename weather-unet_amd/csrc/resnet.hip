// Kernels of the frozen ResNet-101 estimator that the reference's GAN loop calls four times per iteration and differentiates
// once with respect to its input (t_cls_train.py:237,247-250,297,424; the model is torchvision.models.resnet101, built at
// classifier.py:106-112 / estimator.py:143-151, eval mode, BatchNorm therefore a per-channel affine that is folded into the
// conv weights and bias by the host):
//   * conv1x1_mfma_kernel   the pointwise convs of the Bottleneck blocks (2/3 of the network's FLOPs) as a plain GEMM on the
//                           matrix cores: rows = pixels, bias + residual add + ReLU (+ the ReLU gate of the data-gradient
//                           pass) in a direct register epilogue; stride-2 gather (downsample conv) and stride-2 scatter (its
//                           data gradient) are row-address maps;
//   * stem7x7_*             the 7x7 stride-2 stem conv (K = 147) straight from the NCHW fp32 image, and its data gradient;
//   * maxpool3s2_*          the 3x3 stride-2 max-pool with stored arg-max (windows overlap: backward is a gather over the
//                           <= 4 windows that contain a pixel).
// The 3x3 convs of the blocks run on the conv3x3 kernels of this library (conv3x3_mfma_v2.hip / conv3x3_mfma.hip).
#include <algorithm>
#include <type_traits>

#include "wu_common.h"

namespace {

__host__ __device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }

// =================================================================================================
// 1x1 conv = GEMM   y[m][co] = act(sum_ci x[row(m)][ci] * w[co][ci] + bias[co] + residual[m][co]) [* act'(egate[m][co])]
// =================================================================================================
constexpr int kTN = 64;         // output channels per workgroup
constexpr int kKB = 128;        // bytes of K staged per step and row (64 bf16 / 32 fp32 channels)

struct PwArgs {
    const void* x; const void* w; const float* bias; const void* res; void* y; const void* egate;
    int ldx, ldres, ldy, ldegate;
    int N, Hc, Wc, in_stride, Hin, Win, out_stride, Hout, Wout, Cin, Cout, act, egate_act;
    long long M;                // N * Hc * Wc
    int n_tiles;                // Cout / 64
};

template <typename T> struct PwMma;
template <> struct PwMma<bf16_t> {
    static __device__ __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
};
template <> struct PwMma<float> {      // exact fp32: four 32x32x2 steps per 16-byte fragment pair (same k permutation on both operands)
    static __device__ __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
};

// four consecutive channels with ONE load (8 bytes of bf16 / 16 bytes of fp32)
template <typename T> __device__ __forceinline__ void load4(const T* p, float* o);
template <> __device__ __forceinline__ void load4<float>(const float* p, float* o) {
    const float4 v = *(const float4*)p;
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float* o) {
    const uint2 v = *(const uint2*)p;
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
}

// LDS image: 128-byte rows, the eight 16-byte slots of a row XOR-swizzled with ((row >> 1) & 7).  A ds_read_b128 is served in four
// groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32) over 64 banks, i.e. 16 sixteen-byte columns of which
// a 128-byte row covers 8 (even rows the lower, odd rows the upper half): the 8 even and the 8 odd rows of a group must land in 8
// different slots.  (row & 7), the round-2 swizzle, puts rows 12 and 20, 13 and 21 ... of a group in the same slot: 2-way conflicts on
// half the fragment reads (SQ_LDS_BANK_CONFLICT = 31 % of SQ_LDS_IDX_ACTIVE on every shape, scratch/pmc_pw.sh); (row >> 1) & 7 is
// distinct over each group's rows of one parity.
__device__ __forceinline__ int pw_off(int row, int slot) { return row * kKB + ((slot ^ ((row >> 1) & 7)) << 4); }

// TM = pixels per workgroup: 256 (two 40-KiB LDS stages, 2 workgroups per CU), 128 (24 KiB, 3 per CU) or 64 (16 KiB, 4-5 per CU: the default --
// shorter chains of exposed round trips per workgroup and more of them in flight per CU)
template <typename T, int TM>
__global__ __launch_bounds__(256, TM == 256 ? 2 : (TM == 128 ? 3 : 4)) void conv1x1_mfma_kernel(const PwArgs a) {
    // waves: WM along the pixels x WN along the couts (64 rows: 2 x 2, each wave one 32 x 32 block)
    constexpr int WM = TM >= 128 ? 4 : 2, WN = 4 / WM, NI = 2 / WN;
    constexpr int MI = TM / (32 * WM), NA = TM / 32;   // 32-row MFMA blocks per wave; 16-byte activation items per thread and K step
    extern __shared__ __attribute__((aligned(16))) char smem[];                // two stages of 40 KiB: two workgroups per CU
    constexpr int E16 = 16 / (int)sizeof(T);                                    // elements per 16-byte slot
    constexpr int KE = kKB / (int)sizeof(T);                                    // channels per K step
    constexpr int kStage = (TM + kTN) * kKB;
    const int tid = threadIdx.x, lane = tid & 63, wave_ = tid >> 6;
    const int wave = wave_ % WM, wn = wave_ / WM;          // pixel block / cout block of this wave
    const int l31 = lane & 31, lh = lane >> 5;

    // cout tile fastest: the workgroups that share one pixel tile are neighbours in launch order (same XCD after the remap)
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int ct = bid % a.n_tiles;
    const long long m0 = (long long)(bid / a.n_tiles) * TM;
    const int co0 = ct * kTN;

    auto in_pixel = [&](long long m) __attribute__((always_inline)) -> long long {   // GEMM row -> pixel index of x (or -1)
        if (m >= a.M) return -1;
        if (a.in_stride == 1) return m;
        const int wc = (int)(m % a.Wc);
        const long long t = m / a.Wc;
        const int hc = (int)(t % a.Hc);
        const long long n = t / a.Hc;
        return (n * a.Hin + (long long)hc * a.in_stride) * a.Win + (long long)wc * a.in_stride;
    };

    // ---- staging: 8 activation items + 2 weight items of 16 B per thread and K step ----
    const int srow = tid >> 3, sslot = tid & 7;
    const T* asrc[NA];
    bool aok[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const long long pix = in_pixel(m0 + srow + 32 * k);
        aok[k] = pix >= 0;
        asrc[k] = (const T*)a.x + (aok[k] ? pix : 0) * a.ldx + sslot * E16;      // clamped: loads are unconditional, zeroed afterwards
    }
    const T* wsrc[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) wsrc[k] = (const T*)a.w + (size_t)(co0 + srow + 32 * k) * a.Cin + sslot * E16;

    // TWO register sets: the global loads of step c + 2 are issued before step c is computed and are still in flight when step c + 1's
    // are written to LDS (counted waits: the sets are indexed at compile time by an unrolled-by-two loop).  These GEMMs are 1-16 K steps
    // long with one or two workgroups per CU: one step ahead exposed a full memory round trip (~1 us) per K step.
    uint4 areg[2][NA], w00, w01, w10, w11;      // the weight registers are named, not an array: hipcc parks a small array of uint4 in scratch here
    auto load_step = [&](auto SET, int c0) __attribute__((always_inline)) {
        constexpr int S = decltype(SET)::value;
#pragma unroll
        for (int k = 0; k < NA; ++k) areg[S][k] = *(const uint4*)(asrc[k] + c0);
        if constexpr (S == 0) { w00 = *(const uint4*)(wsrc[0] + c0); w01 = *(const uint4*)(wsrc[1] + c0); }
        else { w10 = *(const uint4*)(wsrc[0] + c0); w11 = *(const uint4*)(wsrc[1] + c0); }
    };
    auto store_step = [&](auto SET, int stage) __attribute__((always_inline)) {
        constexpr int S = decltype(SET)::value;
        char* a_lds = smem + stage * kStage;
        char* w_lds = a_lds + TM * kKB;
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            // rows past the end of the GEMM are zeroed with a lane mask (a 128-bit select is lowered through scratch memory)
            const uint32_t m = aok[k] ? 0xffffffffu : 0u;
            *(uint4*)(a_lds + pw_off(srow + 32 * k, sslot)) = make_uint4(areg[S][k].x & m, areg[S][k].y & m, areg[S][k].z & m, areg[S][k].w & m);
        }
        *(uint4*)(w_lds + pw_off(srow, sslot)) = S == 0 ? w00 : w10;
        *(uint4*)(w_lds + pw_off(srow + 32, sslot)) = S == 0 ? w01 : w11;
    };

    // accumulators TRANSPOSED (weights are the MFMA A operand): a lane owns 4 consecutive channels of one pixel per register quad
    f32x16_t acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

    // K loop, two LDS stages + two register sets: step c is computed from stage c & 1 while step c + 1 sits in registers (written to
    // the OTHER stage after the compute: everyone left it when it passed the barrier that published stage c & 1) and step c + 2 is on
    // its way from memory
    const int nsteps = a.Cin / KE;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    auto step = [&](auto CUR, auto NXT, auto LOAD, int c) __attribute__((always_inline)) {
        // CUR: the register set step c came through (free again), NXT: the set holding step c + 1.  LOAD (compile time): step c + 2
        // exists and is requested now -- unconditionally, so that the number of loads in flight at the LDS store below is a constant
        // the compiler can count (behind a run-time `if` it waited for ALL loads there: one step in flight again)
        if constexpr (decltype(LOAD)::value) load_step(CUR, (c + 2) * KE);
        const char* a_lds = smem + (c & 1) * kStage;
        const char* w_lds = a_lds + TM * kKB;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 af[MI], bf[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) af[mi] = *(const uint4*)(a_lds + pw_off(32 * MI * wave + 32 * mi + l31, 2 * ks + lh));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) bf[ni] = *(const uint4*)(w_lds + pw_off(32 * (NI * wn + ni) + l31, 2 * ks + lh));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) PwMma<T>::run(acc[mi][ni], bf[ni], af[mi]);      // D^T = W * X^T
        }
        if (decltype(LOAD)::value || c + 1 < nsteps) {
            store_step(NXT, (c + 1) & 1);
            __syncthreads();
        }
    };
    using YES = std::true_type;
    using NO = std::false_type;
    load_step(S0{}, 0);
    if (nsteps > 1) load_step(S1{}, KE);
    store_step(S0{}, 0);
    __syncthreads();
    int c = 0;
    for (; c + 3 < nsteps; c += 2) {            // both steps of a pair have a step two ahead of them
        step(S0{}, S1{}, YES{}, c);
        step(S1{}, S0{}, YES{}, c + 1);
    }
    if (c + 2 < nsteps) {                       // three steps left
        step(S0{}, S1{}, YES{}, c);
        step(S1{}, S0{}, NO{}, c + 1);
        step(S0{}, S1{}, NO{}, c + 2);
    } else {                                    // one or two
        step(S0{}, S1{}, NO{}, c);
        if (c + 1 < nsteps) step(S1{}, S0{}, NO{}, c + 1);
    }

    // ---- direct register epilogue: bias + residual in fp32, activation, gate, 8-byte (bf16) / 16-byte (fp32) stores ----
    const int os = a.out_stride;
    if constexpr (std::is_same<T, bf16_t>::value) {
        // Fast path (round 3), bf16 and unit output stride -- every pointwise conv of the forward pass: the fp32 sums of two
        // adjacent 4-channel groups are paired across the half-waves with v_permlane32_swap BEFORE residual / activation / gate, so a
        // lane owns 8 consecutive channels of its pixel: residual and gate arrive as ONE 16-byte load each (requested up front,
        // one exposed round trip per tile), the result leaves as one 16-byte store -- half the memory instructions of the general
        // path below, same arithmetic in the same order (bias, + residual, activation, gate; one rounding at the end).
        if (os == 1) {
            uint4 rq[MI][NI][2], eq[MI][NI][2];
            bool okm[MI];
            const T* rp[MI]; const T* ep[MI]; T* yp[MI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const long long m = m0 + 32 * MI * wave + 32 * mi + l31;
                okm[mi] = m < a.M;
                const long long p = okm[mi] ? m : 0;
                rp[mi] = a.res ? (const T*)a.res + p * a.ldres + co0 + 8 * lh : nullptr;
                ep[mi] = a.egate ? (const T*)a.egate + p * a.ldegate + co0 + 8 * lh : nullptr;
                yp[mi] = (T*)a.y + p * a.ldy + co0 + 8 * lh;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        if (a.res) rq[mi][ni][gp] = *(const uint4*)(rp[mi] + 32 * (NI * wn + ni) + 16 * gp);
                        if (a.egate) eq[mi][ni][gp] = *(const uint4*)(ep[mi] + 32 * (NI * wn + ni) + 16 * gp);
                    }
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        float lo[4], hi[4];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int g = 2 * gp + e;
                            const float4 bv = a.bias ? *(const float4*)(a.bias + co0 + 32 * (NI * wn + ni) + 8 * g + 4 * lh) : make_float4(0.f, 0.f, 0.f, 0.f);
                            float* d = e ? hi : lo;
                            d[0] = acc[mi][ni][4 * g + 0] + bv.x; d[1] = acc[mi][ni][4 * g + 1] + bv.y;
                            d[2] = acc[mi][ni][4 * g + 2] + bv.z; d[3] = acc[mi][ni][4 * g + 3] + bv.w;
                        }
                        float o[8];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            // lanes 0-31 keep group 2 gp and receive its upper four channels from lanes 32-63; lanes 32-63 the mirror
                            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo[i]), __float_as_uint(hi[i]), false, false);
                            o[i] = __uint_as_float(sw[0]);
                            o[4 + i] = __uint_as_float(sw[1]);
                        }
                        if (a.res) {
                            float rv[8];
                            unpack16<T>(rq[mi][ni][gp], rv);
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] += rv[e];
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = act_apply(o[e], a.act);
                        if (a.egate) {
                            float ev[8];
                            unpack16<T>(eq[mi][ni][gp], ev);
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = act_gate(o[e], ev[e], a.egate_act);
                        }
                        if (okm[mi]) *(uint4*)(yp[mi] + 32 * (NI * wn + ni) + 16 * gp) = pack16<T>(o);
                    }
            return;
        }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const long long m = m0 + 32 * MI * wave + 32 * mi + l31;
        if (m >= a.M) continue;
        long long opix = m;
        int hc = 0, wc = 0;
        if (os != 1) {
            wc = (int)(m % a.Wc);
            const long long t = m / a.Wc;
            hc = (int)(t % a.Hc);
            const long long n = t / a.Hc;
            opix = (n * a.Hout + (long long)hc * os) * a.Wout + (long long)wc * os;
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = co0 + 32 * (NI * wn + ni) + 8 * g + 4 * lh;
                float v[4];
                const float4 bv = a.bias ? *(const float4*)(a.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
                v[0] = acc[mi][ni][4 * g + 0] + bv.x; v[1] = acc[mi][ni][4 * g + 1] + bv.y;
                v[2] = acc[mi][ni][4 * g + 2] + bv.z; v[3] = acc[mi][ni][4 * g + 3] + bv.w;
                // the pixel itself, then (stride-2 scatter) the other pixels of its block, which receive residual-or-zero
                for (int dy = 0; dy < os; ++dy)
                    for (int dx = 0; dx < os; ++dx) {
                        if (os != 1 && (hc * os + dy >= a.Hout || wc * os + dx >= a.Wout)) continue;
                        const long long p = opix + (long long)dy * a.Wout + dx;
                        float o[4];
                        const bool own = dy == 0 && dx == 0;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = own ? v[e] : 0.f;
                        if (a.res) {
                            float rv[4];
                            load4<T>((const T*)a.res + p * a.ldres + co, rv);
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] += rv[e];
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = act_apply(o[e], a.act);
                        if (a.egate) {
                            float ev[4];
                            load4<T>((const T*)a.egate + p * a.ldegate + co, ev);
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = act_gate(o[e], ev[e], a.egate_act);
                        }
                        T* yp = (T*)a.y + p * a.ldy + co;
                        if constexpr (std::is_same<T, float>::value) {
                            *(float4*)yp = make_float4(o[0], o[1], o[2], o[3]);
                        } else {
                            *(uint2*)yp = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
                        }
                    }
            }
    }
}

// =================================================================================================
// Persistent LDS-DMA form of the pointwise GEMM (round 4; bf16, unit strides, Cout % 128 == 0): conv1x1_pw3_kernel
// =================================================================================================
// Measured on the estimator's shapes (scratch/pw_scaling.py, scratch/pw_vs_blas.py, launches timed inside a captured graph): the 64 x 64
// tiles above move 1 byte from L2 per 32 flops -- the 64 B/clk L2 -> CU port alone caps them at half the matrix rate, the LDS fragment
// traffic (two 1-KiB fragments per MFMA) at the same half -- and every workgroup is a chain of exposed round trips (first loads, K steps
// two ahead, residual, store).  This form: 128 pixels x 128 couts per workgroup (four waves, 64 x 64 each: one fragment per MFMA, 1 byte
// per 64 flops), operands by LDS-DMA through buffer descriptors into a ring of D 32-KiB stages (rows past the end of the GEMM are lanes
// out of the descriptor's range: zeros), PERSISTENT workgroups whose prefetch runs across tile boundaries, and every vector-memory
// operation of the loop (DMA pieces, residual / gate loads, output stores) issued from inline asm so that the waits are COUNTED by the
// kernel (s_waitcnt vmcnt(N), N = operations younger than the stage about to be read: vmcnt retires in issue order) -- the compiler
// never sees an outstanding load it would drain the ring for.  Same MFMA operand mapping, K order and epilogue arithmetic as
// conv1x1_mfma_kernel: bit-identical results (tests/test_gpu_round4.py).
namespace pw3 {
constexpr int TM = 128;                          // pixel rows per tile; cout columns per tile TN = 128 or 64 (template parameter)
constexpr int kMaxCout = 2048;                   // bias image in LDS
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

struct Args {
    const bf16_t* x; const bf16_t* w; const float* bias; const bf16_t* res; const bf16_t* egate; bf16_t* y;
    int ldx, ldres, ldegate, ldy;
    int M, Cin, Cout, act, egate_act;
    int n_ct, items;            // cout tiles; pixel tiles x cout tiles x classes
    // CONV instances.  Rows = the Ho x Wo grid of every image (M = N Ho Wo); tap t of class c reads source pixel (s oy + dy - 1, s ox + dx - 1) of the
    // H x W image and multiplies by slab `slab` of the [9][Cout][Cin] pack; row (n, oy, ox) lands at (oy osy + ooy[c], ox osx + oox[c]) of the OH x OW
    // output image (skipped outside).  Forward conv: one class of nine taps, identity output map.  Data gradient of a stride-2 conv: four parity
    // classes of 4 / 2 / 2 / 1 taps over the dY grid (conv_internal.h).  cls_tap packs (dy, dx, slab) as dy | dx << 2 | slab << 4.
    int H, W, Ho, Wo, stride;
    int ncls, cls_ntaps[4], cls_ooy[4], cls_oox[4];
    unsigned cls_tap[4][9];
    int osy, osx, OH, OW;
};

// Vector-memory operations from inline asm.  The hazard recogniser does not see into asm: an SGPR operand (descriptor, scalar offset) that the
// compiler has just restored from a spill lane with v_readlane_b32 -- it does in the register-hungry instances -- must be 5 wait states old before
// a VMEM instruction reads it ("VALU writes SGPR -> VMEM reads that SGPR"; seen as a wrong first load of a group, the later ones correct), and the
// data registers of a 128-bit store may not be rewritten in the next cycle.  Every block therefore carries its own s_nop.
// PAD: the block carries its own 5 wait states (the CONV instances, where descriptors do get spilled); the pointwise instances keep every descriptor
// in SGPRs and go without -- tests/test_isa_cpu.py checks the shipped machine code for the hazard either way.
template <int OFF, bool PAD> __device__ __forceinline__ u32x4_t ld16(unsigned voff, wu_rsrc_t rs, unsigned soff) {
    u32x4_t v;
    if constexpr (PAD) asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(v) : "v"(voff), "s"(rs), "s"(soff), "n"(OFF) : "memory");
    else asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(v) : "v"(voff), "s"(rs), "s"(soff), "n"(OFF) : "memory");
    return v;
}
template <int OFF, bool PAD> __device__ __forceinline__ void st16(u32x4_t v, unsigned voff, wu_rsrc_t rs, unsigned soff) {
    if constexpr (PAD) asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen offset:%4\n\ts_nop 1" :: "v"(v), "v"(voff), "s"(rs), "s"(soff), "n"(OFF) : "memory");
    else asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen offset:%4\n\ts_nop 1" :: "v"(v), "v"(voff), "s"(rs), "s"(soff), "n"(OFF) : "memory");
}
// wu_dma16b, optionally with the same margin (s_nop 2 + s_mov + s_nop 0 = 5 wait states in front of the load)
template <bool PAD> __device__ __forceinline__ void dma16(unsigned voff, wu_rsrc_t rsrc, unsigned soff, unsigned lds_byte_addr) {
    if constexpr (PAD) asm volatile("s_nop 2\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                                    :: "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_byte_addr) : "memory", "m0");
    else asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                      :: "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_byte_addr) : "memory", "m0");
}
// at most n (even, wave-uniform) vector-memory operations still in flight; anything above 62 waits for 62 (waiting for more is safe)
__device__ __forceinline__ void vm_wait(int n) {
    switch (n >> 1) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
        case 17: asm volatile("s_waitcnt vmcnt(34)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
        case 19: asm volatile("s_waitcnt vmcnt(38)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
        case 21: asm volatile("s_waitcnt vmcnt(42)" ::: "memory"); break;
        case 22: asm volatile("s_waitcnt vmcnt(44)" ::: "memory"); break;
        case 23: asm volatile("s_waitcnt vmcnt(46)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
        case 25: asm volatile("s_waitcnt vmcnt(50)" ::: "memory"); break;
        case 26: asm volatile("s_waitcnt vmcnt(52)" ::: "memory"); break;
        case 27: asm volatile("s_waitcnt vmcnt(54)" ::: "memory"); break;
        case 28: asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); break;
        case 29: asm volatile("s_waitcnt vmcnt(58)" ::: "memory"); break;
        case 30: asm volatile("s_waitcnt vmcnt(60)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(62)" ::: "memory"); break;
    }
}
}  // namespace pw3

// NW waves x TN couts: 8 x 128 (waves 4 x 2, 32 pixels x 64 couts each: two waves per SIMD per workgroup), 4 x 128 (2 x 2, 64 x 64 each), 4 x 64 (4 x 1,
// 32 x 64 each: the 64-cout layers).
// CONV (round 4): the same pipeline as a 3 x 3 conv -- K runs over (64-channel chunk, tap of the tile's class), the weight rows of a step are slab
// `slab` of the [9][Cout][Cin] pack, and its activation rows are GATHERED: row r of the tile is grid pixel (n, oy, ox), the DMA lane that fetches it reads
// source pixel (n, s oy + dy - 1, s ox + dx - 1) or, outside the image, is pushed out of the descriptor's range (zeros).  Every tap re-reads its rows from
// L2 (no halo reuse): 2.25x the input at stride 2, where it replaces conv3x3_mfma_kernel<T, 2> (one workgroup per CU, register staging: 290-510 TFLOP/s),
// and the four parity classes of that conv's data gradient (Args), whose outputs are scattered to their sites by per-lane store offsets.
template <int NW, int TN, int D, bool HAS_RES, bool HAS_GATE, bool CONV = false>
__global__ __launch_bounds__(NW * 64, D == 2 ? 2 : 1) void conv1x1_pw3_kernel(const pw3::Args a) {
    using namespace pw3;
    constexpr int WN = TN / 64, WM = NW / WN, MI = TM / (32 * WM), NI = 2;
    static_assert(TN % 64 == 0 && NW % WN == 0 && MI >= 1 && TM % (32 * WM) == 0, "wave grid");
    constexpr int kStage = (TM + TN) * kKB;      // activation rows, then weight rows
    constexpr int PA = TM / 8 / NW;      // 1-KiB pieces of the activation rows per wave and K step
    constexpr int PW = TN / 8 / NW;      // ... of the weight rows
    static_assert(PA >= 1 && PW >= 1, "every wave moves at least one piece of each operand");
    constexpr int PD = PA + PW;          // DMA operations per wave and K step
    constexpr int PS = MI * NI * 2;      // output stores (residual loads, gate loads) per wave and tile
    static_assert(PD % 2 == 0 && PS % 4 == 0, "vm_wait counts in units of 2");
    constexpr int NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* bias_lds = (float*)(smem + D * kStage);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, lh = lane >> 5;
    const int G = (int)gridDim.x;
    const int wg = xcd_remap(blockIdx.x, G);
    const int nchunks = a.Cin / 64;
    const int ncls = CONV ? a.ncls : 1;

    for (int i = tid; i < a.Cout; i += NT) bias_lds[i] = a.bias ? a.bias[i] : 0.f;
    __syncthreads();

    // ---- descriptors and tile-invariant per-lane byte offsets ----
    // (CONV: the base is shifted back by one row + one pixel so that a tap's offset (dy W + dx) is never negative; may point before the tensor,
    //  never dereferenced there -- such lanes are out of the image and pushed out of range)
    const size_t a_shift = CONV ? ((size_t)a.W + 1) * a.ldx * 2 : 0;
    const size_t a_rows = CONV ? (size_t)(a.M / (a.Ho * a.Wo)) * a.H * a.W : (size_t)a.M;
    const size_t y_rows = CONV ? (size_t)(a.M / (a.Ho * a.Wo)) * a.OH * a.OW : (size_t)a.M;
    const wu_rsrc_t rsA = wu_make_rsrc((const char*)a.x - a_shift, (unsigned)(a_shift + ((a_rows - 1) * a.ldx + a.Cin) * 2));
    const wu_rsrc_t rsW = wu_make_rsrc(a.w, (unsigned)((size_t)(CONV ? 9 : 1) * a.Cout * a.Cin * 2));
    const wu_rsrc_t rsY = wu_make_rsrc(a.y, (unsigned)(((y_rows - 1) * a.ldy + a.Cout) * 2));
    const wu_rsrc_t rsR = wu_make_rsrc(HAS_RES ? a.res : a.y, HAS_RES ? (unsigned)(((y_rows - 1) * a.ldres + a.Cout) * 2) : 0u);
    const wu_rsrc_t rsE = wu_make_rsrc(HAS_GATE ? a.egate : a.y, HAS_GATE ? (unsigned)(((y_rows - 1) * a.ldegate + a.Cout) * 2) : 0u);
    unsigned voA[PA], voW[PW];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int row = 8 * (wave + NW * j) + (lane >> 3);                 // piece wave + NW j holds rows 8 p .. 8 p + 7, lane l lands at + 16 l
        const int sl = (lane & 7) ^ ((row >> 1) & 7);                      // the swizzle of pw_off, applied to the SOURCE slot
        voA[j] = (unsigned)(row * a.ldx * 2 + sl * 16);
    }
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        const int row = 8 * (wave + NW * j) + (lane >> 3);
        const int sl = (lane & 7) ^ ((row >> 1) & 7);
        voW[j] = (unsigned)(row * a.Cin * 2 + sl * 16);
    }
    // epilogue: pixel rows 32 MI wm + 32 mi + l31, channels 64 wn + 32 ni + 16 gp + 8 lh (+ 8) -- (ni, gp) travel in the instruction offset
    // (CONV: recomputed per tile -- the tile's rows are scattered to their output sites, rows without one are pushed out of range)
    unsigned voY[MI], voR[MI], voE[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int r = 32 * MI * wm + 32 * mi + l31, c = 64 * wn + 8 * lh;
        voY[mi] = (unsigned)((r * a.ldy + c) * 2);
        voR[mi] = (unsigned)((r * a.ldres + c) * 2);
        voE[mi] = (unsigned)((r * a.ldegate + c) * 2);
    }
    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    // ---- the two cursors over this workgroup's items (wg, wg + G, ...; digits: class fastest, then cout tile, then pixel tile), carried digit by digit ----
    struct Cur { int cls, ct, pt; };
    const int dcl = G % ncls, g1 = G / ncls, dct = g1 % a.n_ct, dpt = g1 / a.n_ct;
    auto decode = [&](int item) __attribute__((always_inline)) {
        Cur c; c.cls = item % ncls; const int r = item / ncls; c.ct = r % a.n_ct; c.pt = r / a.n_ct; return c;
    };
    auto advance = [&](Cur c) __attribute__((always_inline)) {
        c.cls += dcl; int cy = c.cls >= ncls ? 1 : 0; c.cls -= cy * ncls;
        c.ct += dct + cy; cy = c.ct >= a.n_ct ? 1 : 0; c.ct -= cy * a.n_ct;
        c.pt += dpt + cy;
        return c;
    };
    int f_item = wg, f_k = 0, f_tap = 0;     // (CONV: f_k counts chunks, f_tap the tap inside the chunk)
    Cur fc = decode(wg);
    bool f_alive = f_item < a.items;
    int f_ntaps = CONV ? a.cls_ntaps[fc.cls] : 1;
    unsigned baseA[PA], vbits[PA];       // CONV: byte offset of the fetch tile's rows at tap (0, 0) in the shifted descriptor; bit t = tap t is inside the image
    auto conv_tile = [&]() __attribute__((always_inline)) {
        const int hw = a.Ho * a.Wo;
        f_ntaps = a.cls_ntaps[fc.cls];
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const int row = 8 * (wave + NW * j) + (lane >> 3);
            const int sl = (lane & 7) ^ ((row >> 1) & 7);
            const int p = fc.pt * TM + row;
            const int n = p / hw, rem = p - n * hw, oy = rem / a.Wo, ox = rem - oy * a.Wo;
            const int iy0 = oy * a.stride, ix0 = ox * a.stride;
            baseA[j] = (unsigned)((((n * a.H + iy0) * a.W + ix0) * a.ldx) * 2 + sl * 16);
            unsigned b = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const unsigned tw = a.cls_tap[fc.cls][t];
                const int dy = tw & 3, dx = (tw >> 2) & 3;
                const bool ok = t < f_ntaps && p < a.M && (unsigned)(iy0 + dy - 1) < (unsigned)a.H && (unsigned)(ix0 + dx - 1) < (unsigned)a.W;
                b |= (ok ? 1u : 0u) << t;
            }
            vbits[j] = b;
        }
    };
    if (CONV && f_alive) conv_tile();
    auto fetch = [&](int stage) __attribute__((always_inline)) {
        const unsigned kill = f_alive ? 0u : kWuOOB;
        const unsigned lds = smem_base + stage * kStage + wave * 1024;
        if constexpr (CONV) {
            const unsigned tw = a.cls_tap[fc.cls][f_tap];
            const int dy = tw & 3, dx = (tw >> 2) & 3, slab = tw >> 4;
            const unsigned soA = (unsigned)(((dy * a.W + dx) * a.ldx + f_k * 64) * 2) | kill;
            const unsigned soW = (unsigned)(((slab * a.Cout + fc.ct * TN) * a.Cin + f_k * 64) * 2) | kill;
#pragma unroll
            for (int j = 0; j < PA; ++j)
                dma16<CONV>(((vbits[j] >> f_tap) & 1u) ? baseA[j] : kWuOOB, rsA, soA, __builtin_amdgcn_readfirstlane(lds + j * NW * 1024));
#pragma unroll
            for (int j = 0; j < PW; ++j) dma16<CONV>(voW[j], rsW, soW, __builtin_amdgcn_readfirstlane(lds + TM * kKB + j * NW * 1024));
            if (f_alive && ++f_tap == f_ntaps) {
                f_tap = 0;
                if (++f_k == nchunks) {
                    f_k = 0; f_item += G; fc = advance(fc);
                    f_alive = f_item < a.items;
                    if (f_alive) conv_tile();
                }
            }
        } else {
            const unsigned soA = (unsigned)((fc.pt * TM * a.ldx + f_k * 64) * 2) | kill;
            const unsigned soW = (unsigned)((fc.ct * TN * a.Cin + f_k * 64) * 2) | kill;
#pragma unroll
            for (int j = 0; j < PA; ++j) dma16<CONV>(voA[j], rsA, soA, __builtin_amdgcn_readfirstlane(lds + j * NW * 1024));
#pragma unroll
            for (int j = 0; j < PW; ++j) dma16<CONV>(voW[j], rsW, soW, __builtin_amdgcn_readfirstlane(lds + TM * kKB + j * NW * 1024));
            if (f_alive && ++f_k == nchunks) {
                f_k = 0; f_item += G; fc = advance(fc);
                f_alive = f_item < a.items;
            }
        }
    };

    int st_f = 0, st_c = 0;
#pragma unroll
    for (int i = 0; i < D - 1; ++i) { fetch(st_f); st_f = st_f + 1 == D ? 0 : st_f + 1; }

    // operations issued in the last D - 1 steps besides the PD DMA pieces: [j] = step s - 1 - j; pre = before that step's DMA (residual / gate
    // loads), post = after it (output stores)
    int h_pre[D - 1], h_post[D - 1];
#pragma unroll
    for (int j = 0; j < D - 1; ++j) h_pre[j] = h_post[j] = 0;
    constexpr int NRG = PS * ((HAS_RES ? 1 : 0) + (HAS_GATE ? 1 : 0));
    const bool relu = a.act == WU_ACT_RELU;

    Cur cc = decode(wg);
    for (int c_item = wg; c_item < a.items; c_item += G) {
        const int m0 = cc.pt * TM, co0 = cc.ct * TN;
        const int nk = CONV ? a.cls_ntaps[cc.cls] * nchunks : nchunks;
        unsigned soY = (unsigned)((m0 * a.ldy + co0) * 2), soR = (unsigned)((m0 * a.ldres + co0) * 2), soE = (unsigned)((m0 * a.ldegate + co0) * 2);
        if constexpr (CONV) {            // where this tile's rows land: (n, oy osy + ooy, ox osx + oox) of the OH x OW image, or nowhere
            const int hw = a.Ho * a.Wo, ooy = a.cls_ooy[cc.cls], oox = a.cls_oox[cc.cls];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int p = m0 + 32 * MI * wm + 32 * mi + l31, c = 64 * wn + 8 * lh;
                const int n = p / hw, rem = p - n * hw, oy = rem / a.Wo, ox = rem - oy * a.Wo;
                const int py = oy * a.osy + ooy, px = ox * a.osx + oox;
                const bool ok = p < a.M && py < a.OH && px < a.OW;
                const int q = (n * a.OH + py) * a.OW + px;
                voY[mi] = ok ? (unsigned)((q * a.ldy + c) * 2) : kWuOOB;
                voE[mi] = ok ? (unsigned)((q * a.ldegate + c) * 2) : kWuOOB;
            }
            soY = (unsigned)(co0 * 2); soE = soY; soR = soY;
        }
        f32x16_t acc[MI][NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;
        u32x4_t rq[MI][4], eq[MI][4];
        for (int k = 0; k < nk; ++k) {
            // the stage about to be read has landed: everything younger than its PD pieces may stay in flight
            int n = (D - 2) * PD + h_post[D - 2];
#pragma unroll
            for (int j = 0; j < D - 2; ++j) n += h_pre[j] + h_post[j];
            vm_wait(n);
            __syncthreads();                 // ... for every wave; and every wave has left the stage the DMA below refills
            int pre = 0;
            if (NRG != 0 && k == 0) {        // this tile's residual / gate rows: in flight under its whole K loop
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    if constexpr (HAS_RES) {
                        rq[mi][0] = ld16<0, CONV>(voR[mi], rsR, soR); rq[mi][1] = ld16<32, CONV>(voR[mi], rsR, soR);
                        rq[mi][2] = ld16<64, CONV>(voR[mi], rsR, soR); rq[mi][3] = ld16<96, CONV>(voR[mi], rsR, soR);
                    }
                    if constexpr (HAS_GATE) {
                        eq[mi][0] = ld16<0, CONV>(voE[mi], rsE, soE); eq[mi][1] = ld16<32, CONV>(voE[mi], rsE, soE);
                        eq[mi][2] = ld16<64, CONV>(voE[mi], rsE, soE); eq[mi][3] = ld16<96, CONV>(voE[mi], rsE, soE);
                    }
                }
                pre = NRG;
            }
            fetch(st_f); st_f = st_f + 1 == D ? 0 : st_f + 1;
            {
                // fragments of k-step ks + 1 are requested before the MFMAs of ks (two register sets): the LDS round trip of a step's
                // first fragments is the only one a wave waits out
                const char* a_lds = smem + st_c * kStage;
                const char* w_lds = a_lds + TM * kKB;
                uint4 af[2][MI], bf[2][NI];
                auto frags = [&](int set, int ks) __attribute__((always_inline)) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) af[set][mi] = *(const uint4*)(a_lds + pw_off(32 * MI * wm + 32 * mi + l31, 2 * ks + lh));
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) bf[set][ni] = *(const uint4*)(w_lds + pw_off(64 * wn + 32 * ni + l31, 2 * ks + lh));
                };
                frags(0, 0);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if (ks < 3) frags((ks + 1) & 1, ks + 1);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) PwMma<bf16_t>::run(acc[mi][ni], bf[ks & 1][ni], af[ks & 1][mi]);      // D^T = W * X^T
                }
            }
            st_c = st_c + 1 == D ? 0 : st_c + 1;
#pragma unroll
            for (int j = D - 2; j > 0; --j) { h_pre[j] = h_pre[j - 1]; h_post[j] = h_post[j - 1]; }
            h_pre[0] = pre; h_post[0] = 0;
        }
        // ---- epilogue: conv1x1_mfma_kernel's fast path (bias, + residual, activation, gate; one rounding), 16-byte stores ----
        if (NRG != 0) {
            if (nk < D) vm_wait(PD * nk);    // short K: the loads are younger than every stage waited for so far (PD nk pieces issued since)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                if constexpr (HAS_RES) asm volatile("" : "+v"(rq[mi][0]), "+v"(rq[mi][1]), "+v"(rq[mi][2]), "+v"(rq[mi][3]));
                if constexpr (HAS_GATE) asm volatile("" : "+v"(eq[mi][0]), "+v"(eq[mi][1]), "+v"(eq[mi][2]), "+v"(eq[mi][3]));
            }
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    float lo[4], hi[4];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int g = 2 * gp + e;
                        const float4 bv = *(const float4*)(bias_lds + co0 + 64 * wn + 32 * ni + 8 * g + 4 * lh);
                        float* d = e ? hi : lo;
                        d[0] = acc[mi][ni][4 * g + 0] + bv.x; d[1] = acc[mi][ni][4 * g + 1] + bv.y;
                        d[2] = acc[mi][ni][4 * g + 2] + bv.z; d[3] = acc[mi][ni][4 * g + 3] + bv.w;
                    }
                    float o[8];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo[i]), __float_as_uint(hi[i]), false, false);
                        o[i] = __uint_as_float(sw[0]);
                        o[4 + i] = __uint_as_float(sw[1]);
                    }
                    if constexpr (HAS_RES) {
                        const u32x4_t q = rq[mi][2 * ni + gp];
                        float rv[8];
                        unpack16<bf16_t>(make_uint4(q.x, q.y, q.z, q.w), rv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] += rv[e];
                    }
                    if (relu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = fmaxf(o[e], 0.f);
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = act_apply(o[e], a.act);
                    }
                    if constexpr (HAS_GATE) {
                        const u32x4_t q = eq[mi][2 * ni + gp];
                        float ev[8];
                        unpack16<bf16_t>(make_uint4(q.x, q.y, q.z, q.w), ev);
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = act_gate(o[e], ev[e], a.egate_act);
                    }
                    const uint4 pk = pack16<bf16_t>(o);
                    u32x4_t pv; pv.x = pk.x; pv.y = pk.y; pv.z = pk.z; pv.w = pk.w;
                    if (ni == 0 && gp == 0) st16<0, CONV>(pv, voY[mi], rsY, soY);
                    else if (ni == 0) st16<32, CONV>(pv, voY[mi], rsY, soY);
                    else if (gp == 0) st16<64, CONV>(pv, voY[mi], rsY, soY);
                    else st16<96, CONV>(pv, voY[mi], rsY, soY);
                }
        h_post[0] = PS;
        cc = advance(cc);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the ring's trailing (all-zero) pieces must land before this LDS allocation is released
}

// =================================================================================================
// TWO chained 1x1 convs in one launch (round 4, bf16):
//     y1 = act1(wa . x  + bias_a + res) * act'(gate1)        x: [M][K1], wa: [C1][K1], y1: [M][C1]  (stored)
//     y2 = act2(wb . y1 + bias_b)       * act'(gate2)        wb: [C2][C1],             y2: [M][C2]  (stored)
// Forward of the Bottleneck chain (classifier.py:106-112 / torchvision Bottleneck): conv3 + bn3 + residual + ReLU of block i, then
// conv1 + bn1 + ReLU of block i + 1 -- the 1024-channel block output never comes back from memory for the next block's first conv.
// Backward, the mirror image: conv1^T of block i + 1 (+ identity-path gradient, ReLU gate of the block boundary) then conv3^T of block i
// (ReLU gate of its 3x3 conv's output).  The estimator launched 210 pointwise GEMMs per GAN iteration at 24-80 us each although each is
// 1-4 GFLOP: a dependent launch costs its drain / write-back / ramp (6-10 us) and one HBM round trip of its input; a pair here is one.
//
// One 512-thread workgroup (8 waves, two per SIMD) = 32 GEMM rows, all of C1 and C2.  Every weight element is used for 32 rows only, so
// the weight STREAM (64 B/clk per CU from L2; 1 MB per workgroup for 256 -> 1024 -> 256) bounds the kernel, and what it needs is bytes in
// flight: the weights are read as MFMA A FRAGMENTS straight into registers from a fragment-ordered pack ([cout / 32][k / 16][64 lanes]
// x 16 bytes, made once per frozen plan by the host: every wave-level load is 1 KiB contiguous -- the first version read row-major weights
// in fragment shape, 32 rows x 32 bytes per instruction, and ran 1.6x SLOWER than the two launches it replaces), 16 fragments per wave =
// 128 KiB per CU requested ahead of the MFMAs that use them.
// Stage 1: wave w owns couts {32 (w + 8 j)}; the 32 x K1 x-tile is its MFMA B operand, in REGISTERS (K1 / 16 fragments, loaded once).
// The finished y1 block goes to global AND, as bf16, to a swizzled LDS image [32][C1] (16-byte slot XOR (row & 15): the 32 rows of a
// fragment read hit 16 different slots per 16-lane group).  Stage 2 (after ONE barrier): wave w owns couts {32 (w + 8 j)} of C2, B
// fragments from the LDS image.  Accumulation order = the K order of conv1x1_mfma_kernel: bit-identical to the two separate launches.
constexpr int kChainWaves = 8;
// GATED = false: the forward pair (bias_a, res, bias_b present; no gates).  GATED = true: the backward pair (res, gate1, gate2 present; no
// bias).  Compile-time, and the prefetches below are UNCONDITIONAL (the last block re-requests itself): behind a run-time `if` the
// compiler cannot count the loads in flight and waits for all of them in front of the MFMAs (measured on the first version: vmcnt(1)
// right after the 16 prefetch loads).
template <int K1, bool GATED>
__global__ __launch_bounds__(64 * kChainWaves, 1) void conv1x1_chain_kernel(const bf16_t* __restrict__ x, int ldx, const uint4* __restrict__ wa,
        const float* __restrict__ bias_a, const bf16_t* __restrict__ res, int ldres, int act1, const bf16_t* __restrict__ gate1, int ldg1, int gate1_act,
        bf16_t* __restrict__ y1, int ldy1, const uint4* __restrict__ wb, const float* __restrict__ bias_b, int act2,
        const bf16_t* __restrict__ gate2, int ldg2, int gate2_act, bf16_t* __restrict__ y2, int ldy2, long long M, int C1, int C2) {
    constexpr int KS1 = K1 / 16;                               // MFMA K steps of stage 1
    constexpr int NW = kChainWaves;
    extern __shared__ __attribute__((aligned(16))) char smem[];    // y1 tile: [32][C1] bf16, swizzled
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const long long m0 = (long long)blockIdx.x * 32;
    const long long mrow = m0 + l31;
    const bool ok = mrow < M;
    const long long prow = ok ? mrow : M - 1;                  // clamped: loads unconditional, stores skipped
    const int pitch = C1 * 2;

    // ---- the x tile as B fragments: lane (row l31, k half lh) holds 8 consecutive k of K step ks ----
    uint4 xf[KS1];
    {
        const bf16_t* xp = x + prow * ldx + 8 * lh;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) xf[ks] = *(const uint4*)(xp + 16 * ks);
    }
    // epilogue of one 32-cout block in the transposed-accumulator layout (a lane: 16 couts of row l31): pairs the half-waves'
    // 4-channel groups (v_permlane32_swap) so that residual / gate / output move as 16-byte items
    auto epilogue = [&](const f32x16_t& acc, int co0, const float* __restrict__ bias, bool has_r, const uint4 (&rq)[2], int act,
                        const uint4 (&gq)[2], int gact, bf16_t* __restrict__ yo, int ldy, bool to_lds) __attribute__((always_inline)) {
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            float lo[4], hi[4];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int g = 2 * gp + e;
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (!GATED) bv = *(const float4*)(bias + co0 + 8 * g + 4 * lh);
                float* d = e ? hi : lo;
                d[0] = acc[4 * g + 0] + bv.x; d[1] = acc[4 * g + 1] + bv.y; d[2] = acc[4 * g + 2] + bv.z; d[3] = acc[4 * g + 3] + bv.w;
            }
            float o[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo[i]), __float_as_uint(hi[i]), false, false);
                o[i] = __uint_as_float(sw[0]);
                o[4 + i] = __uint_as_float(sw[1]);
            }
            if (has_r) {
                float rv[8];
                unpack16<bf16_t>(rq[gp], rv);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += rv[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = act_apply(o[e], act);
            if constexpr (GATED) {
                float gv[8];
                unpack16<bf16_t>(gq[gp], gv);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = act_gate(o[e], gv[e], gact);
            }
            const uint4 pk = pack16<bf16_t>(o);
            const int c = co0 + 16 * gp + 8 * lh;                 // this lane's 8 consecutive channels
            if (ok) *(uint4*)(yo + mrow * ldy + c) = pk;
            if (to_lds) *(uint4*)(smem + l31 * pitch + ((((c >> 3) ^ (l31 & 15))) << 4)) = pk;
        }
    };
    // A fragment (cout block cb, K step ks) of a fragment-ordered pack with KS steps per block: 1 KiB contiguous per wave
    auto load_a = [&](const uint4* __restrict__ w, int cb, int KS, int ks) __attribute__((always_inline)) { return w[((size_t)cb * KS + ks) * 64 + lane]; };

    // ---- stage 1: couts 32 (wave + NW j); C1 / 256 = 1, 2 or 4 blocks per wave ----
    const int nblk = C1 / (32 * NW);
    uint4 af[2][16];
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) af[0][ks] = load_a(wa, wave, KS1, ks);
    auto block1 = [&](auto CUR, int cb, int cb_next) __attribute__((always_inline)) {
        constexpr int S = decltype(CUR)::value;
        const int co0 = 32 * cb;
        uint4 rq[2], gq[2];
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {                       // residual / gate of this block: in flight under its MFMAs
            rq[gp] = *(const uint4*)(res + prow * ldres + co0 + 16 * gp + 8 * lh);
            if constexpr (GATED) gq[gp] = *(const uint4*)(gate1 + prow * ldg1 + co0 + 16 * gp + 8 * lh);
        }
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) af[S ^ 1][ks] = load_a(wa, cb_next, KS1, ks);       // the next block's fragments (unconditional)
        f32x16_t acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) PwMma<bf16_t>::run(acc, af[S][ks], xf[ks]);
        epilogue(acc, co0, bias_a, true, rq, act1, gq, gate1_act, y1, ldy1, true);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    const int last1 = wave + NW * (nblk - 1);
    if (nblk == 1) {
        block1(S0{}, wave, wave);
    } else {
        for (int j = 0; j < nblk; j += 2) {
            const int cb = wave + NW * j;
            block1(S0{}, cb, cb + NW);
            block1(S1{}, cb + NW, min(cb + 2 * NW, last1));
        }
    }
    // ---- stage 2: couts 32 (wave + NW j) of C2; K = C1 in groups of 16 K steps (C1 / 256 = 1, 2 or 4 groups) ----
    const int nb2 = C2 / 32;
    const int KS2 = C1 / 16, ngrp = KS2 / 16;
    // the first group of this wave's first block is requested BEFORE the barrier that publishes the y1 image
#pragma unroll
    for (int t = 0; t < 16; ++t) af[0][t] = load_a(wb, min(wave, nb2 - 1), KS2, t);
    __syncthreads();
    auto group2 = [&](auto CUR, f32x16_t& acc, int cb_next, int g, int g_next) __attribute__((always_inline)) {
        constexpr int S = decltype(CUR)::value;
#pragma unroll
        for (int t = 0; t < 16; ++t) af[S ^ 1][t] = load_a(wb, cb_next, KS2, 16 * g_next + t);     // unconditional
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int slot = 2 * (16 * g + t) + lh;                            // 16-byte slot of (K step, k half) in the row
            const uint4 bfrag = *(const uint4*)(smem + l31 * pitch + ((slot ^ (l31 & 15)) << 4));
            PwMma<bf16_t>::run(acc, af[S][t], bfrag);
        }
    };
    for (int cb = wave; cb < nb2; cb += NW) {
        const int co0 = 32 * cb;
        const int cbn = min(cb + NW, nb2 - 1);                 // the block whose first group follows this block's last one
        uint4 rq[2], gq[2];
        if constexpr (GATED) {
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) gq[gp] = *(const uint4*)(gate2 + prow * ldg2 + co0 + 16 * gp + 8 * lh);
        }
        f32x16_t acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        // every block starts with its group 0 in set 0 and ends with the next block's group 0 in set 0
        if (ngrp == 1) {
            group2(S0{}, acc, cbn, 0, 0);
#pragma unroll
            for (int t = 0; t < 16; ++t) af[0][t] = af[1][t];
        } else {
            for (int g = 0; g < ngrp; g += 2) {
                group2(S0{}, acc, cb, g, g + 1);
                const bool lastg = g + 2 >= ngrp;
                group2(S1{}, acc, lastg ? cbn : cb, g + 1, lastg ? 0 : g + 2);
            }
        }
        epilogue(acc, co0, bias_b, false, rq, act2, gq, gate2_act, y2, ldy2, false);
    }
}

// =================================================================================================
// stem: conv 7x7, stride 2, pad 3, 3 -> 64, + bias + ReLU, NCHW fp32 image -> NHWC T
// =================================================================================================
// One workgroup = 8 x 32 output pixels x 64 channels: the 21 x 69 x 3 input patch and the [147][64] weights sit in LDS, every
// thread owns one output pixel and all 64 channels (64 fp32 accumulators); weights are read as wave-uniform (broadcast) 16-byte
// LDS loads, so the inner loop is 1 patch read + 16 broadcast reads per 64 FMAs.
constexpr int kSTH = 8, kSTW = 32, kSPH = 2 * kSTH + 5, kSPW = 2 * kSTW + 5;     // patch 21 x 69

template <typename T>
__global__ __launch_bounds__(256) void stem7x7_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                          T* __restrict__ y, int ldy, int N, int H, int W, int Ho, int Wo, int act) {
    __shared__ float patch[3][kSPH][kSPW + 1];
    __shared__ __attribute__((aligned(16))) float wl[147][64];        // [c*49 + kh*7 + kw][co]
    const int tid = threadIdx.x;
    const int tiles_x = cdiv_dev(Wo, kSTW), tiles_y = cdiv_dev(Ho, kSTH);
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int n = bid / tiles_y;
    const int oh0 = ty * kSTH, ow0 = tx * kSTW;
    const int ih0 = 2 * oh0 - 3, iw0 = 2 * ow0 - 3;
    for (int i = tid; i < 147 * 64; i += 256) {
        const int co = i & 63, k = i >> 6;
        wl[k][co] = w[co * 147 + k];                                  // OIHW [64][3][7][7] -> [k][co]
    }
    for (int i = tid; i < 3 * kSPH * kSPW; i += 256) {
        const int px = i % kSPW, t = i / kSPW;
        const int py = t % kSPH, c = t / kSPH;
        const int ih = ih0 + py, iw = iw0 + px;
        const bool ok = ih >= 0 && ih < H && iw >= 0 && iw < W;
        const float v = x[((size_t)(n * 3 + c) * H + min(max(ih, 0), H - 1)) * W + min(max(iw, 0), W - 1)];
        patch[c][py][px] = ok ? v : 0.f;
    }
    __syncthreads();
    const int py0 = 2 * (tid >> 5), px0 = 2 * (tid & 31);
    float acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = 0.f;
    for (int c = 0; c < 3; ++c)
        for (int kh = 0; kh < 7; ++kh)
#pragma unroll
            for (int kw = 0; kw < 7; ++kw) {
                const float xv = patch[c][py0 + kh][px0 + kw];
                const float4* wr = (const float4*)wl[c * 49 + kh * 7 + kw];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float4 wv = wr[q];
                    acc[4 * q + 0] = fmaf(xv, wv.x, acc[4 * q + 0]); acc[4 * q + 1] = fmaf(xv, wv.y, acc[4 * q + 1]);
                    acc[4 * q + 2] = fmaf(xv, wv.z, acc[4 * q + 2]); acc[4 * q + 3] = fmaf(xv, wv.w, acc[4 * q + 3]);
                }
            }
    const int oh = oh0 + (tid >> 5), ow = ow0 + (tid & 31);
    if (oh < Ho && ow < Wo) {
        T* yp = y + ((size_t)(n * Ho + oh) * Wo + ow) * ldy;
        constexpr int E = ElemTraits<T>::kPer16B;
#pragma unroll
        for (int q = 0; q < 64 / E; ++q) {
            float o[E];
#pragma unroll
            for (int e = 0; e < E; ++e) o[e] = act_apply(acc[q * E + e] + (bias ? bias[q * E + e] : 0.f), act);
            *(uint4*)(yp + q * E) = pack16<T>(o);
        }
    }
}

// bf16 stem on the matrix cores (round 3).  The VALU kernel above reads 147 x 16 float4 weight rows from LDS per output pixel and
// is LDS-bound (510 us for B = 64 at 256x256, 185 MB: 0.36 TB/s); as an implicit GEMM the stem is ~20 GFLOP and HBM-bound.
//   rows (MFMA B operand, transposed scheme of conv3x3_mfma_v2): 32 output pixels of one output row; couts: 2 x 32 (A operand);
//   K ordered (kh, c, kw padded 7 -> 8): 21 groups of 8 + one zero group = 11 k-steps of 16.  A lane's 8 k-values of a group are
//   8 CONSECUTIVE image columns of one channel row -- the bf16 patch row is read with four 4-byte LDS reads (the window of output
//   column ox starts at patch column 2 ox: even, so 4-byte aligned; consecutive lanes read consecutive dwords: conflict-free).
//   The weight fragments (11 k-steps x 2 cout halves) are built once per workgroup from the OIHW fp32 tensor and live in registers;
//   workgroups are persistent over (image, 8 x 32-pixel tile) items.
constexpr int kSP16 = 72;       // bf16 patch pitch: 69 real columns + column 69 (kw = 7, zero weight: must only be finite) + pad

__global__ __launch_bounds__(256, 2) void stem7x7_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                                  bf16_t* __restrict__ y, int ldy, int N, int H, int W, int Ho, int Wo, int act, int ntiles) {
    __shared__ __attribute__((aligned(16))) bf16_t patch[3][kSPH][kSP16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int tiles_x = cdiv_dev(Wo, kSTW), tiles_y = cdiv_dev(Ho, kSTH);

    // ---- weight fragments: k-step s, lane half lh -> group g = 2 s + lh = (kh, c); element j = kw (kw = 7 and g = 21: zero) ----
    uint4 wa[11][2];
#pragma unroll
    for (int s_ = 0; s_ < 11; ++s_)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int g = 2 * s_ + lh, kh = g / 3, c = g - 3 * kh, co = 32 * h + l31;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (g < 21 && j < 7) ? w[(co * 3 + c) * 49 + kh * 7 + j] : 0.f;
            wa[s_][h] = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
        }
    // bias in the transposed-accumulator layout: register quad q of half h holds couts 32 h + 8 q + 4 lh .. + 3
    float4 bq[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[h][q] = bias ? *(const float4*)(bias + 32 * h + 8 * q + 4 * lh) : make_float4(0.f, 0.f, 0.f, 0.f);
    // per-lane patch byte offset of (group row of k-step s, this lane's window start): row = c * 21 + kh (+ 2 * local output row)
    int xoff[11];
#pragma unroll
    for (int s_ = 0; s_ < 11; ++s_) {
        const int g = min(2 * s_ + lh, 20), kh = g / 3, c = g - 3 * kh;      // g = 21 (zero weights) re-reads group 20
        xoff[s_] = ((c * kSPH + kh) * kSP16 + 2 * l31) * 2;
    }

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y;
        const int n = t / tiles_y;
        const int oh0 = ty * kSTH, ow0 = tx * kSTW;
        const int ih0 = 2 * oh0 - 3, iw0 = 2 * ow0 - 3;
        __syncthreads();                                  // the previous tile's fragment reads are done
        for (int i = tid; i < 3 * kSPH * kSP16; i += 256) {
            const int px = i % kSP16, r = i / kSP16;
            const int py = r % kSPH, c = r / kSPH;
            const int ih = ih0 + py, iw = iw0 + px;
            const bool ok = px < kSPW && ih >= 0 && ih < H && iw >= 0 && iw < W;
            const float v = x[((size_t)(n * 3 + c) * H + min(max(ih, 0), H - 1)) * W + min(max(iw, 0), W - 1)];
            patch[c][py][px] = f32_to_bf16(ok ? v : 0.f);
        }
        __syncthreads();
        f32x16_t acc[2][2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[r][h][i] = 0.f;
        const char* pb = (const char*)&patch[0][0][0];
#pragma unroll
        for (int s_ = 0; s_ < 11; ++s_) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const char* p = pb + xoff[s_] + (2 * (2 * wave + r)) * kSP16 * 2;       // output row 2 wave + r -> patch rows 2 (2 wave + r) + kh
                const uint4 xb = make_uint4(*(const uint32_t*)p, *(const uint32_t*)(p + 4), *(const uint32_t*)(p + 8), *(const uint32_t*)(p + 12));
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    acc[r][h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wa[s_][h]), __builtin_bit_cast(bf16x8_t, xb), acc[r][h], 0, 0, 0);
            }
        }
        // ---- epilogue: bias + activation, bf16, v_permlane32_swap pairs the half-waves' 4-channel groups into 16-byte stores ----
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int oh = oh0 + 2 * wave + r, ow = ow0 + l31;
            const bool ok = oh < Ho && ow < Wo;
            bf16_t* yp = y + ((size_t)(n * Ho + min(oh, Ho - 1)) * Wo + min(ow, Wo - 1)) * ldy;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    uint32_t o[2][2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float4 bv = bq[h][q + e];
                        const int r0 = 4 * (q + e);
                        o[e][0] = pack_bf16x2(act_apply(acc[r][h][r0 + 0] + bv.x, act), act_apply(acc[r][h][r0 + 1] + bv.y, act));
                        o[e][1] = pack_bf16x2(act_apply(acc[r][h][r0 + 2] + bv.z, act), act_apply(acc[r][h][r0 + 3] + bv.w, act));
                    }
                    const auto s0 = __builtin_amdgcn_permlane32_swap(o[0][0], o[1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(o[0][1], o[1][1], false, false);
                    // lanes 0-31: channels 32 h + 8 q .. + 7 of their pixel, lanes 32-63: + 8 .. + 15
                    if (ok) *(uint4*)(yp + 32 * h + 8 * q + 8 * lh) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                }
        }
    }
}

// data gradient of the stem wrt the NCHW fp32 image: dx[n][c][ih][iw] = sum_{co, kh, kw : (ih+3-kh, iw+3-kw) even} dy[n][oh][ow][co] w[co][c][kh][kw]
// 8 lanes per input pixel (each 8 of the 64 channels), the partial sums of the 3 image channels folded with xor shuffles.
template <typename T>
__global__ __launch_bounds__(256) void stem7x7_dgrad_kernel(const T* __restrict__ dy, int lddy, const float* __restrict__ w,
                                                            float* __restrict__ dx, int N, int H, int W, int Ho, int Wo, int accumulate) {
    __shared__ float wl[49][3][64];                                    // [kh*7+kw][c][co]
    const int tid = threadIdx.x;
    for (int i = tid; i < 147 * 64; i += 256) {
        const int co = i & 63, k = i >> 6;            // k = c*49 + tap
        wl[k % 49][k / 49][co] = w[co * 147 + k];
    }
    __syncthreads();
    const int sub = tid & 7;                                           // channel octet
    const long long total = (long long)N * H * W;
    for (long long pix = (long long)blockIdx.x * 32 + (tid >> 3); pix < total; pix += (long long)gridDim.x * 32) {
        const unsigned p32 = (unsigned)pix, t = p32 / (unsigned)W;         // N*H*W < 2^31 (host check): no 64-bit division per pixel
        const int iw = (int)(p32 - t * (unsigned)W), n = (int)(t / (unsigned)H), ih = (int)(t - (unsigned)n * (unsigned)H);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        // kh must have the parity of ih + 3 (so that ih + 3 - kh = 2 * oh)
        for (int kh = (ih + 3) & 1; kh < 7; kh += 2) {
            const int oh = (ih + 3 - kh) >> 1;
            if (oh < 0 || oh >= Ho) continue;
            for (int kw = (iw + 3) & 1; kw < 7; kw += 2) {
                const int ow = (iw + 3 - kw) >> 1;
                if (ow < 0 || ow >= Wo) continue;
                const T* gp = dy + ((size_t)(n * Ho + oh) * Wo + ow) * lddy + sub * 8;
                float g[8];
                if constexpr (std::is_same<T, float>::value) {
                    const float4 a0 = *(const float4*)gp, a1 = *(const float4*)(gp + 4);
                    g[0] = a0.x; g[1] = a0.y; g[2] = a0.z; g[3] = a0.w; g[4] = a1.x; g[5] = a1.y; g[6] = a1.z; g[7] = a1.w;
                } else {
                    unpack16<T>(*(const uint4*)gp, g);
                }
                const float* wp = &wl[kh * 7 + kw][0][sub * 8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s0 = fmaf(g[e], wp[e], s0);
                    s1 = fmaf(g[e], wp[64 + e], s1);
                    s2 = fmaf(g[e], wp[128 + e], s2);
                }
            }
        }
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {
            s0 += __shfl_xor(s0, off); s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off);
        }
        if (sub < 3) {
            const float v = sub == 0 ? s0 : (sub == 1 ? s1 : s2);
            float* dp = dx + ((size_t)(n * 3 + sub) * H + ih) * W + iw;
            *dp = accumulate ? *dp + v : v;
        }
    }
}

// bf16 stem data gradient on the matrix cores (round 3; the formulation of conv3x3_c3_dgrad_s2_mfma_kernel, thin.hip).  The VALU kernel
// above spends ~150 LDS-fed FMAs per input pixel and channel octet and is instruction-bound (432 us at B = 32 for 92 MB of traffic).
// Here, per OUTPUT pixel q and image channel c, the 49 products  P_c[k][q] = sum_co W[co][c][k] * dY[q][co]  (k = kh*7 + kw, padded to
// 64 rows) are two 32 x 32 x 64 MFMA blocks per 32 pixels; an INPUT pixel then sums the 9 / 12 / 16 entries of P_c whose taps have its
// parity, in (kh, kw) order -- no atomics, deterministic.  A workgroup takes 16 x 32 input pixels = the 11 x 19 output pixels that reach
// them (dY fragments loaded once into registers, 7 blocks of 32 over 4 waves), channel by channel through one 49 x 225 fp32 LDS
// image; the 24 weight fragments (3 channels x 2 row blocks x 4 k-steps) are built once per workgroup, which is persistent over tiles.
constexpr int kDGH = 16, kDGW = 32, kDQH = kDGH / 2 + 3, kDQW = kDGW / 2 + 3, kDQN = kDQH * kDQW;       // 11 x 19 = 209 output pixels
constexpr int kDQB = (kDQN + 31) / 32, kDPS = kDQB * 32 + 1;                                            // 7 blocks; P row pitch 225

// sum of the taps of parity (PH, PW) for the input pixel whose P base is pb = P + (row / 2) * 19 + col / 2, in (kh, kw) order
template <int PH, int PW>
__device__ __forceinline__ float dg_gather(const float* __restrict__ pb) {
    float sacc = 0.f;
#pragma unroll
    for (int kh = (PH + 3) & 1; kh < 7; kh += 2)
#pragma unroll
        for (int kw = (PW + 3) & 1; kw < 7; kw += 2)
            sacc += pb[(kh * 7 + kw) * kDPS + ((PH + 3 - kh) / 2 + 1) * kDQW + ((PW + 3 - kw) / 2 + 1)];
    return sacc;
}

__global__ __launch_bounds__(256, 2) void stem7x7_dgrad_mfma_kernel(const bf16_t* __restrict__ dy, int lddy, const float* __restrict__ w,
                                                                    float* __restrict__ dx, int N, int H, int W, int Ho, int Wo,
                                                                    int tiles_x, int tiles_y, int accumulate) {
    __shared__ float P[49 * kDPS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    // weights as the MFMA A operand: row k = 32 kb + l31 (49 real rows), reduction index co = 16 ks + 8 lh + j.  The OIHW tensor goes
    // through LDS once (coalesced; 192 strided global loads per lane took longer than a tile), the fragments then live in registers
    static_assert(64 * 147 <= 49 * kDPS, "the weight image must fit the P buffer");
    for (int i = tid; i < 64 * 147; i += 256) P[i] = w[i];
    __syncthreads();
    uint4 wf[3][2][4];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int k = 32 * kb + l31;
                float f[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = k < 49 ? P[((16 * ks + 8 * lh + j) * 3 + c) * 49 + k] : 0.f;
                wf[c][kb][ks] = pack16<bf16_t>(f);
            }
    const int ntiles = N * tiles_x * tiles_y;
    const size_t hw = (size_t)H * W;
    // dY fragments of this wave's blocks (wave, wave + 4) of one tile: pixel q = 32 b + l31 -> (q / 19, q % 19); the NEXT tile's are
    // requested before the current tile is worked on
    auto load_dy = [&](int tile, uint4 (&av)[2][4], unsigned (&mk)[2]) __attribute__((always_inline)) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y, n = t / tiles_y;
        const int oh0 = ty * kDGH / 2 - 1, ow0 = tx * kDGW / 2 - 1;      // first output row / column that reaches the tile
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int b = wave + 4 * i;
            const int q = 32 * b + l31, r = q / kDQW, cc = q - r * kDQW;
            const int oh = oh0 + r, ow = ow0 + cc;
            const bool ok = tile < ntiles && b < kDQB && q < kDQN && oh >= 0 && oh < Ho && ow >= 0 && ow < Wo;
            const size_t op = ((size_t)min(n, N - 1) * Ho + min(max(oh, 0), Ho - 1)) * Wo + min(max(ow, 0), Wo - 1);
            mk[i] = ok ? 0xffffffffu : 0u;                  // lane mask instead of a 128-bit select, applied where the fragment is used
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) av[i][ks] = *(const uint4*)(dy + op * lddy + 16 * ks + 8 * lh);
        }
    };
    uint4 av[2][4], avn[2][4];
    unsigned mka[2], mkn[2];
    load_dy(blockIdx.x, av, mka);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y, n = t / tiles_y;
        const int ih0 = ty * kDGH, iw0 = tx * kDGW;
        load_dy(tile + gridDim.x, avn, mkn);
#pragma unroll
        for (int i = 0; i < 2; ++i)                          // the current tile's fragments: out-of-image pixels contribute zero
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) av[i][ks] = make_uint4(av[i][ks].x & mka[i], av[i][ks].y & mka[i], av[i][ks].z & mka[i], av[i][ks].w & mka[i]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {                         // unrolled: wf[c] must be a register, not an indexed (scratch) array
            __syncthreads();                                  // the previous channel's (tile's) gather is done with P
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int b = wave + 4 * i;
                if (b >= kDQB) break;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    f32x16_t acc;
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[c][kb][ks]), __builtin_bit_cast(bf16x8_t, av[i][ks]), acc, 0, 0, 0);
                    }
                    // lane (pixel l31, half lh) holds rows k = 32 kb + 8 g + 4 lh + e of its pixel's column
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int k = 32 * kb + 8 * g + 4 * lh + e;
                            if (k < 49) P[k * kDPS + 32 * b + l31] = acc[4 * g + e];
                        }
                }
            }
            __syncthreads();
            // ---- input pixels (ih0 + row, iw0 + col): taps of matching parity, summed in (kh, kw) order.  A thread takes the two
            // pixels (row, 2 cb) and (row, 2 cb + 1) -- one 8-byte store -- of a row whose parity is fixed per 16-lane group, so every
            // P address is base + a compile-time offset (the generic loops cost ~500 instructions per pixel: the kernel was VALU-bound)
            {
                const int cb = tid & 15, ph = (tid >> 4) & 1, rb = tid >> 5;
                const int row = 2 * rb + ph, ih = ih0 + row, iw = iw0 + 2 * cb;
                const float* pb = P + rb * kDQW + cb;
                float s0, s1;
                if (ph) { s0 = dg_gather<1, 0>(pb); s1 = dg_gather<1, 1>(pb); }
                else { s0 = dg_gather<0, 0>(pb); s1 = dg_gather<0, 1>(pb); }
                if (ih < H && iw < W) {                          // W is even: the pair is inside or outside together
                    float2* o = (float2*)(dx + ((size_t)n * 3 + c) * hw + (size_t)ih * W + iw);
                    float2 v = make_float2(s0, s1);
                    if (accumulate) { const float2 old = *o; v.x += old.x; v.y += old.y; }
                    *o = v;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) av[i][ks] = avn[i][ks];
        mka[0] = mkn[0]; mka[1] = mkn[1];
    }
}

// =================================================================================================
// MaxPool2d(kernel 3, stride 2, pad 1), NHWC, with the window-local arg-max (0..8, first maximum in scan order) kept for backward
// =================================================================================================
template <typename T>
__global__ void maxpool3s2_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, uint8_t* __restrict__ idx,
                                      int N, int H, int W, int Ho, int Wo, int C) {
    constexpr int E = ElemTraits<T>::kPer16B;
    const int cpp = C / E;
    const long long total = (long long)N * Ho * Wo * cpp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cpp);
        long long p = i / cpp;
        const int ow = (int)(p % Wo); p /= Wo;
        const int oh = (int)(p % Ho);
        const int n = (int)(p / Ho);
        float m[E];
        int am[E];
#pragma unroll
        for (int e = 0; e < E; ++e) { m[e] = -INFINITY; am[e] = 0; }
        bool first = true;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ih = 2 * oh - 1 + kh, iw = 2 * ow - 1 + kw;
                if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
                float v[E];
                unpack16<T>(*(const uint4*)(x + ((size_t)(n * H + ih) * W + iw) * ldx + ch * E), v);
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if (first || v[e] > m[e] || v[e] != v[e]) { m[e] = v[e]; am[e] = kh * 3 + kw; }
                first = false;
            }
        const size_t o = ((size_t)(n * Ho + oh) * Wo + ow);
        *(uint4*)(y + o * ldy + ch * E) = pack16<T>(m);
        if (idx) {
#pragma unroll
            for (int e = 0; e < E; ++e) idx[o * C + ch * E + e] = (uint8_t)am[e];
        }
    }
}

// dx[pixel] = sum over the <= 4 windows containing it whose arg-max is this pixel of dy[window]; optionally * act'(x)
template <typename T>
__global__ void maxpool3s2_bwd_kernel(const T* __restrict__ dy, int lddy, const uint8_t* __restrict__ idx, const T* __restrict__ x, int ldx,
                                      T* __restrict__ dx, int lddx, int N, int H, int W, int Ho, int Wo, int C, int gate_act) {
    constexpr int E = ElemTraits<T>::kPer16B;
    const int cpp = C / E;
    const long long total = (long long)N * H * W * cpp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cpp);
        long long p = i / cpp;
        const int iw = (int)(p % W); p /= W;
        const int ih = (int)(p % H);
        const int n = (int)(p / H);
        float g[E];
#pragma unroll
        for (int e = 0; e < E; ++e) g[e] = 0.f;
        // windows oh with 2*oh - 1 <= ih <= 2*oh + 1
        for (int oh = (ih >> 1); oh <= ((ih + 1) >> 1); ++oh) {
            if (oh < 0 || oh >= Ho) continue;
            const int kh = ih - (2 * oh - 1);
            for (int ow = (iw >> 1); ow <= ((iw + 1) >> 1); ++ow) {
                if (ow < 0 || ow >= Wo) continue;
                const int kw = iw - (2 * ow - 1);
                const size_t o = ((size_t)(n * Ho + oh) * Wo + ow);
                float d[E];
                unpack16<T>(*(const uint4*)(dy + o * lddy + ch * E), d);
                const uint8_t* ip = idx + o * C + ch * E;
#pragma unroll
                for (int e = 0; e < E; ++e) g[e] += (ip[e] == kh * 3 + kw) ? d[e] : 0.f;
            }
        }
        const size_t pin = ((size_t)(n * H + ih) * W + iw);
        if (gate_act != WU_ACT_NONE) {
            float xv[E];
            unpack16<T>(*(const uint4*)(x + pin * ldx + ch * E), xv);
#pragma unroll
            for (int e = 0; e < E; ++e) g[e] = act_gate(g[e], xv[e], gate_act);
        }
        *(uint4*)(dx + pin * lddx + ch * E) = pack16<T>(g);
    }
}

inline bool al16(const void* p) { return ((uintptr_t)p % 16) == 0; }
inline int grid_cap(long long total, int block = 256, int cap = 256 * 16) {
    long long g = (total + block - 1) / block;
    return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace

namespace {
template <int NW, int TN, bool G>
void pw3_conv_go(const pw3::Args& p, int grid, hipStream_t s) {
    constexpr int kSmem = 2 * (pw3::TM + TN) * kKB;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv1x1_pw3_kernel<NW, TN, 2, false, G, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem + pw3::kMaxCout * 4);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv1x1_pw3_kernel<NW, TN, 2, false, G, true>), dim3((unsigned)grid), dim3(NW * 64), (size_t)kSmem + (size_t)p.Cout * 4, s, p);
}
// shared tail of the two launchers below: checks, tile shape (128 couts per tile, or 64 when Cout is an odd multiple of 64), launch
int pw3_conv_launch(pw3::Args& p, int N, int src_pixels, int out_pixels, int max_out_ld, bool hg, hipStream_t s) {
    const int pd = g_wu_opt[WU_OPT_PW3] & 7, allow64 = (g_wu_opt[WU_OPT_PW3] >> 4) & 1, min_items = g_wu_opt[WU_OPT_PW3] >> 5;
    const int tn = p.Cout % 128 == 0 ? 128 : 64;
    const long long items = (((long long)p.M + pw3::TM - 1) / pw3::TM) * (p.Cout / tn) * p.ncls;
    // 64-cout tiles (four waves, 128 x 64) are correct but not faster than the register-staged kernels (profiles/r04_s2_gather_bench.txt: SNDisc's first
    // stride-2 data gradient 80 vs 68 us): off unless option bit 4 asks (tests)
    if (tn == 64 && !allow64) return 1;
    if (pd == 0 || p.Cin % 64 != 0 || p.Cout % 64 != 0 || p.Cout > pw3::kMaxCout || items < min_items || items >= (1ll << 30)) return 1;
    if (((long long)N * src_pixels + p.W + 1 + 2ll * pw3::TM * p.stride * p.stride) * p.ldx * 2 >= (1ll << 31) ||
        ((long long)N * out_pixels + pw3::TM) * max_out_ld * 2 >= (1ll << 31) || 9ll * p.Cout * p.Cin * 2 >= (1ll << 31)) return 1;
    if (((uintptr_t)p.x | (uintptr_t)p.y | (uintptr_t)p.w | (uintptr_t)p.egate) % 16 != 0 || (p.ldx * 2) % 16 != 0 || (p.ldy * 2) % 16 != 0 || (hg && (p.ldegate * 2) % 16 != 0)) return 1;
    p.n_ct = p.Cout / tn; p.items = (int)items;
    const int grid = (int)std::min<long long>(items, 2ll * wu_num_cus());
    if (tn == 128) { if (hg) pw3_conv_go<8, 128, true>(p, grid, s); else pw3_conv_go<8, 128, false>(p, grid, s); }
    else { if (hg) pw3_conv_go<4, 64, true>(p, grid, s); else pw3_conv_go<4, 64, false>(p, grid, s); }
    return 0;
}
}  // namespace

// 3 x 3 conv (stride 1 or 2, bf16, unmasked) on the persistent LDS-DMA GEMM pipeline with gathered activation rows (conv1x1_pw3_kernel<.., CONV>):
// 0 = launched, 1 = not applicable (the caller falls back to conv3x3_mfma_kernel).  w_packed: the [9][Cout][Cin] forward pack.
int conv3x3_gather_launch(const void* x, int ldx, const void* w_packed, const float* bias, void* y, int ldy, const void* egate, int ldegate, int egate_act,
                          int N, int H, int W, int Cin, int Cout, int stride, int act, hipStream_t s) {
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const bool hg = egate != nullptr && egate_act != WU_ACT_NONE;
    pw3::Args p = {};
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w_packed; p.bias = bias; p.res = nullptr; p.egate = (const bf16_t*)egate; p.y = (bf16_t*)y;
    p.ldx = ldx; p.ldres = 0; p.ldegate = hg ? ldegate : 0; p.ldy = ldy;
    p.M = N * Ho * Wo; p.Cin = Cin; p.Cout = Cout; p.act = act; p.egate_act = egate_act;
    p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.stride = stride;
    p.ncls = 1; p.cls_ntaps[0] = 9; p.cls_ooy[0] = p.cls_oox[0] = 0;
    for (int t = 0; t < 9; ++t) p.cls_tap[0][t] = (unsigned)((t / 3) | ((t % 3) << 2) | (t << 4));
    p.osy = p.osx = 1; p.OH = Ho; p.OW = Wo;
    if ((long long)N * Ho * Wo >= (1ll << 30)) return 1;
    return pw3_conv_launch(p, N, H * W, Ho * Wo, std::max(ldy, hg ? ldegate : 0), hg, s);
}

// Data gradient of a stride-2 3 x 3 conv on the same pipeline: the four parity classes of conv_s2_dgrad_parity_launch (conv3x3_mfma.hip) as ONE launch,
// class = fastest digit of the item index.  dy: (N, Cout, Ho, Wo) pre-gated; w_dgrad: the rotated pack [tap'][Cin][Cout]; dx: (N, Cin, H, W), optionally
// multiplied by act'(egate).  0 = launched, 1 = not applicable.
int conv_s2_dgrad_gather_launch(const void* dy, int lddy, const void* w_dgrad, void* dx, int lddx, const void* egate, int ldegate, int egate_act,
                                int N, int H, int W, int Cin, int Cout, hipStream_t s) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const bool hg = egate != nullptr && egate_act != WU_ACT_NONE;
    pw3::Args p = {};
    p.x = (const bf16_t*)dy; p.w = (const bf16_t*)w_dgrad; p.bias = nullptr; p.res = nullptr; p.egate = (const bf16_t*)egate; p.y = (bf16_t*)dx;
    p.ldx = lddy; p.ldres = 0; p.ldegate = hg ? ldegate : 0; p.ldy = lddx;
    p.M = N * Ho * Wo; p.Cin = Cout; p.Cout = Cin;       // the GEMM's K = the forward conv's output channels, its N = the forward conv's input channels
    p.act = WU_ACT_NONE; p.egate_act = egate_act;
    p.H = Ho; p.W = Wo; p.Ho = Ho; p.Wo = Wo; p.stride = 1;
    p.ncls = 4; p.osy = p.osx = 2; p.OH = H; p.OW = W;
    for (int c = 0; c < 4; ++c) {
        const int pp = c < 2 ? 1 : 0, q = (c == 0 || c == 2) ? 1 : 0;
        // forward taps reaching input parity p: kh = 1 from output row i (source row offset dy = 1); kh = 0 from row i + 1 (dy = 2), kh = 2 from row i
        const int nk_y = pp ? 2 : 1, nk_x = q ? 2 : 1;
        const int khs[2] = {pp ? 0 : 1, 2}, dys[2] = {pp ? 2 : 1, 1};
        const int kws[2] = {q ? 0 : 1, 2}, dxs[2] = {q ? 2 : 1, 1};
        p.cls_ntaps[c] = nk_y * nk_x;
        for (int iy = 0; iy < nk_y; ++iy)
            for (int ix = 0; ix < nk_x; ++ix)
                p.cls_tap[c][iy * nk_x + ix] = (unsigned)(dys[iy] | (dxs[ix] << 2) | ((8 - (khs[iy] * 3 + kws[ix])) << 4));   // rotated pack: forward tap (kh, kw) in slab 8 - (3 kh + kw)
        p.cls_ooy[c] = pp; p.cls_oox[c] = q;
    }
    if ((long long)N * Ho * Wo >= (1ll << 30)) return 1;
    return pw3_conv_launch(p, N, Ho * Wo, H * W, std::max(lddx, hg ? ldegate : 0), hg, s);
}

#define DISPATCH_T(dtype, ...)                                      \
    do {                                                            \
        if ((dtype) == WU_BF16) { using T = bf16_t; __VA_ARGS__; }  \
        else { using T = float; __VA_ARGS__; }                      \
    } while (0)

extern "C" int wu_conv1x1_fwd(const void* x, int ldx, const void* w, const float* bias, const void* residual, int ldres,
                              void* y, int ldy, int N, int Hc, int Wc, int in_stride, int Hin, int Win,
                              int out_stride, int Hout, int Wout, int Cin, int Cout, int act,
                              const void* egate, int ldegate, int egate_act, int dtype, void* stream) {
    WU_REQUIRE(dtype == WU_F32 || dtype == WU_BF16, "conv1x1_fwd: bad dtype %d", dtype);
    const int esz = dtype == WU_BF16 ? 2 : 4;
    const int ke = kKB / esz;
    WU_REQUIRE(N > 0 && Hc > 0 && Wc > 0, "conv1x1_fwd: empty shape");
    WU_REQUIRE(Cin > 0 && Cin % ke == 0, "conv1x1_fwd: Cin=%d must be a multiple of %d", Cin, ke);
    WU_REQUIRE(Cout > 0 && Cout % kTN == 0, "conv1x1_fwd: Cout=%d must be a multiple of %d", Cout, kTN);
    WU_REQUIRE((in_stride == 1 || in_stride == 2) && (out_stride == 1 || out_stride == 2) && !(in_stride == 2 && out_stride == 2),
               "conv1x1_fwd: strides (%d,%d)", in_stride, out_stride);
    WU_REQUIRE((Hc - 1) * in_stride < Hin && (Wc - 1) * in_stride < Win && (Hc - 1) * out_stride < Hout && (Wc - 1) * out_stride < Wout,
               "conv1x1_fwd: coarse grid %dx%d does not fit input %dx%d / output %dx%d", Hc, Wc, Hin, Win, Hout, Wout);
    WU_REQUIRE(out_stride == 1 ? (Hout == Hc && Wout == Wc) : (Hout <= 2 * Hc && Wout <= 2 * Wc), "conv1x1_fwd: output %dx%d is not covered by the coarse grid", Hout, Wout);
    WU_REQUIRE(ldx >= Cin && ldy >= Cout && (ldx * esz) % 16 == 0 && (ldy * esz) % 16 == 0 && al16(x) && al16(y) && al16(w), "conv1x1_fwd: bad ld / alignment");
    if (residual) WU_REQUIRE(ldres >= Cout && (ldres * esz) % 16 == 0 && al16(residual), "conv1x1_fwd: bad residual");
    if (egate) WU_REQUIRE(ldegate >= Cout && (ldegate * esz) % 16 == 0 && al16(egate), "conv1x1_fwd: bad egate");
    if (bias) WU_REQUIRE(al16(bias), "conv1x1_fwd: bias alignment");
    PwArgs a;
    a.x = x; a.w = w; a.bias = bias; a.res = residual; a.y = y; a.egate = egate;
    a.ldx = ldx; a.ldres = ldres; a.ldy = ldy; a.ldegate = ldegate;
    a.N = N; a.Hc = Hc; a.Wc = Wc; a.in_stride = in_stride; a.Hin = Hin; a.Win = Win; a.out_stride = out_stride; a.Hout = Hout; a.Wout = Wout;
    a.Cin = Cin; a.Cout = Cout; a.act = act; a.egate_act = egate_act;
    a.M = (long long)N * Hc * Wc;
    a.n_tiles = Cout / kTN;
    // pixel tile: 64 rows (option 12: 1 = 256 rows, the round-2 shape; 2 = 128 rows).  Same-box A/B of the whole GAN iteration: 28.12 ms
    // with 256-row tiles, 27.31-27.55 with 128, 27.13 with 64 -- the kernel's waves are parked 45-58 % of their cycles (SQ_WAIT_ANY,
    // scratch/pmc_pw.sh) behind a short chain of fetch / barrier / epilogue round trips, and five small workgroups per CU hide more
    // of that than two large ones; the weights they re-read come from L2
    const int tm = g_wu_opt[WU_OPT_PW_TILE] == 1 ? 256 : (g_wu_opt[WU_OPT_PW_TILE] == 2 ? 128 : 64);
    const long long grid = ((a.M + tm - 1) / tm) * a.n_tiles;
    WU_REQUIRE(grid < (1ll << 31), "conv1x1_fwd: grid too large");
    hipStream_t s = (hipStream_t)stream;
    wu_prof_pre(WU_FAM_CONV1X1, s);
    // Round 4: the persistent LDS-DMA form (128 x 128 tiles) where it applies.  Option 15: low 3 bits = ring depth D (0 = off), bit 3 = eight waves per workgroup instead
    // of four (D = 2: two workgroups per CU; else one), bit 4 = 64-cout tiles also for the 3 x 3 forms (the pointwise convs always take them), bits 5.. = the least number of tiles (below it the 64 x 64 tiles fill more of the chip)
    {
        const int pd = g_wu_opt[WU_OPT_PW3] & 7, pw8 = (g_wu_opt[WU_OPT_PW3] >> 3) & 1, min_items = g_wu_opt[WU_OPT_PW3] >> 5;
        // 64-cout tiles (four waves) for the layers whose Cout is an odd multiple of 64: 256 -> 64 at 131 k rows 15.7 us against 18.1, 64 -> 64 8.1 against 10.1
        const int tn = Cout % 128 == 0 ? 128 : 64;
        const long long items = ((a.M + pw3::TM - 1) / pw3::TM) * (Cout / tn);
        const long long max_ld = std::max(std::max(ldx, ldy), std::max(residual ? ldres : 0, egate ? ldegate : 0));
        if (pd >= 2 && pd <= 4 && dtype == WU_BF16 && in_stride == 1 && out_stride == 1 && Cout <= pw3::kMaxCout &&
            items >= min_items && items < (1ll << 30) && (a.M + pw3::TM) * max_ld * 2 < (1ll << 31) && (long long)Cout * Cin * 2 < (1ll << 31)) {
            pw3::Args p = {};
            p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.bias = bias; p.res = (const bf16_t*)residual; p.egate = (const bf16_t*)egate; p.y = (bf16_t*)y;
            p.ldx = ldx; p.ldres = residual ? ldres : 0; p.ldegate = egate ? ldegate : 0; p.ldy = ldy;
            p.M = (int)a.M; p.Cin = Cin; p.Cout = Cout; p.act = act; p.egate_act = egate_act; p.n_ct = Cout / tn; p.items = (int)items;
            p.ncls = 1;
            const int dd = tn == 64 ? 2 : pd;                // (the 64-cout form has one configuration: four waves, D = 2)
            const int per_cu = dd == 2 ? 2 : 1;
            const int grid3 = (int)std::min<long long>(items, (long long)per_cu * wu_num_cus());
            const size_t smem3 = (size_t)dd * (pw3::TM + tn) * kKB + (size_t)Cout * 4;
            const bool hr = residual != nullptr, hg = egate != nullptr && egate_act != WU_ACT_NONE;
#define WU_PW3_GO(NW_, TN_, D_, R_, G_)                                                                                                  \
            do {                                                                                                                         \
                static bool attr3 = false;                                                                                               \
                if (!attr3) { (void)hipFuncSetAttribute((const void*)conv1x1_pw3_kernel<NW_, TN_, D_, R_, G_>, hipFuncAttributeMaxDynamicSharedMemorySize, D_ * (pw3::TM + TN_) * kKB + pw3::kMaxCout * 4); attr3 = true; } \
                hipLaunchKernelGGL((conv1x1_pw3_kernel<NW_, TN_, D_, R_, G_>), dim3((unsigned)grid3), dim3(NW_ * 64), smem3, s, p);    \
            } while (0)
#define WU_PW3_D(NW_, TN_, D_)                                                        \
            do {                                                                      \
                if (hr && hg) WU_PW3_GO(NW_, TN_, D_, true, true);                    \
                else if (hr) WU_PW3_GO(NW_, TN_, D_, true, false);                    \
                else if (hg) WU_PW3_GO(NW_, TN_, D_, false, true);                    \
                else WU_PW3_GO(NW_, TN_, D_, false, false);                           \
            } while (0)
            if (tn == 64) WU_PW3_D(4, 64, 2);
            else if (pw8) { if (pd == 2) WU_PW3_D(8, 128, 2); else if (pd == 3) WU_PW3_D(8, 128, 3); else WU_PW3_D(8, 128, 4); }
            else { if (pd == 2) WU_PW3_D(4, 128, 2); else if (pd == 3) WU_PW3_D(4, 128, 3); else WU_PW3_D(4, 128, 4); }
#undef WU_PW3_D
#undef WU_PW3_GO
            wu_prof_post(WU_FAM_CONV1X1, s, 2.0 * (double)a.M * Cin * Cout, ((double)a.M * (Cin + Cout * (residual ? 2 : 1)) + (double)Cin * Cout) * esz);
            WU_LAUNCH_CHECK("conv1x1_pw3");
            return 0;
        }
    }
    static thread_local bool attr_set = false;
    if (!attr_set) {
#define WU_PW_ATTR(TM_) (void)hipFuncSetAttribute((const void*)conv1x1_mfma_kernel<bf16_t, TM_>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (TM_ + kTN) * kKB); \
                        (void)hipFuncSetAttribute((const void*)conv1x1_mfma_kernel<float, TM_>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (TM_ + kTN) * kKB)
        WU_PW_ATTR(256); WU_PW_ATTR(128); WU_PW_ATTR(64);
#undef WU_PW_ATTR
        attr_set = true;
    }
    if (tm == 64) DISPATCH_T(dtype, hipLaunchKernelGGL((conv1x1_mfma_kernel<T, 64>), dim3((unsigned)grid), dim3(256), 2 * (64 + kTN) * kKB, s, a));
    else if (tm == 128) DISPATCH_T(dtype, hipLaunchKernelGGL((conv1x1_mfma_kernel<T, 128>), dim3((unsigned)grid), dim3(256), 2 * (128 + kTN) * kKB, s, a));
    else DISPATCH_T(dtype, hipLaunchKernelGGL((conv1x1_mfma_kernel<T, 256>), dim3((unsigned)grid), dim3(256), 2 * (256 + kTN) * kKB, s, a));
    wu_prof_post(WU_FAM_CONV1X1, s, 2.0 * (double)a.M * Cin * Cout, ((double)a.M * (Cin + Cout * (residual ? 2 : 1)) + (double)Cin * Cout) * esz);
    WU_LAUNCH_CHECK("conv1x1_mfma");
    return 0;
}

extern "C" int wu_conv1x1_chain_supported(int K1, int C1, int C2, int dtype) {
    return (dtype == WU_BF16 && (K1 == 64 || K1 == 128 || K1 == 256) && C1 >= 256 && C1 % 256 == 0 && C1 <= 1024 && C2 >= 32 && C2 % 32 == 0) ? 1 : 0;
}

extern "C" int wu_conv1x1_chain(const void* x, int ldx, const void* wa, const float* bias_a, const void* res, int ldres, int act1,
                                const void* gate1, int ldg1, int gate1_act, void* y1, int ldy1,
                                const void* wb, const float* bias_b, int act2, const void* gate2, int ldg2, int gate2_act, void* y2, int ldy2,
                                long long M, int K1, int C1, int C2, int dtype, void* stream) {
    WU_REQUIRE(wu_conv1x1_chain_supported(K1, C1, C2, dtype), "conv1x1_chain: unsupported K1=%d C1=%d C2=%d dtype=%d (ask wu_conv1x1_chain_supported)", K1, C1, C2, dtype);
    WU_REQUIRE(M > 0 && M < (1ll << 36) && x && wa && wb && y1 && y2, "conv1x1_chain: bad args");
    // two forms only (compile-time epilogues): forward = bias_a + res + bias_b, no gates; backward = res + gate1 + gate2, no bias
    const bool gated = gate1 != nullptr;
    WU_REQUIRE(res && (gated ? (gate2 && !bias_a && !bias_b) : (bias_a && bias_b && !gate2)),
               "conv1x1_chain: operands must be (bias_a, res, bias_b) [forward pair] or (res, gate1, gate2) [backward pair]");
    WU_REQUIRE(ldx >= K1 && ldy1 >= C1 && ldy2 >= C2 && (ldx * 2) % 16 == 0 && (ldy1 * 2) % 16 == 0 && (ldy2 * 2) % 16 == 0 &&
               al16(x) && al16(wa) && al16(wb) && al16(y1) && al16(y2), "conv1x1_chain: bad ld / alignment");
    if (res) WU_REQUIRE(ldres >= C1 && (ldres * 2) % 16 == 0 && al16(res), "conv1x1_chain: bad residual");
    if (gate1) WU_REQUIRE(ldg1 >= C1 && (ldg1 * 2) % 16 == 0 && al16(gate1), "conv1x1_chain: bad gate1");
    if (gate2) WU_REQUIRE(ldg2 >= C2 && (ldg2 * 2) % 16 == 0 && al16(gate2), "conv1x1_chain: bad gate2");
    if (bias_a) WU_REQUIRE(al16(bias_a), "conv1x1_chain: bias alignment");
    if (bias_b) WU_REQUIRE(al16(bias_b), "conv1x1_chain: bias alignment");
    const long long grid = (M + 31) / 32;
    WU_REQUIRE(grid < (1ll << 31), "conv1x1_chain: grid too large");
    const size_t lds = (size_t)32 * C1 * 2;
    hipStream_t s = (hipStream_t)stream;
    static thread_local bool attr_set = false;
    if (!attr_set) {
#define WU_CHAIN_ATTR(K_, G_) (void)hipFuncSetAttribute((const void*)conv1x1_chain_kernel<K_, G_>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)
        WU_CHAIN_ATTR(64, false); WU_CHAIN_ATTR(128, false); WU_CHAIN_ATTR(256, false); WU_CHAIN_ATTR(64, true); WU_CHAIN_ATTR(128, true); WU_CHAIN_ATTR(256, true);
#undef WU_CHAIN_ATTR
        attr_set = true;
    }
    wu_prof_pre(WU_FAM_CONV1X1, s);
#define WU_CHAIN(K_) hipLaunchKernelGGL((conv1x1_chain_kernel<K_, G_>), dim3((unsigned)grid), dim3(64 * kChainWaves), lds, s, (const bf16_t*)x, ldx, (const uint4*)wa, bias_a, \
                                        (const bf16_t*)res, ldres, act1, (const bf16_t*)gate1, ldg1, gate1_act, (bf16_t*)y1, ldy1, (const uint4*)wb, bias_b, act2, \
                                        (const bf16_t*)gate2, ldg2, gate2_act, (bf16_t*)y2, ldy2, M, C1, C2)
    if (gated) {
        constexpr bool G_ = true;
        if (K1 == 64) WU_CHAIN(64); else if (K1 == 128) WU_CHAIN(128); else WU_CHAIN(256);
    } else {
        constexpr bool G_ = false;
        if (K1 == 64) WU_CHAIN(64); else if (K1 == 128) WU_CHAIN(128); else WU_CHAIN(256);
    }
#undef WU_CHAIN
    wu_prof_post(WU_FAM_CONV1X1, s, 2.0 * (double)M * ((double)K1 * C1 + (double)C1 * C2),
                 ((double)M * (K1 + C1 * (res ? 2 : 1) + C2) + (double)K1 * C1 + (double)C1 * C2) * 2);
    WU_LAUNCH_CHECK("conv1x1_chain");
    return 0;
}

extern "C" int wu_stem7x7_fwd(const float* x_nchw, const float* w_oihw, const float* bias, void* y, int ldy,
                              int N, int H, int W, int act, int dtype, void* stream) {
    const int esz = dtype == WU_BF16 ? 2 : 4;
    WU_REQUIRE(N > 0 && H >= 7 && W >= 7 && x_nchw && w_oihw && y, "stem7x7_fwd: bad shape");
    WU_REQUIRE(ldy >= 64 && (ldy * esz) % 16 == 0 && al16(y), "stem7x7_fwd: bad output ld / alignment");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;                     // (H + 6 - 7) / 2 + 1
    const long long grid = (long long)N * cdiv(Ho, kSTH) * cdiv(Wo, kSTW);
    WU_REQUIRE(grid < (1ll << 31), "stem7x7_fwd: grid too large");
    if (dtype == WU_BF16 && (!bias || al16(bias))) {
        // persistent: two workgroups per CU (registers), each builds its weight fragments once
        const long long g2 = 2ll * wu_num_cus();
        hipLaunchKernelGGL(stem7x7_fwd_mfma_kernel, dim3((unsigned)(grid < g2 ? grid : g2)), dim3(256), 0, (hipStream_t)stream,
                           x_nchw, w_oihw, bias, (bf16_t*)y, ldy, N, H, W, Ho, Wo, act, (int)grid);
        WU_LAUNCH_CHECK("stem7x7_fwd (mfma)");
        return 0;
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL(stem7x7_fwd_kernel<T>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream,
                                         x_nchw, w_oihw, bias, (T*)y, ldy, N, H, W, Ho, Wo, act));
    WU_LAUNCH_CHECK("stem7x7_fwd");
    return 0;
}

extern "C" int wu_stem7x7_dgrad(const void* dy, int lddy, const float* w_oihw, float* dx_nchw, int N, int H, int W,
                                int accumulate, int dtype, void* stream) {
    const int esz = dtype == WU_BF16 ? 2 : 4;
    WU_REQUIRE(N > 0 && H >= 7 && W >= 7 && dy && w_oihw && dx_nchw, "stem7x7_dgrad: bad shape");
    WU_REQUIRE(lddy >= 64 && (lddy * esz) % 16 == 0 && al16(dy), "stem7x7_dgrad: bad ld / alignment");
    WU_REQUIRE((long long)N * H * W < (1ll << 31), "stem7x7_dgrad: N*H*W must stay below 2^31");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long pixels = (long long)N * H * W;
    if (dtype == WU_BF16 && lddy % 8 == 0 && H % 2 == 0 && W % 2 == 0) {       // even sizes: a tile's output window starts at ih0 / 2 - 1
        const int tx_ = cdiv_dev(W, kDGW), ty_ = cdiv_dev(H, kDGH);
        const long long nt = (long long)N * tx_ * ty_;
        if (nt < (1ll << 31)) {
            const int g = (int)(nt < 2ll * wu_num_cus() ? nt : 2ll * wu_num_cus());
            hipLaunchKernelGGL(stem7x7_dgrad_mfma_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, lddy, w_oihw, dx_nchw,
                               N, H, W, Ho, Wo, tx_, ty_, accumulate);
            WU_LAUNCH_CHECK("stem7x7_dgrad_mfma");
            return 0;
        }
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL(stem7x7_dgrad_kernel<T>, dim3(grid_cap(pixels, 32, 256 * 32)), dim3(256), 0, (hipStream_t)stream,
                                         (const T*)dy, lddy, w_oihw, dx_nchw, N, H, W, Ho, Wo, accumulate));
    WU_LAUNCH_CHECK("stem7x7_dgrad");
    return 0;
}

extern "C" int wu_maxpool3s2_fwd(const void* x, int ldx, void* y, int ldy, uint8_t* argmax, int N, int H, int W, int C,
                                 int dtype, void* stream) {
    const int esz = dtype == WU_BF16 ? 2 : 4;
    WU_REQUIRE(N > 0 && H > 0 && W > 0 && C % (16 / esz) == 0 && C <= ldx && C <= ldy, "maxpool3s2_fwd: bad shape");
    WU_REQUIRE(al16(x) && al16(y) && (ldx * esz) % 16 == 0 && (ldy * esz) % 16 == 0, "maxpool3s2_fwd: alignment");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long total = (long long)N * Ho * Wo * (C / (16 / esz));
    DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool3s2_fwd_kernel<T>, dim3(grid_cap(total)), dim3(256), 0, (hipStream_t)stream,
                                         (const T*)x, ldx, (T*)y, ldy, argmax, N, H, W, Ho, Wo, C));
    WU_LAUNCH_CHECK("maxpool3s2_fwd");
    return 0;
}

extern "C" int wu_maxpool3s2_bwd(const void* dy, int lddy, const uint8_t* argmax, const void* x, int ldx, void* dx, int lddx,
                                 int N, int H, int W, int C, int gate_act, int dtype, void* stream) {
    const int esz = dtype == WU_BF16 ? 2 : 4;
    WU_REQUIRE(N > 0 && H > 0 && W > 0 && C % (16 / esz) == 0 && argmax, "maxpool3s2_bwd: bad shape");
    WU_REQUIRE(al16(dy) && al16(dx) && (lddy * esz) % 16 == 0 && (lddx * esz) % 16 == 0, "maxpool3s2_bwd: alignment");
    WU_REQUIRE(gate_act == WU_ACT_NONE || (x && al16(x) && (ldx * esz) % 16 == 0), "maxpool3s2_bwd: gate needs x");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long total = (long long)N * H * W * (C / (16 / esz));
    DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool3s2_bwd_kernel<T>, dim3(grid_cap(total)), dim3(256), 0, (hipStream_t)stream,
                                         (const T*)dy, lddy, argmax, (const T*)x, ldx, (T*)dx, lddx, N, H, W, Ho, Wo, C, gate_act));
    WU_LAUNCH_CHECK("maxpool3s2_bwd");
    return 0;
}
