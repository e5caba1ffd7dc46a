// HBM-bound glue of the cUNet hot path: weight repack, MaxPool2d(2), AdaIN statistics, the fused
// AdaIN-apply + bilinear x2 + dropout + concat-slice write (and its backward), sum-pool, layout helpers.
// Every kernel moves 16 B per lane (8 bf16 / 4 fp32 channels of one NHWC pixel) so wave accesses
// are 1 KiB contiguous wherever the tensor is dense.
#include "wu_common.h"

namespace {

// =================================================================================================
// weight repack (nets.py:20,22,28-31 weights, OIHW fp32) -> [tap][Cout][Cin] and [tap'][Cin][Cout]
// =================================================================================================
template <typename T>
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd,
                                    int Cout, int Cin, const float* __restrict__ inv_sigma) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cout * Cin) return;
    const int co = idx / Cin, ci = idx - co * Cin;
    const float s = inv_sigma ? *inv_sigma : 1.f;
    const float* src = w + (size_t)idx * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float v = src[t] * s;
        if (wf) ElemTraits<T>::store(wf + ((size_t)t * Cout + co) * Cin + ci, v);
        if (wd) ElemTraits<T>::store(wd + ((size_t)(8 - t) * Cin + ci) * Cout + co, v);
    }
}

// The same repack for up to 16 weights in ONE launch (a generator forward after an optimizer step repacks all 13 of its MFMA conv
// weights: 13 launches of ~8 us each with the queue's dependency gap between them, against one of ~20 us).  Block b belongs to the
// weight whose block range contains it (prefix sums in the argument struct).
struct PackMulti {
    const float* w[16]; void* wf[16]; void* wd[16];
    int cout[16], cin[16], first_block[17];
    int n;
};
template <typename T>
__global__ __launch_bounds__(256) void pack_conv3x3_multi_kernel(const PackMulti a) {
    // one block = one 32(co) x 32(ci) tile of one weight: the 9 taps of the tile go through LDS so that BOTH images are written in
    // 64-byte row segments (written straight from the OIHW order, the [tap'][ci][co] image is one 2-byte store per lane into 64
    // different cache lines: 54 us for the generator's 13 weights, most of it that scatter)
    __shared__ T tile[9][32][33];
    int k = 0;
#pragma unroll
    for (int i = 1; i < 16; ++i) k += (i < a.n && (int)blockIdx.x >= a.first_block[i]) ? 1 : 0;
    const int Cout = a.cout[k], Cin = a.cin[k];
    const int tiles_ci = (Cin + 31) / 32;
    const int tb = (int)blockIdx.x - a.first_block[k];
    const int co0 = (tb / tiles_ci) * 32, ci0 = (tb % tiles_ci) * 32;
    const float* w = a.w[k];
    T* wf = (T*)a.wf[k];
    T* wd = (T*)a.wd[k];
    for (int idx = threadIdx.x; idx < 1024; idx += 256) {
        const int col = idx >> 5, cil = idx & 31;
        const bool ok = co0 + col < Cout && ci0 + cil < Cin;
        const float* src = w + ((size_t)(co0 + col) * Cin + ci0 + cil) * 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            T v;
            ElemTraits<T>::store(&v, ok ? src[t] : 0.f);
            tile[t][col][cil] = v;
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 9 * 1024; idx += 256) {
        const int t = idx >> 10, r = (idx >> 5) & 31, c = idx & 31;
        // forward image [t][co][ci]: row = co, contiguous over ci
        if (co0 + r < Cout && ci0 + c < Cin) wf[((size_t)t * Cout + co0 + r) * Cin + ci0 + c] = tile[t][r][c];
        // data-gradient image [8 - t][ci][co]: row = ci, contiguous over co
        if (ci0 + r < Cin && co0 + c < Cout) wd[((size_t)(8 - t) * Cin + ci0 + r) * Cout + co0 + c] = tile[t][c][r];
    }
}

// =================================================================================================
// MaxPool2d(2)   (cunet.py:27)
// =================================================================================================
// IDX = unsigned whenever the item count fits (the launchers choose): a 64-bit division / modulo is a ~100-instruction software
// routine, and these kernels decompose their linear index with four of them per 16-byte item
template <typename T, typename IDX>
__global__ void maxpool2_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy,
                                    int N, int H, int W, int C) {
    constexpr int E = ElemTraits<T>::kPer16B;
    const int cpp = C / E, Ho = H / 2, Wo = W / 2;
    const IDX total = (IDX)N * Ho * Wo * cpp;
    for (IDX i = (IDX)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (IDX)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cpp);
        IDX p = i / cpp;
        const int ow = (int)(p % Wo); p /= Wo;
        const int oh = (int)(p % Ho);
        const int n = (int)(p / Ho);
        const T* src = x + ((size_t)(n * H + 2 * oh) * W + 2 * ow) * ldx + ch * E;
        float m[E], v[E];
        unpack16<T>(*(const uint4*)src, m);
        unpack16<T>(*(const uint4*)(src + ldx), v);
#pragma unroll
        for (int e = 0; e < E; ++e) m[e] = (v[e] > m[e] || v[e] != v[e]) ? v[e] : m[e];
        unpack16<T>(*(const uint4*)(src + (size_t)W * ldx), v);
#pragma unroll
        for (int e = 0; e < E; ++e) m[e] = (v[e] > m[e] || v[e] != v[e]) ? v[e] : m[e];
        unpack16<T>(*(const uint4*)(src + (size_t)W * ldx + ldx), v);
#pragma unroll
        for (int e = 0; e < E; ++e) m[e] = (v[e] > m[e] || v[e] != v[e]) ? v[e] : m[e];
        *(uint4*)(y + ((size_t)(n * Ho + oh) * Wo + ow) * ldy + ch * E) = pack16<T>(m);
    }
}

// dx = dskip + route(dy): the window's first maximum (scan order (0,0),(0,1),(1,0),(1,1), strict >,
// PyTorch's rule) receives dy.
template <typename T, typename IDX>
__global__ void maxpool2_bwd_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dy, int lddy,
                                    const T* __restrict__ dskip, int lddskip, T* __restrict__ dx, int lddx,
                                    int N, int H, int W, int C, int gate_act) {
    constexpr int E = ElemTraits<T>::kPer16B;
    const int cpp = C / E, Ho = H / 2, Wo = W / 2;
    const IDX total = (IDX)N * Ho * Wo * cpp;
    for (IDX i = (IDX)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (IDX)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cpp);
        IDX p = i / cpp;
        const int ow = (int)(p % Wo); p /= Wo;
        const int oh = (int)(p % Ho);
        const int n = (int)(p / Ho);
        const size_t pix = (size_t)(n * H + 2 * oh) * W + 2 * ow;
        float v[4][E], g[E];
        const size_t offs[4] = {0, 1, (size_t)W, (size_t)W + 1};
#pragma unroll
        for (int k = 0; k < 4; ++k) unpack16<T>(*(const uint4*)(x + (pix + offs[k]) * ldx + ch * E), v[k]);
        unpack16<T>(*(const uint4*)(dy + ((size_t)(n * Ho + oh) * Wo + ow) * lddy + ch * E), g);
        int arg[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            float m = v[0][e];
            arg[e] = 0;
#pragma unroll
            for (int k = 1; k < 4; ++k)
                if (v[k][e] > m || v[k][e] != v[k][e]) { m = v[k][e]; arg[e] = k; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float o[E];
            if (dskip) unpack16<T>(*(const uint4*)(dskip + (pix + offs[k]) * lddskip + ch * E), o);
#pragma unroll
            for (int e = 0; e < E; ++e) o[e] = act_gate((dskip ? o[e] : 0.f) + (arg[e] == k ? g[e] : 0.f), v[k][e], gate_act);
            *(uint4*)(dx + (pix + offs[k]) * lddx + ch * E) = pack16<T>(o);
        }
    }
}

// The same backward from BITS (round 4): the forward conv's epilogue left, per element of x, the ReLU gate (x > 0) and "first maximum of
// its 2x2 window" (wu_conv3x3_relu_pool_bits_fwd; word layout of the gate bits: uint32 [pixel][C/64][2], bit 8k + i of word (p, ct, hf) =
// channel 64 ct + 16 k + 8 hf + i), so x itself -- 2 bytes per element -- is not read again.  One thread = one 16-byte chunk of one
// full-resolution pixel: dx = gate ? dskip + (sel ? dy[window] : 0) : 0, the arithmetic and the one rounding of the kernel above.
__global__ __launch_bounds__(256) void maxpool2_bwd_bits_kernel(const unsigned* __restrict__ gbits, const unsigned* __restrict__ sbits,
                                                                const bf16_t* __restrict__ dy, int lddy, const bf16_t* __restrict__ dskip, int lddskip,
                                                                bf16_t* __restrict__ dx, int lddx, int N, int H, int W, int C) {
    const int cpp = C >> 3, Ho = H >> 1, Wo = W >> 1;
    const unsigned total = (unsigned)N * H * W * cpp;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const int ch = (int)(i % cpp);
        const unsigned pix = i / cpp;                                   // (n, h, w) linear
        const int w_ = (int)(pix % W);
        const unsigned t = pix / W;
        const int h_ = (int)(t % H), n = (int)(t / H);
        const int c0 = ch * 8;
        const unsigned wi = pix * (unsigned)(C >> 5) + 2 * (c0 >> 6) + ((c0 >> 3) & 1);     // word (pixel, ct, hf); C/64 * 2 words per pixel
        const int sh = 8 * ((c0 & 63) >> 4);
        const unsigned gate = (gbits[wi] >> sh) & 0xffu, sel = (sbits[wi] >> sh) & 0xffu;
        float g[8], o[8];
        unpack16<bf16_t>(*(const uint4*)(dy + ((size_t)(n * Ho + (h_ >> 1)) * Wo + (w_ >> 1)) * lddy + c0), g);
        if (dskip) unpack16<bf16_t>(*(const uint4*)(dskip + (size_t)pix * lddskip + c0), o);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = (dskip ? o[e] : 0.f) + (((sel >> e) & 1u) ? g[e] : 0.f);
            o[e] = ((gate >> e) & 1u) ? v : 0.f;
        }
        *(uint4*)(dx + (size_t)pix * lddx + c0) = pack16<bf16_t>(o);
    }
}

// =================================================================================================
// AdaIN statistics (utils.py:34-39): shifted sums per (n,c), finalised to {mean, rstd}
// =================================================================================================
constexpr int kMaxSplits = 16;   // partial-sum slots per (n,c); scratch buffers are sized for this

// deterministic block reduction: every pixel-lane writes its channel partials to LDS, then 128 threads
// (64 channels x 2 quantities) add the pixel-lanes in fixed order.
template <int PP>
__device__ __forceinline__ void block_reduce_2x64(float (*red)[64][2], int tid, float* __restrict__ dst, int dst_stride) {
    __syncthreads();
    if (tid < 128) {
        const int c = tid >> 1, j = tid & 1;
        float s = 0.f;
#pragma unroll 4
        for (int p = 0; p < PP; ++p) s += red[p][c][j];
        dst[(size_t)c * dst_stride + j] = s;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void adain_stats_kernel(const T* __restrict__ x, int ldx, float* __restrict__ scratch,
                                                          int HW, int C, int splits) {
    constexpr int E = ElemTraits<T>::kPer16B;
    constexpr int LP = 64 / E;         // lanes per pixel (64 channels per workgroup)
    constexpr int PP = 256 / LP;       // pixels per iteration
    __shared__ float red[PP][64][2];
    const int tid = threadIdx.x;
    const int cg = blockIdx.x, split = blockIdx.y, n = blockIdx.z;
    const int cl = tid % LP, pl = tid / LP;
    const T* base = x + (size_t)n * HW * ldx + cg * 64 + cl * E;
    float k[E], s1[E], s2[E], v[E];
    unpack16<T>(*(const uint4*)base, k);   // shift = the value at pixel 0 (kills the cancellation)
#pragma unroll
    for (int e = 0; e < E; ++e) s1[e] = s2[e] = 0.f;
    const int per = (HW + splits - 1) / splits;
    const int p0 = split * per, p1 = min(HW, p0 + per);
    for (int p = p0 + pl; p < p1; p += PP) {
        unpack16<T>(*(const uint4*)(base + (size_t)p * ldx), v);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const float d = v[e] - k[e];
            s1[e] += d;
            s2[e] += d * d;
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        red[pl][cl * E + e][0] = s1[e];
        red[pl][cl * E + e][1] = s2[e];
    }
    // scratch layout [n][c][split][2]
    block_reduce_2x64<PP>(red, tid, scratch + (((size_t)n * C + cg * 64) * splits + split) * 2, splits * 2);
}

template <typename T>
__global__ void adain_stats_final_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ scratch,
                                         float* __restrict__ stats, int N, int HW, int C, float eps, int splits) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    const float k = ElemTraits<T>::load(x + (size_t)n * HW * ldx + c);
    float s1 = 0.f, s2 = 0.f;
    for (int sp = 0; sp < splits; ++sp) {
        s1 += scratch[((size_t)i * splits + sp) * 2];
        s2 += scratch[((size_t)i * splits + sp) * 2 + 1];
    }
    const float cnt = (float)HW;
    const float mean = k + s1 / cnt;
    const float var = fmaxf((s2 - s1 * s1 / cnt) / (cnt - 1.f), 0.f);   // unbiased (torch.var default), utils.py:36
    stats[2 * i] = mean;
    stats[2 * i + 1] = 1.f / sqrtf(var + eps);
}

// =================================================================================================
// fused AdaIN apply + bilinear x2 (align_corners) + dropout -> concat slice   (cunet.py:59-62)
// =================================================================================================
struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp src_index(int dst, float scale, int in_size) {
    // PyTorch area_pixel_compute_source_index(align_corners=True): src = scale * dst, in fp32
    const float src = scale * (float)dst;
    Lerp r;
    r.i0 = (int)src;
    r.i1 = r.i0 + (r.i0 < in_size - 1 ? 1 : 0);
    r.l1 = src - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}

// keep-mask bits for the 8 (or 4) channels starting at NHWC linear index idx (idx % 4 == 0)
template <int E> __device__ __forceinline__ void keep_bits(uint64_t seed, uint64_t idx, uint32_t thr, bool* keep) {
#pragma unroll
    for (int q = 0; q < E / 4; ++q) {
        const uint64_t r = wu_rand4(seed, (idx >> 2) + q);
#pragma unroll
        for (int e = 0; e < 4; ++e) keep[4 * q + e] = (uint32_t)((r >> (16 * e)) & 0xffff) < thr;
    }
}

// Round 4: the same eight keep decisions as LANE MASKS of the packed bf16 chunk (dword i = elements 2i, 2i + 1: 0xffff per kept half) and as the keep byte,
// without materialising eight bools (the compiler turned those into ~50 cndmask / shift / or instructions per chunk, more than the two splitmix64 draws,
// in a kernel that issues VALU instructions back to back).  field < thr (unsigned 16-bit) == (field ^ 0x8000) < (thr ^ 0x8000) signed; the saturating packed
// subtract keeps the sign, the packed arithmetic shift spreads it: three packed-int16 instructions per two elements.  thr <= 0xffff here.
typedef short wu_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t keep_mask2(uint32_t fields, uint32_t thr_biased2) {
    const wu_s16x2 x = __builtin_bit_cast(wu_s16x2, fields ^ 0x80008000u);
    const wu_s16x2 d = __builtin_elementwise_sub_sat(x, __builtin_bit_cast(wu_s16x2, thr_biased2));
    return __builtin_bit_cast(uint32_t, d >> (wu_s16x2)15);
}
__device__ __forceinline__ void keep_masks8(uint64_t seed, uint64_t idx, uint32_t thr, uint32_t (&m)[4], uint32_t& bits) {
    const uint32_t tb = (thr ^ 0x8000u) & 0xffffu, tb2 = tb | (tb << 16);
    const uint64_t r0 = wu_rand4(seed, idx >> 2), r1 = wu_rand4(seed, (idx >> 2) + 1);
    m[0] = keep_mask2((uint32_t)r0, tb2);
    m[1] = keep_mask2((uint32_t)(r0 >> 32), tb2);
    m[2] = keep_mask2((uint32_t)r1, tb2);
    m[3] = keep_mask2((uint32_t)(r1 >> 32), tb2);
    const uint32_t v = (m[0] & 0x00020001u) | (m[1] & 0x00080004u) | (m[2] & 0x00200010u) | (m[3] & 0x00800040u);
    bits = (v | (v >> 16)) & 0xffu;
}
// keep byte -> the four lane masks (caller-supplied masks)
__device__ __forceinline__ void keep_masks_from_bits(uint32_t bits, uint32_t (&m)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t lo = (uint32_t)(-(int)((bits >> (2 * i)) & 1u)), hi = (uint32_t)(-(int)((bits >> (2 * i + 1)) & 1u));
        m[i] = (lo & 0x0000ffffu) | (hi & 0xffff0000u);
    }
}

// load E consecutive floats (E = 4 or 8) with 16-B loads
template <int E> __device__ __forceinline__ void ldf(const float* __restrict__ p, float* o) {
#pragma unroll
    for (int e = 0; e < E; e += 4) {
        const float4 v = *(const float4*)(p + e);
        o[e] = v.x; o[e + 1] = v.y; o[e + 2] = v.z; o[e + 3] = v.w;
    }
}

// grid: x = 256-thread chunks of one output row (W2 * C/E items), y = output rows (n, oh) (strided)
template <typename T>
__global__ __launch_bounds__(256) void adain_upcat_fwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ stats,
                                       const float* __restrict__ y_std, const float* __restrict__ y_mean,
                                       T* __restrict__ y, int ldy, int N, int H, int W, int C,
                                       float sy, float sx, uint32_t thr, float keep_scale, uint64_t seed,
                                       const uint64_t* __restrict__ seed_dev, uint8_t* __restrict__ mbits, int mask_in) {
    constexpr int E = ElemTraits<T>::kPer16B;
    const int cpp = C / E, H2 = 2 * H, W2 = 2 * W;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= W2 * cpp) return;
    // graph-safe per-replay seed: the captured kernel argument is frozen, the device-resident counter is not
    if (seed_dev) seed += *seed_dev;
    const int ow = idx / cpp, ch = idx - ow * cpp;
    const Lerp lx = src_index(ow, sx, W);
    constexpr int RPT = 4;                           // consecutive output rows per thread (same image)
    const int groups = (H2 + RPT - 1) / RPT;
    for (int gy = blockIdx.y; gy < N * groups; gy += gridDim.y) {
        const int n = gy / groups, oh_base = (gy - n * groups) * RPT;
        const int sc = n * C + ch * E;
        float st[2 * E], ys[E], ym[E], ka[E], kb[E];
        ldf<2 * E>(stats + 2 * sc, st);
        ldf<E>(y_std + sc, ys);
        ldf<E>(y_mean + sc, ym);
#pragma unroll
        for (int e = 0; e < E; ++e) {                // out = v * ka + kb   (utils.py:49-50 folded)
            ka[e] = st[2 * e + 1] * ys[e];
            kb[e] = ym[e] - st[2 * e] * ka[e];
        }
        const T* b = x + (size_t)n * H * W * ldx + ch * E;
        // all 4 x RPT neighbour loads are issued before any is consumed (rows past the image are clamped and skipped
        // at the store): one memory round trip per RPT output rows instead of one per row
        Lerp lys[RPT];
        uint4 q[RPT][4];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int oh = min(oh_base + r, H2 - 1);
            lys[r] = src_index(oh, sy, H);
            q[r][0] = *(const uint4*)(b + (size_t)(lys[r].i0 * W + lx.i0) * ldx);
            q[r][1] = *(const uint4*)(b + (size_t)(lys[r].i0 * W + lx.i1) * ldx);
            q[r][2] = *(const uint4*)(b + (size_t)(lys[r].i1 * W + lx.i0) * ldx);
            q[r][3] = *(const uint4*)(b + (size_t)(lys[r].i1 * W + lx.i1) * ldx);
        }
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int oh = oh_base + r;
            if (oh < H2) {
                const Lerp ly = lys[r];
                float v00[E], v01[E], v10[E], v11[E], o[E];
                unpack16<T>(q[r][0], v00);
                unpack16<T>(q[r][1], v01);
                unpack16<T>(q[r][2], v10);
                unpack16<T>(q[r][3], v11);
                const size_t opix = ((size_t)n * H2 + oh) * W2 + ow;
                bool keep[E];
                if (thr < 0x10000u && mask_in) {       // caller-supplied keep bits (a mask captured from the reference)
                    const uint32_t bits = mbits[opix * cpp + ch];
#pragma unroll
                    for (int e = 0; e < E; ++e) keep[e] = (bits >> e) & 1u;
                } else if (thr < 0x10000u) {
                    keep_bits<E>(seed, (uint64_t)opix * C + ch * E, thr, keep);
                    if (mbits) {        // one byte per 16-B chunk: backward reads the mask instead of re-hashing
                        uint32_t bits = 0;
#pragma unroll
                        for (int e = 0; e < E; ++e) bits |= (keep[e] ? 1u : 0u) << e;
                        mbits[opix * cpp + ch] = (uint8_t)bits;
                    }
                }
                // the kernel is VALU-bound: 4 products with pre-multiplied tap weights (+1 FMA for the AdaIN affine) instead
                // of the 6-op nested form; the dropout scale is folded into the affine coefficients
                const float w00 = ly.l0 * lx.l0, w01 = ly.l0 * lx.l1, w10 = ly.l1 * lx.l0, w11 = ly.l1 * lx.l1;
                const float ds = (thr < 0x10000u) ? keep_scale : 1.f;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const float v = w00 * v00[e] + w01 * v01[e] + w10 * v10[e] + w11 * v11[e];
                    const float rr = v * (ka[e] * ds) + kb[e] * ds;
                    o[e] = (thr < 0x10000u && !keep[e]) ? 0.f : rr;               // nn.Dropout(p) train mode
                }
                *(uint4*)(y + opix * ldy + ch * E) = pack16<T>(o);
            }
        }
    }
}

// forward, second formulation ("marching"): a thread owns TWO adjacent output columns of one 16-byte channel chunk and walks down the
// output rows of its strip.  The two columns interpolate from at most three source columns; per SOURCE row those three chunks are
// loaded once, mapped through the AdaIN affine (it commutes with the interpolation: the weights sum to one) and reduced
// horizontally; every output row is then a two-term vertical blend of the current and the next reduced source row.  Per output
// chunk: 0.75 loads instead of 4, one unpack per source value instead of one per tap.  Same values as adain_upcat_fwd_kernel up to
// fp32 rounding order (the affine is applied before instead of after the interpolation).
template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void adain_upcat_fwd_march_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ stats,
                                       const float* __restrict__ y_std, const float* __restrict__ y_mean,
                                       T* __restrict__ y, int ldy, int N, int H, int W, int C, int rows_per_strip, int col_tiles,
                                       float sy, float sx, uint32_t thr, float keep_scale, uint64_t seed,
                                       const uint64_t* __restrict__ seed_dev, uint8_t* __restrict__ mbits, int mask_in) {
    constexpr int E = ElemTraits<T>::kPer16B;
    constexpr int LP = 64 / E, PP = 256 / LP;
    const int tid = threadIdx.x;
    // XCD-aware decode (round 4): workgroups are dealt round-robin over the eight XCDs by their linear id; logical ids are remapped so that an XCD owns a
    // contiguous range -- the channel groups of one pixel tile (they write interleaved 8-byte pieces of the same keep-byte lines: on two XCDs every
    // such line left two L2s as two partial-line writes) and neighbouring tiles (shared source rows / columns) meet in one L2, dispatched together
    // Measured (same box, profiles/r04_upcat_fwd.txt): 45 -> 38.6 us at C = 512 (eight groups per keep-byte line), no change at C = 256, 184 -> 193 us at C = 128
    // (the data stream itself gets slower when an XCD's tiles are neighbours): remapped from eight channel groups up.
    int bid = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
    if (gridDim.x >= 8) bid = xcd_remap(bid, (int)(gridDim.x * gridDim.y * gridDim.z));
    const int cg = bid % (int)gridDim.x; bid /= (int)gridDim.x;
    const int by = bid % (int)gridDim.y, n = bid / (int)gridDim.y;
    const int ctile = by % col_tiles, strip = by / col_tiles;
    const int cl = tid % LP, pl = tid / LP;
    const int c0 = cg * 64 + cl * E;
    const int cpp = C / E, chunk = c0 / E;
    const int H2 = 2 * H, W2 = 2 * W;
    const int j0 = 2 * (ctile * PP + pl), j1 = j0 + 1;                 // this thread's two output columns
    const bool mask_store8 = E == 8 && ((uintptr_t)mbits & 7) == 0;    // (uniform) keep bytes leave as 8-byte words (cpp % 8 == 0: C % 64 == 0 here)
    if (j0 >= W2) return;
    const bool v1 = j1 < W2;
    if (seed_dev) seed += *seed_dev;
    const int r0 = strip * rows_per_strip, r1 = min(H2, r0 + rows_per_strip);
    if (r0 >= r1) return;
    // AdaIN affine of this chunk (utils.py:49-50 folded), with the dropout keep-scale folded in
    float ka[E], kb[E];
    {
        float st[2 * E], ys[E], ym[E];
        const int sc = n * C + c0;
        ldf<2 * E>(stats + 2 * sc, st);
        ldf<E>(y_std + sc, ys);
        ldf<E>(y_mean + sc, ym);
        const float ds = thr < 0x10000u ? keep_scale : 1.f;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const float a = st[2 * e + 1] * ys[e];
            ka[e] = a * ds;
            kb[e] = (ym[e] - st[2 * e] * a) * ds;
        }
    }
    // source columns: c, c+1, c+2 (clamped); column j0 blends (c, c+1) or ... expressed as three weights per output column
    const Lerp lx0 = src_index(j0, sx, W), lx1 = src_index(min(j1, W2 - 1), sx, W);
    const int cb = lx0.i0;
    int cx[3];
    float w0[3], w1[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        cx[k] = min(cb + k, W - 1);
        w0[k] = (lx0.i0 == cb + k ? lx0.l0 : 0.f) + (lx0.i1 == cb + k && lx0.i1 != lx0.i0 ? lx0.l1 : 0.f) + (lx0.i1 == lx0.i0 && lx0.i0 == cb + k ? lx0.l1 : 0.f);
        w1[k] = (lx1.i0 == cb + k ? lx1.l0 : 0.f) + (lx1.i1 == cb + k && lx1.i1 != lx1.i0 ? lx1.l1 : 0.f) + (lx1.i1 == lx1.i0 && lx1.i0 == cb + k ? lx1.l1 : 0.f);
    }
    const T* base = x + (size_t)n * H * W * ldx + c0;
    // reduced source rows: A = row `cur`, Bn = row `cur + 1` (two output columns each)
    float A0[E], A1[E], B0[E], B1[E];
    auto reduce_row = [&](const uint4 (&q)[3], float* o0, float* o1) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < E; ++e) o0[e] = o1[e] = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float v[E];
            unpack16<T>(q[k], v);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float t = fmaf(v[e], ka[e], kb[e]);
                o0[e] = fmaf(w0[k], t, o0[e]);
                o1[e] = fmaf(w1[k], t, o1[e]);
            }
        }
    };
    auto load_row = [&](int row, uint4 (&q)[3]) __attribute__((always_inline)) {
        const int rr = min(row, H - 1);
#pragma unroll
        for (int k = 0; k < 3; ++k) q[k] = *(const uint4*)(base + (size_t)(rr * W + cx[k]) * ldx);
    };
    int cur = src_index(r0, sy, H).i0;
    uint4 q[3], qn[3];
    load_row(cur, q);
    reduce_row(q, A0, A1);
    load_row(cur + 1, q);
    reduce_row(q, B0, B1);
    load_row(cur + 2, qn);                                             // one source row ahead
    for (int r = r0; r < r1; ++r) {
        const Lerp ly = src_index(r, sy, H);
        if (ly.i0 > cur) {                                             // advances by at most one source row per output row
#pragma unroll
            for (int e = 0; e < E; ++e) { A0[e] = B0[e]; A1[e] = B1[e]; }
            reduce_row(qn, B0, B1);
            ++cur;
            load_row(cur + 2, qn);
        }
        const float l1w = ly.i1 != ly.i0 ? ly.l1 : 0.f, l0w = ly.i1 != ly.i0 ? ly.l0 : ly.l0 + ly.l1;
        const size_t rowpix = ((size_t)n * H2 + r) * W2;
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            if (side && !v1) continue;
            const float* a_ = side ? A1 : A0;
            const float* b_ = side ? B1 : B0;
            const size_t opix = rowpix + (side ? j1 : j0);
            float o[E];
#pragma unroll
            for (int e = 0; e < E; ++e) o[e] = fmaf(l1w, b_[e], l0w * a_[e]);
            if constexpr (E == 8) {
                // bf16: the keep decisions as lane masks ANDed onto the PACKED chunk (a dropped element is +0, as `keep ? o : 0.f` packs it)
                uint4 pk = pack16<T>(o);
                if (thr < 0x10000u) {
                    uint32_t m[4], bits;
                    if (mask_in) {
                        bits = mbits[opix * cpp + chunk];
                        keep_masks_from_bits(bits, m);
                    } else {
                        keep_masks8(seed, (uint64_t)opix * C + c0, thr, m, bits);
                        if (mbits) {
                            if (mask_store8) {
                                // the eight chunk lanes of a pixel hold eight consecutive keep bytes: gathered into lane 0 of the group (two quad
                                // DPP ORs + one row shift) and stored as ONE 8-byte word per pixel and 64-channel group instead of 64 byte stores per wave
                                uint32_t v = bits << (8 * (cl & 3));
                                v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]
                                v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);      // quad_perm [2,3,0,1]
                                const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xf, 0xf, true);   // row_shl:4 -> lane l reads lane l + 4
                                if (cl == 0) *(uint2*)(mbits + opix * cpp + chunk) = make_uint2(v, hi);
                            } else {
                                mbits[opix * cpp + chunk] = (uint8_t)bits;
                            }
                        }
                    }
                    pk.x &= m[0]; pk.y &= m[1]; pk.z &= m[2]; pk.w &= m[3];
                }
                *(uint4*)(y + opix * ldy + c0) = pk;
            } else {
                if (thr < 0x10000u) {
                    bool keep[E];
                    if (mask_in) {
                        const uint32_t bits = mbits[opix * cpp + chunk];
#pragma unroll
                        for (int e = 0; e < E; ++e) keep[e] = (bits >> e) & 1u;
                    } else {
                        keep_bits<E>(seed, (uint64_t)opix * C + c0, thr, keep);
                        if (mbits) {
                            uint32_t bits = 0;
#pragma unroll
                            for (int e = 0; e < E; ++e) bits |= (keep[e] ? 1u : 0u) << e;
                            mbits[opix * cpp + chunk] = (uint8_t)bits;
                        }
                    }
#pragma unroll
                    for (int e = 0; e < E; ++e) o[e] = keep[e] ? o[e] : 0.f;
                }
                *(uint4*)(y + opix * ldy + c0) = pack16<T>(o);
            }
        }
    }
}

// backward stage A: g'[n,y,x,c] = sum over the output pixels that interpolate from (y,x) of
// weight * dropout * dy  (the gradient wrt the AdaIN output), plus per-(n,c) sums of g' and g'*xhat.
template <typename T>
__global__ __launch_bounds__(256) void adain_upcat_bwd_gather_kernel(
    const T* __restrict__ dy, int lddy, const T* __restrict__ x, int ldx, const float* __restrict__ stats,
    T* __restrict__ gtmp, float* __restrict__ sums, int H, int W, int C,
    float sy, float sx, uint32_t thr, float keep_scale, uint64_t seed, const uint8_t* __restrict__ mbits) {
    constexpr int E = ElemTraits<T>::kPer16B;
    constexpr int LP = 64 / E, PP = 256 / LP;
    __shared__ float red[PP][64][2];
    // keep-byte -> four dwords of 16-bit lane masks (bf16 only): the dropout mask is ANDed onto the PACKED gradient chunk
    // (1 LDS read + 4 v_and per tap) instead of a bit test + compare + select per element and tap
    __shared__ uint4 lut[E == 8 ? 256 : 1];
    const int tid = threadIdx.x;
    if (E == 8 && thr < 0x10000u && mbits) {
        uint32_t m[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) m[k] = ((tid >> (2 * k)) & 1 ? 0x0000ffffu : 0u) | ((tid >> (2 * k + 1)) & 1 ? 0xffff0000u : 0u);
        lut[tid] = make_uint4(m[0], m[1], m[2], m[3]);
        __syncthreads();
    }
    const int cg = blockIdx.x, n = blockIdx.z;
    const int cl = tid % LP, pl = tid / LP;
    const int c0 = cg * 64 + cl * E;
    const int cpp = C / E, chunk = c0 / E;
    const int H2 = 2 * H, W2 = 2 * W, HW = H * W;
    float s1[E], s2[E], st[2 * E];
    ldf<2 * E>(stats + 2 * (n * C + c0), st);
#pragma unroll
    for (int e = 0; e < E; ++e) s1[e] = s2[e] = 0.f;
    const int per = (HW + gridDim.y - 1) / gridDim.y;
    const int p0 = blockIdx.y * per, p1 = min(HW, p0 + per);
    for (int p = p0 + pl; p < p1; p += PP) {
        const int yy = p / W, xx = p - yy * W;
        float g[E];
#pragma unroll
        for (int e = 0; e < E; ++e) g[e] = 0.f;
        // The output pixels that interpolate from (yy, xx) are rows 2yy-1 .. 2yy+2 x cols 2xx-1 .. 2xx+2 (scale
        // (H-1)/(2H-1) < 1/2): 16 UNCONDITIONAL loads (clamped coordinates, zero weight outside) so all of them
        // are in flight together instead of one dependent load per taken branch.
        float wys[4], wxs[4];
        int iys[4], jxs[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = 2 * yy - 1 + k, j = 2 * xx - 1 + k;
            const int ic = min(max(i, 0), H2 - 1), jc = min(max(j, 0), W2 - 1);
            const Lerp ly = src_index(ic, sy, H), lx = src_index(jc, sx, W);
            wys[k] = (i == ic) ? ((ly.i0 == yy ? ly.l0 : 0.f) + (ly.i1 == yy ? ly.l1 : 0.f)) : 0.f;
            wxs[k] = (j == jc) ? ((lx.i0 == xx ? lx.l0 : 0.f) + (lx.i1 == xx ? lx.l1 : 0.f)) : 0.f;
            iys[k] = ic;
            jxs[k] = jc;
        }
        uint4 dv[16];
        uint32_t bv[16];
#pragma unroll
        for (int ka = 0; ka < 4; ++ka)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const size_t opix = (size_t)(n * H2 + iys[ka]) * W2 + jxs[kb];
                dv[ka * 4 + kb] = *(const uint4*)(dy + opix * lddy + c0);
                if (thr < 0x10000u && mbits) bv[ka * 4 + kb] = mbits[opix * cpp + chunk];
            }
#pragma unroll
        for (int ka = 0; ka < 4; ++ka)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const float wgt = wys[ka] * wxs[kb];
                float d[E];
                unpack16<T>(dv[ka * 4 + kb], d);
                if (thr < 0x10000u) {
                    if (mbits && E == 8) {
                        const uint4 mk = lut[bv[ka * 4 + kb] & 255u];
                        const uint4 dm = make_uint4(dv[ka * 4 + kb].x & mk.x, dv[ka * 4 + kb].y & mk.y, dv[ka * 4 + kb].z & mk.z, dv[ka * 4 + kb].w & mk.w);
                        unpack16<T>(dm, d);
#pragma unroll
                        for (int e = 0; e < E; ++e) g[e] = fmaf(wgt, d[e], g[e]);
                    } else if (mbits) {
                        const uint32_t bits = bv[ka * 4 + kb];
#pragma unroll
                        for (int e = 0; e < E; ++e) g[e] = ((bits >> e) & 1u) ? fmaf(wgt, d[e], g[e]) : g[e];
                    } else {
                        bool keep[E];
                        const size_t opix = (size_t)(n * H2 + iys[ka]) * W2 + jxs[kb];
                        keep_bits<E>(seed, (uint64_t)opix * C + c0, thr, keep);
#pragma unroll
                        for (int e = 0; e < E; ++e) g[e] = keep[e] ? fmaf(wgt, d[e], g[e]) : g[e];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < E; ++e) g[e] = fmaf(wgt, d[e], g[e]);
                }
            }
        if (thr < 0x10000u) {
#pragma unroll
            for (int e = 0; e < E; ++e) g[e] *= keep_scale;
        }
        float xv[E];
        unpack16<T>(*(const uint4*)(x + ((size_t)n * HW + p) * ldx + c0), xv);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            s1[e] += g[e];
            s2[e] += g[e] * ((xv[e] - st[2 * e]) * st[2 * e + 1]);
        }
        // g' is parked in the storage dtype (its statistics above were taken in fp32 before rounding)
        *(uint4*)(gtmp + ((size_t)n * HW + p) * C + c0) = pack16<T>(g);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        red[pl][cl * E + e][0] = s1[e];
        red[pl][cl * E + e][1] = s2[e];
    }
    // partials layout [n][c][split][2] (after the final [N][C][2] block)
    const int splits = gridDim.y;
    block_reduce_2x64<PP>(red, tid, sums + (((size_t)n * C + cg * 64) * splits + blockIdx.y) * 2, splits * 2);
}

// backward stage A, second formulation ("marching"): a thread owns TWO adjacent low-resolution columns of one 16-byte channel chunk and
// walks down the high-resolution rows of its strip.  Per high-res row it loads the 6 columns that touch its two low-res columns
// once (the 16-tap gather above loads 16 chunks per low-res pixel, this one 3), reduces them horizontally with row-independent
// weights, and adds the result into the (at most two) low-res rows that row interpolates from -- the bilinear x2 operator is
// separable, so its transpose is too.  Same sums, same parked g', same per-(n,c) statistics; the partial-sum slot of a workgroup is
// its (column tile, row strip) index.
// NCOL = low-resolution columns per thread.  2 (round 2): 6 loads per high-res row serve two columns (3 per column), but the kernel
// needs 208 registers -> two waves per SIMD, and it is latency-bound (3 TB/s).  1 (round 3): 4 loads per row and column, half the
// accumulators and row buffers -> twice the waves, each with its own row in flight.  Same sums in the same order per column.
template <typename T, int NCOL>
__global__ __launch_bounds__(256) void adain_upcat_bwd_march_kernel(
    const T* __restrict__ dy, int lddy, const T* __restrict__ x, int ldx, const float* __restrict__ stats,
    T* __restrict__ gtmp, float* __restrict__ sums, int H, int W, int C, int rows_per_strip, int col_tiles,
    float sy, float sx, uint32_t thr, float keep_scale, uint64_t seed, const uint8_t* __restrict__ mbits) {
    constexpr int E = ElemTraits<T>::kPer16B;
    constexpr int LP = 64 / E, PP = 256 / LP;          // lanes per pixel (64 channels), column groups per workgroup
    constexpr int NK = 2 * NCOL + 2;                   // high-res columns that touch a thread's NCOL low-res columns
    __shared__ float red[PP][64][2];
    __shared__ uint4 lut[E == 8 ? 256 : 1];
    const int tid = threadIdx.x;
    if (E == 8 && thr < 0x10000u && mbits) {
        uint32_t m[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) m[k] = ((tid >> (2 * k)) & 1 ? 0x0000ffffu : 0u) | ((tid >> (2 * k + 1)) & 1 ? 0xffff0000u : 0u);
        lut[tid] = make_uint4(m[0], m[1], m[2], m[3]);
        __syncthreads();
    }
    const int cg = blockIdx.x, n = blockIdx.z;
    const int ctile = blockIdx.y % col_tiles, strip = blockIdx.y / col_tiles;
    const int cl = tid % LP, pl = tid / LP;
    const int c0 = cg * 64 + cl * E;
    const int cpp = C / E, chunk = c0 / E;
    const int H2 = 2 * H, W2 = 2 * W, HW = H * W;
    const int x0 = NCOL * (ctile * PP + pl);                           // this thread's first low-res column
    const int y0 = strip * rows_per_strip, y1 = min(H, y0 + rows_per_strip);
    float s1[E], s2[E];
#pragma unroll
    for (int e = 0; e < E; ++e) s1[e] = s2[e] = 0.f;
    // the NK high-res columns 2*x0 - 1 .. 2*x0 + 2*NCOL and their weights towards column x0 + c (columns k = 2c .. 2c + 3)
    int jx[NK];
    float wgt[NCOL][4];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int j = 2 * x0 - 1 + k;
        const int jc = min(max(j, 0), W2 - 1);
        const Lerp lx = src_index(jc, sx, W);
        const bool in = j == jc;
#pragma unroll
        for (int c = 0; c < NCOL; ++c)
            if (k >= 2 * c && k < 2 * c + 4) {
                const int xc = x0 + c;
                wgt[c][k - 2 * c] = in ? ((lx.i0 == xc ? lx.l0 : 0.f) + (lx.i1 == xc ? lx.l1 : 0.f)) : 0.f;
            }
        jx[k] = jc;
    }
    // two pending low-res rows per column: acc0 = row `cur`, acc1 = row `cur + 1`
    float acc0[NCOL][E], acc1[NCOL][E];
#pragma unroll
    for (int c = 0; c < NCOL; ++c)
#pragma unroll
        for (int e = 0; e < E; ++e) acc0[c][e] = acc1[c][e] = 0.f;
    const int r_begin = max(0, 2 * y0 - 1), r_end = min(H2 - 1, 2 * y1);           // rows that can touch [y0, y1)
    int cur = src_index(r_begin, sy, H).i0;

    auto emit = [&](int yy, const float (&ga)[NCOL][E]) __attribute__((always_inline)) {
        if (yy < y0 || yy >= y1) return;
        const size_t p = (size_t)n * HW + (size_t)yy * W;
        float st[2 * E];
        ldf<2 * E>(stats + 2 * (n * C + c0), st);
#pragma unroll
        for (int c = 0; c < NCOL; ++c) {
            const int xx = x0 + c;
            if (xx >= W) continue;
            float gs[E], xv[E];
            unpack16<T>(*(const uint4*)(x + (p + xx) * ldx + c0), xv);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                gs[e] = thr < 0x10000u ? ga[c][e] * keep_scale : ga[c][e];
                s1[e] += gs[e];
                s2[e] += gs[e] * ((xv[e] - st[2 * e]) * st[2 * e + 1]);
            }
            *(uint4*)(gtmp + (p + xx) * C + c0) = pack16<T>(gs);
        }
    };

    if (x0 < W) {
        // the loads of row r + 1 are issued before row r is reduced: one memory round trip per row is always in flight
        uint4 dvn[NK];
        uint32_t bvn[NK];
        auto load_row = [&](int r) __attribute__((always_inline)) {
            const size_t rp = (size_t)(n * H2 + r) * W2;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                dvn[k] = *(const uint4*)(dy + (rp + jx[k]) * lddy + c0);
                bvn[k] = (thr < 0x10000u && mbits) ? mbits[(rp + jx[k]) * cpp + chunk] : 0u;
            }
        };
        load_row(r_begin);
        for (int r = r_begin; r <= r_end; ++r) {
            const Lerp ly = src_index(r, sy, H);
            uint4 dv[NK];
            uint32_t bv[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k) { dv[k] = dvn[k]; bv[k] = bvn[k]; }
            load_row(min(r + 1, r_end));
            // retire the rows no later high-res row can touch
            while (cur < ly.i0) {
                emit(cur, acc0);
#pragma unroll
                for (int c = 0; c < NCOL; ++c)
#pragma unroll
                    for (int e = 0; e < E; ++e) { acc0[c][e] = acc1[c][e]; acc1[c][e] = 0.f; }
                ++cur;
            }
            const size_t rowpix = (size_t)(n * H2 + r) * W2;
            float t[NCOL][E];
#pragma unroll
            for (int c = 0; c < NCOL; ++c)
#pragma unroll
                for (int e = 0; e < E; ++e) t[c][e] = 0.f;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                float d[E];
                if (thr < 0x10000u) {
                    if (mbits && E == 8) {
                        const uint4 mk = lut[bv[k] & 255u];
                        unpack16<T>(make_uint4(dv[k].x & mk.x, dv[k].y & mk.y, dv[k].z & mk.z, dv[k].w & mk.w), d);
                    } else if (mbits) {
                        unpack16<T>(dv[k], d);
#pragma unroll
                        for (int e = 0; e < E; ++e) d[e] = ((bv[k] >> e) & 1u) ? d[e] : 0.f;
                    } else {
                        bool keep[E];
                        keep_bits<E>(seed, (uint64_t)(rowpix + jx[k]) * C + c0, thr, keep);
                        unpack16<T>(dv[k], d);
#pragma unroll
                        for (int e = 0; e < E; ++e) d[e] = keep[e] ? d[e] : 0.f;
                    }
                } else {
                    unpack16<T>(dv[k], d);
                }
#pragma unroll
                for (int c = 0; c < NCOL; ++c)
                    if (k >= 2 * c && k < 2 * c + 4) {
#pragma unroll
                        for (int e = 0; e < E; ++e) t[c][e] = fmaf(wgt[c][k - 2 * c], d[e], t[c][e]);
                    }
            }
            // vertical: this row interpolates from low-res rows i0 (weight l0) and i1 (weight l1; i1 == i0 on the last row)
            const float l1w = ly.i1 != ly.i0 ? ly.l1 : 0.f, l0w = ly.i1 != ly.i0 ? ly.l0 : ly.l0 + ly.l1;
#pragma unroll
            for (int c = 0; c < NCOL; ++c)
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    acc0[c][e] = fmaf(l0w, t[c][e], acc0[c][e]);
                    acc1[c][e] = fmaf(l1w, t[c][e], acc1[c][e]);
                }
        }
        emit(cur, acc0);
        emit(cur + 1, acc1);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        red[pl][cl * E + e][0] = s1[e];
        red[pl][cl * E + e][1] = s2[e];
    }
    const int splits = gridDim.y;
    block_reduce_2x64<PP>(red, tid, sums + (((size_t)n * C + cg * 64) * splits + blockIdx.y) * 2, splits * 2);
}

// backward stage A, third formulation (round 4, bf16): the strip is STREAMED THROUGH AN LDS RING by LDS-DMA.
// The marching kernel above is bound by what a wave keeps in flight, not by bytes (3 TB/s; SQ_WAIT_ANY 58-62 %: one high-res row of
// 4-6 register loads per thread, retired with vmcnt(0) every row).  Here a workgroup (8 chunk lanes x 32 low-res columns) owns a
// (64-channel, 32-column, row-strip) tile and walks it in STEPS of two high-res rows; a step's operands -- two 66-pixel x 128-byte dy
// rows, their keep bytes, one 32-pixel row of x -- are fetched D = 2 steps ahead by `buffer_load ... lds` (28 wave instructions per
// step, 7 per wave, every one a full-line coalesced piece; no staging registers, nothing the compiler waits for), so 2 x 22 KiB per
// workgroup and ~90 KiB per CU are always in flight.  One barrier per step; the arithmetic (horizontal 4-tap reduce with
// row-independent weights, two-term vertical blend, per-(n,c) sums in fp32 before the bf16 rounding of g') is the marching kernel's,
// term for term in the same order.
//   step slot (22 528 B) = 2 x { main 64 px x 128 B, swizzled | halo px -1 and 64 (256 B) | keep bytes 66 px x 8 (768 B) } + x row 4 KiB
//   pixel p of the main window sits at slot p ^ ((p >> 1) & 1): the 16-lane groups of ds_read_b128 (MI355X_MICROARCH.md, LDS) then
//   see the eight chunk lanes of four columns on four different 64-byte bank quarters (the swizzle is applied to the DMA SOURCE).
// Low-res row yy is complete when high-res row 2 yy + 3 arrives (scale (H-1)/(2H-1) < 1/2), i.e. in step yy - xbase with
// xbase = (r_begin >> 1) - 1: that step's slot carries x row yy; the last row(s) of a strip are emitted after the loop from the
// slot of the extra step that only brings x.
namespace upbt {
constexpr int TC = 32;                                             // low-res columns per workgroup
constexpr int ROW_MAIN = 8192, ROW_HALO = 256, ROW_MASK = 768;
constexpr int ROW_BYTES = ROW_MAIN + ROW_HALO + ROW_MASK;          // 9216
constexpr int X_BYTES = TC * 128;                                  // 4096
constexpr int STEP_BYTES = 2 * ROW_BYTES + X_BYTES;                // 22528
constexpr int D = 2, RS = D + 1;                                   // steps in flight, ring slots
constexpr int NPIECE = 7;                                          // DMA instructions per wave and step
constexpr int LUT_OFF = RS * STEP_BYTES;                           // 67584
constexpr int SMEM = LUT_OFF + 4096;                               // 71680 B: two workgroups per CU
}  // namespace upbt

__global__ __launch_bounds__(256) void adain_upcat_bwd_tile_kernel(
    const bf16_t* __restrict__ dy, int lddy, const bf16_t* __restrict__ x, int ldx, const float* __restrict__ stats,
    bf16_t* __restrict__ gtmp, float* __restrict__ sums, int H, int W, int C, int rows_per_strip, int col_tiles,
    float sy, float sx, uint32_t thr, float keep_scale, const uint8_t* __restrict__ mbits) {
    using namespace upbt;
    constexpr int E = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool masked = thr < 0x10000u;
    if (masked) {
        uint32_t m[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) m[k] = ((tid >> (2 * k)) & 1 ? 0x0000ffffu : 0u) | ((tid >> (2 * k + 1)) & 1 ? 0xffff0000u : 0u);
        *(uint4*)(smem + LUT_OFF + tid * 16) = make_uint4(m[0], m[1], m[2], m[3]);
    }
    const int cg = blockIdx.x, n = blockIdx.z;
    const int ctile = blockIdx.y % col_tiles, strip = blockIdx.y / col_tiles;
    const int cl = tid & 7, pl = tid >> 3;
    const int c0 = cg * 64 + cl * E;
    const int cpp = C >> 3;
    const int H2 = 2 * H, W2 = 2 * W, HW = H * W;
    const int x0t = ctile * TC, xx = x0t + pl;
    const int y0 = strip * rows_per_strip, y1 = min(H, y0 + rows_per_strip);
    const int r_begin = max(0, 2 * y0 - 1), r_end = min(H2 - 1, 2 * y1);
    const int nsteps = (r_end - r_begin + 2) >> 1;          // pairs of high-res rows
    const int total = nsteps + 1;                           // + the step that only carries the last x row
    const int xbase = (r_begin >> 1) - 1;                   // step s carries x row xbase + s

    // ---- tile-invariant per-lane DMA source offsets (bytes; kWuOOB = this lane fetches nothing and lands zeros) ----
    // dy descriptor: one image, shifted back by ONE pixel so that the left halo column has a non-negative offset
    unsigned vo_main[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = (wave + 4 * j) * 64 + lane, pp = i >> 3, c = i & 7;
        const int p = pp ^ ((pp >> 1) & 1);
        vo_main[j] = 2 * x0t + p < W2 ? (unsigned)(((p + 1) * lddy + cg * 64 + c * 8) * 2) : kWuOOB;
    }
    // third piece of a row: wave 0 the two halo pixels (dword lanes: 32 per pixel), waves 1-3 the keep bytes (dword lanes: 2 per pixel)
    unsigned vo_c;
    if (wave == 0) {
        const int h = lane >> 5, col = h ? 2 * x0t + 64 : 2 * x0t - 1;
        vo_c = (col >= 0 && col < W2) ? (unsigned)(((h ? 65 : 0) * lddy + cg * 64) * 2 + (lane & 31) * 4) : kWuOOB;
    } else {
        const int dd = (wave - 1) * 64 + lane;
        if (dd < 128) {
            const int p = dd >> 1;
            vo_c = 2 * x0t + p < W2 ? (unsigned)((p + 1) * cpp + cg * 8 + (dd & 1) * 4) : kWuOOB;
        } else if (dd < 132) {
            const int h = (dd - 128) >> 1, col = h ? 2 * x0t + 64 : 2 * x0t - 1;
            vo_c = (col >= 0 && col < W2) ? (unsigned)((h ? 65 : 0) * cpp + cg * 8 + (dd & 1) * 4) : kWuOOB;
        } else {
            vo_c = kWuOOB;
        }
    }
    unsigned vo_x;
    {
        const int i = wave * 64 + lane, px = i >> 3, c = i & 7;
        vo_x = x0t + px < W ? (unsigned)((px * ldx + cg * 64 + c * 8) * 2) : kWuOOB;
    }
    const wu_rsrc_t rs_d = wu_make_rsrc(dy + ((long long)n * H2 * W2 - 1) * lddy, (unsigned)(((size_t)H2 * W2 + 1) * lddy * 2));
    const wu_rsrc_t rs_m = wu_make_rsrc(mbits ? mbits + ((long long)n * H2 * W2 - 1) * cpp : nullptr,
                                        (masked && mbits) ? (unsigned)(((size_t)H2 * W2 + 1) * cpp) : 0u);
    const wu_rsrc_t rs_x = wu_make_rsrc(x + (size_t)n * HW * ldx, (unsigned)((size_t)HW * ldx * 2));
    const wu_rsrc_t rs_c = wave == 0 ? rs_d : rs_m;
    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned c_lds = wave == 0 ? (unsigned)ROW_MAIN : (unsigned)(ROW_MAIN + ROW_HALO + (wave - 1) * 256);

    auto issue = [&](int j, int slot) __attribute__((always_inline)) {       // the 7 pieces of step j (wave-uniform arguments)
        const unsigned lds = smem_base + slot * STEP_BYTES;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = r_begin + 2 * j + rr;
            const unsigned kill = r > r_end ? kWuOOB : 0u;
            const unsigned pix = (unsigned)(r * W2 + 2 * x0t);
            const unsigned so_d = (pix * (unsigned)lddy * 2u) | kill;
            const unsigned so_c = (wave == 0 ? pix * (unsigned)lddy * 2u : pix * (unsigned)cpp) | kill;
            const unsigned row = lds + rr * ROW_BYTES;
            wu_dma16b(vo_main[0], rs_d, __builtin_amdgcn_readfirstlane(so_d), __builtin_amdgcn_readfirstlane(row + wave * 1024));
            wu_dma16b(vo_main[1], rs_d, __builtin_amdgcn_readfirstlane(so_d), __builtin_amdgcn_readfirstlane(row + (wave + 4) * 1024));
            wu_dma4b(vo_c, rs_c, __builtin_amdgcn_readfirstlane(so_c), __builtin_amdgcn_readfirstlane(row + c_lds));
        }
        const int xr = xbase + j;
        const unsigned so_x = (xr >= y0 && xr < y1) ? (unsigned)((xr * W + x0t) * ldx * 2) : kWuOOB;
        wu_dma16b(vo_x, rs_x, __builtin_amdgcn_readfirstlane(so_x), __builtin_amdgcn_readfirstlane(lds + 2 * ROW_BYTES + wave * 1024));
    };

    // ---- per-thread LDS read offsets inside a row / step slot, column weights ----
    int daddr[4], maddr[4];
    float wgt[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int q = 2 * pl - 1 + k;                                      // pixel of the main window; -1 / 64 = the halo pixels
        daddr[k] = q < 0 ? ROW_MAIN + cl * 16 : q >= 64 ? ROW_MAIN + 128 + cl * 16 : (q ^ ((q >> 1) & 1)) * 128 + cl * 16;
        maddr[k] = ROW_MAIN + ROW_HALO + (q < 0 ? 512 + cl : q >= 64 ? 520 + cl : q * 8 + cl);
        const int j = 2 * xx - 1 + k;
        const int jc = min(max(j, 0), W2 - 1);
        const Lerp lx = src_index(jc, sx, W);
        wgt[k] = j == jc ? ((lx.i0 == xx ? lx.l0 : 0.f) + (lx.i1 == xx ? lx.l1 : 0.f)) : 0.f;
    }
    const int xaddr = 2 * ROW_BYTES + pl * 128 + cl * 16;

    float st[2 * E];
    ldf<2 * E>(stats + 2 * (n * C + c0), st);
    // the statistics are in registers BEFORE the first DMA is issued: a compiler-inserted vmcnt(0) for them inside the loop would
    // drain the ring at every emit (the compiler does not see the DMA; its waits retire in issue order)
#pragma unroll
    for (int e = 0; e < 2 * E; ++e) asm volatile("" : "+v"(st[e]));

    issue(0, 0);
    if (total > 1) issue(1, 1);

    float s1[E], s2[E], acc0[E], acc1[E];
#pragma unroll
    for (int e = 0; e < E; ++e) s1[e] = s2[e] = acc0[e] = acc1[e] = 0.f;
    int cur = src_index(r_begin, sy, H).i0;

    auto emit = [&](int yy, const float (&ga)[E]) __attribute__((always_inline)) {
        if (yy < y0 || yy >= y1) return;
        const int e_ = yy - xbase;                                         // the step whose slot carries x row yy
        const int slot = e_ - 3 * (e_ / 3);
        float xv[E], gs[E];
        unpack16<bf16_t>(*(const uint4*)(smem + slot * STEP_BYTES + xaddr), xv);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            gs[e] = masked ? ga[e] * keep_scale : ga[e];
            if (xx < W) {
                s1[e] += gs[e];
                s2[e] += gs[e] * ((xv[e] - st[2 * e]) * st[2 * e + 1]);
            }
        }
        if (xx < W) *(uint4*)(gtmp + ((size_t)n * HW + (size_t)yy * W + xx) * C + c0) = pack16<bf16_t>(gs);
    };

    int slot = 0;                                                          // ring slot of the step being computed
    for (int s = 0; s < total; ++s) {
        // this wave's pieces of step s have landed: everything issued after them is step s + 1's 7 pieces (and stores, which are
        // younger still) -- vmcnt retires in issue order
        if (s + 1 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                   // ... everyone's have; everyone is done with step s - 1's slot
        if (s + D < total) issue(s + D, slot == 0 ? 2 : slot - 1);         // (s + 2) % 3 == (s - 1) % 3
        if (s < nsteps) {
            const char* sl = smem + slot * STEP_BYTES;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int r = r_begin + 2 * s + rr;
                if (r > r_end) break;
                const Lerp ly = src_index(r, sy, H);
                while (cur < ly.i0) {                                      // retire the rows no later high-res row can touch
                    emit(cur, acc0);
#pragma unroll
                    for (int e = 0; e < E; ++e) { acc0[e] = acc1[e]; acc1[e] = 0.f; }
                    ++cur;
                }
                const char* row = sl + rr * ROW_BYTES;
                float t[E];
#pragma unroll
                for (int e = 0; e < E; ++e) t[e] = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint4 dv = *(const uint4*)(row + daddr[k]);
                    if (masked) {
                        const uint4 mk = *(const uint4*)(smem + LUT_OFF + 16 * (unsigned)*(const uint8_t*)(row + maddr[k]));
                        dv = make_uint4(dv.x & mk.x, dv.y & mk.y, dv.z & mk.z, dv.w & mk.w);
                    }
                    float d[E];
                    unpack16<bf16_t>(dv, d);
#pragma unroll
                    for (int e = 0; e < E; ++e) t[e] = fmaf(wgt[k], d[e], t[e]);
                }
                const float l1w = ly.i1 != ly.i0 ? ly.l1 : 0.f, l0w = ly.i1 != ly.i0 ? ly.l0 : ly.l0 + ly.l1;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    acc0[e] = fmaf(l0w, t[e], acc0[e]);
                    acc1[e] = fmaf(l1w, t[e], acc1[e]);
                }
            }
        }
        slot = slot == 2 ? 0 : slot + 1;
    }
    emit(cur, acc0);
    emit(cur + 1, acc1);
    __syncthreads();                                                       // the ring becomes the reduction scratch
    float (*red)[64][2] = (float (*)[64][2])smem;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        red[pl][cl * E + e][0] = s1[e];
        red[pl][cl * E + e][1] = s2[e];
    }
    const int splits = gridDim.y;
    block_reduce_2x64<TC>(red, tid, sums + (((size_t)n * C + cg * 64) * splits + blockIdx.y) * 2, splits * 2);
}

// fold the per-split partials in fixed order: sums_final[n][c][2]
__global__ void fold_partials_kernel(const float* __restrict__ part, float* __restrict__ out, int NC2, int splits) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NC2) return;
    const int nc = i >> 1, j = i & 1;
    float s = 0.f;
    for (int sp = 0; sp < splits; ++sp) s += part[((size_t)nc * splits + sp) * 2 + j];
    out[i] = s;
}

// backward stage B: dx = y_std*rstd * (g' - mean(g') - xhat * sum(g'*xhat)/(HW-1));  d_y_mean = sum g',
// d_y_std = sum g'*xhat.
// grid: x = 256-thread chunks of one image (HW * C/E items), y = image
template <typename T>
__global__ __launch_bounds__(256) void adain_upcat_bwd_apply_kernel(const T* __restrict__ gtmp, const float* __restrict__ sums,
                                             const T* __restrict__ x, int ldx, const float* __restrict__ stats,
                                             const float* __restrict__ y_std, T* __restrict__ dx, int lddx,
                                             float* __restrict__ d_y_std, float* __restrict__ d_y_mean,
                                             int N, int HW, int C, int x_gate_act) {
    constexpr int E = ElemTraits<T>::kPer16B;
    const int cpp = C / E;
    const int n = blockIdx.y;
    const float inv_hw = 1.f / (float)HW, inv_hw1 = 1.f / (float)(HW - 1);
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < HW * cpp; idx += gridDim.x * 256) {
        const int pix = idx / cpp, ch = idx - pix * cpp;
        const size_t p = (size_t)n * HW + pix;
        const int sc = n * C + ch * E;
        float xv[E], o[E], g[E], st[2 * E], sm[2 * E], ys[E];
        unpack16<T>(*(const uint4*)(x + p * ldx + ch * E), xv);
        unpack16<T>(*(const uint4*)(gtmp + p * C + ch * E), g);
        ldf<2 * E>(stats + 2 * sc, st);
        ldf<2 * E>(sums + 2 * sc, sm);
        ldf<E>(y_std + sc, ys);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const float mean = st[2 * e], rstd = st[2 * e + 1], S1 = sm[2 * e], S2 = sm[2 * e + 1];
            const float xh = (xv[e] - mean) * rstd;
            o[e] = act_gate(ys[e] * rstd * (g[e] - S1 * inv_hw - xh * S2 * inv_hw1), xv[e], x_gate_act);
            if (pix == 0) {
                d_y_mean[sc + e] = S1;
                d_y_std[sc + e] = S2;
            }
        }
        *(uint4*)(dx + p * lddx + ch * E) = pack16<T>(o);
    }
}

// stage B, round 4: the same arithmetic with a thread bound to ONE 16-byte channel chunk for its whole life (C / E a power of two
// <= 256: consecutive threads = consecutive chunks of one pixel, the block walks pixels).  The kernel above fetches the chunk's ten
// 16-byte coefficient vectors (mean / rstd, the two sums, y_std) for EVERY item next to its two data loads; here they are loaded
// once, and four pixels' operands are requested before the first is used.
template <typename T>
__global__ __launch_bounds__(256) void adain_upcat_bwd_apply_rows_kernel(const T* __restrict__ gtmp, const float* __restrict__ sums,
                                             const T* __restrict__ x, int ldx, const float* __restrict__ stats,
                                             const float* __restrict__ y_std, T* __restrict__ dx, int lddx,
                                             float* __restrict__ d_y_std, float* __restrict__ d_y_mean,
                                             int HW, int C, int x_gate_act) {
    constexpr int E = ElemTraits<T>::kPer16B;
    constexpr int U = 4;
    const int cpp = C / E, ppb = 256 / cpp;
    const int n = blockIdx.y;
    const int ch = threadIdx.x % cpp, pl = threadIdx.x / cpp;
    const int sc = n * C + ch * E;
    const float inv_hw = 1.f / (float)HW, inv_hw1 = 1.f / (float)(HW - 1);
    float st[2 * E], sm[2 * E], ys[E];
    ldf<2 * E>(stats + 2 * sc, st);
    ldf<2 * E>(sums + 2 * sc, sm);
    ldf<E>(y_std + sc, ys);
    if (blockIdx.x == 0 && pl == 0) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            d_y_mean[sc + e] = sm[2 * e];
            d_y_std[sc + e] = sm[2 * e + 1];
        }
    }
    const size_t img = (size_t)n * HW;
    const int stride = gridDim.x * ppb;
    for (int pix0 = blockIdx.x * ppb + pl; pix0 < HW; pix0 += U * stride) {
        uint4 xr[U], gr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pix = min(pix0 + u * stride, HW - 1);
            xr[u] = *(const uint4*)(x + (img + pix) * ldx + ch * E);
            gr[u] = *(const uint4*)(gtmp + (img + pix) * C + ch * E);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pix = pix0 + u * stride;
            if (pix >= HW) break;
            float xv[E], g[E], o[E];
            unpack16<T>(xr[u], xv);
            unpack16<T>(gr[u], g);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float mean = st[2 * e], rstd = st[2 * e + 1], S1 = sm[2 * e], S2 = sm[2 * e + 1];
                const float xh = (xv[e] - mean) * rstd;
                o[e] = act_gate(ys[e] * rstd * (g[e] - S1 * inv_hw - xh * S2 * inv_hw1), xv[e], x_gate_act);
            }
            *(uint4*)(dx + (img + pix) * lddx + ch * E) = pack16<T>(o);
        }
    }
}

// out = g * act'(y): the activation backward of autograd as ONE streaming pass, so that the data-gradient
// and weight-gradient GEMMs both consume a pre-gated gradient (no gating inside their staging loops).
template <typename T, typename IDX>
__global__ void act_gate_kernel(const T* __restrict__ g, int ldg, const T* __restrict__ y, int ldy, T* __restrict__ out, int ldo,
                                long long npix, int C, int act) {
    constexpr int E = ElemTraits<T>::kPer16B;
    const int cpp = C / E;
    const IDX total = (IDX)npix * cpp;
    for (IDX i = (IDX)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (IDX)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cpp);
        const IDX p = i / cpp;
        const uint4 gv = *(const uint4*)(g + (size_t)p * ldg + ch * E);
        const uint4 yv = *(const uint4*)(y + (size_t)p * ldy + ch * E);
        *(uint4*)(out + (size_t)p * ldo + ch * E) = gate16<T>(gv, yv, act);
    }
}

__global__ void dropout_mask_kernel(uint8_t* __restrict__ mask, int N, int H2, int W2, int C, uint32_t thr, uint64_t seed) {
    const long long total = (long long)N * H2 * W2 * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        // i = NHWC linear index (the RNG's index space); write NCHW
        const int c = (int)(i % C);
        long long p = i / C;
        const int w = (int)(p % W2); p /= W2;
        const int h = (int)(p % H2);
        const int n = (int)(p / H2);
        const uint64_t r = wu_rand4(seed, (uint64_t)i >> 2);
        const bool keep = (uint32_t)((r >> (16 * (i & 3))) & 0xffff) < thr;
        mask[((size_t)(n * C + c) * H2 + h) * W2 + w] = keep ? 1 : 0;
    }
}

// =================================================================================================
// sum-pool (disc.py:32) and layout helpers
// =================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void sumpool_fwd_kernel(const T* __restrict__ x, int ldx, float* __restrict__ feat, int HW, int C) {
    constexpr int E = ElemTraits<T>::kPer16B;
    constexpr int LP = 64 / E, PP = 256 / LP;
    __shared__ float red[PP][64];
    const int tid = threadIdx.x, cg = blockIdx.x, n = blockIdx.y;
    const int cl = tid % LP, pl = tid / LP;
    const T* base = x + (size_t)n * HW * ldx + cg * 64 + cl * E;
    float s[E], v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) s[e] = 0.f;
    for (int p = pl; p < HW; p += PP) {
        unpack16<T>(*(const uint4*)(base + (size_t)p * ldx), v);
#pragma unroll
        for (int e = 0; e < E; ++e) s[e] += v[e];
    }
#pragma unroll
    for (int e = 0; e < E; ++e) red[pl][cl * E + e] = s[e];
    __syncthreads();
    if (tid < 64) {
        float t = 0.f;
        for (int p = 0; p < PP; ++p) t += red[p][tid];
        feat[(size_t)n * C + cg * 64 + tid] = t;
    }
}

template <typename T>
__global__ void sumpool_bwd_kernel(const float* __restrict__ dfeat, T* __restrict__ dx, int lddx, int N, int HW, int C) {
    constexpr int E = ElemTraits<T>::kPer16B;
    const int cpp = C / E;
    const long long total = (long long)N * HW * cpp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cpp);
        const long long p = i / cpp;
        const int n = (int)(p / HW);
        float o[E];
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = dfeat[(size_t)n * C + ch * E + e];
        *(uint4*)(dx + (size_t)p * lddx + ch * E) = pack16<T>(o);
    }
}

// NHWC T -> NCHW fp32 through a 32x32 LDS tile (pixels x channels)
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ x, int ldx, float* __restrict__ y, int HW, int C) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        tile[r][tx] = (p < HW && c < C) ? ElemTraits<T>::load(x + ((size_t)n * HW + p) * ldx + c) : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        if (p < HW && c < C) y[((size_t)n * C + c) * HW + p] = tile[tx][r];
    }
}
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int ldy, int HW, int C) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        tile[r][tx] = (p < HW && c < C) ? x[((size_t)n * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        if (p < HW && c < C) ElemTraits<T>::store(y + ((size_t)n * HW + p) * ldy + c, tile[tx][r]);
    }
}

inline int grid_for(long long total, int block = 256, int cap = 256 * 16) {
    long long g = (total + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}
inline bool ok16(const void* p, int ld, int esz) { return ((uintptr_t)p % 16) == 0 && (ld * esz) % 16 == 0; }
inline uint32_t keep_thr(float p) {
    if (p <= 0.f) return 0x10000u;
    return (uint32_t)((1.0 - (double)p) * 65536.0 + 0.5);
}


// =================================================================================================
// mean absolute error (ops.py:22-24) with its gradient in the same pass
// =================================================================================================
// loss = mean |a - b| over n fp32 elements; grad (optional) = sign(a - b) / n, i.e. d loss / d a.  Each workgroup leaves one
// partial sum; l1_mean_fold_kernel adds them in index order (deterministic).  The autograd composition of sub / abs / mean and
// their backward is nine launches and six passes over the data; this is two launches and one pass.
constexpr int kL1MaxBlocks = 1024;
__global__ __launch_bounds__(256) void l1_mean_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ grad,
                                                              float* __restrict__ part, long long n4, long long n, float inv_n) {
    __shared__ float red[4];
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 u = ((const float4*)a)[i], v = ((const float4*)b)[i];
        const float d0 = u.x - v.x, d1 = u.y - v.y, d2 = u.z - v.z, d3 = u.w - v.w;
        acc += (fabsf(d0) + fabsf(d1)) + (fabsf(d2) + fabsf(d3));
        if (grad) {
            // torch.sign semantics: 0 at 0
            ((float4*)grad)[i] = make_float4(d0 > 0.f ? inv_n : (d0 < 0.f ? -inv_n : 0.f), d1 > 0.f ? inv_n : (d1 < 0.f ? -inv_n : 0.f),
                                             d2 > 0.f ? inv_n : (d2 < 0.f ? -inv_n : 0.f), d3 > 0.f ? inv_n : (d3 < 0.f ? -inv_n : 0.f));
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) {          // up to three tail elements
        const long long i = 4 * n4 + threadIdx.x;
        const float d = a[i] - b[i];
        acc += fabsf(d);
        if (grad) grad[i] = d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f);
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(64) void l1_mean_fold_kernel(const float* __restrict__ part, int nblk, float inv_n, float* __restrict__ loss) {
    // 64 lanes x fixed stride, then a fixed xor tree: the same summation order on every run
    float acc = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 64) acc += part[i];
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m);
    if (threadIdx.x == 0) *loss = acc * inv_n;
}

}  // namespace

#define DISPATCH_T(dtype, ...)                         \
    do {                                               \
        if ((dtype) == WU_BF16) { using T = bf16_t; __VA_ARGS__; } \
        else { using T = float; __VA_ARGS__; }         \
    } while (0)

extern "C" int wu_pack_conv3x3(const float* w_oihw, void* w_fwd, void* w_dgrad, int Cout, int Cin,
                               const float* inv_sigma, int dtype, void* stream) {
    WU_REQUIRE(dtype == WU_F32 || dtype == WU_BF16, "pack_conv3x3: bad dtype");
    WU_REQUIRE(Cout > 0 && Cin > 0 && w_oihw, "pack_conv3x3: bad shape");
    const int total = Cout * Cin;
    DISPATCH_T(dtype, hipLaunchKernelGGL(pack_conv3x3_kernel<T>, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                                         w_oihw, (T*)w_fwd, (T*)w_dgrad, Cout, Cin, inv_sigma));
    WU_LAUNCH_CHECK("pack_conv3x3");
    return 0;
}

extern "C" int wu_pack_conv3x3_multi(int n, const float* const* w_oihw, void* const* w_fwd, void* const* w_dgrad,
                                     const int* Cout, const int* Cin, int dtype, void* stream) {
    WU_REQUIRE(n >= 1 && n <= 16 && w_oihw && w_fwd && w_dgrad && Cout && Cin, "pack_conv3x3_multi: 1..16 weights per call");
    PackMulti a;
    a.n = n;
    long long blocks = 0;
    for (int i = 0; i < 16; ++i) {
        const int j = i < n ? i : n - 1;
        WU_REQUIRE(w_oihw[j] && w_fwd[j] && w_dgrad[j] && Cout[j] > 0 && Cin[j] > 0, "pack_conv3x3_multi: bad entry %d", j);
        a.w[i] = w_oihw[j]; a.wf[i] = w_fwd[j]; a.wd[i] = w_dgrad[j]; a.cout[i] = Cout[j]; a.cin[i] = Cin[j];
        a.first_block[i] = (int)blocks;
        if (i < n) blocks += (long long)((Cout[j] + 31) / 32) * ((Cin[j] + 31) / 32);
    }
    a.first_block[16] = (int)blocks;
    WU_REQUIRE(blocks < (1ll << 31), "pack_conv3x3_multi: too many elements");
    DISPATCH_T(dtype, hipLaunchKernelGGL(pack_conv3x3_multi_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a));
    WU_LAUNCH_CHECK("pack_conv3x3_multi");
    return 0;
}

extern "C" int wu_maxpool2_fwd(const void* x, int ldx, void* y, int ldy, int N, int H, int W, int C, int dtype, void* stream) {
    const int esz = dtype == WU_BF16 ? 2 : 4;
    WU_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % (16 / esz) == 0 && C <= ldx && C <= ldy, "maxpool2_fwd: bad shape H=%d W=%d C=%d", H, W, C);
    WU_REQUIRE(ok16(x, ldx, esz) && ok16(y, ldy, esz), "maxpool2_fwd: alignment");
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / (16 / esz));
    if (total < (1ll << 31))
        DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool2_fwd_kernel<T, unsigned>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                                             (const T*)x, ldx, (T*)y, ldy, N, H, W, C));
    else
        DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool2_fwd_kernel<T, long long>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                                             (const T*)x, ldx, (T*)y, ldy, N, H, W, C));
    WU_LAUNCH_CHECK("maxpool2_fwd");
    return 0;
}

extern "C" int wu_maxpool2_bwd(const void* x, int ldx, const void* dy, int lddy, const void* dskip, int lddskip,
                               void* dx, int lddx, int N, int H, int W, int C, int gate_act, int dtype, void* stream) {
    const int esz = dtype == WU_BF16 ? 2 : 4;
    WU_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % (16 / esz) == 0, "maxpool2_bwd: bad shape");
    WU_REQUIRE(ok16(x, ldx, esz) && ok16(dy, lddy, esz) && ok16(dx, lddx, esz) && (!dskip || ok16(dskip, lddskip, esz)), "maxpool2_bwd: alignment");
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / (16 / esz));
    if (total < (1ll << 31))
        DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool2_bwd_kernel<T, unsigned>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                                             (const T*)x, ldx, (const T*)dy, lddy, (const T*)dskip, lddskip, (T*)dx, lddx, N, H, W, C, gate_act));
    else
        DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool2_bwd_kernel<T, long long>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                                             (const T*)x, ldx, (const T*)dy, lddy, (const T*)dskip, lddskip, (T*)dx, lddx, N, H, W, C, gate_act));
    WU_LAUNCH_CHECK("maxpool2_bwd");
    return 0;
}

extern "C" int wu_maxpool2_bwd_bits(const unsigned* gate_bits, const unsigned* sel_bits, const void* dy, int lddy, const void* dskip, int lddskip,
                                    void* dx, int lddx, int N, int H, int W, int C, int dtype, void* stream) {
    WU_REQUIRE(dtype == WU_BF16, "maxpool2_bwd_bits: bf16 only (the bits come from the LDS-DMA conv)");
    WU_REQUIRE(N > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && C % 64 == 0 && gate_bits && sel_bits && dy && dx, "maxpool2_bwd_bits: bad args");
    WU_REQUIRE(ok16(dy, lddy, 2) && ok16(dx, lddx, 2) && (!dskip || ok16(dskip, lddskip, 2)) && lddy >= C && lddx >= C, "maxpool2_bwd_bits: alignment");
    const long long total = (long long)N * H * W * (C / 8);
    WU_REQUIRE(total < (1ll << 32), "maxpool2_bwd_bits: too many items for 32-bit indices");
    hipLaunchKernelGGL(maxpool2_bwd_bits_kernel, dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, (hipStream_t)stream, gate_bits, sel_bits,
                       (const bf16_t*)dy, lddy, (const bf16_t*)dskip, lddskip, (bf16_t*)dx, lddx, N, H, W, C);
    WU_LAUNCH_CHECK("maxpool2_bwd_bits");
    return 0;
}

extern "C" int wu_adain_stats(const void* x, int ldx, float* stats, float* scratch, int N, int H, int W, int C,
                              float eps, int dtype, void* stream) {
    const int esz = dtype == WU_BF16 ? 2 : 4;
    WU_REQUIRE(C % 64 == 0 && H * W > 1 && ok16(x, ldx, esz), "adain_stats: bad shape C=%d HW=%d", C, H * W);
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W;
    int splits = cdiv(1024, N * (C / 64));
    if (splits > cdiv(HW, 256)) splits = cdiv(HW, 256);
    if (splits > kMaxSplits) splits = kMaxSplits;
    if (splits < 1) splits = 1;
    DISPATCH_T(dtype, {
        hipLaunchKernelGGL(adain_stats_kernel<T>, dim3(C / 64, splits, N), dim3(256), 0, s, (const T*)x, ldx, scratch, HW, C, splits);
        hipLaunchKernelGGL(adain_stats_final_kernel<T>, dim3(cdiv(N * C, 256)), dim3(256), 0, s, (const T*)x, ldx, scratch, stats, N, HW, C, eps, splits);
    });
    WU_LAUNCH_CHECK("adain_stats");
    return 0;
}

extern "C" int wu_adain_upcat_fwd(const void* x, int ldx, const float* stats, const float* y_std, const float* y_mean,
                                  void* y, int ldy, int N, int H, int W, int C, float p_drop, uint64_t seed,
                                  const uint64_t* seed_dev, uint8_t* mask_bits, int mask_is_input, int dtype, void* stream) {
    const int esz = dtype == WU_BF16 ? 2 : 4;
    WU_REQUIRE(C % (16 / esz) == 0 && C <= ldx && C <= ldy && H > 1 && W > 1, "adain_upcat_fwd: bad shape");
    WU_REQUIRE(ok16(x, ldx, esz) && ok16(y, ldy, esz), "adain_upcat_fwd: alignment");
    WU_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "adain_upcat_fwd: p_drop");
    WU_REQUIRE(!mask_is_input || mask_bits, "adain_upcat_fwd: mask_is_input without mask_bits");
    const float sy = (float)(H - 1) / (float)(2 * H - 1), sx = (float)(W - 1) / (float)(2 * W - 1);
    WU_REQUIRE(((uintptr_t)stats % 16) == 0 && ((uintptr_t)y_std % 16) == 0 && ((uintptr_t)y_mean % 16) == 0, "adain_upcat_fwd: stats alignment");
    const int LPv = 64 / (16 / esz), PPv = 256 / LPv;          // marching formulation: 2 * PPv output columns per workgroup
    if (g_wu_opt[WU_OPT_ADAIN_FWD_MARCH] && C % 64 == 0) {
        const int col_tiles = cdiv(2 * W, 2 * PPv);
        int strips = cdiv(4096, N * (C / 64) * col_tiles);      // enough workgroups to fill the chip ...
        if (strips > cdiv(2 * H, 8)) strips = cdiv(2 * H, 8);   // ... but at least 8 output rows per strip (each strip reloads 2-3 source rows)
        if (strips < 1) strips = 1;
        const int rows_per_strip = cdiv(2 * H, strips);
        strips = cdiv(2 * H, rows_per_strip);
        WU_REQUIRE((long long)col_tiles * strips < 65536, "adain_upcat_fwd: grid too large");
        DISPATCH_T(dtype, hipLaunchKernelGGL(adain_upcat_fwd_march_kernel<T>, dim3(C / 64, col_tiles * strips, N), dim3(256), 0, (hipStream_t)stream,
                                             (const T*)x, ldx, stats, y_std, y_mean, (T*)y, ldy, N, H, W, C, rows_per_strip, col_tiles, sy, sx,
                                             keep_thr(p_drop), 1.f / (1.f - p_drop), seed, seed_dev, mask_bits, mask_is_input));
        WU_LAUNCH_CHECK("adain_upcat_fwd (march)");
        return 0;
    }
    const int rows = N * cdiv(2 * H, 4);            // 4 output rows per thread
    const dim3 grid(cdiv(2 * W * (C / (16 / esz)), 256), rows < 32768 ? rows : 32768);
    DISPATCH_T(dtype, hipLaunchKernelGGL(adain_upcat_fwd_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream,
                                         (const T*)x, ldx, stats, y_std, y_mean, (T*)y, ldy, N, H, W, C, sy, sx,
                                         keep_thr(p_drop), 1.f / (1.f - p_drop), seed, seed_dev, mask_bits, mask_is_input));
    WU_LAUNCH_CHECK("adain_upcat_fwd");
    return 0;
}

extern "C" int wu_adain_upcat_bwd(const void* dy, int lddy, const void* x, int ldx, const float* stats, const float* y_std,
                                  void* dx, int lddx, float* d_y_std, float* d_y_mean, void* gtmp, float* sums,
                                  int N, int H, int W, int C, float p_drop, uint64_t seed, const uint8_t* mask_bits,
                                  int x_gate_act, int dtype, void* stream) {
    const int esz = dtype == WU_BF16 ? 2 : 4;
    WU_REQUIRE(C % 64 == 0 && H > 1 && W > 1, "adain_upcat_bwd: bad shape");
    WU_REQUIRE(((uintptr_t)stats % 16) == 0 && ((uintptr_t)sums % 16) == 0 && ((uintptr_t)y_std % 16) == 0, "adain_upcat_bwd: stats alignment");
    WU_REQUIRE(ok16(x, ldx, esz) && ok16(dy, lddy, esz) && ok16(dx, lddx, esz) && ((uintptr_t)gtmp % 16) == 0, "adain_upcat_bwd: alignment");
    hipStream_t s = (hipStream_t)stream;
    const float sy = (float)(H - 1) / (float)(2 * H - 1), sx = (float)(W - 1) / (float)(2 * W - 1);
    const int HW = H * W;
    const int pp = 256 / (64 / (16 / esz));       // pixels (gather) / column pairs (march) per workgroup
    float* partials = sums + (size_t)N * C * 2;
    int splits;
    // marching formulation: workgroup = (64 channels, ncol*pp low-res columns, a strip of rows); its partial-sum slot = (column tile, strip)
    // option 8: 0 = 16-tap gather, 1 = march with the column count chosen by size (default), 2 / 3 = always one / two columns per thread.
    // One column per thread (3 waves per SIMD instead of 2) wins on the small levels, where a strip is short and the prologue and the
    // row round trips dominate (64 -> 32: 108 -> 88 us, 128 -> 64: 160 -> 149 us); two columns (3 loads per column instead of 4) on 256 -> 128
    // (292 vs 310 us)
    const int opt8 = g_wu_opt[WU_OPT_ADAIN_BWD_MARCH];
    // round 4: bf16 with keep BITS (or no dropout) streams its strips through an LDS ring (adain_upcat_bwd_tile_kernel); option 8 = 5
    // keeps the marching kernel (A/B).  The re-hash mode (dropout without stored bits) stays on the marching kernel.
    const bool drop = keep_thr(p_drop) < 0x10000u;
    const int tile_ct = cdiv(W, upbt::TC);
    const bool tile = dtype == WU_BF16 && opt8 != 0 && opt8 != 5 && (!drop || mask_bits) && tile_ct <= kMaxSplits &&
                      (size_t)4 * H * W * lddy * 2 + (size_t)lddy * 2 < 0x7fffffffull && (size_t)H * W * ldx * 2 < 0x7fffffffull &&
                      ((uintptr_t)mask_bits % 4) == 0;
    const int ncol = (opt8 == 2 || ((opt8 == 1 || opt8 == 5) && W <= 64)) ? 1 : 2;
    const int col_tiles = cdiv(W, ncol * pp);
    const bool march = g_wu_opt[WU_OPT_ADAIN_BWD_MARCH] && col_tiles <= kMaxSplits;
    if (tile) {
        // strips: two workgroups fit a CU, so the grid should be a multiple of 2 x CUs -- 768 workgroups ran as one full wave of 512 and
        // a half-empty one (measured: no faster than the marching kernel); 1024 at B = 32 on all three levels.  At least 8 low-res rows
        // per strip (a strip re-reads a one-row halo and fills / drains its ring once), at most kMaxSplits partial-sum slots per (n, c)
        int strips = kMaxSplits / tile_ct;
        const int want = cdiv(4 * wu_num_cus(), N * (C / 64) * tile_ct);
        if (strips > want) strips = want;
        if (strips > cdiv(H, 8)) strips = cdiv(H, 8);
        if (strips < 1) strips = 1;
        const int rows_per_strip = cdiv(H, strips);
        strips = cdiv(H, rows_per_strip);
        splits = tile_ct * strips;
        static bool attr_done = false;
        if (!attr_done) {
            (void)hipFuncSetAttribute((const void*)adain_upcat_bwd_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, upbt::SMEM);
            attr_done = true;
        }
        hipLaunchKernelGGL(adain_upcat_bwd_tile_kernel, dim3(C / 64, splits, N), dim3(256), upbt::SMEM, s, (const bf16_t*)dy, lddy, (const bf16_t*)x, ldx,
                           stats, (bf16_t*)gtmp, partials, H, W, C, rows_per_strip, tile_ct, sy, sx, keep_thr(p_drop), 1.f / (1.f - p_drop), mask_bits);
    } else if (march) {
        int strips = kMaxSplits / col_tiles;
        const int want = cdiv(2048 * (3 - ncol), N * (C / 64) * col_tiles);      // enough workgroups to fill the chip
        if (strips > want) strips = want;
        if (strips > cdiv(H, 4)) strips = cdiv(H, 4);               // at least 4 low-res rows per strip (each strip re-reads a 1-row halo)
        if (strips < 1) strips = 1;
        const int rows_per_strip = cdiv(H, strips);
        strips = cdiv(H, rows_per_strip);
        splits = col_tiles * strips;
#define WU_MARCH(NC) DISPATCH_T(dtype, hipLaunchKernelGGL((adain_upcat_bwd_march_kernel<T, NC>), dim3(C / 64, splits, N), dim3(256), 0, s, (const T*)dy, lddy, (const T*)x, ldx, \
                                             stats, (T*)gtmp, partials, H, W, C, rows_per_strip, col_tiles, sy, sx, keep_thr(p_drop), 1.f / (1.f - p_drop), seed, mask_bits))
        if (ncol == 1) WU_MARCH(1); else WU_MARCH(2);
#undef WU_MARCH
    } else {
        splits = cdiv(2048, N * (C / 64));
        if (splits > cdiv(HW, pp)) splits = cdiv(HW, pp);
        if (splits > kMaxSplits) splits = kMaxSplits;
        if (splits < 1) splits = 1;
        DISPATCH_T(dtype, hipLaunchKernelGGL(adain_upcat_bwd_gather_kernel<T>, dim3(C / 64, splits, N), dim3(256), 0, s, (const T*)dy, lddy, (const T*)x, ldx,
                                             stats, (T*)gtmp, partials, H, W, C, sy, sx, keep_thr(p_drop), 1.f / (1.f - p_drop), seed, mask_bits));
    }
    const int cpp_ = C / (16 / esz);
    const bool rows = opt8 != 5 && cpp_ <= 256 && (cpp_ & (cpp_ - 1)) == 0;
    DISPATCH_T(dtype, {
        hipLaunchKernelGGL(fold_partials_kernel, dim3(cdiv(N * C * 2, 256)), dim3(256), 0, s, partials, sums, N * C * 2, splits);
        if (rows)
            hipLaunchKernelGGL(adain_upcat_bwd_apply_rows_kernel<T>, dim3(grid_for(cdiv((long long)HW * cpp_, 4), 256, 2048), N), dim3(256), 0, s, (const T*)gtmp, sums,
                               (const T*)x, ldx, stats, y_std, (T*)dx, lddx, d_y_std, d_y_mean, HW, C, x_gate_act);
        else
            hipLaunchKernelGGL(adain_upcat_bwd_apply_kernel<T>, dim3(grid_for((long long)HW * (C / (16 / esz)), 256, 1024), N), dim3(256), 0, s, (const T*)gtmp, sums, (const T*)x, ldx, stats, y_std,
                               (T*)dx, lddx, d_y_std, d_y_mean, N, HW, C, x_gate_act);
    });
    WU_LAUNCH_CHECK("adain_upcat_bwd");
    return 0;
}

extern "C" int wu_dropout_mask(uint8_t* mask_nchw, int N, int H2, int W2, int C, float p_drop, uint64_t seed, void* stream) {
    WU_REQUIRE(C % 4 == 0, "dropout_mask: C %% 4");
    const long long total = (long long)N * H2 * W2 * C;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, mask_nchw, N, H2, W2, C, keep_thr(p_drop), seed);
    WU_LAUNCH_CHECK("dropout_mask");
    return 0;
}

extern "C" size_t wu_l1_mean_scratch_floats(void) { return (size_t)kL1MaxBlocks; }

extern "C" int wu_l1_mean(const float* a, const float* b, float* grad, float* scratch, float* loss, long long n, void* stream) {
    WU_REQUIRE(a && b && scratch && loss && n > 0, "l1_mean: bad args");
    WU_REQUIRE(((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0 && (!grad || ((uintptr_t)grad % 16) == 0), "l1_mean: 16-byte alignment");
    const long long n4 = n / 4;
    long long g = (n4 + 256 * 8 - 1) / (256 * 8);          // ~8 float4 per thread
    if (g > kL1MaxBlocks) g = kL1MaxBlocks;
    if (g < 1) g = 1;
    const float inv_n = (float)(1.0 / (double)n);
    hipLaunchKernelGGL(l1_mean_partial_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, a, b, grad, scratch, n4, n, inv_n);
    hipLaunchKernelGGL(l1_mean_fold_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scratch, (int)g, inv_n, loss);
    WU_LAUNCH_CHECK("l1_mean");
    return 0;
}

extern "C" int wu_sumpool_fwd(const void* x, int ldx, float* feat, int N, int H, int W, int C, int dtype, void* stream) {
    const int esz = dtype == WU_BF16 ? 2 : 4;
    WU_REQUIRE(C % 64 == 0 && ok16(x, ldx, esz), "sumpool_fwd: bad shape");
    DISPATCH_T(dtype, hipLaunchKernelGGL(sumpool_fwd_kernel<T>, dim3(C / 64, N), dim3(256), 0, (hipStream_t)stream, (const T*)x, ldx, feat, H * W, C));
    WU_LAUNCH_CHECK("sumpool_fwd");
    return 0;
}

extern "C" int wu_sumpool_bwd(const float* dfeat, void* dx, int lddx, int N, int H, int W, int C, int dtype, void* stream) {
    const int esz = dtype == WU_BF16 ? 2 : 4;
    WU_REQUIRE(C % (16 / esz) == 0 && ok16(dx, lddx, esz), "sumpool_bwd: bad shape");
    const long long total = (long long)N * H * W * (C / (16 / esz));
    DISPATCH_T(dtype, hipLaunchKernelGGL(sumpool_bwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dfeat, (T*)dx, lddx, N, H * W, C));
    WU_LAUNCH_CHECK("sumpool_bwd");
    return 0;
}

extern "C" int wu_nhwc_to_nchw_f32(const void* x, int ldx, float* y_nchw, int N, int H, int W, int C, int dtype, void* stream) {
    WU_REQUIRE(N > 0 && H * W > 0 && C > 0, "nhwc_to_nchw: bad shape");
    DISPATCH_T(dtype, hipLaunchKernelGGL(nhwc_to_nchw_kernel<T>, dim3(cdiv(H * W, 32), cdiv(C, 32), N), dim3(256), 0, (hipStream_t)stream,
                                         (const T*)x, ldx, y_nchw, H * W, C));
    WU_LAUNCH_CHECK("nhwc_to_nchw");
    return 0;
}

extern "C" int wu_nchw_f32_to_nhwc(const float* x_nchw, void* y, int ldy, int N, int H, int W, int C, int dtype, void* stream) {
    WU_REQUIRE(N > 0 && H * W > 0 && C > 0, "nchw_to_nhwc: bad shape");
    DISPATCH_T(dtype, hipLaunchKernelGGL(nchw_to_nhwc_kernel<T>, dim3(cdiv(H * W, 32), cdiv(C, 32), N), dim3(256), 0, (hipStream_t)stream,
                                         x_nchw, (T*)y, ldy, H * W, C));
    WU_LAUNCH_CHECK("nchw_to_nhwc");
    return 0;
}

extern "C" int wu_act_gate(const void* g, int ldg, const void* y, int ldy, void* out, int ldo,
                           int N, int H, int W, int C, int act, int dtype, void* stream) {
    const int esz = dtype == WU_BF16 ? 2 : 4;
    WU_REQUIRE(C % (16 / esz) == 0 && ok16(g, ldg, esz) && ok16(y, ldy, esz) && ok16(out, ldo, esz), "act_gate: alignment");
    const long long npix = (long long)N * H * W;
    const long long total = npix * (C / (16 / esz));
    if (total < (1ll << 31))
        DISPATCH_T(dtype, hipLaunchKernelGGL((act_gate_kernel<T, unsigned>), dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, (hipStream_t)stream,
                                             (const T*)g, ldg, (const T*)y, ldy, (T*)out, ldo, npix, C, act));
    else
        DISPATCH_T(dtype, hipLaunchKernelGGL((act_gate_kernel<T, long long>), dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, (hipStream_t)stream,
                                             (const T*)g, ldg, (const T*)y, ldy, (T*)out, ldo, npix, C, act));
    WU_LAUNCH_CHECK("act_gate");
    return 0;
}
