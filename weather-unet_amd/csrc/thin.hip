// Thin, HBM-bound ends of the networks (no MFMA: K = 27 or N = 3 cannot feed a matrix core):
//   - 3-input-channel conv3x3 read straight from the NCHW fp32 image: dconv_down1[0] (cunet.py:45 via
//     nets.py:20), disc.conv1[0] (3->3, s1) and disc.conv1[1] (3->64, s2) (disc.py:28 via nets.py:28-31);
//   - conv_last 1x1 (64->3) + tanh (cunet.py:39-40,80-82), forward and backward.
#include "wu_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------
// conv3x3, Cin = 3, NCHW fp32 input.  One thread = one output pixel x all COUT channels; the 27 x COUT
// weights sit in LDS and are read as wave-uniform (broadcast) float4s.
// ---------------------------------------------------------------------------------------------------
template <typename T, int COUT, int STRIDE, bool OUT_NCHW, int ACT>
__global__ __launch_bounds__(256) void conv3x3_c3_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, const float* __restrict__ inv_sigma,
                                                             void* __restrict__ yv, int ldy, int N, int H, int W) {
    constexpr int CP = (COUT + 3) & ~3;
    __shared__ __attribute__((aligned(16))) float wl[27][CP];
    __shared__ float bl[CP];
    const float s = inv_sigma ? *inv_sigma : 1.f;
    for (int i = threadIdx.x; i < 27 * CP; i += 256) {
        const int k = i / CP, co = i - k * CP;          // k = ci*9 + tap  (OIHW: w[co][ci][kh][kw])
        wl[k][co] = co < COUT ? w[co * 27 + k] * s : 0.f;
    }
    if (threadIdx.x < CP) bl[threadIdx.x] = (bias && threadIdx.x < COUT) ? bias[threadIdx.x] : 0.f;
    __syncthreads();
    const int Ho = (H - 1) / STRIDE + 1, Wo = (W - 1) / STRIDE + 1;
    const long long total = (long long)N * Ho * Wo;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ow = (int)(i % Wo);
        const int oh = (int)((i / Wo) % Ho);
        const int n = (int)(i / ((long long)Wo * Ho));
        float acc[CP];
#pragma unroll
        for (int c = 0; c < CP; ++c) acc[c] = bl[c];
        const float* xn = x + (size_t)n * 3 * H * W;
#pragma unroll 1
        for (int k = 0; k < 27; ++k) {
            const int ci = k / 9, t = k - ci * 9, kh = t / 3, kw = t - kh * 3;
            const int ih = oh * STRIDE + kh - 1, iw = ow * STRIDE + kw - 1;
            const float v = (ih >= 0 && ih < H && iw >= 0 && iw < W) ? xn[((size_t)ci * H + ih) * W + iw] : 0.f;
            const float4* wr = (const float4*)wl[k];
#pragma unroll
            for (int c4 = 0; c4 < CP / 4; ++c4) {
                const float4 ww = wr[c4];
                acc[4 * c4 + 0] += v * ww.x; acc[4 * c4 + 1] += v * ww.y;
                acc[4 * c4 + 2] += v * ww.z; acc[4 * c4 + 3] += v * ww.w;
            }
        }
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[c] = act_apply(acc[c], ACT);   // ACT is a compile-time constant here
        if (OUT_NCHW) {
            float* y = (float*)yv;
#pragma unroll
            for (int c = 0; c < COUT; ++c) y[((size_t)(n * COUT + c) * Ho + oh) * Wo + ow] = acc[c];
        } else {
            constexpr int E = ElemTraits<T>::kPer16B;
            T* y = (T*)yv + (size_t)i * ldy;
#pragma unroll
            for (int c = 0; c + E <= COUT; c += E) *(uint4*)(y + c) = pack16<T>(acc + c);
        }
    }
}

// bf16 production path of the 3->64 conv: the K = 27 (padded to 32) reduction on the matrix cores.
//   D[co][pixel] = W[co][k] * patch[k][pixel]   (v_mfma_f32_32x32x16_bf16, two K-steps, two 32-channel halves)
// A wave takes 32 consecutive output pixels per iteration: lane (pixel = lane & 31, K-half = lane >> 5) gathers its 2 x 8
// patch values straight from the NCHW fp32 image (lanes = consecutive pixels: coalesced 128-B rows), rounds them to bf16
// into the MFMA B operand -- no LDS, no barrier.  The 64 x 32 weight matrix lives in 16 registers per lane for the whole
// kernel.  Accumulators are transposed (lane owns 4 consecutive channels of its pixel); the epilogue transposes through a
// wave-private LDS patch (no barrier) so the stores cover whole pixel rows.  The VALU form above needs
// 1728 fp32 FMAs per pixel (fp32 VALU peak = 1/16 of the bf16 matrix rate); this one is bound by the 268 MB it writes.
// BITS: also write the gate bits of the output (ACT == RELU).  A template flag, not a run-time one: the loop must end in exactly ONE
// counted-wait asm statement (see below).
template <int STRIDE, int ACT, bool FULL, bool BITS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void conv3x3_c3_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, const float* __restrict__ inv_sigma,
                                                                  bf16_t* __restrict__ y, int ldy, int N, int H, int W,
                                                                  unsigned char* __restrict__ gbits) {
    __shared__ __attribute__((aligned(16))) char tile[4 * 32 * 144];
    __shared__ __attribute__((aligned(16))) float bias_s[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lh = lane >> 5;
    const float sg = inv_sigma ? *inv_sigma : 1.f;
    if (threadIdx.x < 64) bias_s[threadIdx.x] = bias ? bias[threadIdx.x] : 0.f;
    // weights as the MFMA A operand: row = co = 32 m + l31, k = 16 ks + 8 lh + j
    uint4 wf[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 16 * ks + 8 * lh + j;
                f[j] = k < 27 ? w[(32 * m + l31) * 27 + k] * sg : 0.f;
            }
            wf[m][ks] = pack16<bf16_t>(f);
        }
    // this lane's 16 K slots: k = 16 (t >> 3) + 8 lh + (t & 7) -> channel byte offset ci * H * W * 4 (one register per slot);
    // the (kh, kw) of a slot are compile-time constants per half-wave.  The five padding slots (k >= 27) point at the centre
    // pixel of channel 0: their weights are zero and that pixel is in the patch anyway.
    const int HW = H * W;
    unsigned chan[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int k = 16 * (t >> 3) + 8 * lh + (t & 7);
        chan[t] = k < 27 ? (unsigned)((k / 9) * HW) * 4u : 0u;
    }
    __syncthreads();       // bias_s

    const int Ho = (H - 1) / STRIDE + 1, Wo = (W - 1) / STRIDE + 1;
    const int total = N * Ho * Wo;                 // < 2^31 (checked by the host): 32-bit index arithmetic
    const int nblk = (total + 31) / 32;
    const int bstep = gridDim.x * 4;
    // The image is read through a buffer descriptor (x is < 2^30 bytes, host check): a tap outside the image gets a byte
    // offset >= 2^30 -- out of range, the load returns 0 -- so the padding needs neither clamped coordinates nor a select
    // afterwards.  Per block a lane computes 3 row terms and 3 column terms; a slot's offset is one v_add3 of the two terms
    // picked by half-wave and the slot's channel offset (the first version spent ~20 VALU instructions per slot on clamps,
    // 64-bit addresses and masks and was VALU-bound at 107 us).
    // The patch of block i + 1 is requested BEFORE block i is multiplied and stored -- with three waves per SIMD nothing else
    // covers the ~2 us round trip -- from inline asm, retired by ONE counted wait at the end of the iteration: vmcnt(4) leaves
    // this iteration's four stores in flight (left to the compiler, the loop header waits vmcnt(0), write acknowledgements
    // included).
    const wu_rsrc_t rs = wu_make_rsrc(x, (unsigned)((size_t)N * 3 * HW * 4));
    auto gather = [&](int blk, float (&v)[16]) __attribute__((always_inline)) {
        const int pix = blk * 32 + l31;
        const int pc = min(pix, total - 1);
        const int rowi = pc / Wo, ow = pc - rowi * Wo, n = rowi / Ho, oh = rowi - n * Ho;
        const int ih0 = oh * STRIDE - 1, iw0 = ow * STRIDE - 1;
        const unsigned nb = (unsigned)(n * 3 * HW) * 4u;
        unsigned rowb[3], colb[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            rowb[d] = (unsigned)(ih0 + d) < (unsigned)H ? nb + (unsigned)((ih0 + d) * W) * 4u : 0x40000000u;
            colb[d] = (unsigned)(iw0 + d) < (unsigned)W ? (unsigned)(iw0 + d) * 4u : 0x80000000u;
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int kA = 16 * (t >> 3) + (t & 7), kB = kA + 8;
            const int rA = kA < 27 ? kA % 9 : 4, rB = kB < 27 ? kB % 9 : 4;
            const unsigned off = (lh ? rowb[rB / 3] : rowb[rA / 3]) + (lh ? colb[rB % 3] : colb[rA % 3]) + chan[t];
            asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=&v"(v[t]) : "v"(off), "s"(rs) : "memory");
        }
    };
    float vn[16];
    int blk = blockIdx.x * 4 + wave;
#define WU_V16(v) "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), \
                  "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
    if (blk < nblk) {
        gather(blk, vn);
        asm volatile("s_waitcnt vmcnt(0)" : WU_V16(vn) : : "memory");
    }
    for (; blk < nblk; blk += bstep) {
        const uint4 pf0 = pack16<bf16_t>(vn), pf1 = pack16<bf16_t>(vn + 8);
        gather(min(blk + bstep, nblk - 1), vn);      // past the end: a harmless re-read of the last block
        f32x16_t acc[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[m][0]), __builtin_bit_cast(bf16x8_t, pf0), acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[m][1]), __builtin_bit_cast(bf16x8_t, pf1), acc[m], 0, 0, 0);
        }
        // transposed accumulators -> [pixel][64 ch] rows in this wave's private LDS patch (144-B rows), then 16-byte stores
        // that cover whole 128-byte pixel rows (8 lanes per pixel, 1 KiB contiguous per instruction when ldy == 64): the
        // output stream is the kernel's roofline, and row-per-lane stores (32 lines touched per instruction) ran at 1.2 TB/s
        char* tp = tile + wave * (32 * 144);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *(const float4*)(bias_s + 32 * m + 8 * g + 4 * lh);
                const int r0 = 4 * g;
                const uint32_t lo = pack_bf16x2(act_apply(acc[m][r0 + 0] + bv.x, ACT), act_apply(acc[m][r0 + 1] + bv.y, ACT));
                const uint32_t hi = pack_bf16x2(act_apply(acc[m][r0 + 2] + bv.z, ACT), act_apply(acc[m][r0 + 3] + bv.w, ACT));
                *(uint2*)(tp + l31 * 144 + (32 * m + 8 * g + 4 * lh) * 2) = make_uint2(lo, hi);
            }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // same wave: LDS operations execute in order
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = 8 * k + (lane >> 3), sl = lane & 7;
            const int p2 = blk * 32 + r;
            const uint4 v4 = *(const uint4*)(tp + r * 144 + sl * 16);
            // FULL (total % 32 == 0): unconditional stores.  Behind a branch the compiler cannot count them and waits vmcnt(0)
            // -- their write acknowledgements -- before it touches the prefetched patch at the top of the next iteration.
            // gate bits (wu_kernels.h): this lane's 8 channels 8 sl .. 8 sl + 7 are byte sl >> 1 of word (pixel, 0, sl & 1).  Computed
            // BEFORE the 16-byte store is issued: placed after it, the VALU results were allocated onto the store's address / data
            // registers and the stored rows came out corrupted at full size (more than one iteration per wave) although the
            // compiler's hazard rules were met -- nothing may write those registers while the store is young.
            uint32_t gate_byte = 0;
            if (BITS) {
                uint32_t t0, t1, t2, t3;
                const uint32_t one = 0x00010001u;          // v_pk_min_u16 with 1: 0/1 per non-negative bf16 half (see conv3x3_mfma_v2.hip)
                asm("v_pk_min_u16 %0, %1, %2" : "=&v"(t0) : "v"(v4.x), "v"(one));
                asm("v_pk_min_u16 %0, %1, %2" : "=&v"(t1) : "v"(v4.y), "v"(one));
                asm("v_pk_min_u16 %0, %1, %2" : "=&v"(t2) : "v"(v4.z), "v"(one));
                asm("v_pk_min_u16 %0, %1, %2" : "=&v"(t3) : "v"(v4.w), "v"(one));
                const uint32_t xx = t0 | (t1 << 2) | (t2 << 4) | (t3 << 6);
                gate_byte = (xx | (xx >> 15)) & 0xffu;
            }
            __builtin_amdgcn_sched_barrier(0);
            if (FULL || p2 < total) *(uint4*)(y + (size_t)p2 * ldy + sl * 8) = v4;
            if (BITS && (FULL || p2 < total)) gbits[(size_t)p2 * 8 + (sl & 1) * 4 + (sl >> 1)] = (unsigned char)gate_byte;
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_wave_barrier();
        // The next block's patch has landed.  ONE asm statement with a compile-time count: its tied operands must be the very
        // registers the asm loads wrote.  With several alternative wait statements (run-time if / else) the register allocator
        // copies the patch registers into each statement's operands BEFORE the wait -- i.e. before the data has arrived -- and the
        // kernel computes on the previous block's patch (found at full size only: one iteration per wave hides it).
        // tests/test_isa_cpu.py checks in the generated code that nothing reads those registers between the loads and the wait.
        // vmcnt(4) leaves this iteration's four stores in flight, vmcnt(8) the four stores + four gate-byte stores.
        asm volatile("s_waitcnt vmcnt(%16)" : WU_V16(vn) : "n"(FULL ? (BITS ? 8 : 4) : 0) : "memory");
    }
#undef WU_V16
}

// NHWC-output form of the conv above (Cout = 64), production path.  One thread = one output pixel with its 27 inputs in
// registers; the output channels are walked in the OUTER loop, so the weights w[co][0..26] are wave-uniform and come
// through the scalar cache (s_load) straight into the FMAs: no LDS weight traffic (the LDS-broadcast form above spends
// 432 ds_read_b128 per pixel).  Results are packed 8 channels at a time into an LDS tile [256 px][64 ch] (144-B rows) and
// copied out with 16-B stores that cover 1 KiB contiguous per wave instruction.
template <typename T, int STRIDE, int ACT>
__global__ __launch_bounds__(256) void conv3x3_c3_fwd_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, const float* __restrict__ inv_sigma,
                                                                  T* __restrict__ y, int ldy, int N, int H, int W) {
    constexpr int E = ElemTraits<T>::kPer16B;
    constexpr int ROW = 64 * (int)sizeof(T) + 16;
    __shared__ __attribute__((aligned(16))) char tile[256 * ROW];
    const float s = inv_sigma ? *inv_sigma : 1.f;
    const int Ho = (H - 1) / STRIDE + 1, Wo = (W - 1) / STRIDE + 1;
    const long long total = (long long)N * Ho * Wo;
    const long long nblk = (total + 255) / 256;
    for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const long long pix = blk * 256 + threadIdx.x;
        float xv[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) xv[k] = 0.f;
        if (pix < total) {
            const int ow = (int)(pix % Wo), oh = (int)((pix / Wo) % Ho), n = (int)(pix / ((long long)Wo * Ho));
            const float* xn = x + (size_t)n * 3 * H * W;
#pragma unroll
            for (int k = 0; k < 27; ++k) {
                const int ci = k / 9, t = k % 9, ih = oh * STRIDE + t / 3 - 1, iw = ow * STRIDE + t % 3 - 1;
                if (ih >= 0 && ih < H && iw >= 0 && iw < W) xv[k] = xn[((size_t)ci * H + ih) * W + iw];
            }
        }
#pragma unroll 1
        for (int cg = 0; cg < 64 / E; ++cg) {
            float o[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int co = cg * E + e;                       // wave-uniform: scalar loads
                const float* wr = w + co * 27;
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < 27; ++k) a += xv[k] * wr[k];
                o[e] = act_apply(a * s + (bias ? bias[co] : 0.f), ACT);
            }
            *(uint4*)(tile + threadIdx.x * ROW + cg * 16) = pack16<T>(o);
        }
        __syncthreads();
        constexpr int SLOTS = 64 * (int)sizeof(T) / 16;
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
            const int q = threadIdx.x + 256 * k;
            const int r = q / SLOTS, sl = q % SLOTS;
            const long long p2 = blk * 256 + r;
            if (p2 < total) *(uint4*)(y + (size_t)p2 * ldy + sl * E) = *(const uint4*)(tile + r * ROW + sl * 16);
        }
        __syncthreads();
    }
}

// weight/bias gradient: dW[co][k] += sum_pix dy'[pix][co] * patch[pix][k], k = ci*9+tap (27).
// Workgroup = 64 couts x 4 k-groups; pixels staged through LDS in batches of 64.
template <typename T, int STRIDE, bool DY_NCHW>
__global__ __launch_bounds__(256) void conv3x3_c3_wgrad_kernel(const float* __restrict__ x, const void* __restrict__ dyv, int lddy,
                                                               const void* __restrict__ yv, int ldy, int act,
                                                               float* __restrict__ dw, float* __restrict__ dbias, float* __restrict__ part,
                                                               int N, int H, int W, int Cout) {
    __shared__ float patch[27][64];
    __shared__ float dyl[64][65];
    const int tid = threadIdx.x, co = tid & 63, q = tid >> 6;
    const int Ho = (H - 1) / STRIDE + 1, Wo = (W - 1) / STRIDE + 1;
    const long long total = (long long)N * Ho * Wo;
    const int cog = blockIdx.y * 64;     // cout group
    float acc[7], bsum = 0.f;
#pragma unroll
    for (int j = 0; j < 7; ++j) acc[j] = 0.f;
    for (long long base = (long long)blockIdx.x * 64; base < total; base += (long long)gridDim.x * 64) {
        for (int i = tid; i < 27 * 64; i += 256) {
            const int k = i >> 6, p = i & 63;
            const long long pix = base + p;
            float v = 0.f;
            if (pix < total) {
                const int ow = (int)(pix % Wo), oh = (int)((pix / Wo) % Ho), n = (int)(pix / ((long long)Wo * Ho));
                const int ci = k / 9, t = k - ci * 9, ih = oh * STRIDE + t / 3 - 1, iw = ow * STRIDE + t % 3 - 1;
                if (ih >= 0 && ih < H && iw >= 0 && iw < W) v = x[(((size_t)n * 3 + ci) * H + ih) * W + iw];
            }
            patch[k][p] = v;
        }
        for (int i = tid; i < 64 * 64; i += 256) {
            const int p = i >> 6, c = i & 63;
            const long long pix = base + p;
            float g = 0.f;
            if (pix < total && cog + c < Cout) {
                if (DY_NCHW) {
                    const int hw = Ho * Wo;
                    const size_t o = ((size_t)(pix / hw) * Cout + cog + c) * hw + (pix % hw);
                    g = ((const float*)dyv)[o];
                    if (yv) g = act_gate(g, ((const float*)yv)[o], act);
                } else {
                    g = ElemTraits<T>::load((const T*)dyv + (size_t)pix * lddy + cog + c);
                    if (yv) g = act_gate(g, ElemTraits<T>::load((const T*)yv + (size_t)pix * ldy + cog + c), act);
                }
            }
            dyl[p][c] = g;
        }
        __syncthreads();
#pragma unroll 4
        for (int p = 0; p < 64; ++p) {
            const float d = dyl[p][co];
            if (q == 0) bsum += d;
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int k = q + 4 * j;
                if (k < 27) acc[j] += d * patch[k][p];
            }
        }
        __syncthreads();
    }
    if (cog + co < Cout) {
        // part != NULL (Cout <= 64, one cout group): this workgroup's slice of the partial-sum slab, folded in workgroup order by
        // thin_fold_kernel (dW at [co*27 + k], dbias behind it at [Cout*27 + co]); else fp32 atomics
        float* mine = part ? part + (size_t)blockIdx.x * (64 * 27 + 64) : nullptr;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int k = q + 4 * j;
            if (k < 27) { if (mine) mine[co * 27 + k] = acc[j]; else atomicAdd(&dw[(size_t)(cog + co) * 27 + k], acc[j]); }
        }
        if (q == 0 && dbias) { if (mine) mine[Cout * 27 + co] = bsum; else atomicAdd(&dbias[cog + co], bsum); }
    }
}

// Deterministic combination of per-workgroup partial sums: slab[nblk][stride] -> out0[0..n0) (+ out1[0..n1) from offset n0),
// summed over the workgroups in index order (4 independent chains for load-level parallelism, combined in a fixed tree).
constexpr int kC3Slab = 64 * 27 + 64;        // dW (1728) + dbias (64) per workgroup of conv3x3_c3_wgrad_mfma_kernel
constexpr int kC1Slab = 3 * 64 + 3;          // dW (192) + dbias (3) per workgroup of conv1x1_tanh_bwd_kernel
constexpr int kThinMaxBlocks = 1024;
__global__ __launch_bounds__(256) void thin_fold_kernel(const float* __restrict__ slab, int nblk, int stride, float* __restrict__ out0, int n0,
                                                        float* __restrict__ out1, int n1, int accumulate) {
    // 8 outputs x 32 segments per workgroup: segment s sums workgroups s, s+32, ... (independent loads), the 32 segment sums
    // are combined in segment order through LDS -> deterministic, and ~1000 partials per output no longer sit on one thread
    __shared__ float red[32][8];
    const int oi = threadIdx.x & 7, seg = threadIdx.x >> 3;
    const int i = blockIdx.x * 8 + oi;
    float a = 0.f;
    if (i < n0 + n1) {
#pragma unroll 4
        for (int b = seg; b < nblk; b += 32) a += slab[(size_t)b * stride + i];
    }
    red[seg][oi] = a;
    __syncthreads();
    if (seg == 0 && i < n0 + n1) {
        float v = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < 32; ++s_) v += red[s_][oi];
        if (i < n0) out0[i] = accumulate ? out0[i] + v : v;
        else if (out1) out1[i - n0] = accumulate ? out1[i - n0] + v : v;
    }
}

// bf16 production path of the weight gradient above: dW[64 co][27 -> 32 k] = dY'^T [64 x P] * patch [P x 32] on the
// matrix cores.  Per 256-pixel tile (linear pixel order, any image shape) the gated dY' tile and a per-pixel
// 27-element patch row (bf16, padded to 32) are staged in LDS; both MFMA operands have the pixel (K) as their
// slow index, so fragments come through ds_read_b64_tr_b16 as in conv3x3_wgrad.  4 waves split the 16 K-steps;
// accumulators persist over a grid-stride loop of tiles, then one LDS reduction + fp32 atomics per workgroup.
template <int STRIDE, bool GATE>
__global__ __launch_bounds__(256) void conv3x3_c3_wgrad_mfma_kernel(const float* __restrict__ x, const bf16_t* __restrict__ dy, int lddy,
                                                                    const bf16_t* __restrict__ y, int ldy, int act,
                                                                    float* __restrict__ dw, float* __restrict__ dbias,
                                                                    float* __restrict__ part, int N, int H, int W) {
    __shared__ __attribute__((aligned(16))) char smem[256 * 128 + 256 * 64];
    char* dy_lds = smem;                 // [256 px][64 co] bf16, 64-B halves swapped on odd pixel pairs
    char* pa_lds = smem + 256 * 128;     // [256 px][32 k] bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Ho = (H - 1) / STRIDE + 1, Wo = (W - 1) / STRIDE + 1;
    const long long total = (long long)N * Ho * Wo;
    const long long ntiles = (total + 255) / 256;
    f32x16_t acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
    float bs8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bs8[e] = 0.f;
    const int g16 = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3, h = g16 >> 1;

    // Register prefetch: the global loads of tile t+1 (8 dY' chunks [+ 8 gate chunks], 27 image values) are issued before the
    // MFMAs of tile t and staged to LDS after them; every load is unconditional from clamped coordinates (a conditional
    // load compiles to a branch with its own s_waitcnt: one serialized round trip per item).
    uint4 dv[8], yv[8];
    float pv[32];
#pragma unroll
    for (int k = 27; k < 32; ++k) pv[k] = 0.f;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((size_t)N * 3 * H * W * 4), 0x00020000);
    auto fetch = [&](long long tile) __attribute__((always_inline)) {
        const long long base = tile * 256;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int item = tid + 256 * k;
            const int r = item >> 3, s_ = item & 7;
            const long long pc = min(base + r, total - 1);
            dv[k] = *(const uint4*)(dy + (size_t)pc * lddy + s_ * 8);
            if (GATE) yv[k] = *(const uint4*)(y + (size_t)pc * ldy + s_ * 8);
        }
        const long long pc = min(base + tid, total - 1);
        // 32-bit decomposition (the host keeps N*Ho*Wo below 2^31 on this path: a 64-bit division is a ~100-instruction routine)
        const unsigned pc32 = (unsigned)pc, rowi = pc32 / (unsigned)Wo;
        const int ow = (int)(pc32 - rowi * (unsigned)Wo), n = (int)(rowi / (unsigned)Ho), oh = (int)(rowi - (unsigned)n * (unsigned)Ho);
        // the image goes through a buffer descriptor (< 2^30 bytes, host check): a tap outside the image -- or a pixel past the
        // end -- gets an out-of-range offset and loads 0, so the 27 offsets are one add each on top of 3 row + 3 column terms
        // (clamped 64-bit addresses + selects were ~20 VALU instructions per tap)
        const bool pok = base + tid < total;
        const unsigned nb = (unsigned)(n * 3 * H * W) * 4u, hw4 = (unsigned)(H * W) * 4u;
        unsigned rc[9];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int ih = oh * STRIDE + d - 1;
            const unsigned rowb = (pok && (unsigned)ih < (unsigned)H) ? nb + (unsigned)(ih * W) * 4u : 0x40000000u;
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const int iw = ow * STRIDE + e - 1;
                rc[3 * d + e] = rowb + ((unsigned)iw < (unsigned)W ? (unsigned)iw * 4u : 0x80000000u);
            }
        }
#pragma unroll
        for (int k = 0; k < 27; ++k)
            pv[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, rc[k % 9] + (unsigned)(k / 9) * hw4, 0, 0));
    };
    auto stage = [&](long long tile) __attribute__((always_inline)) {
        const long long base = tile * 256;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int item = tid + 256 * k;
            const int r = item >> 3, s_ = item & 7;
            uint4 v = dv[k];
            if (GATE) v = gate16<bf16_t>(v, yv[k], act);
            if (base + r >= total) v = make_uint4(0, 0, 0, 0);
            *(uint4*)(dy_lds + r * 128 + ((s_ * 16) ^ (((r >> 1) & 1) << 6))) = v;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) *(uint4*)(pa_lds + tid * 64 + c * 16) = pack16<bf16_t>(pv + 8 * c);
    };
    if ((long long)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        stage(tile);
        __syncthreads();
        if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);
        if (dbias) {       // 16-byte reads: 8 per thread and tile (was 64 two-byte ones)
            const int c8 = tid & 7;
#pragma unroll
            for (int r = tid >> 3; r < 256; r += 32) {
                float f[8];
                unpack16<bf16_t>(*(const uint4*)(dy_lds + r * 128 + ((c8 ^ (((r >> 1) & 1) << 2)) << 4)), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) bs8[e] += f[e];
            }
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ks = 4 * kk + wave;
            const int r0 = 16 * ks + 8 * h + q;
            const char* pb = pa_lds + r0 * 64 + (16 * (g16 & 1) + 4 * pp) * 2;
            const s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)pb);
            const s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(pb + 4 * 64));
            const uint4 bf = make_uint4(((const uint32_t*)&b0)[0], ((const uint32_t*)&b0)[1], ((const uint32_t*)&b1)[0], ((const uint32_t*)&b1)[1]);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int cb = (32 * m + 16 * (g16 & 1) + 4 * pp) * 2;
                const char* pa = dy_lds + r0 * 128 + (cb ^ (((q >> 1) & 1) << 6));
                const s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)pa);
                const s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(pa + 4 * 128));
                const uint4 af = make_uint4(((const uint32_t*)&a0)[0], ((const uint32_t*)&a0)[1], ((const uint32_t*)&a1)[0], ((const uint32_t*)&a1)[1]);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af), __builtin_bit_cast(bf16x8_t, bf), acc[m], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // ---- reduce the 4 waves through LDS, then one atomic per output element ----
    float* red = (float*)smem;           // [4 waves][32 regs][64 lanes]
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) red[(wave * 32 + m * 16 + i) * 64 + lane] = acc[m][i];
    __syncthreads();
    for (int e = tid; e < 32 * 64; e += 256) {
        const int reg = e >> 6, ln = e & 63;
        const float s = red[(0 * 32 + reg) * 64 + ln] + red[(1 * 32 + reg) * 64 + ln] + red[(2 * 32 + reg) * 64 + ln] + red[(3 * 32 + reg) * 64 + ln];
        const int m = reg >> 4, i = reg & 15;
        const int co = 32 * m + (i & 3) + 8 * (i >> 2) + 4 * (ln >> 5), k = ln & 31;
        // part != NULL: this workgroup's slice of the partial-sum slab (summed in fixed order by thin_fold_kernel)
        if (k < 27) { if (part) part[(size_t)blockIdx.x * kC3Slab + co * 27 + k] = s; else atomicAdd(&dw[co * 27 + k], s); }
    }
    if (dbias) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[tid * 8 + e] = bs8[e];
        __syncthreads();
        if (tid < 64) {
            float sb = 0.f;
            for (int j = 0; j < 32; ++j) sb += red[(j * 8 + (tid >> 3)) * 8 + (tid & 7)];
            if (part) part[(size_t)blockIdx.x * kC3Slab + 1728 + tid] = sb; else atomicAdd(&dbias[tid], sb);
        }
    }
}

// data gradient wrt the NCHW fp32 image (D differentiated wrt G's output): one thread = one input pixel.
template <typename T, int STRIDE, bool DY_NCHW>
__global__ __launch_bounds__(256) void conv3x3_c3_dgrad_kernel(const void* __restrict__ dyv, int lddy, const void* __restrict__ yv, int ldy,
                                                               int act, const float* __restrict__ w, const float* __restrict__ inv_sigma,
                                                               float* __restrict__ dx, int N, int H, int W, int Cout, int accumulate) {
    extern __shared__ float wl[];   // [27][Cout]
    const float s = inv_sigma ? *inv_sigma : 1.f;
    for (int i = threadIdx.x; i < 27 * Cout; i += 256) {
        const int k = i / Cout, co = i - k * Cout;
        wl[i] = w[co * 27 + k] * s;
    }
    __syncthreads();
    const int Ho = (H - 1) / STRIDE + 1, Wo = (W - 1) / STRIDE + 1;
    const long long total = (long long)N * H * W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int iw = (int)(i % W), ih = (int)((i / W) % H), n = (int)(i / ((long long)W * H));
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int kh = 0; kh < 3; ++kh) {
            const int th = ih + 1 - kh;
            if (th < 0 || th % STRIDE) continue;
            const int oh = th / STRIDE;
            if (oh >= Ho) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int tw = iw + 1 - kw;
                if (tw < 0 || tw % STRIDE) continue;
                const int ow = tw / STRIDE;
                if (ow >= Wo) continue;
                const int t = kh * 3 + kw;
                const size_t pix = ((size_t)n * Ho + oh) * Wo + ow;
                for (int co = 0; co < Cout; ++co) {
                    float g;
                    if (DY_NCHW) {
                        const size_t o = (((size_t)n * Cout + co) * Ho + oh) * Wo + ow;
                        g = ((const float*)dyv)[o];
                        if (yv) g = act_gate(g, ((const float*)yv)[o], act);
                    } else {
                        g = ElemTraits<T>::load((const T*)dyv + pix * lddy + co);
                        if (yv) g = act_gate(g, ElemTraits<T>::load((const T*)yv + pix * ldy + co), act);
                    }
                    a0 += g * wl[(0 * 9 + t) * Cout + co];
                    a1 += g * wl[(1 * 9 + t) * Cout + co];
                    a2 += g * wl[(2 * 9 + t) * Cout + co];
                }
            }
        }
        const size_t hw = (size_t)H * W, o = (size_t)n * 3 * hw + (size_t)ih * W + iw;
        if (accumulate) { dx[o] += a0; dx[o + hw] += a1; dx[o + 2 * hw] += a2; }
        else { dx[o] = a0; dx[o + hw] = a1; dx[o + 2 * hw] = a2; }
    }
}

// Data gradient wrt the image for NHWC dy with Cout % (elements per 16 B) == 0 (the 3->64 convs): LP lanes share
// one INPUT pixel, each owning 16 B of dy channels; per contributing tap one coalesced 16-B load + 3 partial dot
// products against the LDS-resident weights, finished with xor-shuffles.  Memory-bound (dy read ~once from L2).
template <typename T, int STRIDE>
__global__ __launch_bounds__(256) void conv3x3_c3_dgrad_nhwc_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ y, int ldy,
                                                                    int act, const float* __restrict__ w, const float* __restrict__ inv_sigma,
                                                                    float* __restrict__ dx, int N, int H, int W, int Cout, int accumulate) {
    constexpr int E = ElemTraits<T>::kPer16B;
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [tap][ci][Cout]
    const float sg = inv_sigma ? *inv_sigma : 1.f;
    for (int i = threadIdx.x; i < 27 * Cout; i += 256) {
        const int t = i / (3 * Cout), r = i - t * 3 * Cout, ci = r / Cout, co = r - ci * Cout;
        wl[i] = w[co * 27 + ci * 9 + t] * sg;
    }
    __syncthreads();
    const int LP = Cout / E;                     // lanes per pixel (8 for bf16 / 64 channels)
    const int cl = threadIdx.x % LP;
    const int Ho = (H - 1) / STRIDE + 1, Wo = (W - 1) / STRIDE + 1;
    const long long total = (long long)N * H * W;
    const int ppi = 256 / LP;
    const long long iters = (total + ppi - 1) / ppi;
    for (long long it = blockIdx.x; it < iters; it += gridDim.x) {
        const long long pix = it * ppi + threadIdx.x / LP;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        int iw = 0, ih = 0, n = 0;
        if (pix < total) {
            {   // 32-bit index arithmetic (64-bit divisions cost several hundred instructions per pixel)
                const unsigned p32 = (unsigned)pix, rowi = p32 / (unsigned)W;
                iw = (int)(p32 - rowi * (unsigned)W); n = (int)(rowi / (unsigned)H); ih = (int)(rowi - (unsigned)n * (unsigned)H);
            }
            // Candidate taps per dimension: all three at stride 1; at stride 2 only the taps of matching parity (kh = 1, or
            // kh in {0, 2}).  Every candidate is loaded UNCONDITIONALLY from clamped coordinates and zeroed when it falls
            // outside (a conditional load compiles to a branch with its own s_waitcnt: serialized round trips).
            constexpr int NS = STRIDE == 1 ? 3 : 2;
            uint4 gvs[NS * NS];
            int taps[NS * NS];
#pragma unroll
            for (int a = 0; a < NS; ++a) {
                const int ph = (ih + 1) & 1;
                const int kh = STRIDE == 1 ? a : (ph ? (a == 0 ? 1 : 3) : 2 * a);
                const int th = ih + 1 - kh;
                const bool okh = kh < 3 && th >= 0 && th / STRIDE < Ho;
                const int ohc = min(max(th, 0) / STRIDE, Ho - 1);
#pragma unroll
                for (int b = 0; b < NS; ++b) {
                    const int pw = (iw + 1) & 1;
                    const int kw = STRIDE == 1 ? b : (pw ? (b == 0 ? 1 : 3) : 2 * b);
                    const int tw = iw + 1 - kw;
                    const bool okw = kw < 3 && tw >= 0 && tw / STRIDE < Wo;
                    const int owc = min(max(tw, 0) / STRIDE, Wo - 1);
                    const size_t op = ((size_t)n * Ho + ohc) * Wo + owc;
                    uint4 gv = *(const uint4*)(dy + op * lddy + cl * E);
                    if (y) gv = gate16<T>(gv, *(const uint4*)(y + op * ldy + cl * E), act);
                    const bool okk = okh && okw;
                    gvs[a * NS + b] = okk ? gv : make_uint4(0u, 0u, 0u, 0u);
                    taps[a * NS + b] = okk ? kh * 3 + kw : 0;
                }
            }
#pragma unroll
            for (int sidx = 0; sidx < NS * NS; ++sidx) {
                float g[E];
                unpack16<T>(gvs[sidx], g);
                const float* wt = wl + taps[sidx] * 3 * Cout + cl * E;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    a0 += g[e] * wt[e];
                    a1 += g[e] * wt[Cout + e];
                    a2 += g[e] * wt[2 * Cout + e];
                }
            }
        }
        for (int m = LP >> 1; m > 0; m >>= 1) {
            a0 += __shfl_xor(a0, m); a1 += __shfl_xor(a1, m); a2 += __shfl_xor(a2, m);
        }
        if (cl == 0 && pix < total) {
            const size_t hw = (size_t)H * W, o = (size_t)n * 3 * hw + (size_t)ih * W + iw;
            if (accumulate) { dx[o] += a0; dx[o + hw] += a1; dx[o + 2 * hw] += a2; }
            else { dx[o] = a0; dx[o + hw] = a1; dx[o + 2 * hw] = a2; }
        }
    }
}

// bf16 production path of the stride-2 data gradient above (disc.conv1[1], 64 channels -> the 3-channel image), on the matrix
// cores.  Per OUTPUT pixel q the 27 products P[k][q] = sum_co W[co][k] * dY'[q][co] are one 32 x 32 x 64 MFMA block per 32
// pixels (k = ci*9 + kh*3 + kw, padded to 32); an input pixel then sums the 1, 2 or 4 entries of P that reach it (stride 2: the
// taps of matching parity), in a fixed order -- no atomics.  A workgroup takes an 8 x 32 tile of output pixels plus one halo row
// and column (the odd input rows / columns at the far edge read them), parks P in LDS as [k][q] and writes the 16 x 64 input
// pixels of its tile with row-contiguous stores.  The VALU formulation above spends ~400 instructions per 8 input pixels (LDS
// weight reads, shuffles, 64-bit addresses): 402 us at B = 32 256x256 against ~100 MB of traffic.
template <bool GATE>
__global__ __launch_bounds__(256) void conv3x3_c3_dgrad_s2_mfma_kernel(const bf16_t* __restrict__ dy, int lddy, const bf16_t* __restrict__ y, int ldy,
                                                                       int act, const float* __restrict__ w, const float* __restrict__ inv_sigma,
                                                                       float* __restrict__ dx, int H, int W, int tiles_x, int tiles_y, int accumulate) {
    constexpr int TH = 8, TW = 32, QW = TW + 1, QN = (TH + 1) * QW, NB = (QN + 31) / 32, PS = NB * 32 + 1;
    __shared__ float P[27 * PS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const float sg = inv_sigma ? *inv_sigma : 1.f;
    // weights as the MFMA A operand: row = k = l31 (27 real rows), reduction index co = 16 ks + 8 lh + j
    uint4 wf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = l31 < 27 ? w[(16 * ks + 8 * lh + j) * 27 + l31] * sg : 0.f;
        wf[ks] = pack16<bf16_t>(f);
    }
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y, n = t / tiles_y;
    const int oh0 = ty * TH, ow0 = tx * TW;
    // ---- P = W^T dY' for the (TH + 1) x (TW + 1) output pixels, 32 at a time (blocks wave, wave + 4, wave + 8) ----
    constexpr int MAXB = (NB + 3) / 4;
    uint4 av[MAXB][4];
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        const int b = wave + 4 * i;
        const int q = 32 * b + l31, r = q / QW, c = q - r * QW;
        const int oh = oh0 + r, ow = ow0 + c;
        const bool ok = b < NB && q < QN && oh < Ho && ow < Wo;
        const size_t op = ((size_t)n * Ho + min(oh, Ho - 1)) * Wo + min(ow, Wo - 1);
        const unsigned m = ok ? 0xffffffffu : 0u;       // lane mask instead of a 128-bit select (which goes through scratch)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 v = *(const uint4*)(dy + op * lddy + 16 * ks + 8 * lh);
            if (GATE) v = gate16<bf16_t>(v, *(const uint4*)(y + op * ldy + 16 * ks + 8 * lh), act);
            av[i][ks] = make_uint4(v.x & m, v.y & m, v.z & m, v.w & m);
        }
    }
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        const int b = wave + 4 * i;
        if (b >= NB) break;
        f32x16_t acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[ks]), __builtin_bit_cast(bf16x8_t, av[i][ks]), acc, 0, 0, 0);
        // lane (pixel l31, half lh) holds rows k = 8 g + 4 lh + e of its pixel's column
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 8 * g + 4 * lh + e;
                if (k < 27) P[k * PS + 32 * b + l31] = acc[4 * g + e];
            }
    }
    __syncthreads();
    // ---- input pixels (2 oh0 + row, 2 ow0 + col), row < 16, col < 64: taps of matching parity, summed in (kh, kw) order ----
    const int col = tid & 63, iw = 2 * ow0 + col;
    const bool pw = col & 1;
    const int kwA = pw ? 0 : 1, cA = (col + 1) >> 1, cB = col >> 1;
    const size_t hw = (size_t)H * W;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (tid >> 6) + 4 * j, ih = 2 * oh0 + row;
        const bool ph = row & 1;
        const int khA = ph ? 0 : 1, rA = (row + 1) >> 1, rB = row >> 1;
        if (ih < H && iw < W) {
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                const float* pk = P + ci * 9 * PS;
                float sacc = pk[(khA * 3 + kwA) * PS + rA * QW + cA];
                const float v1 = pk[(khA * 3 + 2) * PS + rA * QW + cB];
                const float v2 = pk[(6 + kwA) * PS + rB * QW + cA];
                const float v3 = pk[8 * PS + rB * QW + cB];
                sacc += pw ? v1 : 0.f;
                sacc += ph ? v2 : 0.f;
                sacc += (ph && pw) ? v3 : 0.f;
                float* o = dx + ((size_t)n * 3 + ci) * hw + (size_t)ih * W + iw;
                *o = accumulate ? *o + sacc : sacc;
            }
        }
    }
}

// Weight gradient of the 3->3 image-layout conv (disc.conv1[0]: NCHW fp32 dy, Cout <= 4): one thread per output
// pixel accumulates all Cout*27 products in registers over a grid-stride loop; wave shuffle + LDS reduce; atomics.
template <int STRIDE>
__global__ __launch_bounds__(256) void conv3x3_c3_wgrad_small_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                     const float* __restrict__ y, int act, float* __restrict__ dw,
                                                                     float* __restrict__ dbias, float* __restrict__ part, int N, int H, int W, int Cout) {
    __shared__ float red[4][84];
    const int Ho = (H - 1) / STRIDE + 1, Wo = (W - 1) / STRIDE + 1;
    const long long total = (long long)N * Ho * Wo;
    float acc[3][27], bs[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        bs[c] = 0.f;
#pragma unroll
        for (int k = 0; k < 27; ++k) acc[c][k] = 0.f;
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ow = (int)(i % Wo), oh = (int)((i / Wo) % Ho), n = (int)(i / ((long long)Wo * Ho));
        float g[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            g[c] = 0.f;
            if (c < Cout) {
                const size_t o = (((size_t)n * Cout + c) * Ho + oh) * Wo + ow;
                g[c] = dy[o];
                if (y) g[c] = act_gate(g[c], y[o], act);
            }
            bs[c] += g[c];
        }
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            const int ci = k / 9, t = k % 9, ih = oh * STRIDE + t / 3 - 1, iw = ow * STRIDE + t % 3 - 1;
            const float v = (ih >= 0 && ih < H && iw >= 0 && iw < W) ? x[(((size_t)n * 3 + ci) * H + ih) * W + iw] : 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c][k] += g[c] * v;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            float v = acc[c][k];
            for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
            if (lane == 0) red[wave][c * 27 + k] = v;
        }
        float b = bs[c];
        for (int m = 32; m > 0; m >>= 1) b += __shfl_xor(b, m);
        if (lane == 0) red[wave][81 + c] = b;
    }
    __syncthreads();
    if (threadIdx.x < 84) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        float* mine = part ? part + (size_t)blockIdx.x * (64 * 27 + 64) : nullptr;      // slab slice: dW [Cout*27], then dbias [Cout]
        if (threadIdx.x < 81) {
            if (threadIdx.x / 27 < Cout) { if (mine) mine[threadIdx.x] = v; else atomicAdd(&dw[threadIdx.x], v); }
        } else if (dbias && threadIdx.x - 81 < Cout) {
            if (mine) mine[Cout * 27 + threadIdx.x - 81] = v; else atomicAdd(&dbias[threadIdx.x - 81], v);
        }
    }
}

// sum over the LP (power of two) consecutive lanes that share a pixel; every lane of the group ends with the total.
// DPP lane moves folded into the adds (no LDS crossbar): xor 1, xor 2, then mirrors (lane i <-> 7 - i, i <-> 15 - i), which
// pair equal-valued groups just like xor 4 / xor 8 would.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_group_sum(float v, int LP) {
    if (LP >= 2) v += dpp_mov<0xB1>(v);         // quad_perm [1,0,3,2]
    if (LP >= 4) v += dpp_mov<0x4E>(v);         // quad_perm [2,3,0,1]
    if (LP >= 8) v += dpp_mov<0x141>(v);        // row_half_mirror
    if (LP >= 16) v += dpp_mov<0x140>(v);       // row_mirror
    if (LP >= 32) v += __shfl_xor(v, 16);
    if (LP >= 64) v += __shfl_xor(v, 32);
    return v;
}

// ---------------------------------------------------------------------------------------------------
// The image-layout 3 -> 3 stride-1 conv of the discriminator (disc.conv1[0], nets.py:28: NCHW fp32 in and out, no activation behind
// it) through an LDS tile: a workgroup stages a 16 x 64 pixel tile of the three input planes with its one-pixel halo (3 x 18 x 66
// floats, coalesced rows) and every thread computes FOUR adjacent pixels of one row from it -- 54 LDS reads for 4 x 81 FMAs instead of
// 27 dependent global loads per pixel (the one-thread-per-pixel kernels above: 74 / 93 / 54 us forward / weight gradient / data
// gradient at B = 32, 256 x 256, for 50 MB of traffic).
//   img3_conv_kernel<ACT, false>: y = act(conv(x, w) + bias);   <ACT_NONE, true>: the data gradient = the same conv of dy with the
//   weights transposed and flipped (w'[ci][co][kh][kw] = w[co][ci][2-kh][2-kw]), no bias.
//   img3_wgrad_kernel: dW, dbias partials per workgroup over a grid-stride loop of tiles (fixed order: deterministic).
// ---------------------------------------------------------------------------------------------------
constexpr int kI3H = 16, kI3W = 64, kI3P = 68;          // tile rows / columns, LDS row pitch (floats; 16-byte aligned rows)

// acc += a * b as ONE v_fmac_f32, never half of a v_pk_fma_f32.  Measured on MI355X (scratch/img3_concurrency_probe.py, round 3): with
// the compiler's packed form (162 v_pk_fma_f32 in the conv kernel) a few 16-lane groups per launch came back with ONE accumulator
// wrong (the low half of a packed pair, off by about one tap's product) whenever the kernel ran on a second stream BESIDE the stem's
// MFMA kernel (stem7x7_fwd_mfma_kernel: wave64 MFMAs with VGPR accumulators, 220 VGPRs) -- 10 of 10 launches, on several boxes, never
// alone and never beside VALU-only kernels or the other MFMA kernels; with scalar FMAs 0 of 10.  The library is therefore also built
// without packed-FP32 VALU ops (wu/_build.py: -target-feature -packed-fp32-ops); this helper keeps the property if that flag is dropped.
__device__ __forceinline__ void fmac1(float& acc, float a, float b) { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b)); }

// 54 halo rows (3 planes x 18) of 66 floats: wave w takes rows w, w + 4, ... -- one coalesced 256-byte request per row and wave, all 14
// issued before the first is stored (the one-loop form waited for every load in turn); the two right-hand halo columns go to 108 threads.
__device__ __forceinline__ void img3_stage(float (*xs)[kI3H + 2][kI3P], const float* __restrict__ xn, int H, int W, int h0, int w0) {
    constexpr int R = 3 * (kI3H + 2), NK = (R + 3) / 4;
    const int lane = threadIdx.x & 63, rsub = threadIdx.x >> 6;
    const int iw = w0 - 1 + lane;
    const bool cok = iw >= 0 && iw < W;
    float v[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int row = rsub + 4 * k, ci = row / (kI3H + 2), r = row - ci * (kI3H + 2), ih = h0 - 1 + r;
        v[k] = (row < R && cok && ih >= 0 && ih < H) ? xn[((size_t)ci * H + ih) * W + iw] : 0.f;
    }
    float ve = 0.f;
    const int erow = threadIdx.x >> 1, ec = kI3W + (threadIdx.x & 1);
    if (threadIdx.x < 2 * R) {
        const int ci = erow / (kI3H + 2), r = erow - ci * (kI3H + 2), ih = h0 - 1 + r, iwe = w0 - 1 + ec;
        if (ih >= 0 && ih < H && iwe < W) ve = xn[((size_t)ci * H + ih) * W + iwe];
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int row = rsub + 4 * k, ci = row / (kI3H + 2), r = row - ci * (kI3H + 2);
        if (row < R) xs[ci][r][lane] = v[k];
    }
    if (threadIdx.x < 2 * R) xs[erow / (kI3H + 2)][erow % (kI3H + 2)][ec] = ve;
}

template <int ACT, bool FLIP>
__global__ __launch_bounds__(256) void img3_conv_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                        const float* __restrict__ inv_sigma, float* __restrict__ y, int H, int W,
                                                        int tiles_x, int tiles_y, int accumulate) {
    __shared__ __attribute__((aligned(16))) float xs[3][kI3H + 2][kI3P];
    __shared__ float wl[81], bl[3];
    const float sc = inv_sigma ? *inv_sigma : 1.f;
    if (threadIdx.x < 81) {
        // wl[(ci*9 + tap)*3 + co]
        const int co = threadIdx.x % 3, k = threadIdx.x / 3, ci = k / 9, t = k % 9;
        wl[threadIdx.x] = (FLIP ? w[(ci * 3 + co) * 9 + (8 - t)] : w[(co * 3 + ci) * 9 + t]) * sc;
    }
    if (threadIdx.x < 3) bl[threadIdx.x] = (!FLIP && bias) ? bias[threadIdx.x] : 0.f;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int h0 = ty * kI3H, w0 = tx * kI3W;
    const size_t hw = (size_t)H * W;
    img3_stage(xs, x + (size_t)n * 3 * hw, H, W, h0, w0);
    __syncthreads();
    const int r = threadIdx.x >> 4, c4 = (threadIdx.x & 15) * 4;
    float acc[3][4];
#pragma unroll
    for (int co = 0; co < 3; ++co)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[co][p] = bl[co];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const float4 v4 = *(const float4*)&xs[ci][r + kh][c4];
            const float v[6] = {v4.x, v4.y, v4.z, v4.w, xs[ci][r + kh][c4 + 4], xs[ci][r + kh][c4 + 5]};
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int co = 0; co < 3; ++co) {
                    const float ww = wl[((ci * 9) + kh * 3 + kw) * 3 + co];
#pragma unroll
                    for (int p = 0; p < 4; ++p) fmac1(acc[co][p], v[p + kw], ww);
                }
        }
    const int oh = h0 + r, ow = w0 + c4;
    if (oh >= H || ow >= W) return;
#pragma unroll
    for (int co = 0; co < 3; ++co) {
        float* o = y + ((size_t)n * 3 + co) * hw + (size_t)oh * W + ow;
        float q[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) q[p] = act_apply(acc[co][p], ACT);
        if (ow + 3 < W && (W & 3) == 0 && (((uintptr_t)y) & 15) == 0) {          // 16-byte stores need an aligned base too (advisor, round 3)
            float4 ov = make_float4(q[0], q[1], q[2], q[3]);
            if (accumulate) { const float4 old = *(const float4*)o; ov.x += old.x; ov.y += old.y; ov.z += old.z; ov.w += old.w; }
            *(float4*)o = ov;
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (ow + p < W) o[p] = accumulate ? o[p] + q[p] : q[p];
        }
    }
}

// part: this workgroup's slice of the partial-sum slab (kC3Slab floats: dW[co*27 + ci*9 + tap] for co < 3, dbias at [81..84)), folded in
// workgroup order by thin_fold_kernel; NULL: fp32 atomics.  y / act: optional activation gate of dy (unused by the discriminator).
__global__ __launch_bounds__(256) void img3_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ yact,
                                                         int act, float* __restrict__ dw, float* __restrict__ dbias, float* __restrict__ part,
                                                         int N, int H, int W, int tiles_x, int tiles_y) {
    __shared__ __attribute__((aligned(16))) float xs[3][kI3H + 2][kI3P];
    __shared__ float red[4][84];
    const size_t hw = (size_t)H * W;
    const int r = threadIdx.x >> 4, c4 = (threadIdx.x & 15) * 4;
    float acc[3][27], bs[3];
#pragma unroll
    for (int co = 0; co < 3; ++co) {
        bs[co] = 0.f;
#pragma unroll
        for (int k = 0; k < 27; ++k) acc[co][k] = 0.f;
    }
    const int ntiles = N * tiles_y * tiles_x;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int b = tile;
        const int tx = b % tiles_x; b /= tiles_x;
        const int ty = b % tiles_y;
        const int n = b / tiles_y;
        const int h0 = ty * kI3H, w0 = tx * kI3W;
        __syncthreads();                                   // the previous tile's readers are done
        img3_stage(xs, x + (size_t)n * 3 * hw, H, W, h0, w0);
        const int oh = h0 + r, ow = w0 + c4;
        float g[3][4];
        const bool vec = (W & 3) == 0 && oh < H && ow + 3 < W && ((((uintptr_t)dy) | ((uintptr_t)yact)) & 15) == 0;   // aligned bases (NULL yact: 0)
#pragma unroll
        for (int co = 0; co < 3; ++co) {
            const size_t o = ((size_t)n * 3 + co) * hw + (size_t)oh * W + ow;
            if (vec) {
                const float4 d4 = *(const float4*)(dy + o);
                g[co][0] = d4.x; g[co][1] = d4.y; g[co][2] = d4.z; g[co][3] = d4.w;
                if (yact) {
                    const float4 y4 = *(const float4*)(yact + o);
                    g[co][0] = act_gate(g[co][0], y4.x, act); g[co][1] = act_gate(g[co][1], y4.y, act);
                    g[co][2] = act_gate(g[co][2], y4.z, act); g[co][3] = act_gate(g[co][3], y4.w, act);
                }
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float v = 0.f;
                    if (oh < H && ow + p < W) {
                        v = dy[o + p];
                        if (yact) v = act_gate(v, yact[o + p], act);
                    }
                    g[co][p] = v;
                }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) bs[co] += g[co][p];
        }
        __syncthreads();
#pragma unroll
        for (int ci = 0; ci < 3; ++ci)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const float4 v4 = *(const float4*)&xs[ci][r + kh][c4];
                const float v[6] = {v4.x, v4.y, v4.z, v4.w, xs[ci][r + kh][c4 + 4], xs[ci][r + kh][c4 + 5]};
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int co = 0; co < 3; ++co)
#pragma unroll
                        for (int p = 0; p < 4; ++p) fmac1(acc[co][ci * 9 + kh * 3 + kw], g[co][p], v[p + kw]);
            }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int co = 0; co < 3; ++co) {
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            float v = acc[co][k];
            v = lane_group_sum(v, 64);
            if (lane == 0) red[wave][co * 27 + k] = v;
        }
        float bsum = lane_group_sum(bs[co], 64);
        if (lane == 0) red[wave][81 + co] = bsum;
    }
    __syncthreads();
    if (threadIdx.x < 84) {
        const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        float* mine = part ? part + (size_t)blockIdx.x * kC3Slab : nullptr;
        if (threadIdx.x < 81) { if (mine) mine[threadIdx.x] = v; else atomicAdd(&dw[threadIdx.x], v); }
        else if (dbias) { if (mine) mine[threadIdx.x] = v; else atomicAdd(&dbias[threadIdx.x - 81], v); }
    }
}

// ---------------------------------------------------------------------------------------------------
// conv_last (1x1, Cin -> 3) + tanh.  LP lanes share one pixel (16 B of channels each): a wave load is
// 64/LP full pixel rows; the three dot products are finished with xor-shuffles.
// ---------------------------------------------------------------------------------------------------
// tanh(s) = 1 - 2 / (exp(2 s) + 1) on the hardware exp2 / rcp units: absolute error ~2e-7 (the library tanhf is ~40 VALU
// instructions, which made the forward head VALU-bound); saturates correctly (exp -> inf gives 1, exp -> 0 gives -1)
__device__ __forceinline__ float fast_tanh(float s) {
    const float e = __builtin_amdgcn_exp2f(s * 2.885390081777927f);       // exp(2 s) = 2^(2 s log2 e)
    return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}

// grid: x = pixel groups of one image (strided), y = image.  32-bit index arithmetic throughout (the host checks H*W*ldx < 2^31).
// LPT: lanes per pixel when known at compile time (Cin = 64: 8 for bf16, 16 for fp32), 0 = Cin / E at run time
template <typename T, int LPT>
__global__ __launch_bounds__(256) void conv1x1_tanh_fwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ out,
                                                               int HW, int Cin) {
    constexpr int E = ElemTraits<T>::kPer16B;
    const int LP = LPT ? LPT : Cin / E;
    const int tid = threadIdx.x, cl = tid & (LP - 1), pl = tid / LP;
    float wr[3][E];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int e = 0; e < E; ++e) wr[k][e] = w[k * Cin + cl * E + e];
    const float bk = cl < 3 ? bias[cl] : 0.f;
    const int ppi = 256 / LP;
    const T* xn = x + (size_t)blockIdx.y * HW * ldx;
    float* on = out + (size_t)blockIdx.y * 3 * HW + (cl < 3 ? cl : 0) * HW;
    // two pixel groups per iteration: two independent 16-byte loads in flight per lane
    for (int p0 = blockIdx.x * 2 * ppi; p0 < HW; p0 += gridDim.x * 2 * ppi) {
        const int pa = p0 + pl, pb = pa + ppi;
        // unconditional loads from clamped pixels (a conditional load is a branch with its own wait); the tail's results are dropped
        const uint4 ra = *(const uint4*)(xn + (unsigned)(min(pa, HW - 1) * ldx + cl * E));
        const uint4 rb = *(const uint4*)(xn + (unsigned)(min(pb, HW - 1) * ldx + cl * E));
        float va[E], vb[E], sa[3] = {0.f, 0.f, 0.f}, sb[3] = {0.f, 0.f, 0.f};
        unpack16<T>(ra, va);
        unpack16<T>(rb, vb);
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int e = 0; e < E; ++e) { sa[k] += va[e] * wr[k][e]; sb[k] += vb[e] * wr[k][e]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { sa[k] = lane_group_sum(sa[k], LP); sb[k] = lane_group_sum(sb[k], LP); }
        // every lane of the pixel holds the three sums: lanes 0..2 finish one output channel each
        const float ta = fast_tanh((cl == 0 ? sa[0] : (cl == 1 ? sa[1] : sa[2])) + bk);
        const float tb = fast_tanh((cl == 0 ? sb[0] : (cl == 1 ? sb[1] : sb[2])) + bk);
        if (cl < 3) {
            if (pa < HW) on[pa] = ta;
            if (pb < HW) on[pb] = tb;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void conv1x1_tanh_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                               const T* __restrict__ x, int ldx, const float* __restrict__ w,
                                                               T* __restrict__ dx, int lddx, float* __restrict__ dw, float* __restrict__ dbias,
                                                               float* __restrict__ part, int N, int HW, int Cin, int x_gate_act) {
    constexpr int E = ElemTraits<T>::kPer16B;
    // per-thread partials are combined in a FIXED order: thread t owns (k, channel) and walks the pixel groups 0, 1, 2, ...
    __shared__ float pl[256][3 * E + 1];
    __shared__ float pb[64][3];
    const int LP = Cin / E;
    const int tid = threadIdx.x, cl = tid % LP;
    float wr[3][E], aw[3][E], ab[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int e = 0; e < E; ++e) { wr[k][e] = w[k * Cin + cl * E + e]; aw[k][e] = 0.f; }
    const int ppi = 256 / LP;
    const long long total = (long long)N * HW;
    const long long iters = (total + ppi - 1) / ppi;
    for (long long it = blockIdx.x; it < iters; it += gridDim.x) {
        const long long pix = it * ppi + tid / LP;
        if (pix >= total) continue;
        const unsigned p32 = (unsigned)pix, n = p32 / (unsigned)HW, p = p32 - n * (unsigned)HW;      // N*HW < 2^31 (host check)
        const size_t o = (size_t)n * 3 * HW + p;
        float g[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float t = out[o + (size_t)k * HW];
            g[k] = dout[o + (size_t)k * HW] * (1.f - t * t);    // d tanh
        }
        float v[E], d[E];
        unpack16<T>(*(const uint4*)(x + (size_t)pix * ldx + cl * E), v);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            d[e] = act_gate(g[0] * wr[0][e] + g[1] * wr[1][e] + g[2] * wr[2][e], v[e], x_gate_act);
            aw[0][e] += g[0] * v[e]; aw[1][e] += g[1] * v[e]; aw[2][e] += g[2] * v[e];
        }
        if (cl == 0) { ab[0] += g[0]; ab[1] += g[1]; ab[2] += g[2]; }
        *(uint4*)(dx + (size_t)pix * lddx + cl * E) = pack16<T>(d);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int e = 0; e < E; ++e) pl[tid][k * E + e] = aw[k][e];
        if (cl == 0) pb[tid / LP][k] = ab[k];
    }
    __syncthreads();
    if (tid < 3 * Cin) {             // Cin == 64: 192 threads
        const int k = tid / Cin, c = tid - k * Cin, ccl = c / E, e = c - ccl * E;
        float sacc = 0.f;
        for (int pg = 0; pg < ppi; ++pg) sacc += pl[pg * LP + ccl][k * E + e];
        // part != NULL: this workgroup's slice of the partial-sum slab (summed in fixed order by thin_fold_kernel)
        if (part) part[(size_t)blockIdx.x * kC1Slab + tid] = sacc; else atomicAdd(&dw[tid], sacc);
    }
    if (tid < 3) {
        float sacc = 0.f;
        for (int pg = 0; pg < ppi; ++pg) sacc += pb[pg][tid];
        if (part) part[(size_t)blockIdx.x * kC1Slab + 3 * Cin + tid] = sacc; else atomicAdd(&dbias[tid], sacc);
    }
}

inline int grid_cap(long long work_items, int per_block, int cap) {
    long long g = (work_items + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

// the matrix-core form of the 3 -> 64 forward applies (only it can write gate bits)
static bool c3_fwd_mfma_ok(int N, int H, int W, int Cout, int stride, const float* bias, int dtype) {
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    return dtype == WU_BF16 && Cout == 64 && g_wu_opt[WU_OPT_C3_ROWS] == 0 && (!bias || ((uintptr_t)bias % 16) == 0) &&
           (long long)N * Ho * Wo < (1ll << 31) - 64 && (long long)N * 3 * H * W < (1ll << 28);
}

static int c3_fwd_impl(const float* x_nchw, const float* w_oihw, const float* bias, const float* inv_sigma,
                       void* y, int ldy, int out_nchw, int N, int H, int W, int Cout, int stride, int act,
                       int dtype, void* stream, void* gate_bits_out) {
    WU_REQUIRE(stride == 1 || stride == 2, "conv3x3_c3_fwd: stride");
    WU_REQUIRE((Cout == 64 && !out_nchw) || (Cout == 3 && out_nchw && stride == 1), "conv3x3_c3_fwd: supported (Cout,layout): (64,NHWC) or (3,NCHW s1); got Cout=%d", Cout);
    const int esz = dtype == WU_BF16 ? 2 : 4;
    if (!out_nchw) WU_REQUIRE(((uintptr_t)y % 16) == 0 && (ldy * esz) % 16 == 0 && ldy >= Cout, "conv3x3_c3_fwd: alignment");
    hipStream_t s = (hipStream_t)stream;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const int grid = grid_cap((long long)N * Ho * Wo, 256, 256 * 32);
#define C3_LAUNCH_A(T, CO, ST, NCHW, A) hipLaunchKernelGGL((conv3x3_c3_fwd_kernel<T, CO, ST, NCHW, A>), dim3(grid), dim3(256), 0, s, x_nchw, w_oihw, bias, inv_sigma, y, ldy, N, H, W)
#define C3_LAUNCH(T, CO, ST, NCHW) do { if (act == WU_ACT_RELU) C3_LAUNCH_A(T, CO, ST, NCHW, WU_ACT_RELU); else if (act == WU_ACT_LEAKY) C3_LAUNCH_A(T, CO, ST, NCHW, WU_ACT_LEAKY); else C3_LAUNCH_A(T, CO, ST, NCHW, WU_ACT_NONE); } while (0)
#define C3_LANES_A(T, ST, A) hipLaunchKernelGGL((conv3x3_c3_fwd_rows_kernel<T, ST, A>), dim3(grid_l), dim3(256), 0, s, x_nchw, w_oihw, bias, inv_sigma, (T*)y, ldy, N, H, W)
#define C3_LANES(T, ST) do { if (act == WU_ACT_RELU) C3_LANES_A(T, ST, WU_ACT_RELU); else if (act == WU_ACT_LEAKY) C3_LANES_A(T, ST, WU_ACT_LEAKY); else C3_LANES_A(T, ST, WU_ACT_NONE); } while (0)
    const int grid_l = grid_cap((long long)N * Ho * Wo, 256, 256 * 8);
    // fp32 path: the LDS-broadcast one-thread-per-pixel kernel (239 us at B=32 256x256 in bf16 storage; the scalar-weight /
    // transposed-store variant conv3x3_c3_fwd_rows_kernel measured 326 us and is kept selectable for A/B work)
#define C3_MFMA_F(ST, A, F, B) hipLaunchKernelGGL((conv3x3_c3_fwd_mfma_kernel<ST, A, F, B>), dim3(grid_m), dim3(256), 0, s, x_nchw, w_oihw, bias, inv_sigma, (bf16_t*)y, ldy, N, H, W, (unsigned char*)gate_bits_out)
#define C3_MFMA_A(ST, A) do { const bool full_ = ((long long)N * Ho * Wo) % 32 == 0;                                              \
        if (gate_bits_out && A == WU_ACT_RELU) { if (full_) C3_MFMA_F(ST, WU_ACT_RELU, true, true); else C3_MFMA_F(ST, WU_ACT_RELU, false, true); } \
        else { if (full_) C3_MFMA_F(ST, A, true, false); else C3_MFMA_F(ST, A, false, false); } } while (0)
#define C3_MFMA(ST) do { if (act == WU_ACT_RELU) C3_MFMA_A(ST, WU_ACT_RELU); else if (act == WU_ACT_LEAKY) C3_MFMA_A(ST, WU_ACT_LEAKY); else C3_MFMA_A(ST, WU_ACT_NONE); } while (0)
    const int grid_m = grid_cap((long long)N * Ho * Wo, 128, 256 * 16);
    // option 5: 0 = bf16 on the matrix cores (default), 1 = scalar-weight rows kernel, 2 = one-thread-per-pixel VALU kernel
    if (out_nchw && g_wu_opt[WU_OPT_IMG3_TILED] && (long long)N * cdiv(H, kI3H) * cdiv(W, kI3W) < (1ll << 31)) {          // the LDS-tiled image-layout 3 -> 3 conv
        const int tx_ = cdiv(W, kI3W), ty_ = cdiv(H, kI3H);
#define I3F(A) hipLaunchKernelGGL((img3_conv_kernel<A, false>), dim3((unsigned)(N * tx_ * ty_)), dim3(256), 0, s, x_nchw, w_oihw, bias, inv_sigma, (float*)y, H, W, tx_, ty_, 0)
        if (act == WU_ACT_RELU) I3F(WU_ACT_RELU); else if (act == WU_ACT_LEAKY) I3F(WU_ACT_LEAKY); else I3F(WU_ACT_NONE);
#undef I3F
    } else if (out_nchw) C3_LAUNCH(float, 3, 1, true);
    else if (c3_fwd_mfma_ok(N, H, W, Cout, stride, bias, dtype)) { if (stride == 1) C3_MFMA(1); else C3_MFMA(2); }
    else if (g_wu_opt[WU_OPT_C3_ROWS] == 1) {
        if (dtype == WU_BF16) { if (stride == 1) C3_LANES(bf16_t, 1); else C3_LANES(bf16_t, 2); }
        else { if (stride == 1) C3_LANES(float, 1); else C3_LANES(float, 2); }
    } else if (dtype == WU_BF16) { if (stride == 1) C3_LAUNCH(bf16_t, 64, 1, false); else C3_LAUNCH(bf16_t, 64, 2, false); }
    else { if (stride == 1) C3_LAUNCH(float, 64, 1, false); else C3_LAUNCH(float, 64, 2, false); }
#undef C3_MFMA
#undef C3_MFMA_A
#undef C3_MFMA_F
#undef C3_LANES
#undef C3_LANES_A
#undef C3_LAUNCH_A
#undef C3_LAUNCH
    WU_LAUNCH_CHECK("conv3x3_c3_fwd");
    return 0;
}

extern "C" int wu_conv3x3_c3_fwd(const float* x_nchw, const float* w_oihw, const float* bias, const float* inv_sigma,
                                 void* y, int ldy, int out_nchw, int N, int H, int W, int Cout, int stride, int act,
                                 int dtype, void* stream) {
    return c3_fwd_impl(x_nchw, w_oihw, bias, inv_sigma, y, ldy, out_nchw, N, H, W, Cout, stride, act, dtype, stream, nullptr);
}

// wu_conv3x3_c3_fwd (NHWC output, ReLU) that also writes the gate bits of its output (wu_kernels.h, "gate bits")
extern "C" int wu_conv3x3_c3_gate_bits_supported(int N, int H, int W, int Cout, int stride, const float* bias, int dtype) {
    return c3_fwd_mfma_ok(N, H, W, Cout, stride, bias, dtype) ? 1 : 0;
}
extern "C" int wu_conv3x3_c3_fwd_bits(const float* x_nchw, const float* w_oihw, const float* bias, const float* inv_sigma,
                                      void* y, int ldy, void* gate_bits_out, int N, int H, int W, int Cout, int stride, int dtype, void* stream) {
    WU_REQUIRE(gate_bits_out && c3_fwd_mfma_ok(N, H, W, Cout, stride, bias, dtype), "conv3x3_c3_fwd_bits: outside the matrix-core path (ask wu_conv3x3_c3_gate_bits_supported)");
    return c3_fwd_impl(x_nchw, w_oihw, bias, inv_sigma, y, ldy, 0, N, H, W, Cout, stride, WU_ACT_RELU, dtype, stream, gate_bits_out);
}

extern "C" size_t wu_thin_workspace_bytes(void) { return (size_t)kThinMaxBlocks * kC3Slab * sizeof(float); }

extern "C" int wu_conv3x3_c3_wgrad(const float* x_nchw, const void* dy, int lddy, int dy_nchw, const void* y, int ldy_,
                                   int act, float* dw_oihw, float* dbias, void* workspace, size_t workspace_bytes,
                                   int N, int H, int W, int Cout, int stride, int accumulate, int dtype, void* stream) {
    WU_REQUIRE(stride == 1 || stride == 2, "conv3x3_c3_wgrad: stride");
    WU_REQUIRE(Cout > 0 && dw_oihw, "conv3x3_c3_wgrad: bad args");
    hipStream_t s = (hipStream_t)stream;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const bool mfma_path = !dy_nchw && dtype == WU_BF16 && Cout == 64 && (long long)N * Ho * Wo < (1ll << 31) && (long long)N * 3 * H * W < (1ll << 28) && ((uintptr_t)dy % 16) == 0 && (lddy % 8) == 0 &&
                           (!y || (((uintptr_t)y % 16) == 0 && (ldy_ % 8) == 0));
    // deterministic mode (every variant with Cout <= 64): a caller-owned slab receives the per-workgroup partials, thin_fold_kernel sums
    // them in workgroup order; without the slab the kernels fall back to fp32 atomics
    float* part = (Cout <= 64 && workspace && workspace_bytes >= wu_thin_workspace_bytes() && ((uintptr_t)workspace % 16) == 0) ? (float*)workspace : nullptr;
    if (!accumulate && !part) {
        hipMemsetAsync(dw_oihw, 0, (size_t)Cout * 27 * sizeof(float), s);
        if (dbias) hipMemsetAsync(dbias, 0, (size_t)Cout * sizeof(float), s);
    }
    dim3 grid(grid_cap((long long)N * Ho * Wo, 64, 1024), cdiv(Cout, 64));
#define C3W(T, ST, NCHW) hipLaunchKernelGGL((conv3x3_c3_wgrad_kernel<T, ST, NCHW>), grid, dim3(256), 0, s, x_nchw, dy, lddy, y, ldy_, act, dw_oihw, dbias, part, N, H, W, Cout)
    if (mfma_path) {
        const long long ntiles = ((long long)N * Ho * Wo + 255) / 256;
        const int g = (int)(ntiles < kThinMaxBlocks ? ntiles : kThinMaxBlocks);
#define C3WM(ST, G) hipLaunchKernelGGL((conv3x3_c3_wgrad_mfma_kernel<ST, G>), dim3(g), dim3(256), 0, s, x_nchw, (const bf16_t*)dy, lddy, (const bf16_t*)y, ldy_, act, dw_oihw, dbias, part, N, H, W)
        if (stride == 1) { if (y) C3WM(1, true); else C3WM(1, false); }
        else { if (y) C3WM(2, true); else C3WM(2, false); }
#undef C3WM
        if (part) hipLaunchKernelGGL(thin_fold_kernel, dim3(cdiv(kC3Slab, 8)), dim3(256), 0, s, part, g, kC3Slab, dw_oihw, 1728, dbias, 64, accumulate);
        WU_LAUNCH_CHECK("conv3x3_c3_wgrad_mfma");
        return 0;
    }
    if (dy_nchw && Cout == 3 && stride == 1 && g_wu_opt[WU_OPT_IMG3_TILED] && (long long)N * cdiv(H, kI3H) * cdiv(W, kI3W) < (1ll << 31)) {
        const int tx_ = cdiv(W, kI3W), ty_ = cdiv(H, kI3H);
        const int g = grid_cap((long long)N * tx_ * ty_, 1, kThinMaxBlocks);
        hipLaunchKernelGGL(img3_wgrad_kernel, dim3(g), dim3(256), 0, s, x_nchw, (const float*)dy, (const float*)y, act, dw_oihw, dbias, part, N, H, W, tx_, ty_);
        if (part) hipLaunchKernelGGL(thin_fold_kernel, dim3(cdiv(84, 8)), dim3(256), 0, s, part, g, kC3Slab, dw_oihw, 81, dbias, dbias ? 3 : 0, accumulate);
        WU_LAUNCH_CHECK("img3_wgrad");
        return 0;
    }
    if (dy_nchw && Cout <= 3) {
        const int g = grid_cap((long long)N * Ho * Wo, 256, 1024);
        if (stride == 1) hipLaunchKernelGGL(conv3x3_c3_wgrad_small_kernel<1>, dim3(g), dim3(256), 0, s, x_nchw, (const float*)dy, (const float*)y, act, dw_oihw, dbias, part, N, H, W, Cout);
        else hipLaunchKernelGGL(conv3x3_c3_wgrad_small_kernel<2>, dim3(g), dim3(256), 0, s, x_nchw, (const float*)dy, (const float*)y, act, dw_oihw, dbias, part, N, H, W, Cout);
        if (part) hipLaunchKernelGGL(thin_fold_kernel, dim3(cdiv(Cout * 28, 8)), dim3(256), 0, s, part, g, kC3Slab, dw_oihw, Cout * 27, dbias, dbias ? Cout : 0, accumulate);
        WU_LAUNCH_CHECK("conv3x3_c3_wgrad_small");
        return 0;
    }
    if (dy_nchw) { if (stride == 1) C3W(float, 1, true); else C3W(float, 2, true); }
    else if (dtype == WU_BF16) { if (stride == 1) C3W(bf16_t, 1, false); else C3W(bf16_t, 2, false); }
    else { if (stride == 1) C3W(float, 1, false); else C3W(float, 2, false); }
#undef C3W
    if (part) hipLaunchKernelGGL(thin_fold_kernel, dim3(cdiv(Cout * 28, 8)), dim3(256), 0, s, part, (int)grid.x, kC3Slab, dw_oihw, Cout * 27, dbias, dbias ? Cout : 0, accumulate);
    WU_LAUNCH_CHECK("conv3x3_c3_wgrad");
    return 0;
}

extern "C" int wu_conv3x3_c3_dgrad(const void* dy, int lddy, int dy_nchw, const void* y, int ldy_, int act,
                                   const float* w_oihw, const float* inv_sigma, float* dx_nchw, int N, int H, int W,
                                   int Cout, int stride, int accumulate, int dtype, void* stream) {
    WU_REQUIRE(stride == 1 || stride == 2, "conv3x3_c3_dgrad: stride");
    WU_REQUIRE(Cout > 0 && Cout <= 64, "conv3x3_c3_dgrad: Cout");
    hipStream_t s = (hipStream_t)stream;
    const int grid = grid_cap((long long)N * H * W, 256, 256 * 32);
    const size_t lds = (size_t)27 * Cout * sizeof(float);
    const int esz_ = dtype == WU_BF16 ? 2 : 4;
    const int lp_ = dy_nchw ? 0 : Cout / (16 / esz_);
    if (!dy_nchw && dtype == WU_BF16 && stride == 2 && Cout == 64 && ((uintptr_t)dy % 16) == 0 && (lddy % 8) == 0 &&
        (!y || (((uintptr_t)y % 16) == 0 && (ldy_ % 8) == 0))) {
        const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
        const int tiles_x = cdiv(Wo, 32), tiles_y = cdiv(Ho, 8);
        const long long nwg = (long long)N * tiles_x * tiles_y;
        if (nwg < (1ll << 31)) {
#define C3DM(G) hipLaunchKernelGGL((conv3x3_c3_dgrad_s2_mfma_kernel<G>), dim3((unsigned)nwg), dim3(256), 0, s, (const bf16_t*)dy, lddy, (const bf16_t*)y, ldy_, act, w_oihw, inv_sigma, dx_nchw, H, W, tiles_x, tiles_y, accumulate)
            if (y) C3DM(true); else C3DM(false);
#undef C3DM
            WU_LAUNCH_CHECK("conv3x3_c3_dgrad_s2_mfma");
            return 0;
        }
    }
    if (!dy_nchw && (long long)N * H * W < (1ll << 32) && Cout % (16 / esz_) == 0 && lp_ >= 1 && lp_ <= 64 && (lp_ & (lp_ - 1)) == 0 && ((uintptr_t)dy % 16) == 0 && (lddy * esz_) % 16 == 0 &&
        (!y || (((uintptr_t)y % 16) == 0 && (ldy_ * esz_) % 16 == 0))) {
        const int g = grid_cap((long long)N * H * W, 256 / lp_, 256 * 16);
#define C3DN(T, ST) hipLaunchKernelGGL((conv3x3_c3_dgrad_nhwc_kernel<T, ST>), dim3(g), dim3(256), lds, s, (const T*)dy, lddy, (const T*)y, ldy_, act, w_oihw, inv_sigma, dx_nchw, N, H, W, Cout, accumulate)
        if (dtype == WU_BF16) { if (stride == 1) C3DN(bf16_t, 1); else C3DN(bf16_t, 2); }
        else { if (stride == 1) C3DN(float, 1); else C3DN(float, 2); }
#undef C3DN
        WU_LAUNCH_CHECK("conv3x3_c3_dgrad_nhwc");
        return 0;
    }
    if (dy_nchw && Cout == 3 && stride == 1 && !y && g_wu_opt[WU_OPT_IMG3_TILED] && (long long)N * cdiv(H, kI3H) * cdiv(W, kI3W) < (1ll << 31)) {
        const int tx_ = cdiv(W, kI3W), ty_ = cdiv(H, kI3H);       // = the forward conv of dy with transposed, flipped weights
        hipLaunchKernelGGL((img3_conv_kernel<WU_ACT_NONE, true>), dim3((unsigned)(N * tx_ * ty_)), dim3(256), 0, s, (const float*)dy, w_oihw, (const float*)nullptr,
                           inv_sigma, dx_nchw, H, W, tx_, ty_, accumulate);
        WU_LAUNCH_CHECK("img3_dgrad");
        return 0;
    }
#define C3D(T, ST, NCHW) hipLaunchKernelGGL((conv3x3_c3_dgrad_kernel<T, ST, NCHW>), dim3(grid), dim3(256), lds, s, dy, lddy, y, ldy_, act, w_oihw, inv_sigma, dx_nchw, N, H, W, Cout, accumulate)
    if (dy_nchw) { if (stride == 1) C3D(float, 1, true); else C3D(float, 2, true); }
    else if (dtype == WU_BF16) { if (stride == 1) C3D(bf16_t, 1, false); else C3D(bf16_t, 2, false); }
    else { if (stride == 1) C3D(float, 1, false); else C3D(float, 2, false); }
#undef C3D
    WU_LAUNCH_CHECK("conv3x3_c3_dgrad");
    return 0;
}

extern "C" int wu_conv1x1_tanh_fwd(const void* x, int ldx, const float* w, const float* bias, float* out_nchw,
                                   int N, int H, int W, int Cin, int dtype, void* stream) {
    const int esz = dtype == WU_BF16 ? 2 : 4;
    const int E = 16 / esz;
    const int LP = Cin / E;
    // lanes cl = 0..2 of a pixel group each write one of the 3 output channels: a group needs at least 4 lanes (LP is a power of two)
    WU_REQUIRE(Cin % E == 0 && LP >= 4 && LP <= 64 && (LP & (LP - 1)) == 0, "conv1x1_tanh_fwd: Cin=%d unsupported (Cin/%d must be a power of two in [4, 64])", Cin, E);
    WU_REQUIRE(((uintptr_t)x % 16) == 0 && (ldx * esz) % 16 == 0 && bias, "conv1x1_tanh_fwd: alignment/bias");
    WU_REQUIRE((long long)N * H * W < (1ll << 31), "conv1x1_tanh_fwd: N*H*W must stay below 2^31 (32-bit pixel arithmetic)");
    WU_REQUIRE((long long)H * W * ldx < (1ll << 31) && N <= 65535, "conv1x1_tanh_fwd: H*W*ldx must stay below 2^31, N below 65536");
    // x = pixel groups of one image (two groups of 256/LP pixels per workgroup iteration), y = image; ~16 workgroups per CU in total
    const int per_img = grid_cap((long long)H * W, 2 * (256 / LP), (256 * 16 + N - 1) / N);
    const dim3 grid(per_img, N);
#define C1F(T, L) hipLaunchKernelGGL((conv1x1_tanh_fwd_kernel<T, L>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, ldx, w, bias, out_nchw, H * W, Cin)
    if (dtype == WU_BF16) { if (LP == 8) C1F(bf16_t, 8); else C1F(bf16_t, 0); }
    else { if (LP == 16) C1F(float, 16); else C1F(float, 0); }
#undef C1F
    WU_LAUNCH_CHECK("conv1x1_tanh_fwd");
    return 0;
}

extern "C" int wu_conv1x1_tanh_bwd(const float* dout_nchw, const float* out_nchw, const void* x, int ldx, const float* w,
                                   void* dx, int lddx, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                                   int N, int H, int W, int Cin, int accumulate, int x_gate_act, int dtype, void* stream) {
    const int esz = dtype == WU_BF16 ? 2 : 4;
    const int E = 16 / esz;
    const int LP = Cin / E;
    WU_REQUIRE(Cin == 64, "conv1x1_tanh_bwd: Cin must be 64 (got %d)", Cin);
    WU_REQUIRE((long long)N * H * W < (1ll << 31), "conv1x1_tanh_bwd: N*H*W must stay below 2^31 (32-bit pixel arithmetic)");
    WU_REQUIRE(((uintptr_t)x % 16) == 0 && (ldx * esz) % 16 == 0 && ((uintptr_t)dx % 16) == 0 && (lddx * esz) % 16 == 0, "conv1x1_tanh_bwd: alignment");
    hipStream_t s = (hipStream_t)stream;
    // deterministic mode (caller-owned slab for the per-workgroup partials); otherwise fp32 atomics across workgroups
    float* part = (workspace && workspace_bytes >= (size_t)kThinMaxBlocks * kC1Slab * sizeof(float) && ((uintptr_t)workspace % 16) == 0) ? (float*)workspace : nullptr;
    if (!accumulate && !part) {
        hipMemsetAsync(dw, 0, (size_t)3 * Cin * sizeof(float), s);
        hipMemsetAsync(dbias, 0, 3 * sizeof(float), s);
    }
    const int grid = grid_cap((long long)N * H * W, 256 / LP, kThinMaxBlocks);
    if (dtype == WU_BF16) hipLaunchKernelGGL(conv1x1_tanh_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, dout_nchw, out_nchw, (const bf16_t*)x, ldx, w, (bf16_t*)dx, lddx, dw, dbias, part, N, H * W, Cin, x_gate_act);
    else hipLaunchKernelGGL(conv1x1_tanh_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, dout_nchw, out_nchw, (const float*)x, ldx, w, (float*)dx, lddx, dw, dbias, part, N, H * W, Cin, x_gate_act);
    if (part) hipLaunchKernelGGL(thin_fold_kernel, dim3(cdiv(kC1Slab, 8)), dim3(256), 0, s, part, grid, kC1Slab, dw, 3 * Cin, dbias, 3, accumulate);
    WU_LAUNCH_CHECK("conv1x1_tanh_bwd");
    return 0;
}
