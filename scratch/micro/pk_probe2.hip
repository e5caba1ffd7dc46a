// Minimal reproducer attempt for the cross-stream hazard: register-only (no LDS, no memory traffic in the loop), built
// around the INSTRUCTION SHAPE the compiler emitted in the failing kernel (scratch/micro/img3_variants.hip, packed build):
//
//     v_mov_b32    v114, v29                                   ; ONE half of a 64-bit pair is (re)written ...
//     v_mov_b32    v26,  v7
//     v_fmac_f32   v70,  v75, v7
//     v_pk_fma_f32 v[56:57], v[114:115], v[26:27], v[56:57] op_sel_hi:[1,0,1]   ; ... and the pair is read 1-3 instructions later,
//                                                                               ;     the weight broadcast from the LOW half
//
// round 3's register-only probe (pk_probe.hip) fed v_pk_fma_f32 with whole pairs written by earlier v_pk_fma_f32 and was exact beside
// every kernel.  Variants (one asm block each, fixed registers so the halves can be named):
//   0: the shape above: v_mov lo(x), v_mov lo(w), one independent v_fmac, then the packed FMA with op_sel_hi:[1,0,1]
//   1: the same with "s_nop 3" (4 wait states) between the last v_mov and the packed FMA
//   2: both halves of both pairs written by v_mov FAR ahead (16 wait states), then the packed FMA        (no fresh partial write)
//   3: shape 0 but the packed FMA has no op_sel broadcast (w pair = {w, w} as data)
//   4: shape 0 with the arithmetic as two v_fmac_f32 (control: same data flow, no packed op)
//   5: LDS-fed whole pair: {t, u} and w go through LDS (ds_write / ds_read_b64 straight into v[100:101]), lgkmcnt(0), packed FMA
//   6: LDS-fed HALF pair: the high half by v_mov far ahead, the low halves by ds_read_b32 into v100 / v102, lgkmcnt(0), packed FMA
// Every lane iterates t <- 0.75 t + 0.125 (so the value moved into the low half CHANGES every iteration: a stale read is visible) and
// accumulates acc += {t, u} * w.  Deterministic: any difference between two launches is a hardware effect.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC scratch/micro/pk_probe2.hip -o scratch/micro/libpk_probe2.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float float2_t __attribute__((ext_vector_type(2)));

template <int V>
__global__ __launch_bounds__(256) void pk_probe2_kernel(float* __restrict__ out, const float* __restrict__ in, int iters) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    float t = in[i], u = in[i + 1], w = in[i + 2] * 1e-3f, side = in[i + 3];
    float2_t acc = {0.f, 0.f};
    __shared__ __attribute__((aligned(16))) float lbuf[256 * 4];
    const unsigned lds_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)(lbuf + 4 * threadIdx.x);
    for (int it = 0; it < iters; ++it) {
        t = __builtin_fmaf(t, 0.75f, 0.125f);
        u = __builtin_fmaf(u, -0.5f, 0.25f);
        // v[100:101] = x pair {t, u}, v[102:103] = w pair {w, (junk or w)}
        if (V == 0)
            asm volatile("v_mov_b32 v101, %[u]\n v_mov_b32 v103, %[u]\n s_nop 7\n s_nop 7\n"
                         "v_mov_b32 v100, %[t]\n v_mov_b32 v102, %[w]\n v_fmac_f32 %[s], %[t], %[w]\n"
                         "v_pk_fma_f32 %[acc], v[100:101], v[102:103], %[acc] op_sel_hi:[1,0,1]"
                         : [acc] "+v"(acc), [s] "+v"(side) : [t] "v"(t), [u] "v"(u), [w] "v"(w) : "v100", "v101", "v102", "v103");
        else if (V == 1)
            asm volatile("v_mov_b32 v101, %[u]\n v_mov_b32 v103, %[u]\n s_nop 7\n s_nop 7\n"
                         "v_mov_b32 v100, %[t]\n v_mov_b32 v102, %[w]\n v_fmac_f32 %[s], %[t], %[w]\n s_nop 3\n"
                         "v_pk_fma_f32 %[acc], v[100:101], v[102:103], %[acc] op_sel_hi:[1,0,1]"
                         : [acc] "+v"(acc), [s] "+v"(side) : [t] "v"(t), [u] "v"(u), [w] "v"(w) : "v100", "v101", "v102", "v103");
        else if (V == 2)
            asm volatile("v_mov_b32 v101, %[u]\n v_mov_b32 v103, %[u]\n v_mov_b32 v100, %[t]\n v_mov_b32 v102, %[w]\n s_nop 7\n s_nop 7\n"
                         "v_fmac_f32 %[s], %[t], %[w]\n"
                         "v_pk_fma_f32 %[acc], v[100:101], v[102:103], %[acc] op_sel_hi:[1,0,1]"
                         : [acc] "+v"(acc), [s] "+v"(side) : [t] "v"(t), [u] "v"(u), [w] "v"(w) : "v100", "v101", "v102", "v103");
        else if (V == 3)
            asm volatile("v_mov_b32 v101, %[u]\n v_mov_b32 v103, %[w]\n s_nop 7\n s_nop 7\n"
                         "v_mov_b32 v100, %[t]\n v_mov_b32 v102, %[w]\n v_fmac_f32 %[s], %[t], %[w]\n"
                         "v_pk_fma_f32 %[acc], v[100:101], v[102:103], %[acc]"
                         : [acc] "+v"(acc), [s] "+v"(side) : [t] "v"(t), [u] "v"(u), [w] "v"(w) : "v100", "v101", "v102", "v103");
        else if (V == 5)
            asm volatile("ds_write_b32 %[la], %[t]\n ds_write_b32 %[la], %[u] offset:4\n ds_write_b32 %[la], %[w] offset:8\n s_waitcnt lgkmcnt(0)\n"
                         "v_mov_b32 v103, %[u]\n ds_read_b64 v[100:101], %[la]\n ds_read_b32 v102, %[la] offset:8\n v_fmac_f32 %[s], %[t], %[w]\n s_waitcnt lgkmcnt(0)\n"
                         "v_pk_fma_f32 %[acc], v[100:101], v[102:103], %[acc] op_sel_hi:[1,0,1]"
                         : [acc] "+v"(acc), [s] "+v"(side) : [t] "v"(t), [u] "v"(u), [w] "v"(w), [la] "v"(lds_addr) : "v100", "v101", "v102", "v103", "memory");
        else if (V == 6)
            asm volatile("ds_write_b32 %[la], %[t]\n ds_write_b32 %[la], %[w] offset:8\n s_waitcnt lgkmcnt(0)\n"
                         "v_mov_b32 v101, %[u]\n v_mov_b32 v103, %[u]\n s_nop 7\n ds_read_b32 v100, %[la]\n ds_read_b32 v102, %[la] offset:8\n v_fmac_f32 %[s], %[t], %[w]\n s_waitcnt lgkmcnt(0)\n"
                         "v_pk_fma_f32 %[acc], v[100:101], v[102:103], %[acc] op_sel_hi:[1,0,1]"
                         : [acc] "+v"(acc), [s] "+v"(side) : [t] "v"(t), [u] "v"(u), [w] "v"(w), [la] "v"(lds_addr) : "v100", "v101", "v102", "v103", "memory");
        else {
            float a0 = acc.x, a1 = acc.y;
            asm volatile("v_mov_b32 v101, %[u]\n v_mov_b32 v103, %[u]\n s_nop 7\n s_nop 7\n"
                         "v_mov_b32 v100, %[t]\n v_mov_b32 v102, %[w]\n v_fmac_f32 %[s], %[t], %[w]\n"
                         "v_fmac_f32 %[a0], v100, v102\n v_fmac_f32 %[a1], v101, v102"
                         : [a0] "+v"(a0), [a1] "+v"(a1), [s] "+v"(side) : [t] "v"(t), [u] "v"(u), [w] "v"(w) : "v100", "v101", "v102", "v103");
            acc.x = a0; acc.y = a1;
        }
    }
    out[i] = acc.x; out[i + 1] = acc.y; out[i + 2] = side; out[i + 3] = t + u;
}

extern "C" int pk_probe2(float* out, const float* in, long long n_floats, int iters, int variant, void* stream) {
    const dim3 grid((unsigned)(n_floats / (256 * 4))), blk(256);
    hipStream_t s = (hipStream_t)stream;
    switch (variant) {
        case 0: hipLaunchKernelGGL(pk_probe2_kernel<0>, grid, blk, 0, s, out, in, iters); break;
        case 1: hipLaunchKernelGGL(pk_probe2_kernel<1>, grid, blk, 0, s, out, in, iters); break;
        case 2: hipLaunchKernelGGL(pk_probe2_kernel<2>, grid, blk, 0, s, out, in, iters); break;
        case 3: hipLaunchKernelGGL(pk_probe2_kernel<3>, grid, blk, 0, s, out, in, iters); break;
        case 4: hipLaunchKernelGGL(pk_probe2_kernel<4>, grid, blk, 0, s, out, in, iters); break;
        case 5: hipLaunchKernelGGL(pk_probe2_kernel<5>, grid, blk, 0, s, out, in, iters); break;
        case 6: hipLaunchKernelGGL(pk_probe2_kernel<6>, grid, blk, 0, s, out, in, iters); break;
        default: return -2;
    }
    return (int)hipGetLastError();
}
