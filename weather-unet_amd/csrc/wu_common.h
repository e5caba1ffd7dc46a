// Shared device/host helpers for libwu_kernels.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/wu_kernels.h"

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));

// ---- error plumbing ---------------------------------------------------------------------------
extern thread_local char g_wu_err[256];
#define WU_FAIL(code, ...)                                   \
    do {                                                     \
        snprintf(g_wu_err, sizeof(g_wu_err), __VA_ARGS__);   \
        return (code);                                       \
    } while (0)
#define WU_REQUIRE(cond, ...) \
    do {                      \
        if (!(cond)) WU_FAIL(-1, __VA_ARGS__); \
    } while (0)
#define WU_LAUNCH_CHECK(name)                                                        \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) WU_FAIL((int)e_, "%s: %s", name, hipGetErrorString(e_)); \
    } while (0)

// ---- tuning switches (wu_prof.hip) ----
extern int g_wu_opt[16];
extern void* g_wu_dbg_ptr;
#define WU_OPT_CONV_V2 0
#define WU_OPT_CONV_PERSISTENT 1
#define WU_OPT_WGRAD_V2 2
#define WU_OPT_CONV_SMALL 3
#define WU_OPT_WGRAD_DMA_INTERLEAVE 4
#define WU_OPT_C3_ROWS 5
#define WU_OPT_CONV_PRIO 6
#define WU_OPT_CONV_STRIDED 7
#define WU_OPT_ADAIN_BWD_MARCH 8
#define WU_OPT_ADAIN_FWD_MARCH 9
#define WU_OPT_GRID_CUS 10
#define WU_OPT_CONV_W_RESIDENT 11
#define WU_OPT_PW_TILE 12
#define WU_OPT_IMG3_TILED 13
#define WU_OPT_S2_DGRAD_PARITY 14
#define WU_OPT_PW3 15

int wu_num_cus();   // compute units of the current device (wu_prof.hip), cached

// ---- profiling hooks (wu_prof.hip) --------------------------------------------------------------
void wu_prof_pre(int family, hipStream_t s);
void wu_prof_post(int family, hipStream_t s, double flops, double bytes);

// ---- bf16 <-> f32 -------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    // plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 (round-to-nearest-even, NaN preserved)
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    // ONE v_cvt_pk_bf16_f32 for the pair (two scalar casts + or compile to two single-operand cvt_pk and a v_perm)
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
    const bf16x2v r = __builtin_convertvector(f32x2v{lo, hi}, bf16x2v);
    return __builtin_bit_cast(uint32_t, r);
}

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
    static constexpr int kPer16B = 4;
    __device__ static __forceinline__ float load(const float* p) { return *p; }
    __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct ElemTraits<bf16_t> {
    static constexpr int kPer16B = 8;
    __device__ static __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// unpack a 16-byte chunk into floats (4 for f32, 8 for bf16) and back
template <typename T> __device__ __forceinline__ void unpack16(const uint4& v, float* f);
template <> __device__ __forceinline__ void unpack16<float>(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ uint4 pack16(const float* f);
template <> __device__ __forceinline__ uint4 pack16<float>(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}
template <> __device__ __forceinline__ uint4 pack16<bf16_t>(const float* f) {
    return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
}

// activation-derivative gate: g * act'(y)   (ReLU: nets.py:21,23; LeakyReLU(0.2): nets.py:32)
__device__ __forceinline__ float act_gate(float g, float y, int act) {
    if (act == WU_ACT_RELU) return y > 0.f ? g : 0.f;
    if (act == WU_ACT_LEAKY) return y > 0.f ? g : 0.2f * g;
    return g;
}
__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == WU_ACT_RELU) return fmaxf(v, 0.f);
    if (act == WU_ACT_LEAKY) return v > 0.f ? v : 0.2f * v;
    return v;
}
template <typename T> __device__ __forceinline__ uint4 gate16(const uint4& g, const uint4& y, int act) {
    // the activation kind is tested once per 16-byte chunk, not per element (scalar branch per value otherwise)
    constexpr int n = ElemTraits<T>::kPer16B;
    float gf[n], yf[n];
    unpack16<T>(g, gf);
    unpack16<T>(y, yf);
    if (act == WU_ACT_RELU) {
#pragma unroll
        for (int i = 0; i < n; ++i) gf[i] = yf[i] > 0.f ? gf[i] : 0.f;
    } else if (act == WU_ACT_LEAKY) {
#pragma unroll
        for (int i = 0; i < n; ++i) gf[i] = yf[i] > 0.f ? gf[i] : 0.2f * gf[i];
    }
    return pack16<T>(gf);
}

// ---- LDS-DMA through a buffer descriptor (buffer_load_dwordx4 ... lds) -----------------------------------------------
// Issued from inline asm: invisible to hipcc's waitcnt bookkeeping, so issuing it inside an MFMA loop does not make the
// compiler drain it before the next ds_read; the kernel waits for it itself (s_waitcnt vmcnt) ahead of the barrier that
// publishes the buffer.  M0 = wave-uniform LDS byte address of the 1-KiB piece (lane l lands at +16*l); memory address =
// descriptor base + voff (per lane) + soff (scalar); a lane with voff + soff >= num_records writes ZEROS to LDS (probed on
// gfx950: scratch/micro/buflds.hip) -- zero padding costs no select and no branch.
typedef int wu_rsrc_t __attribute__((ext_vector_type(4)));
constexpr unsigned kWuOOB = 0x80000000u;
// raw buffer descriptor (stride 0, word 3 = DATA_FORMAT 32) from provably wave-uniform words (cdna guide T20)
__device__ __forceinline__ wu_rsrc_t wu_make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long u = (unsigned long long)(uintptr_t)base;
    wu_rsrc_t r;
    r.x = __builtin_amdgcn_readfirstlane((int)(u & 0xffffffffull));
    r.y = __builtin_amdgcn_readfirstlane((int)((u >> 32) & 0xffffull));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ void wu_dma16b(unsigned voff, wu_rsrc_t rsrc, unsigned soff, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :: "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_byte_addr) : "memory", "m0");
}
// the 4-byte form: a 256-byte piece, lane l lands at M0 + 4*l (narrow rows: halo columns, one-byte-per-chunk keep masks)
__device__ __forceinline__ void wu_dma4b(unsigned voff, wu_rsrc_t rsrc, unsigned soff, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %0, %1, %2 offen lds"
                 :: "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_byte_addr) : "memory", "m0");
}

// ---- XCD-aware block remap ------------------------------------------------------------------------
// Blocks are dealt round-robin over the 8 XCDs (private 4 MiB L2 each).  Remap so each XCD owns a
// contiguous range of logical tile ids: neighbouring tiles (shared halos / same input tile for
// several Cout tiles) then hit the same L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// ---- counter-based RNG for dropout ----------------------------------------------------------------
// 64-bit mix (splitmix64 finaliser) of (seed, counter); 16 bits per element, 4 elements per call.
__host__ __device__ __forceinline__ uint64_t wu_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// keep decision for element `idx` (a linear NCHW index of the dropout tensor): 16-bit uniform < thr
__host__ __device__ __forceinline__ uint64_t wu_rand4(uint64_t seed, uint64_t group) {
    return wu_mix64(seed * 0x9E3779B97F4A7C15ull + group + 0x632BE59BD9B4E019ull);
}

__host__ __device__ static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
