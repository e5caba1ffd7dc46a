// Instruction count of the Winograd F(2x2,3x3) INPUT transform as an MFMA B-operand producer: a lane holds the 4x4 input patch of its tile
// for 8 channels (16 x 16-byte bf16 chunks, as read from the LDS halo tile), forms V = B^T d B in fp32 and packs the 16 results to bf16
// fragments.  (The forward conv would run 16 MFMAs per K step of 16 channels on these fragments instead of 36.)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk(float a, float b) { const bf16x2v r = __builtin_convertvector(f32x2v{a, b}, bf16x2v); return __builtin_bit_cast(uint32_t, r); }
__device__ __forceinline__ void unpack(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u); f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u); f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__global__ __launch_bounds__(256) void wino_xform_kernel(const uint4* __restrict__ in, uint4* __restrict__ out) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    float d[4][4][8];
#pragma unroll
    for (int i = 0; i < 16; ++i) unpack(in[t * 16 + i], d[i >> 2][i & 3]);
    float tmp[4][4][8], v[4][4][8];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) {       // B^T d: rows (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
            tmp[0][c][e] = d[0][c][e] - d[2][c][e]; tmp[1][c][e] = d[1][c][e] + d[2][c][e];
            tmp[2][c][e] = d[2][c][e] - d[1][c][e]; tmp[3][c][e] = d[1][c][e] - d[3][c][e];
        }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) {       // (B^T d) B
            v[r][0][e] = tmp[r][0][e] - tmp[r][2][e]; v[r][1][e] = tmp[r][1][e] + tmp[r][2][e];
            v[r][2][e] = tmp[r][2][e] - tmp[r][1][e]; v[r][3][e] = tmp[r][1][e] - tmp[r][3][e];
        }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float* f = v[i >> 2][i & 3];
        out[t * 16 + i] = make_uint4(pk(f[0], f[1]), pk(f[2], f[3]), pk(f[4], f[5]), pk(f[6], f[7]));
    }
}
