// Probe for the cross-stream hazard of DESIGN.md 4: a register-only kernel (no LDS, no memory traffic inside the loop) whose inner loop
// is packed-FP32 FMAs (v_pk_fma_f32, PACKED=1) or single FMAs (v_fmac_f32, PACKED=0).  Every lane iterates a contractive affine map on 16
// values, so the result is a deterministic function of the input: any difference between two launches is a hardware effect.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC scratch/micro/pk_probe.hip -o scratch/micro/libpk_probe.so
#include <hip/hip_runtime.h>

typedef float float2_t __attribute__((ext_vector_type(2)));

template <int PACKED>
__global__ __launch_bounds__(256) void pk_probe_kernel(float* __restrict__ out, const float* __restrict__ in, int iters) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = in[i + k];
    for (int it = 0; it < iters; ++it) {
        if (PACKED) {
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
                float2_t a = {v[k], v[k + 1]}, b = {0.75f, -0.5f}, c = {0.125f, 0.25f};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(a), "v"(b), "v"(c));
                v[k] = a.x; v[k + 1] = a.y;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
                float a0 = 0.125f, a1 = 0.25f;
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(v[k]), "v"(0.75f));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a1) : "v"(v[k + 1]), "v"(-0.5f));
                v[k] = a0; v[k + 1] = a1;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) out[i + k] = v[k];
}

extern "C" int pk_probe(float* out, const float* in, long long n_floats, int iters, int packed, void* stream) {
    const int grid = (int)(n_floats / (256 * 16));
    if (packed) hipLaunchKernelGGL(pk_probe_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, in, iters);
    else hipLaunchKernelGGL(pk_probe_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, in, iters);
    return (int)hipGetLastError();
}
