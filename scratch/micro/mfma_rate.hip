// micro-probe: cycles per v_mfma_f32_32x32x16_bf16 in a compiler-scheduled back-to-back stream, one or two waves per SIMD,
// NACC independent accumulators, with / without LDS fragment reads in the gaps
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
template <int NACC, int READS>
__global__ __launch_bounds__(512) void probe(const uint4* src, float* out, unsigned long long* cyc, int iters) {
    __shared__ uint4 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = src[i];
    __syncthreads();
    f32x16_t acc[NACC];
    for (int t = 0; t < NACC; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    uint4 a = lds[threadIdx.x & 1023], b = lds[(threadIdx.x + 64) & 1023];
    const int lane = threadIdx.x & 63;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < NACC; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc[t], 0, 0, 0);
            if (READS) { if (t & 1) a = lds[(lane + 64 * t + it) & 4095]; else b = lds[(lane + 64 * t + it + 7) & 4095]; }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int t = 0; t < NACC; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int NACC, int READS> void run(const char* name, int threads, const uint4* d, float* o, unsigned long long* c) {
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<NACC, READS>), dim3(256), dim3(threads), 0, 0, d, o, c, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<NACC, READS>), dim3(256), dim3(threads), 0, 0, d, o, c, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256 * 8];
    hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0; int n = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) { sum += (double)h[b * 8 + w]; ++n; }
    const double per_wave = sum / n / (iters * NACC);
    const double flops = 2.0 * 32 * 32 * 16 * (double)iters * NACC * (threads / 64) * 256;
    const double counter_ghz = sum / n / (ms * 1e-3) / 1e9;
    printf("%-44s %d waves/SIMD: %.1f counter-cycles per MFMA per SIMD; wall %.3f ms -> %.0f TFLOP/s; counter runs at %.2f GHz -> %.1f ns per MFMA per SIMD\n",
           name, threads / 256, per_wave / (threads / 256), ms, flops / (ms * 1e-3) / 1e12, counter_ghz, per_wave / (threads / 256) / counter_ghz);
}
int main() {
    uint4* d; float* o; unsigned long long* c;
    hipMalloc(&d, 4096 * 16); hipMalloc(&o, 256 * 512 * 4); hipMalloc(&c, 256 * 8 * 8);
    uint32_t* h = (uint32_t*)malloc(4096 * 16);
    for (int i = 0; i < 4096 * 4; ++i) { uint32_t r = (uint32_t)rand(); h[i] = (r & 0x807f807fu) | 0x3f003f00u; }   // bf16 pairs ~[0.5,1)
    hipMemcpy(d, h, 4096 * 16, hipMemcpyHostToDevice);
    run<9, 0>("9 accumulators, no reads", 256, d, o, c);
    run<9, 0>("9 accumulators, no reads", 512, d, o, c);
    run<4, 0>("4 accumulators, no reads", 256, d, o, c);
    run<4, 0>("4 accumulators, no reads", 512, d, o, c);
    run<8, 1>("8 accumulators, 1 ds_read_b128 per MFMA", 256, d, o, c);
    run<4, 1>("4 accumulators, 1 ds_read_b128 per MFMA", 512, d, o, c);
    return 0;
}
