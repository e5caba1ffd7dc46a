// micro-probe: MFMA rate vs LDS fragment-read density, software-pipelined (reads of step s+1 issued before the MFMAs of
// step s, double-buffered registers), 8 MFMAs per step, R ds_read_b128 per step, one or two waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
template <int R, int M, int BAR>
__global__ __launch_bounds__(512) void probe(const uint4* src, float* out, unsigned long long* cyc, int iters) {
    extern __shared__ uint4 lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = src[i & 4095];
    __syncthreads();
    f32x16_t acc[8];
    for (int t = 0; t < 8; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4 f[2][8];
    for (int j = 0; j < 8; ++j) { f[0][j] = lds[(lane + 64 * j) & 8191]; f[1][j] = f[0][j]; }
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it += 2) {
        if (BAR && (it % 18) == 0) __syncthreads();
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int cur = half, nxt = half ^ 1;
#pragma unroll
            for (int j = 0; j < R; ++j) f[nxt][j] = lds[(lane + 64 * (j + 8 * wave) + (it + half) * 64) & 8191];   // conflict-free 1-KiB rows
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < M; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, f[cur][t & 3]), __builtin_bit_cast(bf16x8_t, f[cur][4 + (t >> 2)]), acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int t = 0; t < 8; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int R, int M, int BAR>
__global__ __launch_bounds__(512) void probe16(const uint4* src, float* out, unsigned long long* cyc, int iters) {
    extern __shared__ uint4 lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = src[i & 4095];
    __syncthreads();
    f32x4_t acc[2 * M];
    for (int t = 0; t < 2 * M; ++t) for (int i = 0; i < 4; ++i) acc[t][i] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4 f[2][8];
    for (int j = 0; j < 8; ++j) { f[0][j] = lds[(lane + 64 * j) & 8191]; f[1][j] = f[0][j]; }
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it += 2) {
        if (BAR && (it % 18) == 0) __syncthreads();
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int cur = half, nxt = half ^ 1;
#pragma unroll
            for (int j = 0; j < R; ++j) f[nxt][j] = lds[(lane + 64 * (j + 8 * wave) + (it + half) * 64) & 8191];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 2 * M; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, f[cur][t & 3]), __builtin_bit_cast(bf16x8_t, f[cur][4 + ((t >> 2) & 3)]), acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int t = 0; t < 2 * M; ++t) for (int i = 0; i < 4; ++i) s += acc[t][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int R, int M, int BAR> void run16(int threads, const uint4* d, float* o, unsigned long long* c) {
    const int iters = 4000;
    hipFuncSetAttribute((const void*)probe16<R, M, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 16);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe16<R, M, BAR>), dim3(256), dim3(threads), 8192 * 16, 0, d, o, c, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe16<R, M, BAR>), dim3(256), dim3(threads), 8192 * 16, 0, d, o, c, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256 * 8];
    hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0; int n = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) { sum += (double)h[b * 8 + w]; ++n; }
    const double flops = 2.0 * 16 * 16 * 32 * (double)iters * 2 * M * (threads / 64) * 256;
    printf("16x16x32: %d ds_read_b128 per %d MFMAs%s, %d waves/SIMD: %.1f counter-cycles per 16x16x32 per SIMD, %.0f TFLOP/s, counter %.2f GHz\n",
           R, 2 * M, BAR ? ", barrier every 18 steps" : "", threads / 256, sum / n / (iters * 2.0 * M) / (threads / 256), flops / (ms * 1e-3) / 1e12, sum / n / (ms * 1e-3) / 1e9);
}
template <int R, int M, int BAR> void run(int threads, const uint4* d, float* o, unsigned long long* c) {
    const int iters = 4000;
    hipFuncSetAttribute((const void*)probe<R, M, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 16);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<R, M, BAR>), dim3(256), dim3(threads), 8192 * 16, 0, d, o, c, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<R, M, BAR>), dim3(256), dim3(threads), 8192 * 16, 0, d, o, c, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256 * 8];
    hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0; int n = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < threads / 64; ++w) { sum += (double)h[b * 8 + w]; ++n; }
    const double per_simd = sum / n / (iters * (double)M) / (threads / 256);
    const double flops = 2.0 * 32 * 32 * 16 * (double)iters * M * (threads / 64) * 256;
    printf("%d ds_read_b128 per %d MFMAs%s, %d waves/SIMD: %.1f counter-cycles per MFMA per SIMD, %.0f TFLOP/s, counter %.2f GHz\n",
           R, M, BAR ? ", barrier every 18 steps" : "", threads / 256, per_simd, flops / (ms * 1e-3) / 1e12, sum / n / (ms * 1e-3) / 1e9);
}
int main() {
    uint4* d; float* o; unsigned long long* c;
    hipMalloc(&d, 4096 * 16); hipMalloc(&o, 256 * 512 * 4); hipMalloc(&c, 256 * 8 * 8);
    uint32_t* h = (uint32_t*)malloc(4096 * 16);
    for (int i = 0; i < 4096 * 4; ++i) { uint32_t r = (uint32_t)rand(); h[i] = (r & 0x807f807fu) | 0x3f003f00u; }
    hipMemcpy(d, h, 4096 * 16, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<4, 4, 1>(512, d, o, c); run16<4, 4, 1>(512, d, o, c);
        run<8, 8, 1>(512, d, o, c); run16<8, 8, 1>(512, d, o, c);
        run<6, 8, 1>(512, d, o, c); run16<6, 8, 1>(512, d, o, c);
    }
    return 0;
}
