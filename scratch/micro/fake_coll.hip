// Stand-in for a collective's kernel in the one-GPU overlap experiment (scratch/fake_collective_probe.py): G workgroups of 256 threads that
// hold `lds_bytes` of LDS and stay resident for `usec` microseconds (s_memrealtime, 100 MHz).  What it models: a kernel that cannot share a
// CU with a persistent 160-KiB-LDS conv workgroup and occupies the CUs it gets for the duration of a bucket's all-reduce.
// Build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC scratch/micro/fake_coll.hip -o scratch/micro/fake_coll.so
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void fake_coll_kernel(long long ticks, int* sink) {
    extern __shared__ int lds[];
    lds[threadIdx.x] = (int)threadIdx.x;
    __syncthreads();
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);     // every wave reaches the exit: bounded by `ticks`
    if (sink && lds[(threadIdx.x + 1) & 255] == -1) *sink = 1;
}

extern "C" int fake_coll_launch(int grid, int lds_bytes, int usec, void* stream) {
    if (grid <= 0 || grid > 1024 || lds_bytes < 1024 || lds_bytes > 65536 || usec < 0 || usec > 5000) return -1;
    hipLaunchKernelGGL(fake_coll_kernel, dim3(grid), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, (long long)usec * 100, (int*)nullptr);
    return (int)hipGetLastError();
}
