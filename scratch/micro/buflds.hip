// micro-probe: semantics of buffer_load_dwordx4 ... lds on gfx950 (OOB lanes, soffset, imm offset, M0 > 64 KiB)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void probe(const uint32_t* src, uint32_t* out, int nbytes, int lds_base, int soff) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    uint32_t* l32 = (uint32_t*)smem;
    for (int i = lane; i < 160 * 1024 / 4; i += 64) l32[i] = 0xFFFFFFFFu;
    __syncthreads();
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    const unsigned voff = lane * 16;
    const unsigned m0v = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + lds_base;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds\n\ts_waitcnt vmcnt(0)"
                 :: "v"(voff), "s"(rsrc), "s"(m0v), "s"(soff) : "memory", "m0");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = l32[lds_base / 4 + lane * 4 + i];
}
int main() {
    std::vector<uint32_t> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i;
    uint32_t *d, *o;
    hipMalloc(&d, 4096 * 4); hipMalloc(&o, 1024);
    hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    struct { int nbytes, lds_base, soff; } cases[] = {{512, 0, 0}, {512, 0, 256}, {4096, 150 * 1024, 64}, {0, 1024, 0}};
    for (auto c : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 160 * 1024, 0, d, o, c.nbytes, c.lds_base, c.soff);
        uint32_t r[256];
        hipError_t e = hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
        printf("nbytes=%d lds_base=%d soff=%d (%s):\n", c.nbytes, c.lds_base, c.soff, hipGetErrorString(e));
        for (int l : {0, 1, 15, 16, 31, 32, 33, 47, 48, 63}) printf("  lane %2d: %08x %08x %08x %08x\n", l, r[l*4], r[l*4+1], r[l*4+2], r[l*4+3]);
    }
    return 0;
}
