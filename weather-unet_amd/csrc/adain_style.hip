// AdaIN style statistics (reference utils.py:41-48): y_ = l1(y).view(N, C, 4); y_mean = mean_k y_; y_std = sqrt(var_k(unbiased) + eps)
// and their backward into l1.weight / l1.bias.  In stock PyTorch this is ~15 tiny kernels per decoder level and direction
// (GEMM, bias, Welford, add, sqrt, mean and their autograd twins): ~0.25 ms of a 9.5 ms training step in launch-sized
// kernels.  Here: one launch forward, one backward, deterministic (fixed summation order).
#include "wu_common.h"

namespace {

constexpr int kMaxNc = 32;

// one thread per (n, c): the four rows 4c..4c+3 of W against y[n, :]
__global__ __launch_bounds__(256) void adain_style_fwd_kernel(const float* __restrict__ y, const float* __restrict__ w, const float* __restrict__ b,
                                                              float eps, float* __restrict__ y_std, float* __restrict__ y_mean,
                                                              float* __restrict__ y4, int N, int C, int nc) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= N * C) return;
    const int n = idx / C, c = idx - n * C;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float* wr = w + (size_t)(4 * c + k) * nc;
        float s = b ? b[4 * c + k] : 0.f;
        for (int j = 0; j < nc; ++j) s += wr[j] * y[n * nc + j];
        v[k] = s;
    }
    const float m = (v[0] + v[1] + v[2] + v[3]) * 0.25f;
    const float var = ((v[0] - m) * (v[0] - m) + (v[1] - m) * (v[1] - m) + (v[2] - m) * (v[2] - m) + (v[3] - m) * (v[3] - m)) * (1.f / 3.f);
    y_mean[idx] = m;
    y_std[idx] = sqrtf(var + eps);
    if (y4) *(float4*)(y4 + (size_t)idx * 4) = make_float4(v[0], v[1], v[2], v[3]);
}

// EIGHT lanes per weight row r = 4c + k: dW[r][:] = sum_n g[n] y[n][:], db[r] = sum_n g[n],
// g[n] = d_mean[n,c]/4 + d_std[n,c] * (y4[n,c,k] - mean) / (3 std).  Lane `sub` takes the samples n = sub, sub + 8, ...; the eight
// partial sums are combined with a fixed xor-shuffle tree (deterministic).  One thread walking all N samples was a chain of N
// dependent memory round trips: 38 us per launch for half a megabyte, three times at the very end of every backward pass.
__global__ __launch_bounds__(256) void adain_style_bwd_kernel(const float* __restrict__ d_std, const float* __restrict__ d_mean,
                                                              const float* __restrict__ y, const float* __restrict__ y4,
                                                              const float* __restrict__ y_std, const float* __restrict__ y_mean,
                                                              float* __restrict__ dw, float* __restrict__ db, int N, int C, int nc, int accumulate) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int r = t >> 3, sub = t & 7;
    const bool valid = r < 4 * C;                   // every lane stays for the shuffles
    const int c = (valid ? r : 0) >> 2, k = r & 3;
    float acc[kMaxNc];
#pragma unroll
    for (int j = 0; j < kMaxNc; ++j) acc[j] = 0.f;
    float bs = 0.f;
    for (int n = sub; n < N; n += 8) {
        const int i = n * C + c;
        const float g = d_mean[i] * 0.25f + d_std[i] * (y4[(size_t)i * 4 + k] - y_mean[i]) / (3.f * y_std[i]);
        bs += g;
#pragma unroll
        for (int j = 0; j < kMaxNc; ++j)
            if (j < nc) acc[j] += g * y[n * nc + j];
    }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
        bs += __shfl_xor(bs, m);
#pragma unroll
        for (int j = 0; j < kMaxNc; ++j)
            if (j < nc) acc[j] += __shfl_xor(acc[j], m);
    }
    if (!valid || sub != 0) return;
    if (db) db[r] = accumulate ? db[r] + bs : bs;
#pragma unroll
    for (int j = 0; j < kMaxNc; ++j)
        if (j < nc) dw[(size_t)r * nc + j] = accumulate ? dw[(size_t)r * nc + j] + acc[j] : acc[j];
}

// The three AdaIN levels of a U-Net forward / backward (cunet.py:59,66,73) in ONE launch each: blockIdx.y = level, the level's pointers and channel
// count come from a by-value table, the arithmetic per (n, c) / per weight row is the single-level kernel's (bit-identical results).  Per step the six
// launch-sized kernels sat on the critical path at the very top of forward and the very end of backward (three dependent launches each).
constexpr int kMaxLevels = 4;
struct StyleMultiFwd {
    const float* w[kMaxLevels]; const float* b[kMaxLevels]; float* y_std[kMaxLevels]; float* y_mean[kMaxLevels]; float* y4[kMaxLevels];
    float eps[kMaxLevels]; int C[kMaxLevels];
};
struct StyleMultiBwd {
    const float* d_std[kMaxLevels]; const float* d_mean[kMaxLevels]; const float* y4[kMaxLevels]; const float* y_std[kMaxLevels]; const float* y_mean[kMaxLevels];
    float* dw[kMaxLevels]; float* db[kMaxLevels]; int C[kMaxLevels];
};

__global__ __launch_bounds__(256) void adain_style_fwd_multi_kernel(const float* __restrict__ y, const StyleMultiFwd a, int N, int nc) {
    const int lv = blockIdx.y;
    const int C = a.C[lv];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= N * C) return;
    const float* __restrict__ w = a.w[lv];
    const float* __restrict__ b = a.b[lv];
    const int n = idx / C, c = idx - n * C;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float* wr = w + (size_t)(4 * c + k) * nc;
        float s = b ? b[4 * c + k] : 0.f;
        for (int j = 0; j < nc; ++j) s += wr[j] * y[n * nc + j];
        v[k] = s;
    }
    const float m = (v[0] + v[1] + v[2] + v[3]) * 0.25f;
    const float var = ((v[0] - m) * (v[0] - m) + (v[1] - m) * (v[1] - m) + (v[2] - m) * (v[2] - m) + (v[3] - m) * (v[3] - m)) * (1.f / 3.f);
    a.y_mean[lv][idx] = m;
    a.y_std[lv][idx] = sqrtf(var + a.eps[lv]);
    if (a.y4[lv]) *(float4*)(a.y4[lv] + (size_t)idx * 4) = make_float4(v[0], v[1], v[2], v[3]);
}

__global__ __launch_bounds__(256) void adain_style_bwd_multi_kernel(const float* __restrict__ y, const StyleMultiBwd a, int N, int nc, int accumulate) {
    const int lv = blockIdx.y;
    const int C = a.C[lv];
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (((t & ~63) >> 3) >= 4 * C) return;          // whole waves beyond this level's rows (wave-uniform: the shuffles below always see full 8-lane groups)
    const float* __restrict__ d_std = a.d_std[lv];
    const float* __restrict__ d_mean = a.d_mean[lv];
    const float* __restrict__ y4 = a.y4[lv];
    const float* __restrict__ y_std = a.y_std[lv];
    const float* __restrict__ y_mean = a.y_mean[lv];
    const int r = t >> 3, sub = t & 7;
    const bool valid = r < 4 * C;
    const int c = (valid ? r : 0) >> 2, k = r & 3;
    float acc[kMaxNc];
#pragma unroll
    for (int j = 0; j < kMaxNc; ++j) acc[j] = 0.f;
    float bs = 0.f;
    for (int n = sub; n < N; n += 8) {
        const int i = n * C + c;
        const float g = d_mean[i] * 0.25f + d_std[i] * (y4[(size_t)i * 4 + k] - y_mean[i]) / (3.f * y_std[i]);
        bs += g;
#pragma unroll
        for (int j = 0; j < kMaxNc; ++j)
            if (j < nc) acc[j] += g * y[n * nc + j];
    }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
        bs += __shfl_xor(bs, m);
#pragma unroll
        for (int j = 0; j < kMaxNc; ++j)
            if (j < nc) acc[j] += __shfl_xor(acc[j], m);
    }
    if (!valid || sub != 0) return;
    float* __restrict__ dw = a.dw[lv];
    float* __restrict__ db = a.db[lv];
    if (db) db[r] = accumulate ? db[r] + bs : bs;
#pragma unroll
    for (int j = 0; j < kMaxNc; ++j)
        if (j < nc) dw[(size_t)r * nc + j] = accumulate ? dw[(size_t)r * nc + j] + acc[j] : acc[j];
}

}  // namespace

extern "C" int wu_adain_style_fwd_multi(int levels, const float* y, const float* const* w, const float* const* b, const float* eps,
                                        float* const* y_std, float* const* y_mean, float* const* y4, int N, const int* C, int nc, void* stream) {
    WU_REQUIRE(levels >= 1 && levels <= kMaxLevels && y && w && b && eps && y_std && y_mean && y4 && C && N > 0 && nc > 0 && nc <= kMaxNc,
               "adain_style_fwd_multi: bad args (levels <= %d, nc <= %d)", kMaxLevels, kMaxNc);
    StyleMultiFwd a{};
    int cmax = 0;
    for (int i = 0; i < levels; ++i) {
        WU_REQUIRE(w[i] && y_std[i] && y_mean[i] && C[i] > 0, "adain_style_fwd_multi: level %d", i);
        WU_REQUIRE(!y4[i] || ((uintptr_t)y4[i] % 16) == 0, "adain_style_fwd_multi: y4 alignment");
        a.w[i] = w[i]; a.b[i] = b[i]; a.y_std[i] = y_std[i]; a.y_mean[i] = y_mean[i]; a.y4[i] = y4[i]; a.eps[i] = eps[i]; a.C[i] = C[i];
        cmax = C[i] > cmax ? C[i] : cmax;
    }
    hipLaunchKernelGGL(adain_style_fwd_multi_kernel, dim3(cdiv(N * cmax, 256), levels), dim3(256), 0, (hipStream_t)stream, y, a, N, nc);
    WU_LAUNCH_CHECK("adain_style_fwd_multi");
    return 0;
}

extern "C" int wu_adain_style_bwd_multi(int levels, const float* const* d_std, const float* const* d_mean, const float* y, const float* const* y4,
                                        const float* const* y_std, const float* const* y_mean, float* const* dw, float* const* db,
                                        int N, const int* C, int nc, int accumulate, void* stream) {
    WU_REQUIRE(levels >= 1 && levels <= kMaxLevels && d_std && d_mean && y && y4 && y_std && y_mean && dw && db && C && N > 0 && nc > 0 && nc <= kMaxNc,
               "adain_style_bwd_multi: bad args");
    StyleMultiBwd a{};
    int cmax = 0;
    for (int i = 0; i < levels; ++i) {
        WU_REQUIRE(d_std[i] && d_mean[i] && y4[i] && y_std[i] && y_mean[i] && dw[i] && C[i] > 0, "adain_style_bwd_multi: level %d", i);
        a.d_std[i] = d_std[i]; a.d_mean[i] = d_mean[i]; a.y4[i] = y4[i]; a.y_std[i] = y_std[i]; a.y_mean[i] = y_mean[i]; a.dw[i] = dw[i]; a.db[i] = db[i];
        a.C[i] = C[i];
        cmax = C[i] > cmax ? C[i] : cmax;
    }
    hipLaunchKernelGGL(adain_style_bwd_multi_kernel, dim3(cdiv(4 * cmax * 8, 256), levels), dim3(256), 0, (hipStream_t)stream, y, a, N, nc, accumulate);
    WU_LAUNCH_CHECK("adain_style_bwd_multi");
    return 0;
}

extern "C" int wu_adain_style_fwd(const float* y, const float* w, const float* b, float eps, float* y_std, float* y_mean, float* y4,
                                  int N, int C, int nc, void* stream) {
    WU_REQUIRE(y && w && y_std && y_mean && N > 0 && C > 0 && nc > 0 && nc <= kMaxNc, "adain_style_fwd: bad args (nc <= %d)", kMaxNc);
    WU_REQUIRE(!y4 || ((uintptr_t)y4 % 16) == 0, "adain_style_fwd: y4 alignment");
    hipLaunchKernelGGL(adain_style_fwd_kernel, dim3(cdiv(N * C, 256)), dim3(256), 0, (hipStream_t)stream, y, w, b, eps, y_std, y_mean, y4, N, C, nc);
    WU_LAUNCH_CHECK("adain_style_fwd");
    return 0;
}

extern "C" int wu_adain_style_bwd(const float* d_std, const float* d_mean, const float* y, const float* y4, const float* y_std,
                                  const float* y_mean, float* dw, float* db, int N, int C, int nc, int accumulate, void* stream) {
    WU_REQUIRE(d_std && d_mean && y && y4 && y_std && y_mean && dw && N > 0 && C > 0 && nc > 0 && nc <= kMaxNc, "adain_style_bwd: bad args");
    hipLaunchKernelGGL(adain_style_bwd_kernel, dim3(cdiv(4 * C * 8, 256)), dim3(256), 0, (hipStream_t)stream, d_std, d_mean, y, y4, y_std, y_mean,
                       dw, db, N, C, nc, accumulate);
    WU_LAUNCH_CHECK("adain_style_bwd");
    return 0;
}
