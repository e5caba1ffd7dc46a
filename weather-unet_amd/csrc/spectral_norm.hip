// Spectral normalisation of a conv weight (torch.nn.utils.spectral_norm, nets.py:28-31): one power iteration
//   v <- normalize(W^T u),  u <- normalize(W v),  sigma = u . (W v),  W_eff = W / sigma
// and its backward  dW = G / sigma - (<G, W> / sigma^2) u v^T   (u, v are constants of the graph, as in torch).
// The reference runs this through ~25 tiny torch kernels per layer and forward; here it is 5 launches forward and 2
// backward, all deterministic (fixed-order block reductions), W (rows x cols, fp32, OIHW flattened) read twice.
#include <algorithm>

#include "wu_common.h"

namespace {

constexpr int kPartMax = 1024;   // max partial-sum slots in scratch

constexpr int kRowGroup = 32;    // rows per partial of W^T u

// W^T u in two deterministic stages (one thread per column walking ALL rows left 18 workgroups on a 256-CU chip):
//   A: pv[g][j] = sum_{i in row group g} W[i][j] u[i]                       grid (cols/256, rows/32)
//   B: v_raw[j] = sum_g pv[g][j];  part[blockIdx.x] = sum_j v_raw[j]^2      grid (cols/256)
__device__ __forceinline__ void sn_wt_u_partial_body(const float* __restrict__ w, const float* __restrict__ u, float* __restrict__ pv,
                                                     int rows, int cols, int bx, int by) {
    const int j = bx * 256 + threadIdx.x;
    if (j >= cols) return;
    const int i0 = by * kRowGroup, i1 = min(rows, i0 + kRowGroup);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = i0;
    for (; i + 4 <= i1; i += 4) {
        s0 += w[(size_t)i * cols + j] * u[i];
        s1 += w[(size_t)(i + 1) * cols + j] * u[i + 1];
        s2 += w[(size_t)(i + 2) * cols + j] * u[i + 2];
        s3 += w[(size_t)(i + 3) * cols + j] * u[i + 3];
    }
    for (; i < i1; ++i) s0 += w[(size_t)i * cols + j] * u[i];
    pv[(size_t)by * cols + j] = (s0 + s1) + (s2 + s3);
}
__global__ __launch_bounds__(256) void sn_wt_u_partial_kernel(const float* __restrict__ w, const float* __restrict__ u, float* __restrict__ pv,
                                                              int rows, int cols) {
    sn_wt_u_partial_body(w, u, pv, rows, cols, blockIdx.x, blockIdx.y);
}
__device__ __forceinline__ void sn_wt_u_fold_body(const float* __restrict__ pv, int groups, float* __restrict__ v_raw,
                                                  float* __restrict__ part, int cols, int bx, float* red) {
    const int j = bx * 256 + threadIdx.x;
    float s = 0.f;
    if (j < cols) {
        for (int g = 0; g < groups; ++g) s += pv[(size_t)g * cols + j];
        v_raw[j] = s;
    }
    red[threadIdx.x] = j < cols ? s * s : 0.f;
    __syncthreads();
    for (int m = 128; m > 0; m >>= 1) {
        if (threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[bx] = red[0];
}
__global__ __launch_bounds__(256) void sn_wt_u_fold_kernel(const float* __restrict__ pv, int groups, float* __restrict__ v_raw,
                                                           float* __restrict__ part, int cols) {
    __shared__ float red[256];
    sn_wt_u_fold_body(pv, groups, v_raw, part, cols, blockIdx.x, red);
}

// one workgroup per row: u_raw[i] = sum_j W[i][j] * (v_raw[j] * inv_norm_v)   (inv_norm_v from the partials, or 1 if !normalize_v)
// v_save (may be NULL): a second copy of the normalised v for the backward pass (the buffer itself advances on the next forward)
__device__ __forceinline__ void sn_w_v_body(const float* __restrict__ w, const float* __restrict__ v_in, const float* __restrict__ part_v,
                                            int nparts_v, int normalize_v, float eps, float* __restrict__ v_out, float* __restrict__ v_save,
                                            float* __restrict__ u_raw, int rows, int cols, int i, float* red) {
    float inv = 1.f;
    if (normalize_v) {
        float n2 = 0.f;
        for (int k = 0; k < nparts_v; ++k) n2 += part_v[k];
        inv = 1.f / fmaxf(sqrtf(n2), eps);
    }
    float s = 0.f;
    for (int j = threadIdx.x; j < cols; j += 256) {
        const float vj = v_in[j] * inv;
        if (i == 0 && v_out) v_out[j] = vj;                 // row 0's workgroup also publishes the normalised v
        if (i == 0 && v_save) v_save[j] = vj;
        s += w[(size_t)i * cols + j] * vj;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int m = 128; m > 0; m >>= 1) {
        if (threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) u_raw[i] = red[0];
}
__global__ __launch_bounds__(256) void sn_w_v_kernel(const float* __restrict__ w, const float* __restrict__ v_in, const float* __restrict__ part_v,
                                                     int nparts_v, int normalize_v, float eps, float* __restrict__ v_out,
                                                     float* __restrict__ u_raw, int rows, int cols) {
    __shared__ float red[256];
    sn_w_v_body(w, v_in, part_v, nparts_v, normalize_v, eps, v_out, nullptr, u_raw, rows, cols, blockIdx.x, red);
}

// single workgroup: power iteration: u = u_raw / max(|u_raw|, eps), sigma = u . u_raw;  else sigma = u_old . u_raw
// u_save (may be NULL): a copy of the u that defines sigma, for the backward pass
__device__ __forceinline__ void sn_finish_body(const float* __restrict__ u_raw, float* __restrict__ u, float* __restrict__ u_save, int rows,
                                               int power_iter, float eps, float* __restrict__ sigma_out, float* red) {
    float s = 0.f;
    for (int i = threadIdx.x; i < rows; i += 256) s += power_iter ? u_raw[i] * u_raw[i] : u[i] * u_raw[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int m = 128; m > 0; m >>= 1) {
        if (threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    const float tot = red[0];
    if (power_iter) {
        const float inv = 1.f / fmaxf(sqrtf(tot), eps);
        for (int i = threadIdx.x; i < rows; i += 256) {
            const float ui = u_raw[i] * inv;
            u[i] = ui;
            if (u_save) u_save[i] = ui;
        }
        if (threadIdx.x == 0) { sigma_out[0] = tot * inv; sigma_out[1] = 1.f / (tot * inv); }   // sigma = u . u_raw
    } else {
        if (u_save)
            for (int i = threadIdx.x; i < rows; i += 256) u_save[i] = u[i];
        if (threadIdx.x == 0) { sigma_out[0] = tot; sigma_out[1] = 1.f / tot; }
    }
}
__global__ __launch_bounds__(256) void sn_finish_kernel(const float* __restrict__ u_raw, float* __restrict__ u, int rows, int power_iter, float eps,
                                                        float* __restrict__ sigma_out) {
    __shared__ float red[256];
    sn_finish_body(u_raw, u, nullptr, rows, power_iter, eps, sigma_out, red);
}

__device__ __forceinline__ void sn_scale_body(const float* __restrict__ w, const float* __restrict__ sigma, float* __restrict__ w_eff, long long n,
                                              int bx, int nbx) {
    const float inv = sigma[1];
    for (long long i = (long long)bx * blockDim.x + threadIdx.x; i < n; i += (long long)nbx * blockDim.x) w_eff[i] = w[i] * inv;
}
__global__ void sn_scale_kernel(const float* __restrict__ w, const float* __restrict__ sigma, float* __restrict__ w_eff, long long n) {
    sn_scale_body(w, sigma, w_eff, n, blockIdx.x, gridDim.x);
}

// backward: partial <G, W> per block (nb blocks of this layer: the stride of the walk, hence the summation order)
__device__ __forceinline__ void sn_dot_body(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ part, long long n,
                                            int bx, int nb, float* red) {
    float s = 0.f;
    for (long long i = (long long)bx * 256 + threadIdx.x; i < n; i += (long long)nb * 256) s += g[i] * w[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int m = 128; m > 0; m >>= 1) {
        if (threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[bx] = red[0];
}
__global__ __launch_bounds__(256) void sn_dot_kernel(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ part, long long n) {
    __shared__ float red[256];
    sn_dot_body(g, w, part, n, blockIdx.x, gridDim.x, red);
}
__device__ __forceinline__ void sn_bwd_body(const float* __restrict__ g, const float* __restrict__ u, const float* __restrict__ v,
                                            const float* __restrict__ sigma, const float* __restrict__ part, int nparts, float* __restrict__ dw,
                                            int rows, int cols, int bx, int nbx, float* red) {
    // <G, W>: the per-block partials summed by the whole workgroup in a fixed tree (every thread used to walk all of them)
    float ps = 0.f;
    for (int k = threadIdx.x; k < nparts; k += 256) ps += part[k];
    red[threadIdx.x] = ps;
    __syncthreads();
    for (int m = 128; m > 0; m >>= 1) {
        if (threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    const float dot = red[0];
    const float inv = sigma[1], c = dot * inv * inv;          // <G,W> / sigma^2
    const long long n = (long long)rows * cols;
    for (long long i = (long long)bx * blockDim.x + threadIdx.x; i < n; i += (long long)nbx * blockDim.x) {
        const int r = (int)(i / cols), j = (int)(i - (long long)r * cols);
        dw[i] = g[i] * inv - c * u[r] * v[j];
    }
}
__global__ __launch_bounds__(256) void sn_bwd_kernel(const float* __restrict__ g, const float* __restrict__ u, const float* __restrict__ v, const float* __restrict__ sigma,
                              const float* __restrict__ part, int nparts, float* __restrict__ dw, int rows, int cols) {
    __shared__ float red[256];
    sn_bwd_body(g, u, v, sigma, part, nparts, dw, rows, cols, blockIdx.x, gridDim.x, red);
}

// ---- the same stages over up to kSnMax weights per launch (blockIdx.z / .y = layer): a discriminator forward normalises ten weights,
// and five launches of a few microseconds each per layer were 50 kernel boundaries per pass.  Every workgroup does exactly what the
// single-weight kernels do for its layer (same partial counts, same trees): results are bit-identical.
constexpr int kSnMax = 16;
struct SnMulti {
    const float* w[kSnMax];
    float* u[kSnMax]; float* v[kSnMax]; float* sigma[kSnMax]; float* w_eff[kSnMax]; float* scratch[kSnMax];
    float* u_save[kSnMax]; float* v_save[kSnMax];
    int rows[kSnMax], cols[kSnMax];
    int power_iter; float eps;
};
struct SnPtrs { float *v_raw, *u_raw, *part, *pv; };
__device__ __forceinline__ SnPtrs sn_ptrs(float* scratch, int rows, int cols) {
    SnPtrs p;
    p.v_raw = scratch; p.u_raw = scratch + cols; p.part = p.u_raw + rows; p.pv = p.part + 2 * kPartMax;
    return p;
}
__global__ __launch_bounds__(256) void sn_multi_wt_u_partial_kernel(const SnMulti a) {
    const int l = blockIdx.z, rows = a.rows[l], cols = a.cols[l];
    if ((int)blockIdx.x >= cdiv(cols, 256) || (int)blockIdx.y >= cdiv(rows, kRowGroup)) return;
    sn_wt_u_partial_body(a.w[l], a.u[l], sn_ptrs(a.scratch[l], rows, cols).pv, rows, cols, blockIdx.x, blockIdx.y);
}
__global__ __launch_bounds__(256) void sn_multi_wt_u_fold_kernel(const SnMulti a) {
    __shared__ float red[256];
    const int l = blockIdx.y, rows = a.rows[l], cols = a.cols[l];
    if ((int)blockIdx.x >= cdiv(cols, 256)) return;
    const SnPtrs p = sn_ptrs(a.scratch[l], rows, cols);
    sn_wt_u_fold_body(p.pv, cdiv(rows, kRowGroup), p.v_raw, p.part, cols, blockIdx.x, red);
}
__global__ __launch_bounds__(256) void sn_multi_w_v_kernel(const SnMulti a) {
    __shared__ float red[256];
    const int l = blockIdx.y, rows = a.rows[l], cols = a.cols[l];
    if ((int)blockIdx.x >= rows) return;
    const SnPtrs p = sn_ptrs(a.scratch[l], rows, cols);
    if (a.power_iter) sn_w_v_body(a.w[l], p.v_raw, p.part, cdiv(cols, 256), 1, a.eps, a.v[l], a.v_save[l], p.u_raw, rows, cols, blockIdx.x, red);
    else sn_w_v_body(a.w[l], a.v[l], p.part, 0, 0, a.eps, nullptr, a.v_save[l], p.u_raw, rows, cols, blockIdx.x, red);
}
__global__ __launch_bounds__(256) void sn_multi_finish_kernel(const SnMulti a) {
    __shared__ float red[256];
    const int l = blockIdx.x, rows = a.rows[l], cols = a.cols[l];
    sn_finish_body(sn_ptrs(a.scratch[l], rows, cols).u_raw, a.u[l], a.u_save[l], rows, a.power_iter, a.eps, a.sigma[l], red);
}
__global__ void sn_multi_scale_kernel(const SnMulti a) {
    const int l = blockIdx.y;
    if (!a.w_eff[l]) return;
    const long long n = (long long)a.rows[l] * a.cols[l];
    sn_scale_body(a.w[l], a.sigma[l], a.w_eff[l], n, blockIdx.x, gridDim.x);
}

struct SnBwdMulti {
    const float* g[kSnMax]; const float* w[kSnMax]; const float* u[kSnMax]; const float* v[kSnMax]; const float* sigma[kSnMax];
    float* dw[kSnMax]; float* scratch[kSnMax];
    int rows[kSnMax], cols[kSnMax];
};
__device__ __forceinline__ int sn_dot_blocks(long long n) {
    const long long nb = (n + 255) / 256;
    return (int)(nb > kPartMax ? kPartMax : nb);
}
__global__ __launch_bounds__(256) void sn_multi_dot_kernel(const SnBwdMulti a) {
    __shared__ float red[256];
    const int l = blockIdx.y;
    const long long n = (long long)a.rows[l] * a.cols[l];
    const int nb = sn_dot_blocks(n);
    if ((int)blockIdx.x >= nb) return;
    sn_dot_body(a.g[l], a.w[l], a.scratch[l], n, blockIdx.x, nb, red);
}
__global__ __launch_bounds__(256) void sn_multi_bwd_kernel(const SnBwdMulti a) {
    __shared__ float red[256];
    const int l = blockIdx.y;
    const long long n = (long long)a.rows[l] * a.cols[l];
    if ((long long)blockIdx.x * 256 >= n) return;                       // elementwise: any grid gives the same numbers
    sn_bwd_body(a.g[l], a.u[l], a.v[l], a.sigma[l], a.scratch[l], sn_dot_blocks(n), a.dw[l], a.rows[l], a.cols[l], blockIdx.x, gridDim.x, red);
}

}  // namespace

extern "C" size_t wu_spectral_norm_scratch_floats(int rows, int cols) {
    return (size_t)cols + rows + 2 * kPartMax + (size_t)cdiv(rows, kRowGroup) * cols;
}

// u (rows), v (cols): power-iteration buffers, updated in place when power_iter != 0.  sigma_out[0] = sigma,
// sigma_out[1] = 1/sigma.  w_eff (may be NULL) receives W / sigma.  scratch: wu_spectral_norm_scratch_floats().
extern "C" int wu_spectral_norm_fwd(const float* w, int rows, int cols, float* u, float* v, int power_iter, float eps,
                                    float* sigma_out, float* w_eff, float* scratch, void* stream) {
    WU_REQUIRE(w && u && v && sigma_out && scratch && rows > 0 && cols > 0, "spectral_norm_fwd: bad args");
    hipStream_t s = (hipStream_t)stream;
    float* v_raw = scratch;
    float* u_raw = scratch + cols;
    float* part = u_raw + rows;
    const int nb = cdiv(cols, 256);
    WU_REQUIRE(nb <= kPartMax, "spectral_norm_fwd: cols too large");
    if (power_iter) {
        float* pv = part + 2 * kPartMax;
        const int groups = cdiv(rows, kRowGroup);
        hipLaunchKernelGGL(sn_wt_u_partial_kernel, dim3(nb, groups), dim3(256), 0, s, w, u, pv, rows, cols);
        hipLaunchKernelGGL(sn_wt_u_fold_kernel, dim3(nb), dim3(256), 0, s, pv, groups, v_raw, part, cols);
        hipLaunchKernelGGL(sn_w_v_kernel, dim3(rows), dim3(256), 0, s, w, v_raw, part, nb, 1, eps, v, u_raw, rows, cols);
    } else {
        hipLaunchKernelGGL(sn_w_v_kernel, dim3(rows), dim3(256), 0, s, w, v, part, 0, 0, eps, (float*)nullptr, u_raw, rows, cols);
    }
    hipLaunchKernelGGL(sn_finish_kernel, dim3(1), dim3(256), 0, s, u_raw, u, rows, power_iter, eps, sigma_out);
    if (w_eff) {
        const long long n = (long long)rows * cols;
        long long g = (n + 255) / 256; if (g > 1024) g = 1024;
        hipLaunchKernelGGL(sn_scale_kernel, dim3((int)g), dim3(256), 0, s, w, sigma_out, w_eff, n);
    }
    WU_LAUNCH_CHECK("spectral_norm_fwd");
    return 0;
}

// dw = g / sigma - (<g, w> / sigma^2) u v^T   (u, v: the buffers used by the forward; sigma: its sigma_out)
extern "C" int wu_spectral_norm_bwd(const float* g, const float* w, const float* u, const float* v, const float* sigma,
                                    float* dw, int rows, int cols, float* scratch, void* stream) {
    WU_REQUIRE(g && w && u && v && sigma && dw && scratch && rows > 0 && cols > 0, "spectral_norm_bwd: bad args");
    hipStream_t s = (hipStream_t)stream;
    const long long n = (long long)rows * cols;
    long long nb = (n + 255) / 256; if (nb > kPartMax) nb = kPartMax;
    hipLaunchKernelGGL(sn_dot_kernel, dim3((int)nb), dim3(256), 0, s, g, w, scratch, n);
    long long gb = (n + 255) / 256; if (gb > 2048) gb = 2048;
    hipLaunchKernelGGL(sn_bwd_kernel, dim3((int)gb), dim3(256), 0, s, g, u, v, sigma, scratch, (int)nb, dw, rows, cols);
    WU_LAUNCH_CHECK("spectral_norm_bwd");
    return 0;
}

// wu_spectral_norm_fwd for n <= 16 weights in 5 launches (4 without a power iteration) instead of 5 n.  Arrays of n entries (host memory,
// read during the call): entry i as in the single-weight call.  u_save / v_save (arrays may be NULL, entries may be NULL): copies of
// the u, v that define sigma[i], written by the kernels themselves (what the backward needs once the buffers have moved on).
extern "C" int wu_spectral_norm_fwd_multi(int n, const float* const* w, const int* rows, const int* cols, float* const* u, float* const* v,
                                          int power_iter, float eps, float* const* sigma_out, float* const* w_eff, float* const* scratch,
                                          float* const* u_save, float* const* v_save, void* stream) {
    WU_REQUIRE(n > 0 && n <= kSnMax && w && rows && cols && u && v && sigma_out && scratch, "spectral_norm_fwd_multi: bad args");
    SnMulti a{};
    int max_nb = 0, max_groups = 0, max_rows = 0;
    long long max_n = 0;
    for (int i = 0; i < n; ++i) {
        WU_REQUIRE(w[i] && u[i] && v[i] && sigma_out[i] && scratch[i] && rows[i] > 0 && cols[i] > 0, "spectral_norm_fwd_multi: bad entry %d", i);
        WU_REQUIRE(cdiv(cols[i], 256) <= kPartMax, "spectral_norm_fwd_multi: cols too large");
        a.w[i] = w[i]; a.u[i] = u[i]; a.v[i] = v[i]; a.sigma[i] = sigma_out[i]; a.scratch[i] = scratch[i];
        a.w_eff[i] = w_eff ? w_eff[i] : nullptr;
        a.u_save[i] = u_save ? u_save[i] : nullptr;
        a.v_save[i] = v_save ? v_save[i] : nullptr;
        a.rows[i] = rows[i]; a.cols[i] = cols[i];
        max_nb = std::max(max_nb, cdiv(cols[i], 256));
        max_groups = std::max(max_groups, cdiv(rows[i], kRowGroup));
        max_rows = std::max(max_rows, rows[i]);
        if (a.w_eff[i]) max_n = std::max(max_n, (long long)rows[i] * cols[i]);
    }
    a.power_iter = power_iter; a.eps = eps;
    hipStream_t s = (hipStream_t)stream;
    if (power_iter) {
        hipLaunchKernelGGL(sn_multi_wt_u_partial_kernel, dim3(max_nb, max_groups, n), dim3(256), 0, s, a);
        hipLaunchKernelGGL(sn_multi_wt_u_fold_kernel, dim3(max_nb, n), dim3(256), 0, s, a);
    }
    hipLaunchKernelGGL(sn_multi_w_v_kernel, dim3(max_rows, n), dim3(256), 0, s, a);
    hipLaunchKernelGGL(sn_multi_finish_kernel, dim3(n), dim3(256), 0, s, a);
    if (max_n > 0) {
        long long g = (max_n + 255) / 256; if (g > 1024) g = 1024;
        hipLaunchKernelGGL(sn_multi_scale_kernel, dim3((int)g, n), dim3(256), 0, s, a);
    }
    WU_LAUNCH_CHECK("spectral_norm_fwd_multi");
    return 0;
}

// wu_spectral_norm_bwd for n <= 16 weights in 2 launches.
extern "C" int wu_spectral_norm_bwd_multi(int n, const float* const* g, const float* const* w, const float* const* u, const float* const* v,
                                          const float* const* sigma, float* const* dw, const int* rows, const int* cols, float* const* scratch,
                                          void* stream) {
    WU_REQUIRE(n > 0 && n <= kSnMax && g && w && u && v && sigma && dw && rows && cols && scratch, "spectral_norm_bwd_multi: bad args");
    SnBwdMulti a{};
    long long max_n = 0;
    for (int i = 0; i < n; ++i) {
        WU_REQUIRE(g[i] && w[i] && u[i] && v[i] && sigma[i] && dw[i] && scratch[i] && rows[i] > 0 && cols[i] > 0, "spectral_norm_bwd_multi: bad entry %d", i);
        a.g[i] = g[i]; a.w[i] = w[i]; a.u[i] = u[i]; a.v[i] = v[i]; a.sigma[i] = sigma[i]; a.dw[i] = dw[i]; a.scratch[i] = scratch[i];
        a.rows[i] = rows[i]; a.cols[i] = cols[i];
        max_n = std::max(max_n, (long long)rows[i] * cols[i]);
    }
    hipStream_t s = (hipStream_t)stream;
    long long nb = (max_n + 255) / 256; if (nb > kPartMax) nb = kPartMax;
    hipLaunchKernelGGL(sn_multi_dot_kernel, dim3((int)nb, n), dim3(256), 0, s, a);
    long long gb = (max_n + 255) / 256; if (gb > 2048) gb = 2048;
    hipLaunchKernelGGL(sn_multi_bwd_kernel, dim3((int)gb, n), dim3(256), 0, s, a);
    WU_LAUNCH_CHECK("spectral_norm_bwd_multi");
    return 0;
}
