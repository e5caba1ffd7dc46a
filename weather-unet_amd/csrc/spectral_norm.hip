// Spectral normalisation of a conv weight (torch.nn.utils.spectral_norm, nets.py:28-31): one power iteration
//   v <- normalize(W^T u),  u <- normalize(W v),  sigma = u . (W v),  W_eff = W / sigma
// and its backward  dW = G / sigma - (<G, W> / sigma^2) u v^T   (u, v are constants of the graph, as in torch).
// The reference runs this through ~25 tiny torch kernels per layer and forward; here it is 5 launches forward and 2
// backward, all deterministic (fixed-order block reductions), W (rows x cols, fp32, OIHW flattened) read twice.
#include "wu_common.h"

namespace {

constexpr int kPartMax = 1024;   // max partial-sum slots in scratch

constexpr int kRowGroup = 32;    // rows per partial of W^T u

// W^T u in two deterministic stages (one thread per column walking ALL rows left 18 workgroups on a 256-CU chip):
//   A: pv[g][j] = sum_{i in row group g} W[i][j] u[i]                       grid (cols/256, rows/32)
//   B: v_raw[j] = sum_g pv[g][j];  part[blockIdx.x] = sum_j v_raw[j]^2      grid (cols/256)
__global__ __launch_bounds__(256) void sn_wt_u_partial_kernel(const float* __restrict__ w, const float* __restrict__ u, float* __restrict__ pv,
                                                              int rows, int cols) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= cols) return;
    const int i0 = blockIdx.y * kRowGroup, i1 = min(rows, i0 + kRowGroup);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = i0;
    for (; i + 4 <= i1; i += 4) {
        s0 += w[(size_t)i * cols + j] * u[i];
        s1 += w[(size_t)(i + 1) * cols + j] * u[i + 1];
        s2 += w[(size_t)(i + 2) * cols + j] * u[i + 2];
        s3 += w[(size_t)(i + 3) * cols + j] * u[i + 3];
    }
    for (; i < i1; ++i) s0 += w[(size_t)i * cols + j] * u[i];
    pv[(size_t)blockIdx.y * cols + j] = (s0 + s1) + (s2 + s3);
}
__global__ __launch_bounds__(256) void sn_wt_u_fold_kernel(const float* __restrict__ pv, int groups, float* __restrict__ v_raw,
                                                           float* __restrict__ part, int cols) {
    __shared__ float red[256];
    const int j = blockIdx.x * 256 + threadIdx.x;
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
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// one workgroup per row: u_raw[i] = sum_j W[i][j] * (v_raw[j] * inv_norm_v)   (inv_norm_v from the partials, or 1 if !normalize_v)
__global__ __launch_bounds__(256) void sn_w_v_kernel(const float* __restrict__ w, const float* __restrict__ v_in, const float* __restrict__ part_v,
                                                     int nparts_v, int normalize_v, float eps, float* __restrict__ v_out,
                                                     float* __restrict__ u_raw, int rows, int cols) {
    __shared__ float red[256];
    float inv = 1.f;
    if (normalize_v) {
        float n2 = 0.f;
        for (int k = 0; k < nparts_v; ++k) n2 += part_v[k];
        inv = 1.f / fmaxf(sqrtf(n2), eps);
    }
    const int i = blockIdx.x;
    float s = 0.f;
    for (int j = threadIdx.x; j < cols; j += 256) {
        const float vj = v_in[j] * inv;
        if (i == 0 && v_out) v_out[j] = vj;                 // row 0's workgroup also publishes the normalised v
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

// single workgroup: power iteration: u = u_raw / max(|u_raw|, eps), sigma = u . u_raw;  else sigma = u_old . u_raw
__global__ __launch_bounds__(256) void sn_finish_kernel(const float* __restrict__ u_raw, float* __restrict__ u, int rows, int power_iter, float eps,
                                                        float* __restrict__ sigma_out) {
    __shared__ float red[256];
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
        for (int i = threadIdx.x; i < rows; i += 256) u[i] = u_raw[i] * inv;
        if (threadIdx.x == 0) { sigma_out[0] = tot * inv; sigma_out[1] = 1.f / (tot * inv); }   // sigma = u . u_raw
    } else if (threadIdx.x == 0) {
        sigma_out[0] = tot; sigma_out[1] = 1.f / tot;
    }
}

__global__ void sn_scale_kernel(const float* __restrict__ w, const float* __restrict__ sigma, float* __restrict__ w_eff, long long n) {
    const float inv = sigma[1];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) w_eff[i] = w[i] * inv;
}

// backward: partial <G, W> per block
__global__ __launch_bounds__(256) void sn_dot_kernel(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ part, long long n) {
    __shared__ float red[256];
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += g[i] * w[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int m = 128; m > 0; m >>= 1) {
        if (threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void sn_bwd_kernel(const float* __restrict__ g, const float* __restrict__ u, const float* __restrict__ v, const float* __restrict__ sigma,
                              const float* __restrict__ part, int nparts, float* __restrict__ dw, int rows, int cols) {
    // <G, W>: the per-block partials summed by the whole workgroup in a fixed tree (every thread used to walk all of them)
    __shared__ float red[256];
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
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), j = (int)(i - (long long)r * cols);
        dw[i] = g[i] * inv - c * u[r] * v[j];
    }
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
