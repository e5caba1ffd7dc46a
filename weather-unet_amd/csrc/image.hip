// GPU-side input pipeline: the per-image transforms the reference runs on the host through torchvision / PIL
// before every step (t_cls_train.py:81-108: Resize | RandomResizedCrop, RandomRotation(10), RandomHorizontalFlip, ColorJitter,
// ToTensor, Normalize(0.5, 0.5)), applied to a whole batch of decoded uint8 images that are already in HBM.
//
// The arithmetic follows Pillow's (torchvision's transforms are thin wrappers over it), so results are bit-identical for the
// geometric part and the colour jitter:
//   * resize: ImagingResample's two-pass (horizontal, then vertical) separable filter with 22-bit fixed-point coefficients and an
//     8-bit intermediate (libImaging/Resample.c): bilinear (triangle) filter whose support grows with the down-scaling factor;
//   * rotation: Image.rotate(angle, NEAREST, expand=False) = an affine map evaluated in 16.16 fixed point (Geometry.c affine_fixed),
//     coefficients prepared by the host exactly as Image.rotate / ImagingTransformAffine prepare them;
//   * flip: index reversal;
//   * colour jitter: ImageEnhance.{Brightness,Contrast,Color}.enhance = Image.blend(degenerate, image, factor) with C-float
//     arithmetic and truncation, grey = (19595 R + 38470 G + 7471 B + 0x8000) >> 16, contrast's degenerate = the rounded mean grey.
// One thread per output pixel walks the chain BACKWARDS (flip -> [rotate ->] resize taps -> [rotate ->] source), so no intermediate
// image exists except the 8-bit S x S staging buffer the colour jitter works on.
#include "wu_common.h"

namespace {

constexpr int kPrec = 22;          // Resample.c PRECISION_BITS = 32 - 8 - 2

struct ImgGeo {                    // one per image (host-prepared, see wu/input_pipeline.py)
    long long src_off;             // byte offset of the image inside the source buffer
    int src_h, src_w, src_ld;      // size and row stride (pixels) of the source image
    int crop_top, crop_left, crop_h, crop_w;   // window that is resized to S x S (the whole image for transforms.Resize)
    int flip;                      // RandomHorizontalFlip drew "flip"
    int rot[6];                    // 16.16 fixed-point affine coefficients a0..a5 of Image.rotate (Geometry.c affine_fixed)
    int do_rot;
};

// ---- Resample.c precompute_coeffs + normalize_coeffs_8bpc for ONE output coordinate ----------------------------------------
// table layout per (image, axis): bounds[S][2] = {xmin, count}, coeffs[S][ksize]
__global__ void resample_coeffs_kernel(const ImgGeo* __restrict__ geo, int* __restrict__ bounds, int* __restrict__ coeffs,
                                       int N, int S, int ksize) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * 2 * S) return;
    const int xx = i % S, axis = (i / S) & 1, n = i / (2 * S);
    const ImgGeo g = geo[n];
    const int in_size = axis ? g.crop_h : g.crop_w;           // the window is an image of its own (img.crop(...).resize(...))
    const double scale = (double)in_size / (double)S;
    double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;                 // bilinear: support 1
    const double center = (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    int* k = coeffs + ((size_t)(n * 2 + axis) * S + xx) * ksize;
    double ww = 0.0;
    for (int x = 0; x < xmax && x < ksize; ++x) {
        double t = (x + xmin - center + 0.5) * ss;
        if (t < 0.0) t = -t;
        ww += t < 1.0 ? 1.0 - t : 0.0;
    }
    for (int x = 0; x < ksize; ++x) {
        int q = 0;
        if (x < xmax) {
            double t = (x + xmin - center + 0.5) * ss;
            if (t < 0.0) t = -t;
            double w = t < 1.0 ? 1.0 - t : 0.0;
            if (ww != 0.0) w /= ww;
            q = w < 0 ? (int)(-0.5 + w * (double)(1 << kPrec)) : (int)(0.5 + w * (double)(1 << kPrec));
        }
        k[x] = q;
    }
    bounds[((size_t)(n * 2 + axis) * S + xx) * 2] = xmin;
    bounds[((size_t)(n * 2 + axis) * S + xx) * 2 + 1] = xmax < ksize ? xmax : ksize;
}

__device__ __forceinline__ int clip8(int v) {
    v >>= kPrec;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// Geometry.c affine_fixed, nearest: source pixel of output (x, y) of an xsize x ysize image, or "outside"
__device__ __forceinline__ bool rot_map(const int* a, int x, int y, int xsize, int ysize, int& xin, int& yin) {
    // xx = a2 + a0*x + a1*y ; yy = a5 + a3*x + a4*y  (32-bit wrap-around arithmetic, as in C)
    const int xx = (int)((unsigned)a[2] + (unsigned)a[0] * (unsigned)x + (unsigned)a[1] * (unsigned)y);
    const int yy = (int)((unsigned)a[5] + (unsigned)a[3] * (unsigned)x + (unsigned)a[4] * (unsigned)y);
    xin = xx >> 16;
    yin = yy >> 16;
    return xin >= 0 && xin < xsize && yin >= 0 && yin < ysize;
}

// out[n][oy][ox] for every pixel; dst_u8 (N,S,S,3) and / or dst_nchw (N,3,S,S) normalised with (v/255 - 0.5) / 0.5
__global__ __launch_bounds__(256) void image_geometry_kernel(const uint8_t* __restrict__ src, const ImgGeo* __restrict__ geo,
                                                             const int* __restrict__ bounds, const int* __restrict__ coeffs,
                                                             uint8_t* __restrict__ dst_u8, float* __restrict__ dst_nchw,
                                                             int N, int S, int ksize, int rot_first) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * S * S) return;
    const unsigned i32 = (unsigned)i, row = i32 / (unsigned)S;              // N*S*S < 2^31 (host check)
    const int ox = (int)(i32 - row * (unsigned)S), n = (int)(row / (unsigned)S), oy = (int)(row - (unsigned)n * (unsigned)S);
    const ImgGeo g = geo[n];
    const uint8_t* im = src + g.src_off;
    int px = g.flip ? S - 1 - ox : ox, py = oy;               // F.hflip is the last geometric op in both pipelines
    bool inside = true;
    if (!rot_first && g.do_rot) inside = rot_map(g.rot, px, py, S, S, px, py);      // Resize -> RandomRotation: rotate the S x S image
    int out[3] = {0, 0, 0};                                    // fill colour of Image.rotate: black
    if (inside) {
        const int* bx = bounds + ((size_t)(n * 2 + 0) * S + px) * 2;
        const int* by = bounds + ((size_t)(n * 2 + 1) * S + py) * 2;
        const int* kx = coeffs + ((size_t)(n * 2 + 0) * S + px) * ksize;
        const int* ky = coeffs + ((size_t)(n * 2 + 1) * S + py) * ksize;
        const int xmin = bx[0], nx = bx[1], ymin = by[0], ny = by[1];
        int acc[3] = {1 << (kPrec - 1), 1 << (kPrec - 1), 1 << (kPrec - 1)};
        for (int j = 0; j < ny; ++j) {
            const int sy = g.crop_top + ymin + j;
            int h[3] = {1 << (kPrec - 1), 1 << (kPrec - 1), 1 << (kPrec - 1)};
            for (int k = 0; k < nx; ++k) {
                const int sx = g.crop_left + xmin + k;
                int ux = sx, uy = sy;
                bool ok = true;
                if (rot_first && g.do_rot) ok = rot_map(g.rot, sx, sy, g.src_w, g.src_h, ux, uy);   // RandomRotation -> RandomResizedCrop
                if (ok) {
                    const uint8_t* p = im + ((size_t)uy * g.src_ld + ux) * 3;
                    const int c = kx[k];
                    h[0] += p[0] * c; h[1] += p[1] * c; h[2] += p[2] * c;
                }
            }
            const int c = ky[j];
            acc[0] += clip8(h[0]) * c; acc[1] += clip8(h[1]) * c; acc[2] += clip8(h[2]) * c;      // 8-bit intermediate of the two-pass resample
        }
        out[0] = clip8(acc[0]); out[1] = clip8(acc[1]); out[2] = clip8(acc[2]);
    }
    if (dst_u8) {
        uint8_t* d = dst_u8 + (size_t)i * 3;
        d[0] = (uint8_t)out[0]; d[1] = (uint8_t)out[1]; d[2] = (uint8_t)out[2];
    }
    if (dst_nchw) {
        const size_t plane = (size_t)S * S;
        float* d = dst_nchw + (size_t)n * 3 * plane + (size_t)oy * S + ox;
#pragma unroll
        for (int c = 0; c < 3; ++c) d[c * plane] = ((float)out[c] / 255.0f - 0.5f) / 0.5f;       // ToTensor, Normalize(0.5, 0.5)
    }
}

// ---- colour jitter: one workgroup per image, the three enhancers in the image's own random order -----------------------------
__device__ __forceinline__ int grey_l(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }   // Convert.c rgb2l
__device__ __forceinline__ int blend8(int degenerate, int v, float alpha) {                                                  // Blend.c
    const float t = (float)degenerate + alpha * ((float)v - (float)degenerate);
    if (alpha >= 0.f && alpha <= 1.f) return (int)(uint8_t)t;
    return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)(uint8_t)t);
}

__global__ __launch_bounds__(1024) void image_jitter_kernel(uint8_t* __restrict__ img, const float* __restrict__ factors,
                                                            const int* __restrict__ order, float* __restrict__ dst_nchw, int S) {
    __shared__ unsigned long long red[1024];
    __shared__ int mean_grey;
    const int n = blockIdx.x, tid = threadIdx.x;
    const int npix = S * S;
    uint8_t* im = img + (size_t)n * npix * 3;
    for (int k = 0; k < 3; ++k) {
        const int op = order[n * 3 + k];
        if (op < 0 || op > 2) continue;                        // wave-uniform (one image per workgroup)
        const float f = factors[n * 3 + op];
        if (op == 1) {                                         // Contrast: degenerate = int(mean(L) + 0.5), an image-wide reduction
            unsigned long long s = 0;
            for (int p = tid; p < npix; p += 1024) s += (unsigned long long)grey_l(im[p * 3], im[p * 3 + 1], im[p * 3 + 2]);
            red[tid] = s;
            __syncthreads();
            for (int off = 512; off > 0; off >>= 1) {
                if (tid < off) red[tid] += red[tid + off];
                __syncthreads();
            }
            if (tid == 0) mean_grey = (int)((double)red[0] / (double)npix + 0.5);
            __syncthreads();
        }
        for (int p = tid; p < npix; p += 1024) {
            const int r = im[p * 3], g = im[p * 3 + 1], b = im[p * 3 + 2];
            const int d = op == 0 ? 0 : (op == 1 ? mean_grey : grey_l(r, g, b));
            im[p * 3] = (uint8_t)blend8(d, r, f);
            im[p * 3 + 1] = (uint8_t)blend8(d, g, f);
            im[p * 3 + 2] = (uint8_t)blend8(d, b, f);
        }
        __syncthreads();
    }
    if (dst_nchw) {
        float* d = dst_nchw + (size_t)n * 3 * npix;
        for (int p = tid; p < npix; p += 1024)
#pragma unroll
            for (int c = 0; c < 3; ++c) d[(size_t)c * npix + p] = ((float)im[p * 3 + c] / 255.0f - 0.5f) / 0.5f;
    }
}

}  // namespace

extern "C" size_t wu_image_geo_bytes(void) { return sizeof(ImgGeo); }

extern "C" size_t wu_image_workspace_bytes(int N, int S, int ksize) {
    if (N <= 0 || S <= 0 || ksize <= 0) return 0;
    return ((size_t)N * 2 * S * 2 + (size_t)N * 2 * S * ksize) * sizeof(int);
}

extern "C" int wu_image_geometry(const uint8_t* src, const void* geo, void* workspace, size_t workspace_bytes,
                                 uint8_t* dst_u8, float* dst_nchw, int N, int S, int ksize, int rot_first, void* stream) {
    WU_REQUIRE(src && geo && workspace && (dst_u8 || dst_nchw), "image_geometry: null argument");
    WU_REQUIRE(N > 0 && S > 0 && S < 32768 && ksize >= 3 && ksize <= 4099, "image_geometry: bad shape N=%d S=%d ksize=%d", N, S, ksize);
    WU_REQUIRE(workspace_bytes >= wu_image_workspace_bytes(N, S, ksize), "image_geometry: workspace too small");
    WU_REQUIRE((long long)N * S * S < (1ll << 31), "image_geometry: N*S*S must stay below 2^31");
    hipStream_t s = (hipStream_t)stream;
    int* bounds = (int*)workspace;
    int* coeffs = bounds + (size_t)N * 2 * S * 2;
    hipLaunchKernelGGL(resample_coeffs_kernel, dim3(cdiv(N * 2 * S, 256)), dim3(256), 0, s, (const ImgGeo*)geo, bounds, coeffs, N, S, ksize);
    const long long total = (long long)N * S * S;
    hipLaunchKernelGGL(image_geometry_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, (const ImgGeo*)geo, bounds, coeffs,
                       dst_u8, dst_nchw, N, S, ksize, rot_first);
    WU_LAUNCH_CHECK("image_geometry");
    return 0;
}

extern "C" int wu_image_color_jitter(uint8_t* img_u8, const float* factors, const int* order, float* dst_nchw, int N, int S, void* stream) {
    WU_REQUIRE(img_u8 && factors && order && N > 0 && S > 0, "image_color_jitter: bad argument");
    hipLaunchKernelGGL(image_jitter_kernel, dim3(N), dim3(1024), 0, (hipStream_t)stream, img_u8, factors, order, dst_nchw, S);
    WU_LAUNCH_CHECK("image_color_jitter");
    return 0;
}
