// K4 MaxPool2d(2) (OV:67), K5/K6 ConvTranspose2d pixel shuffle + F.pad + cat placement (OV:86-100),
// K5' bilinear x2 align_corners=True (OV:83), K7 clip(1-X+bias) (OV:180), plus fill/axpy/copy.
// All HBM-bound streaming kernels: lanes run along W (coalesced), grid-stride, no LDS.
#include "common.hpp"

using namespace onet;

static inline unsigned grid_for(int64_t n, int per_block = 256) {
    int64_t b = (n + per_block - 1) / per_block;
    if (b > 16384) b = 16384;   // grid-stride beyond this
    if (b < 1) b = 1;
    return (unsigned)b;
}

// ---------------------------------------------------------------- maxpool
__global__ void maxpool2_fwd_kernel(const float* __restrict__ x, int64_t x_bs, float* __restrict__ y,
                                    int64_t y_bs, int B, int C, int H, int W, int Ho, int Wo) {
    const int64_t n = (int64_t)B * C * Ho * Wo;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % Wo);
        int64_t r = i / Wo;
        const int oy = (int)(r % Ho);
        r /= Ho;
        const int c = (int)(r % C), b = (int)(r / C);
        const float* p = x + (int64_t)b * x_bs + (int64_t)c * H * W + (int64_t)(2 * oy) * W + 2 * ox;
        const float2 t = *reinterpret_cast<const float2*>(p);          // 2*ox even, W even or odd: see note
        const float2 u = *reinterpret_cast<const float2*>(p + W);
        const float m = fmaxf(fmaxf(t.x, t.y), fmaxf(u.x, u.y));
        y[(int64_t)b * y_bs + (int64_t)c * Ho * Wo + (int64_t)oy * Wo + ox] = m;
    }
}

// scalar variant for odd W or unaligned strides (float2 loads need 8-B alignment)
__global__ void maxpool2_fwd_kernel_s(const float* __restrict__ x, int64_t x_bs, float* __restrict__ y,
                                      int64_t y_bs, int B, int C, int H, int W, int Ho, int Wo) {
    const int64_t n = (int64_t)B * C * Ho * Wo;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % Wo);
        int64_t r = i / Wo;
        const int oy = (int)(r % Ho);
        r /= Ho;
        const int c = (int)(r % C), b = (int)(r / C);
        const float* p = x + (int64_t)b * x_bs + (int64_t)c * H * W + (int64_t)(2 * oy) * W + 2 * ox;
        y[(int64_t)b * y_bs + (int64_t)c * Ho * Wo + (int64_t)oy * Wo + ox] =
            fmaxf(fmaxf(p[0], p[1]), fmaxf(p[W], p[W + 1]));
    }
}

// one thread per INPUT pixel: dx = dy of its window if it is the window's first maximum, else 0
__global__ void maxpool2_bwd_kernel(const float* __restrict__ x, int64_t x_bs, const float* __restrict__ dy,
                                    int64_t dy_bs, float* __restrict__ dx, int64_t dx_bs, int B, int C, int H,
                                    int W, int Ho, int Wo, int accumulate, const float* __restrict__ add,
                                    int64_t add_bs, const float* __restrict__ add2, int64_t add2_bs) {
    const int64_t n = (int64_t)B * C * H * W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int ix = (int)(i % W);
        int64_t r = i / W;
        const int iy = (int)(r % H);
        r /= H;
        const int c = (int)(r % C), b = (int)(r / C);
        const int oy = iy >> 1, ox = ix >> 1;
        float g = 0.f;
        if (oy < Ho && ox < Wo) {
            const float* p = x + (int64_t)b * x_bs + (int64_t)c * H * W + (int64_t)(2 * oy) * W + 2 * ox;
            const float v0 = p[0], v1 = p[1], v2 = p[W], v3 = p[W + 1];
            // first maximum in scan order (dy-major), like ATen's max_pool2d
            int am = 0;
            float m = v0;
            if (v1 > m) { m = v1; am = 1; }
            if (v2 > m) { m = v2; am = 2; }
            if (v3 > m) { m = v3; am = 3; }
            const int me = ((iy & 1) << 1) | (ix & 1);
            if (me == am) g = dy[(int64_t)b * dy_bs + (int64_t)c * Ho * Wo + (int64_t)oy * Wo + ox];
        }
        float* o = dx + (int64_t)b * dx_bs + (int64_t)c * H * W + (int64_t)iy * W + ix;
        if (add) g += add[(int64_t)b * add_bs + (int64_t)c * H * W + (int64_t)iy * W + ix];
        if (add2) g += add2[(int64_t)b * add2_bs + (int64_t)c * H * W + (int64_t)iy * W + ix];
        *o = accumulate ? *o + g : g;
    }
}

// fast path (W % 4 == 0, H even, 16-B aligned planes): one thread per 2 x 4 input patch = two pooling windows;
// float4 row loads / stores, one index decomposition per 8 pixels (the per-pixel kernel above spends its time on
// four redundant window loads and three runtime divisions per element: 2.3 TB/s effective)
__global__ __launch_bounds__(256) void maxpool2_bwd_v4_kernel(const float* __restrict__ x, int64_t x_bs,
                                                              const float* __restrict__ dy, int64_t dy_bs,
                                                              float* __restrict__ dx, int64_t dx_bs, int B, int C, int H,
                                                              int W, int accumulate, const float* __restrict__ add,
                                                              int64_t add_bs, const float* __restrict__ add2,
                                                              int64_t add2_bs) {
    const int Ho = H >> 1, Wo = W >> 1, W4 = W >> 2;
    const int64_t n = (int64_t)B * C * Ho * W4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % W4);
        int64_t r = i / W4;
        const int oy = (int)(r % Ho);
        r /= Ho;
        const int c = (int)(r % C), b = (int)(r / C);
        const int64_t in_off = (int64_t)c * H * W + (int64_t)(2 * oy) * W + 4 * q;
        const float4 r0 = *reinterpret_cast<const float4*>(x + (int64_t)b * x_bs + in_off);
        const float4 r1 = *reinterpret_cast<const float4*>(x + (int64_t)b * x_bs + in_off + W);
        const float2 g = *reinterpret_cast<const float2*>(dy + (int64_t)b * dy_bs + (int64_t)c * Ho * Wo + (int64_t)oy * Wo + 2 * q);
        // first maximum in scan order (dy-major), like ATen's max_pool2d
        int a0 = 0, a1 = 0;
        float m0 = r0.x, m1 = r0.z;
        if (r0.y > m0) { m0 = r0.y; a0 = 1; }
        if (r1.x > m0) { m0 = r1.x; a0 = 2; }
        if (r1.y > m0) { m0 = r1.y; a0 = 3; }
        if (r0.w > m1) { m1 = r0.w; a1 = 1; }
        if (r1.z > m1) { m1 = r1.z; a1 = 2; }
        if (r1.w > m1) { m1 = r1.w; a1 = 3; }
        float4 o0 = make_float4(a0 == 0 ? g.x : 0.f, a0 == 1 ? g.x : 0.f, a1 == 0 ? g.y : 0.f, a1 == 1 ? g.y : 0.f);
        float4 o1 = make_float4(a0 == 2 ? g.x : 0.f, a0 == 3 ? g.x : 0.f, a1 == 2 ? g.y : 0.f, a1 == 3 ? g.y : 0.f);
        float* o = dx + (int64_t)b * dx_bs + in_off;
        if (add) {           // the other gradient of a tensor that also feeds a skip connection: dx = pool grad + add
            const float4 p0 = *reinterpret_cast<const float4*>(add + (int64_t)b * add_bs + in_off);
            const float4 p1 = *reinterpret_cast<const float4*>(add + (int64_t)b * add_bs + in_off + W);
            o0.x += p0.x; o0.y += p0.y; o0.z += p0.z; o0.w += p0.w;
            o1.x += p1.x; o1.y += p1.y; o1.z += p1.z; o1.w += p1.w;
        }
        if (add2) {          // and a third one (the U-Net's first output is pooled, concatenated AND returned to the head)
            const float4 p0 = *reinterpret_cast<const float4*>(add2 + (int64_t)b * add2_bs + in_off);
            const float4 p1 = *reinterpret_cast<const float4*>(add2 + (int64_t)b * add2_bs + in_off + W);
            o0.x += p0.x; o0.y += p0.y; o0.z += p0.z; o0.w += p0.w;
            o1.x += p1.x; o1.y += p1.y; o1.z += p1.z; o1.w += p1.w;
        }
        if (accumulate) {
            const float4 p0 = *reinterpret_cast<const float4*>(o), p1 = *reinterpret_cast<const float4*>(o + W);
            o0.x += p0.x; o0.y += p0.y; o0.z += p0.z; o0.w += p0.w;
            o1.x += p1.x; o1.y += p1.y; o1.z += p1.z; o1.w += p1.w;
        }
        *reinterpret_cast<float4*>(o) = o0;
        *reinterpret_cast<float4*>(o + W) = o1;
    }
}

// ---------------------------------------------------------------- convT pixel shuffle / space-to-depth
__global__ void pixel_shuffle2_bias_kernel(const float* __restrict__ sub, const float* __restrict__ bias,
                                           float* __restrict__ y, int64_t y_bs, int B, int C, int h, int w,
                                           int Ho, int Wo, int pt, int pl) {
    const int64_t n = (int64_t)B * C * Ho * Wo;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % Wo);
        int64_t r = i / Wo;
        const int oy = (int)(r % Ho);
        r /= Ho;
        const int co = (int)(r % C), b = (int)(r / C);
        const int uy = oy - pt, ux = ox - pl;
        float v = 0.f;   // F.pad border
        if (uy >= 0 && uy < 2 * h && ux >= 0 && ux < 2 * w) {
            const int q = ((uy & 1) << 1) | (ux & 1);
            v = sub[(((int64_t)b * 4 * C + (int64_t)q * C + co) * h + (uy >> 1)) * w + (ux >> 1)] +
                (bias ? bias[co] : 0.f);
        }
        y[(int64_t)b * y_bs + (int64_t)co * Ho * Wo + (int64_t)oy * Wo + ox] = v;
    }
}

__global__ void space_to_depth2_kernel(const float* __restrict__ dy, int64_t dy_bs, float* __restrict__ sub,
                                       int B, int C, int h, int w, int Ho, int Wo, int pt, int pl) {
    const int64_t n = (int64_t)B * 4 * C * h * w;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % w);
        int64_t r = i / w;
        const int y = (int)(r % h);
        r /= h;
        const int k = (int)(r % (4 * C)), b = (int)(r / (4 * C));
        const int q = k / C, co = k % C;
        sub[i] = dy[(int64_t)b * dy_bs + (int64_t)co * Ho * Wo + (int64_t)(pt + 2 * y + (q >> 1)) * Wo + pl + 2 * x + (q & 1)];
    }
}

// dbias[co] (+)= sum over b and the un-padded 2h x 2w window of dy.  Two deterministic stages:
// (channel, image) partials in fp64, then one wave per channel folds the B partials.
__global__ __launch_bounds__(256) void convT_dbias_partial_kernel(const float* __restrict__ dy, int64_t dy_bs,
                                                                  double* __restrict__ scratch, int B, int C, int h2,
                                                                  int w2, int Ho, int Wo, int pt, int pl) {
    __shared__ double red[4];
    const int co = blockIdx.x % C, b = blockIdx.x / C;
    double v[1] = {0.0};
    const float* src = dy + (int64_t)b * dy_bs + (int64_t)co * Ho * Wo;
    const int n = h2 * w2;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int x = i % w2, y = i / w2;
        v[0] += (double)src[(int64_t)(pt + y) * Wo + pl + x];
    }
    block_sum_256<double, 1>(v, red);
    if (threadIdx.x == 0) scratch[(int64_t)b * C + co] = v[0];
}

__global__ __launch_bounds__(64) void convT_dbias_final_kernel(const double* __restrict__ scratch,
                                                               float* __restrict__ dbias, int accumulate, int B, int C) {
    const int co = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < B; b += 64) s += scratch[(int64_t)b * C + co];
    s = wave_sum(s);
    if (threadIdx.x == 0) dbias[co] = accumulate ? dbias[co] + (float)s : (float)s;
}

__global__ void copy_strided_kernel(const float* __restrict__ src, int64_t src_bs, float* __restrict__ dst,
                                    int64_t dst_bs, int B, int64_t n4) {
    const int64_t tot = (int64_t)B * n4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < tot; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / n4);
        const int64_t j = i % n4;
        reinterpret_cast<float4*>(dst + (int64_t)b * dst_bs)[j] = reinterpret_cast<const float4*>(src + (int64_t)b * src_bs)[j];
    }
}

__global__ void copy_strided_kernel_s(const float* __restrict__ src, int64_t src_bs, float* __restrict__ dst,
                                      int64_t dst_bs, int B, int64_t n) {
    const int64_t tot = (int64_t)B * n;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < tot; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / n);
        const int64_t j = i % n;
        dst[(int64_t)b * dst_bs + j] = src[(int64_t)b * src_bs + j];
    }
}

// ---------------------------------------------------------------- bilinear x2 (align_corners=True)
__device__ __forceinline__ void bil_src(int o, int in, int out, int& i0, int& i1, float& l1) {
    // ATen area_pixel_compute_source_index(align_corners=True): src = o * (in-1)/(out-1)
    const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    const float s = scale * (float)o;
    i0 = (int)s;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

__global__ void bilinear2x_fwd_kernel(const float* __restrict__ x, int64_t x_bs, float* __restrict__ y,
                                      int64_t y_bs, int B, int C, int h, int w, int Ho, int Wo, int pt, int pl) {
    const int64_t n = (int64_t)B * C * Ho * Wo;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % Wo);
        int64_t r = i / Wo;
        const int oy = (int)(r % Ho);
        r /= Ho;
        const int c = (int)(r % C), b = (int)(r / C);
        const int uy = oy - pt, ux = ox - pl;
        float v = 0.f;
        if (uy >= 0 && uy < 2 * h && ux >= 0 && ux < 2 * w) {
            int y0, y1, x0, x1;
            float ly, lx;
            bil_src(uy, h, 2 * h, y0, y1, ly);
            bil_src(ux, w, 2 * w, x0, x1, lx);
            const float* p = x + (int64_t)b * x_bs + (int64_t)c * h * w;
            const float hy = 1.f - ly, hx = 1.f - lx;
            v = hy * (hx * p[y0 * w + x0] + lx * p[y0 * w + x1]) + ly * (hx * p[y1 * w + x0] + lx * p[y1 * w + x1]);
        }
        y[(int64_t)b * y_bs + (int64_t)c * Ho * Wo + (int64_t)oy * Wo + ox] = v;
    }
}

// GATHER form of the backward: one thread per dx element sums, in a fixed order (rows, then columns of the up-sampled plane), the
// contributions of the output pixels whose interpolation stencil touches it -- those o with floor(o * scale) in {i - 1, i}, found by
// running the FORWARD's own index computation (bil_src) over a conservative range, so forward and backward agree on every rounding.
// No atomics: results are bitwise reproducible and leave as plain coalesced stores; dx needs no zero-initialisation.
constexpr int BIL_MAXC = 8;       // candidates per axis: o in [(i - 1) / scale - 1, (i + 1) / scale + 1], scale = (n - 1) / (2 n - 1) in (1/3, 1/2)
__device__ __forceinline__ int bil_candidates(int i, int in, int out, int& lo, float* wgt) {
    const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    int hi;
    if (scale == 0.f) {           // in == 1: every output pixel reads input 0
        lo = 0;
        hi = out - 1;
    } else {
        lo = max(0, (int)floorf((float)(i - 1) / scale) - 1);
        hi = min(out - 1, (int)ceilf((float)(i + 1) / scale) + 1);
    }
    // trim to the pixels that actually touch i (both ends), so that at most BIL_MAXC remain
    int n = 0, first = lo;
    for (int o = lo; o <= hi; ++o) {
        int i0, i1;
        float l1;
        bil_src(o, in, out, i0, i1, l1);
        const float wv = (i0 == i ? 1.f - l1 : 0.f) + (i1 == i ? l1 : 0.f);
        const bool touches = i0 == i || i1 == i;
        if (!touches) {
            if (n == 0) first = o + 1;
            continue;
        }
        if (n < BIL_MAXC) wgt[n] = wv;
        // (pixels between two touching ones always touch: floor(o * scale) is monotone)
        ++n;
    }
    lo = first;
    return min(n, BIL_MAXC);
}

__global__ void bilinear2x_bwd_kernel(const float* __restrict__ dy, int64_t dy_bs, float* __restrict__ dx,
                                      int64_t dx_bs, int B, int C, int h, int w, int Ho, int Wo, int pt, int pl) {
    const int64_t n = (int64_t)B * C * h * w;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int ix = (int)(i % w);
        int64_t r = i / w;
        const int iy = (int)(r % h);
        r /= h;
        const int c = (int)(r % C), b = (int)(r / C);
        float wy[BIL_MAXC], wx[BIL_MAXC];
        int ylo, xlo;
        const int ny = bil_candidates(iy, h, 2 * h, ylo, wy), nx = bil_candidates(ix, w, 2 * w, xlo, wx);
        const float* g = dy + (int64_t)b * dy_bs + (int64_t)c * Ho * Wo + (int64_t)(ylo + pt) * Wo + xlo + pl;
        float acc = 0.f;
        for (int ky = 0; ky < ny; ++ky) {
            float row = 0.f;
            for (int kx = 0; kx < nx; ++kx) row = fmaf(wx[kx], g[(int64_t)ky * Wo + kx], row);
            acc = fmaf(wy[ky], row, acc);
        }
        dx[(int64_t)b * dx_bs + (int64_t)c * h * w + (int64_t)iy * w + ix] = acc;
    }
}

// ---------------------------------------------------------------- elementwise
__global__ void complement_clip_kernel(const float* __restrict__ x, float* __restrict__ y, float bias, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = fminf(fmaxf(1.f - x[i] + bias, 0.f), 1.f);
}

__global__ void fill_kernel(float* __restrict__ p, float v, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void axpy_kernel(const float* __restrict__ x, int64_t x_bs, float* __restrict__ y, int64_t y_bs, float a,
                            int B, int64_t n, int accumulate) {
    const int64_t tot = (int64_t)B * n;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < tot; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / n);
        const int64_t j = i % n;
        float* o = y + (int64_t)b * y_bs + j;
        const float v = a * x[(int64_t)b * x_bs + j];
        *o = accumulate ? *o + v : v;
    }
}


// maxpool2_bwd_v4_kernel + the first BatchNorm-backward pass of the layer that PRODUCED x: x = a = relu(bn(z)) is an
// encoder output (OV:58 -> 67, 100, 152), so the sum written to dx IS that layer's activation gradient and the
// (sum dy, sum dy * xhat) records of bn_relu_bwd_reduce_kernel (bn.hip; same mask expression, fp64 sums, hi/lo float
// pairs, [image * bands + band][C][4]) can be taken on the way out instead of re-reading dx.  One block per (image,
// channel, band of rows).
template <typename ZT> __device__ __forceinline__ float4 pool_ldz4(const ZT* p);      // z may be stored as bf16 (bn.hip: bn_ldz4)
template <> __device__ __forceinline__ float4 pool_ldz4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 pool_ldz4<__bf16>(const __bf16* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                       __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u));
}
template <typename ZT = float>
__global__ __launch_bounds__(256) void maxpool2_bwd_bn_kernel(const float* __restrict__ x, int64_t x_bs,
                                                              const float* __restrict__ dy, int64_t dy_bs,
                                                              const float* __restrict__ add, int64_t add_bs,
                                                              const float* __restrict__ add2, int64_t add2_bs,
                                                              float* __restrict__ dx, int64_t dx_bs,
                                                              const ZT* __restrict__ z, int64_t z_bs,
                                                              const float* __restrict__ save, int group_images,
                                                              float* __restrict__ part2, int C, int H, int W, int bands,
                                                              int band_rows, unsigned* __restrict__ amax = nullptr) {
    __shared__ double red[8];
    float vmax = 0.f;                           // amax: magnitude slots of dx (= the producing layer's da; bn.hip: bn_bwd_bound_kernel)
    const int c = blockIdx.x % C, p = blockIdx.x / C;
    const int b = p / bands, band = p % bands;
    const int Ho = H >> 1, Wo = W >> 1, W4 = W >> 2;
    const float* sv = save + (int64_t)(b / group_images) * 4 * C;
    const float mean = sv[c], invstd = sv[C + c], sc = sv[2 * C + c], sh = sv[3 * C + c];
    const double meand = mean, invd = invstd;
    const int oy0 = band * (band_rows >> 1), oy1 = min(oy0 + (band_rows >> 1), Ho);
    const int npatch = (oy1 - oy0) * W4;
    double v[2] = {0.0, 0.0};
    for (int i = threadIdx.x; i < npatch; i += 256) {
        const int q = i % W4, oy = oy0 + i / W4;
        const int64_t in_off = (int64_t)c * H * W + (int64_t)(2 * oy) * W + 4 * q;
        const float4 z0 = pool_ldz4<ZT>(z + (int64_t)b * z_bs + in_off);
        const float4 z1 = pool_ldz4<ZT>(z + (int64_t)b * z_bs + in_off + W);
        float4 r0, r1;
        if (x) {
            r0 = *reinterpret_cast<const float4*>(x + (int64_t)b * x_bs + in_off);
            r1 = *reinterpret_cast<const float4*>(x + (int64_t)b * x_bs + in_off + W);
        } else {
            // x == NULL (pre-split storage: the activation exists only as the split operand of its consumers): x = relu(bn(z)) is
            // recomputed with bn_relu_apply_kernel's expression -- the same bits, and one tensor read less
            r0 = make_float4(fmaxf(fmaf(z0.x - mean, sc, sh), 0.f), fmaxf(fmaf(z0.y - mean, sc, sh), 0.f),
                             fmaxf(fmaf(z0.z - mean, sc, sh), 0.f), fmaxf(fmaf(z0.w - mean, sc, sh), 0.f));
            r1 = make_float4(fmaxf(fmaf(z1.x - mean, sc, sh), 0.f), fmaxf(fmaf(z1.y - mean, sc, sh), 0.f),
                             fmaxf(fmaf(z1.z - mean, sc, sh), 0.f), fmaxf(fmaf(z1.w - mean, sc, sh), 0.f));
        }
        const float2 g = *reinterpret_cast<const float2*>(dy + (int64_t)b * dy_bs + (int64_t)c * Ho * Wo + (int64_t)oy * Wo + 2 * q);
        int a0 = 0, a1 = 0;
        float m0 = r0.x, m1 = r0.z;
        if (r0.y > m0) { m0 = r0.y; a0 = 1; }
        if (r1.x > m0) { m0 = r1.x; a0 = 2; }
        if (r1.y > m0) { m0 = r1.y; a0 = 3; }
        if (r0.w > m1) { m1 = r0.w; a1 = 1; }
        if (r1.z > m1) { m1 = r1.z; a1 = 2; }
        if (r1.w > m1) { m1 = r1.w; a1 = 3; }
        float o[8] = {a0 == 0 ? g.x : 0.f, a0 == 1 ? g.x : 0.f, a1 == 0 ? g.y : 0.f, a1 == 1 ? g.y : 0.f,
                      a0 == 2 ? g.x : 0.f, a0 == 3 ? g.x : 0.f, a1 == 2 ? g.y : 0.f, a1 == 3 ? g.y : 0.f};
        if (add) {
            const float4 p0 = *reinterpret_cast<const float4*>(add + (int64_t)b * add_bs + in_off);
            const float4 p1 = *reinterpret_cast<const float4*>(add + (int64_t)b * add_bs + in_off + W);
            o[0] += p0.x; o[1] += p0.y; o[2] += p0.z; o[3] += p0.w; o[4] += p1.x; o[5] += p1.y; o[6] += p1.z; o[7] += p1.w;
        }
        if (add2) {
            const float4 p0 = *reinterpret_cast<const float4*>(add2 + (int64_t)b * add2_bs + in_off);
            const float4 p1 = *reinterpret_cast<const float4*>(add2 + (int64_t)b * add2_bs + in_off + W);
            o[0] += p0.x; o[1] += p0.y; o[2] += p0.z; o[3] += p0.w; o[4] += p1.x; o[5] += p1.y; o[6] += p1.z; o[7] += p1.w;
        }
        float* dst = dx + (int64_t)b * dx_bs + in_off;
        *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(dst + W) = make_float4(o[4], o[5], o[6], o[7]);
        const float zz[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double dyk = fmaf(zz[k] - mean, sc, sh) > 0.f ? (double)o[k] : 0.0;
            v[0] += dyk;
            v[1] += dyk * (((double)zz[k] - meand) * invd);
            vmax = fmaxf(vmax, fabsf(o[k]));
        }
    }
    if (amax) {                                 // block maximum into one of the 64 slots (as bn.hip: amax_commit)
        __shared__ float wmax[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = vmax;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            unsigned* sl = amax + (blockIdx.x & 63) * 32;
            const unsigned bits = __builtin_bit_cast(unsigned, m);
            if (m == m && __builtin_nontemporal_load(sl) < bits) atomicMax(sl, bits);
        }
    }
    block_sum_256<double, 2>(v, red);
    if (threadIdx.x == 0) {
        float* o = part2 + ((int64_t)p * C + c) * 4;
        o[0] = (float)v[0];
        o[1] = (float)(v[0] - (double)o[0]);
        o[2] = (float)v[1];
        o[3] = (float)(v[1] - (double)o[2]);
    }
}

static void pool_bn_plan(int H, int W, int& bands, int& band_rows) {
    band_rows = std::max(2, (16384 / std::max(W, 1)) & ~1);
    bands = cdiv(H, band_rows);
}

static int maxpool2_bwd_impl(const float* x, int64_t x_bs, const float* dy, int64_t dy_bs, float* dx, int64_t dx_bs, int B,
                             int C, int H, int W, int accumulate, const float* add, int64_t add_bs, const float* add2,
                             int64_t add2_bs, void* stream) {
    const int64_t n = (int64_t)B * C * H * W;
    const bool v4 = ((W & 3) == 0) && ((H & 1) == 0) && ((x_bs & 3) == 0) && ((dx_bs & 3) == 0) && ((dy_bs & 1) == 0) &&
                    ((add_bs & 3) == 0) && ((reinterpret_cast<uintptr_t>(add) & 15) == 0) &&
                    ((add2_bs & 3) == 0) && ((reinterpret_cast<uintptr_t>(add2) & 15) == 0) &&
                    ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(dx) & 15) == 0) &&
                    ((reinterpret_cast<uintptr_t>(dy) & 7) == 0);
    if (v4) {
        hipLaunchKernelGGL(maxpool2_bwd_v4_kernel, dim3(grid_for(n / 8)), dim3(256), 0, as_stream(stream), x, x_bs, dy, dy_bs,
                           dx, dx_bs, B, C, H, W, accumulate, add, add_bs, add2, add2_bs);
        return check_launch("maxpool2_bwd_v4_kernel");
    }
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), x, x_bs, dy, dy_bs, dx,
                       dx_bs, B, C, H, W, H / 2, W / 2, accumulate, add, add_bs, add2, add2_bs);
    return check_launch("maxpool2_bwd_kernel");
}

extern "C" {

int onet_maxpool2_fwd(const float* x, int64_t x_bs, float* y, int64_t y_bs, int B, int C, int H, int W,
                      void* stream) {
    ONET_REQUIRE(x && y && B > 0 && C > 0 && H >= 2 && W >= 2, "maxpool2_fwd: bad args (H=%d W=%d)", H, W);
    const int Ho = H / 2, Wo = W / 2;
    const int64_t n = (int64_t)B * C * Ho * Wo;
    const bool aligned = ((W & 1) == 0) && ((x_bs & 1) == 0) && ((reinterpret_cast<uintptr_t>(x) & 7) == 0);
    if (aligned)
        hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), x, x_bs, y, y_bs, B, C, H, W, Ho, Wo);
    else
        hipLaunchKernelGGL(maxpool2_fwd_kernel_s, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), x, x_bs, y, y_bs, B, C, H, W, Ho, Wo);
    return check_launch("maxpool2_fwd_kernel");
}

int onet_maxpool2_bwd(const float* x, int64_t x_bs, const float* dy, int64_t dy_bs, const float* add, int64_t add_bs,
                      const float* add2, int64_t add2_bs, float* dx, int64_t dx_bs, int B, int C, int H, int W, int accumulate,
                      void* stream) {
    ONET_REQUIRE(x && dy && dx && (add || !add2) && !(add && accumulate) && B > 0 && C > 0 && H >= 2 && W >= 2, "maxpool2_bwd: bad args");
    return maxpool2_bwd_impl(x, x_bs, dy, dy_bs, dx, dx_bs, B, C, H, W, accumulate, add, add_bs, add2, add2_bs, stream);
}

int onet_maxpool2_bwd_bn_bands(int H, int W) {
    if (H < 2 || W < 4 || (W & 3) || (H & 1)) return 0;       // the fused form needs the 2 x 4 patch layout
    int bands, rows;
    pool_bn_plan(H, W, bands, rows);
    return bands;
}

static int maxpool2_bwd_add_bnreduce_impl(const float* x, int64_t x_bs, const float* dy, int64_t dy_bs, const float* add, int64_t add_bs,
                                          const float* add2, int64_t add2_bs, float* dx, int64_t dx_bs, const void* z, int z_bf16, int64_t z_bs,
                                          const float* save, int group_images, float* part2, unsigned* amax, int B, int C, int H, int W,
                                          void* stream);

int onet_maxpool2_bwd_add_bnreduce(const float* x, int64_t x_bs, const float* dy, int64_t dy_bs, const float* add, int64_t add_bs,
                                        const float* add2, int64_t add2_bs, float* dx, int64_t dx_bs, const void* z, int z_bf16, int64_t z_bs,
                                        const float* save, int group_images, float* part2, void* dx_amax, int B, int C, int H, int W,
                                        void* stream) {
    return maxpool2_bwd_add_bnreduce_impl(x, x_bs, dy, dy_bs, add, add_bs, add2, add2_bs, dx, dx_bs, z, z_bf16, z_bs, save, group_images, part2,
                                          (unsigned*)dx_amax, B, C, H, W, stream);
}

static int maxpool2_bwd_add_bnreduce_impl(const float* x, int64_t x_bs, const float* dy, int64_t dy_bs, const float* add, int64_t add_bs,
                                          const float* add2, int64_t add2_bs, float* dx, int64_t dx_bs, const void* z, int z_bf16, int64_t z_bs,
                                          const float* save, int group_images, float* part2, unsigned* amax, int B, int C, int H, int W,
                                          void* stream) {
    ONET_REQUIRE(dy && dx && z && save && part2 && B > 0 && C > 0, "maxpool2_bwd_add_bnreduce: bad args");
    ONET_REQUIRE(group_images > 0 && (B % group_images) == 0, "maxpool2_bwd_add_bnreduce: bad statistics groups");
    const bool ok = H >= 2 && W >= 4 && ((W & 3) == 0) && ((H & 1) == 0) && ((x_bs & 3) == 0) && ((dx_bs & 3) == 0) &&
                    ((z_bs & 3) == 0) && ((dy_bs & 1) == 0) && ((add_bs & 3) == 0) && ((add2_bs & 3) == 0) &&
                    ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(dx) & 15) == 0) &&
                    ((reinterpret_cast<uintptr_t>(z) & 15) == 0) && ((reinterpret_cast<uintptr_t>(add) & 15) == 0) &&
                    ((reinterpret_cast<uintptr_t>(add2) & 15) == 0) && ((reinterpret_cast<uintptr_t>(dy) & 7) == 0);
    ONET_REQUIRE(ok, "maxpool2_bwd_add_bnreduce: W %% 4 == 0, even H and 16-byte aligned planes required (onet_maxpool2_bwd_bn_bands() == 0 elsewhere)");
    int bands, rows;
    pool_bn_plan(H, W, bands, rows);
    const int64_t blocks = (int64_t)B * bands * C;
    ONET_REQUIRE(blocks < (1ll << 31), "maxpool2_bwd_add_bnreduce: grid too large");
    if (z_bf16)
        hipLaunchKernelGGL(maxpool2_bwd_bn_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, x_bs, dy, dy_bs, add,
                           add_bs, add2, add2_bs, dx, dx_bs, (const __bf16*)z, z_bs, save, group_images, part2, C, H, W, bands, rows, amax);
    else
        hipLaunchKernelGGL(maxpool2_bwd_bn_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, x_bs, dy, dy_bs, add,
                           add_bs, add2, add2_bs, dx, dx_bs, (const float*)z, z_bs, save, group_images, part2, C, H, W, bands, rows, amax);
    return check_launch("maxpool2_bwd_bn_kernel");
}

int onet_pixel_shuffle2_bias(const float* sub, const float* bias, float* y, int64_t y_bs, int B, int C, int h,
                             int w, int Ho, int Wo, int pt, int pl, void* stream) {
    ONET_REQUIRE(sub && y && B > 0 && C > 0 && h > 0 && w > 0, "pixel_shuffle2: bad args");
    ONET_REQUIRE(pt >= 0 && pl >= 0 && pt + 2 * h <= Ho && pl + 2 * w <= Wo, "pixel_shuffle2: window outside plane");
    const int64_t n = (int64_t)B * C * Ho * Wo;
    hipLaunchKernelGGL(pixel_shuffle2_bias_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), sub, bias, y,
                       y_bs, B, C, h, w, Ho, Wo, pt, pl);
    return check_launch("pixel_shuffle2_bias_kernel");
}

int onet_copy_strided(const float* src, int64_t src_bs, float* dst, int64_t dst_bs, int B, int64_t n,
                      void* stream) {
    ONET_REQUIRE(src && dst && B > 0 && n > 0, "copy_strided: bad args");
    const bool v4 = ((n & 3) == 0) && ((src_bs & 3) == 0) && ((dst_bs & 3) == 0) &&
                    ((reinterpret_cast<uintptr_t>(src) & 15) == 0) && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);
    if (v4)
        hipLaunchKernelGGL(copy_strided_kernel, dim3(grid_for((int64_t)B * n / 4)), dim3(256), 0, as_stream(stream), src, src_bs, dst, dst_bs, B, n / 4);
    else
        hipLaunchKernelGGL(copy_strided_kernel_s, dim3(grid_for((int64_t)B * n)), dim3(256), 0, as_stream(stream), src, src_bs, dst, dst_bs, B, n);
    return check_launch("copy_strided_kernel");
}

int onet_bilinear2x_fwd(const float* x, int64_t x_bs, float* y, int64_t y_bs, int B, int C, int h, int w, int Ho,
                        int Wo, int pt, int pl, void* stream) {
    ONET_REQUIRE(x && y && B > 0 && C > 0 && h > 0 && w > 0, "bilinear2x_fwd: bad args");
    ONET_REQUIRE(pt >= 0 && pl >= 0 && pt + 2 * h <= Ho && pl + 2 * w <= Wo, "bilinear2x_fwd: window outside plane");
    const int64_t n = (int64_t)B * C * Ho * Wo;
    hipLaunchKernelGGL(bilinear2x_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), x, x_bs, y, y_bs, B, C,
                       h, w, Ho, Wo, pt, pl);
    return check_launch("bilinear2x_fwd_kernel");
}

int onet_complement_clip(const float* x, float* y, float bias, int64_t n, void* stream) {
    ONET_REQUIRE(x && y && n > 0, "complement_clip: bad args");
    hipLaunchKernelGGL(complement_clip_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), x, y, bias, n);
    return check_launch("complement_clip_kernel");
}

int onet_fill(float* p, float value, int64_t n, void* stream) {
    ONET_REQUIRE(p && n > 0, "fill: bad args");
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), p, value, n);
    return check_launch("fill_kernel");
}

int onet_axpy(const float* x, int64_t x_bs, float* y, int64_t y_bs, float a, int B, int64_t n, int accumulate,
              void* stream) {
    ONET_REQUIRE(x && y && B > 0 && n > 0, "axpy: bad args");
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for((int64_t)B * n)), dim3(256), 0, as_stream(stream), x, x_bs, y, y_bs, a, B, n, accumulate);
    return check_launch("axpy_kernel");
}

}  // extern "C"

extern "C" int onet_space_to_depth2(const float* dy, int64_t dy_bs, float* sub, float* dbias, double* scratch,
                                    int accumulate, int B, int C, int h, int w, int Ho, int Wo, int pt, int pl,
                                    void* stream) {
    ONET_REQUIRE(dy && sub && B > 0 && C > 0 && h > 0 && w > 0, "space_to_depth2: bad args");
    ONET_REQUIRE(pt >= 0 && pl >= 0 && pt + 2 * h <= Ho && pl + 2 * w <= Wo, "space_to_depth2: window outside plane");
    const int64_t n = (int64_t)B * 4 * C * h * w;
    hipLaunchKernelGGL(space_to_depth2_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), dy, dy_bs, sub, B, C, h,
                       w, Ho, Wo, pt, pl);
    int rc = check_launch("space_to_depth2_kernel");
    if (rc) return rc;
    if (dbias) {
        ONET_REQUIRE(scratch, "space_to_depth2: dbias needs a scratch buffer of B*C doubles");
        hipLaunchKernelGGL(convT_dbias_partial_kernel, dim3((unsigned)((int64_t)B * C)), dim3(256), 0, as_stream(stream),
                           dy, dy_bs, scratch, B, C, 2 * h, 2 * w, Ho, Wo, pt, pl);
        rc = check_launch("convT_dbias_partial_kernel");
        if (rc) return rc;
        hipLaunchKernelGGL(convT_dbias_final_kernel, dim3(C), dim3(64), 0, as_stream(stream), (const double*)scratch,
                           dbias, accumulate, B, C);
        rc = check_launch("convT_dbias_final_kernel");
    }
    return rc;
}

extern "C" int onet_convT2x2_dbias(const float* dy, int64_t dy_bs, float* dbias, double* scratch, int accumulate, int B,
                                   int C, int h, int w, int Ho, int Wo, int pt, int pl, void* stream) {
    ONET_REQUIRE(dy && dbias && scratch && B > 0 && C > 0 && h > 0 && w > 0, "convT2x2_dbias: bad args");
    ONET_REQUIRE(pt >= 0 && pl >= 0 && pt + 2 * h <= Ho && pl + 2 * w <= Wo, "convT2x2_dbias: window outside plane");
    hipLaunchKernelGGL(convT_dbias_partial_kernel, dim3((unsigned)((int64_t)B * C)), dim3(256), 0, as_stream(stream), dy,
                       dy_bs, scratch, B, C, 2 * h, 2 * w, Ho, Wo, pt, pl);
    int rc = check_launch("convT_dbias_partial_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(convT_dbias_final_kernel, dim3(C), dim3(64), 0, as_stream(stream), (const double*)scratch, dbias,
                       accumulate, B, C);
    return check_launch("convT_dbias_final_kernel");
}

// every element of dx is written (gather form: no atomics, no zero-initialisation needed)
extern "C" int onet_bilinear2x_bwd(const float* dy, int64_t dy_bs, float* dx, int64_t dx_bs, int B, int C, int h,
                                   int w, int Ho, int Wo, int pt, int pl, void* stream) {
    ONET_REQUIRE(dy && dx && B > 0 && C > 0 && h > 0 && w > 0, "bilinear2x_bwd: bad args");
    ONET_REQUIRE(pt >= 0 && pl >= 0 && pt + 2 * h <= Ho && pl + 2 * w <= Wo, "bilinear2x_bwd: window outside plane");
    const int64_t n = (int64_t)B * C * h * w;
    hipLaunchKernelGGL(bilinear2x_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), dy, dy_bs, dx, dx_bs, B,
                       C, h, w, Ho, Wo, pt, pl);
    return check_launch("bilinear2x_bwd_kernel");
}
