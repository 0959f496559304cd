// "Next" row f-4: the synthetic K-distributed sea-clutter generator on the GPU, so that a weak-scaling run
// can make each rank's frames in place instead of on the host (onet_amd/data.py is the NumPy statement of the same recipe,
// itself restating the reference's generators: KD = K_distributed_SeaClutter_Simulation_20210919.py:469-526, RG =
// Rayleigh_bg_Gaussian_EOT_generator_20230208.py:63-216):
//
//   white   w  = N(0,1) field from a counter-based generator (Philox4x32-10 + Box-Muller): reproducible from (seed, frame,
//                pixel) alone, no state, any launch geometry
//   colour  g  = Re ifft2( fft2(w) * sqrt(PSD) )                 (KD:70-81 `generate_GP_via_gaussianACF`)
//   texture tau = gammaincinv(nu, Phi(g / std g))                (KD:83-91 `mnlt`; nu = 5: closed-form Gamma(5) CDF, Newton)
//   speckle s  = ifft2( fft2(w') * sqrt(|f|^-0.6) ), complex     (KD:270-297)
//   clutter a  = |s| * sqrt(tau)                                 (KD:519-520)
//   targets    rotated 2-D Gaussian blobs added on top, label = blob > e^-2   (RG:63-175; parameters drawn on the host)
//
// The 2-D FFT is hipFFT-free: n x n frames with n = 512, one 256-thread block per line, radix-2 Stockham in LDS, rows
// then columns (the column pass reads a line with stride n: 2 MB per frame, L2-resident).  Memory-bound helper code, not
// on the training path.
#include <cmath>
#include "common.hpp"

using namespace onet;

namespace {

constexpr int FN = 512, FLOG = 9;

// ---------------------------------------------------------------- Philox4x32-10
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox4(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }   // (0, 1)

// data[f][i] = (N(0,1), 0): four normals per counter
__global__ __launch_bounds__(256) void clutter_white_kernel(float2* __restrict__ data, int64_t n_per_frame, int frames, uint32_t seed_lo,
                                                            uint32_t seed_hi, uint32_t stream) {
    const int64_t quads = (n_per_frame + 3) / 4;
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= quads * frames) return;
    const int f = (int)(i / quads);
    const int64_t q = i % quads;
    uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)f, stream};
    philox4(c, seed_lo, seed_hi);
    float z[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float r = sqrtf(-2.0f * logf(u01(c[2 * h]))), a = 6.28318530717958647692f * u01(c[2 * h + 1]);
        z[2 * h] = r * cosf(a);
        z[2 * h + 1] = r * sinf(a);
    }
    float2* o = data + (int64_t)f * n_per_frame + 4 * q;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (4 * q + k < n_per_frame) o[k] = make_float2(z[k], 0.f);
}

// ---------------------------------------------------------------- line FFT (512 points, radix-2 Stockham, in LDS)
// line l of `lines` starts at data[(l / inner) * outer_stride + (l % inner) * inner_stride]; element i at + i * elem_stride
__global__ __launch_bounds__(256) void fft512_kernel(float2* __restrict__ data, int64_t elem_stride, int inner, int64_t inner_stride,
                                                     int64_t outer_stride, int inverse) {
    __shared__ float2 buf[2][FN];
    const int l = blockIdx.x, t = threadIdx.x;
    float2* base = data + (int64_t)(l / inner) * outer_stride + (int64_t)(l % inner) * inner_stride;
    buf[0][t] = base[(int64_t)t * elem_stride];
    buf[0][t + 256] = base[(int64_t)(t + 256) * elem_stride];
    __syncthreads();
    const float sgn = inverse ? 1.f : -1.f;
    int cur = 0;
    // Stockham autosort: stage s combines sub-transforms of length Ls = 2^s
#pragma unroll
    for (int s = 0; s < FLOG; ++s) {
        const int Ls = 1 << s;
        const int j = t & (Ls - 1), k = t >> s;                  // butterfly t: element j of block k
        const float ang = sgn * 3.14159265358979323846f * (float)j / (float)Ls;
        float sn, cs;
        sincosf(ang, &sn, &cs);
        const float2 a = buf[cur][k * Ls + j], b = buf[cur][k * Ls + j + FN / 2];
        const float2 wb = make_float2(b.x * cs - b.y * sn, b.x * sn + b.y * cs);
        buf[cur ^ 1][k * 2 * Ls + j] = make_float2(a.x + wb.x, a.y + wb.y);
        buf[cur ^ 1][k * 2 * Ls + j + Ls] = make_float2(a.x - wb.x, a.y - wb.y);
        cur ^= 1;
        __syncthreads();
    }
    const float sc = inverse ? 1.0f / FN : 1.0f;
    base[(int64_t)t * elem_stride] = make_float2(buf[cur][t].x * sc, buf[cur][t].y * sc);
    base[(int64_t)(t + 256) * elem_stride] = make_float2(buf[cur][t + 256].x * sc, buf[cur][t + 256].y * sc);
}

// spec[f][y][x] *= filt[y][x]   (filt = sqrt(PSD), real)
__global__ __launch_bounds__(256) void clutter_filter_kernel(float2* __restrict__ spec, const float* __restrict__ filt, int64_t n, int frames) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n * frames) return;
    const float w = filt[i % n];
    float2 v = spec[i];
    v.x *= w; v.y *= w;
    spec[i] = v;
}

// sqrt(PSD) of the texture: PSD = max(Re fft2(acf), 0) with acf = exp(-(dx + dy) / corr_len), d = wrapped distance.
// acf_spec holds fft2(acf) (computed with the kernels above); out = sqrt(max(Re, 0))
__global__ __launch_bounds__(256) void clutter_acf_kernel(float2* __restrict__ acf, float corr_len) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= FN * FN) return;
    const int y = i / FN, x = i % FN;
    const float dy = (float)min(y, FN - y), dx = (float)min(x, FN - x);
    acf[i] = make_float2(expf(-(dx + dy) / corr_len), 0.f);
}
__global__ __launch_bounds__(256) void clutter_sqrt_psd_kernel(const float2* __restrict__ acf_spec, float* __restrict__ filt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < FN * FN) filt[i] = sqrtf(fmaxf(acf_spec[i].x, 0.f));
}
// speckle: sqrt(PSD) = (fx^2 + fy^2)^(-0.15), f = linspace(0.1, n / 10, n)   (KD:270-297)
__global__ __launch_bounds__(256) void clutter_speckle_filter_kernel(float* __restrict__ filt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= FN * FN) return;
    const float fs = FN / 10.0f, step = (fs - 0.1f) / (FN - 1);
    const float fy = 0.1f + step * (i / FN), fx = 0.1f + step * (i % FN);
    filt[i] = powf(fx * fx + fy * fy, -0.15f);
}

// per-frame sums of (re, re^2) [mode 0] or (|z|^2 * extra) ... -> double partials[f][2]; one block per (frame, slice)
__global__ __launch_bounds__(256) void clutter_moments_kernel(const float2* __restrict__ g, const float* __restrict__ amp, double* __restrict__ out,
                                                              int64_t n, int slices) {
    const int f = blockIdx.x / slices, sl = blockIdx.x % slices;
    const int64_t per = (n + slices - 1) / slices, i0 = sl * per, i1 = min(n, i0 + per);
    double s1 = 0.0, s2 = 0.0;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const float v = g ? g[(int64_t)f * n + i].x : amp[(int64_t)f * n + i];
        s1 += v;
        s2 += (double)v * v;
    }
    __shared__ double sh[8];
    double v2[2] = {s1, s2};
    block_sum_256<double, 2>(v2, sh);
    if (threadIdx.x == 0) {
        atomicAdd(out + 2 * f, v2[0]);          // a handful of adds per frame; statistics only (the frames themselves are
        atomicAdd(out + 2 * f + 1, v2[1]);      // reproducible up to the rounding of these two doubles' summation order)
    }
}

// Gamma(5) quantile by Newton on P(5, x) = 1 - e^-x (1 + x + x^2/2 + x^3/6 + x^4/24),  p(x) = x^4 e^-x / 24
__device__ __forceinline__ float gammaincinv5(float u) {
    // start: Wilson-Hilferty
    const float zq = -1.41421356f * erfcinvf(2.0f * u);                       // Phi^-1(u)
    const float c = 1.0f - 1.0f / 45.0f + zq * 0.14907119849998599f;         // 1 - 1/(9 nu) + z / (3 sqrt(nu))
    float x = fmaxf(5.0f * c * c * c, 1e-3f);
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const float e = expf(-x);
        const float poly = 1.0f + x * (1.0f + x * (0.5f + x * (1.0f / 6.0f + x * (1.0f / 24.0f))));
        const float cdf = 1.0f - e * poly;
        const float pdf = e * x * x * x * x * (1.0f / 24.0f);
        x = fmaxf(x - (cdf - u) / fmaxf(pdf, 1e-30f), 1e-6f);
    }
    return x;
}

// amp[f][i] = |speckle[f][i]| * sqrt(gammaincinv(5, Phi(g / std)))      (KD:83-91, 519-520)
__global__ __launch_bounds__(256) void clutter_combine_kernel(const float2* __restrict__ g, const float2* __restrict__ speckle, const double* __restrict__ mom,
                                                              float* __restrict__ amp, int64_t n, int frames) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n * frames) return;
    const int f = (int)(i / n);
    const double m = mom[2 * f] / (double)n, var = mom[2 * f + 1] / (double)n - m * m;
    const float z = (g[i].x) * (float)(1.0 / sqrt(var > 1e-30 ? var : 1e-30));          // data.py divides by std only (mean ~ 0)
    float u = 0.5f * erfcf(-z * 0.70710678118654752f);
    u = fminf(fmaxf(u, 1e-7f), 1.0f - 1e-7f);
    const float2 s = speckle[i];
    amp[i] = sqrtf(s.x * s.x + s.y * s.y) * sqrtf(gammaincinv5(u));
}

// out[f][c = 0][y][x] (crop H x W at (y0, x0)) = amp + peak_f * sum_t blob_t;  label = any blob > e^-2
// targets[f][t] = (cx, cy, sx, sy, cos th, sin th); peak_f = sqrt(10^(snr_f / 10) * mean(amp_f^2))
__global__ __launch_bounds__(256) void clutter_targets_crop_kernel(const float* __restrict__ amp, const double* __restrict__ mom2, const float* __restrict__ targets,
                                                                   const float* __restrict__ snr_db, int n_targets, float* __restrict__ out,
                                                                   float* __restrict__ label, int frames, int H, int W, int y0, int x0) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= (int64_t)frames * H * W) return;
    const int f = (int)(i / ((int64_t)H * W)), p = (int)(i % ((int64_t)H * W));
    const int y = y0 + p / W, x = x0 + p % W;
    const float power = (float)(mom2[2 * f + 1] / ((double)FN * FN));
    const float peak = sqrtf(powf(10.0f, snr_db[f] * 0.1f) * power);
    float v = amp[(int64_t)f * FN * FN + (int64_t)y * FN + x], lab = 0.f;
    for (int t = 0; t < n_targets; ++t) {
        const float* tg = targets + ((int64_t)f * n_targets + t) * 6;
        const float dx = (float)x - tg[0], dy = (float)y - tg[1];
        const float xr = dx * tg[4] + dy * tg[5], yr = -dx * tg[5] + dy * tg[4];
        const float e = -0.5f * ((xr / tg[2]) * (xr / tg[2]) + (yr / tg[3]) * (yr / tg[3]));
        v += peak * expf(e);
        if (e > -2.0f) lab = 1.f;
    }
    out[i] = v;
    if (label) label[i] = lab;
}

int fft2(float2* data, int frames, int inverse, hipStream_t st) {
    // rows: line = (frame, y), elements contiguous; columns: line = (frame, x), element stride n
    hipLaunchKernelGGL(fft512_kernel, dim3((unsigned)(frames * FN)), dim3(256), 0, st, data, (int64_t)1, FN, (int64_t)FN, (int64_t)FN * FN, inverse);
    int rc = check_launch("fft512_kernel(rows)");
    if (rc) return rc;
    hipLaunchKernelGGL(fft512_kernel, dim3((unsigned)(frames * FN)), dim3(256), 0, st, data, (int64_t)FN, FN, (int64_t)1, (int64_t)FN * FN, inverse);
    return check_launch("fft512_kernel(columns)");
}

}  // namespace

extern "C" {

int onet_clutter_frame_size(void) { return FN; }

int64_t onet_clutter_ws_bytes(int frames) {
    // two complex fields [frames][n][n] + amplitude [frames][n][n] + two filters [n][n] + one complex [n][n] + moments
    return (int64_t)frames * FN * FN * (8 + 8 + 4) + (int64_t)FN * FN * (4 + 4 + 8) + (int64_t)frames * 4 * 8 + 256;
}

int onet_clutter_generate(float* out, float* label, const float* targets, const float* snr_db, int n_targets, int frames, int H, int W,
                          uint64_t seed, float corr_len, void* ws, int64_t ws_bytes, void* stream) {
    ONET_REQUIRE(out && ws && (n_targets == 0 || (targets && snr_db)), "clutter_generate: null pointer");
    ONET_REQUIRE(frames > 0 && H > 0 && W > 0 && H <= FN && W <= FN && n_targets >= 0, "clutter_generate: bad shape (H, W <= %d)", FN);
    ONET_REQUIRE(ws_bytes >= onet_clutter_ws_bytes(frames), "clutter_generate: workspace too small");
    ONET_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15) == 0, "clutter_generate: workspace must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    const int64_t n = (int64_t)FN * FN;
    char* p = static_cast<char*>(ws);
    float2* gfield = reinterpret_cast<float2*>(p);  p += frames * n * 8;
    float2* sfield = reinterpret_cast<float2*>(p);  p += frames * n * 8;
    float* amp = reinterpret_cast<float*>(p);       p += frames * n * 4;
    float* filt_t = reinterpret_cast<float*>(p);    p += n * 4;
    float* filt_s = reinterpret_cast<float*>(p);    p += n * 4;
    float2* acf = reinterpret_cast<float2*>(p);     p += n * 8;
    double* mom = reinterpret_cast<double*>(p);                              // [frames][2] texture, then [frames][2] amplitude
    int rc;
    const unsigned gb = (unsigned)cdiv(n, 256);
    // filters
    hipLaunchKernelGGL(clutter_acf_kernel, dim3(gb), dim3(256), 0, st, acf, corr_len);
    if ((rc = check_launch("clutter_acf_kernel"))) return rc;
    if ((rc = fft2(acf, 1, 0, st))) return rc;
    hipLaunchKernelGGL(clutter_sqrt_psd_kernel, dim3(gb), dim3(256), 0, st, (const float2*)acf, filt_t);
    hipLaunchKernelGGL(clutter_speckle_filter_kernel, dim3(gb), dim3(256), 0, st, filt_s);
    if ((rc = check_launch("clutter filters"))) return rc;
    // white fields (independent Philox streams 0 / 1), coloured in the frequency domain
    const int64_t quads = (n + 3) / 4 * frames;
    hipLaunchKernelGGL(clutter_white_kernel, dim3((unsigned)cdiv(quads, 256)), dim3(256), 0, st, gfield, n, frames, (uint32_t)seed, (uint32_t)(seed >> 32), 0u);
    hipLaunchKernelGGL(clutter_white_kernel, dim3((unsigned)cdiv(quads, 256)), dim3(256), 0, st, sfield, n, frames, (uint32_t)seed, (uint32_t)(seed >> 32), 1u);
    if ((rc = check_launch("clutter_white_kernel"))) return rc;
    float2* fields[2] = {gfield, sfield};
    const float* filts[2] = {filt_t, filt_s};
    for (int k = 0; k < 2; ++k) {
        if ((rc = fft2(fields[k], frames, 0, st))) return rc;
        hipLaunchKernelGGL(clutter_filter_kernel, dim3((unsigned)cdiv(n * frames, 256)), dim3(256), 0, st, fields[k], filts[k], n, frames);
        if ((rc = check_launch("clutter_filter_kernel"))) return rc;
        if ((rc = fft2(fields[k], frames, 1, st))) return rc;
    }
    if (hipMemsetAsync(mom, 0, (size_t)frames * 4 * 8, st) != hipSuccess) { set_error("clutter_generate: memset failed"); return ONET_EHIP; }
    const int slices = 16;
    hipLaunchKernelGGL(clutter_moments_kernel, dim3((unsigned)(frames * slices)), dim3(256), 0, st, (const float2*)gfield, (const float*)nullptr, mom, n, slices);
    hipLaunchKernelGGL(clutter_combine_kernel, dim3((unsigned)cdiv(n * frames, 256)), dim3(256), 0, st, (const float2*)gfield, (const float2*)sfield, (const double*)mom,
                       amp, n, frames);
    hipLaunchKernelGGL(clutter_moments_kernel, dim3((unsigned)(frames * slices)), dim3(256), 0, st, (const float2*)nullptr, (const float*)amp, mom + 2 * frames, n, slices);
    if ((rc = check_launch("clutter_combine"))) return rc;
    const int y0 = (FN - H) / 2, x0 = (FN - W) / 2;
    hipLaunchKernelGGL(clutter_targets_crop_kernel, dim3((unsigned)cdiv((int64_t)frames * H * W, 256)), dim3(256), 0, st, (const float*)amp,
                       (const double*)(mom + 2 * frames), targets, snr_db, n_targets, out, label, frames, H, W, y0, x0);
    return check_launch("clutter_targets_crop_kernel");
}

}  // extern "C"
