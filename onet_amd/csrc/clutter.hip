// "Next" row f-4: the synthetic K-distributed sea-clutter generator on the GPU, so that a weak-scaling run
// can make each rank's frames in place instead of on the host.  It follows the FRAME RECIPE of the reference's generators step
// by step (KD = K_distributed_SeaClutter_Simulation_20210919.py, RG = Rayleigh_bg_Gaussian_EOT_generator_20230208.py;
// onet_amd/data.py is the NumPy statement of the same recipe and builds the two filters), and its statistics are pinned to
// frames made by the reference's own functions (tests/golden/clutter_stats.npz):
//
//   white   w   = N(0,1) field from a counter-based generator (Philox4x32-10 + Box-Muller): reproducible from (seed, frame,
//                 pixel) alone, no state, any launch geometry
//   texture g   = Re ifft2( fft2(w) * Ht ),  tau = gammaincinv(5, Phi(g))     (KD:499-503, KD:83-91 `mnlt`; g is NOT normalised:
//                 the recipe feeds a field of variance 1.109 to the unit-variance transform).  Ht = sqrt(fft2(R_G)) with R_G the
//                 per-pixel root field of the Hermite-coefficient polynomial (KD:121-164, 483-497), built once on the host
//   speckle s   = ifft2( fft2(w') * Hs ), complex,  Hs = sqrt((fx^2 + fy^2)^-0.3) on the recipe's index grid   (KD:270-297)
//   clutter a   = | s * sqrt(tau) |                                            (KD:519-520)
//   targets     20 rotated un-normalised Gaussians, placed ONE AFTER THE OTHER:  bg += (template > bg) * template, template =
//               sqrt(10^(snr/10) * mean(a^2)) * kgauss;  label |= kgauss > max - 2 std   (RG:63-175, 189-209; parameters are
//               drawn and reduced to window / quadratic form / threshold on the host: data.target_params)
//   frame       400 x 400 (RG:186), centre crop H x W (RG:302)
//
// The reference colours on the 400-torus; the 2-D FFT here is hipFFT-free radix-2 (512 points, one 256-thread block per line,
// Stockham in LDS, rows then columns), so both filters arrive as the frequency responses of the recipe's convolution kernels
// placed on the 512-torus with wrapped lags (data._embed_kernel): inside the 400 x 400 window the fields have the recipe's
// covariance up to the kernels' wrap-around tails (< 0.1 % of the texture kernel's energy).  Memory-bound helper code, not on
// the training path.
#include <cmath>
#include "common.hpp"

using namespace onet;

namespace {

constexpr int FN = 512, FLOG = 9;     // FFT torus
constexpr int FR = 400;               // the recipe's frame (RG:186), a window of the torus

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

// spec[f][i] *= filt[i]   (complex: the frequency response of the recipe's convolution kernel on the 512-torus)
__global__ __launch_bounds__(256) void clutter_filter_kernel(float2* __restrict__ spec, const float2* __restrict__ filt, int64_t n, int frames) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n * frames) return;
    const float2 w = filt[i % n];
    const float2 v = spec[i];
    spec[i] = make_float2(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
}

// Gamma(5) quantile of Phi(z) (KD:83-91 with v = 5), by Newton from the Wilson-Hilferty start.  The lower half solves
// P(5, x) = Phi(z), the upper half Q(5, x) = 1 - Phi(z), each from erfc of the NEAR tail, so neither loses the tail to the
// fp32 spacing of numbers next to 1;  Q(5, x) = e^-x (1 + x + x^2/2 + x^3/6 + x^4/24),  P = 1 - Q (series below x = 1),
// density x^4 e^-x / 24.  fp32, 5 iterations: within 1.5e-5 of scipy's gammaincinv / gammainccinv for |z| <= 7.5 (emulated in
// NumPy float32 when the kernel was written).
__device__ __forceinline__ float gamma5_quantile_of_normal(float z) {
    const float c = 1.0f - 1.0f / 45.0f + z * 0.14907119849998599f;           // 1 - 1/(9 nu) + z / (3 sqrt(nu))
    const bool upper = z > 0.f;
    const float p = 0.5f * erfcf(fabsf(z) * 0.70710678118654752f);            // the nearer tail's probability, <= 1/2
    // far lower tail: P(5, x) ~ x^5 / 120 (Wilson-Hilferty is off by 60 % at z = -6)
    float x = (z < -2.5f) ? powf(120.0f * p, 0.2f) : fmaxf(5.0f * c * c * c, 0.02f);
#pragma unroll
    for (int it = 0; it < 5; ++it) {
        const float e = expf(-x);
        const float x4 = x * x * x * x;
        const float pdf = fmaxf(e * x4 * (1.0f / 24.0f), 1e-37f);
        float step;
        if (upper) {
            const float q = e * (1.0f + x * (1.0f + x * (0.5f + x * (1.0f / 6.0f + x * (1.0f / 24.0f)))));
            step = q * logf(q / p) / pdf;                                        // Newton on log Q (Q ~ e^-x: linear Newton crawls)
        } else {
            const float cdf = (x < 1.0f)
                ? e * x4 * x * (1.0f / 120.0f) * (1.0f + x * (1.0f / 6.0f + x * (1.0f / 42.0f + x * (1.0f / 336.0f + x * (1.0f / 3024.0f +
                                                                                                       x * (1.0f / 30240.0f))))))
                : 1.0f - e * (1.0f + x * (1.0f + x * (0.5f + x * (1.0f / 6.0f + x * (1.0f / 24.0f)))));
            step = -(cdf - p) / pdf;
        }
        x = fmaxf(x + fminf(fmaxf(step, -0.5f * x), 4.0f), 1e-4f);               // damped: never below half, never a wild jump
    }
    return x;
}

// One block per (frame, row) of the 400 x 400 window: amp = |s| sqrt(tau) (KD:519-520), the row's sum of amp^2 (-> mean clutter
// power, RG:192) in a fixed order, and on request the three fields themselves (statistics tests).
__global__ __launch_bounds__(256) void clutter_combine_kernel(const float2* __restrict__ g, const float2* __restrict__ speckle,
                                                              float* __restrict__ amp, double* __restrict__ row_power,
                                                              float* __restrict__ fields) {
    const int f = blockIdx.x / FR, y = blockIdx.x % FR;
    double s2 = 0.0;
    for (int x = threadIdx.x; x < FR; x += 256) {
        const int64_t src = (int64_t)f * FN * FN + (int64_t)y * FN + x;
        const float tau = gamma5_quantile_of_normal(g[src].x);
        const float2 s = speckle[src];
        const float a = sqrtf((s.x * s.x + s.y * s.y) * tau);
        const int64_t dst = ((int64_t)f * FR + y) * FR + x;
        amp[dst] = a;
        s2 += (double)a * a;
        if (fields) {
            float* fp = fields + (int64_t)f * 4 * FR * FR + (int64_t)y * FR + x;
            fp[0] = tau; fp[(int64_t)FR * FR] = s.x; fp[2ll * FR * FR] = s.y; fp[3ll * FR * FR] = a;
        }
    }
    __shared__ double sh[4];
    double v[1] = {s2};
    block_sum_256<double, 1>(v, sh);
    if (threadIdx.x == 0) row_power[blockIdx.x] = v[0];
}

// erc[f] = mean(amp_f^2): the 400 row sums of a frame added in a fixed order (bit-reproducible frames, unlike atomics)
__global__ __launch_bounds__(64) void clutter_power_kernel(const double* __restrict__ row_power, double* __restrict__ erc) {
    const int f = blockIdx.x;
    double s = 0.0;
    for (int y = threadIdx.x; y < FR; y += 64) s += row_power[(int64_t)f * FR + y];
    s = wave_sum(s);
    if (threadIdx.x == 0) erc[f] = s / ((double)FR * FR);
}

// out[f][0][y][x] = the (y0 + y, x0 + x) pixel of the frame after its targets (RG:207-209: placed in order, each one seeing the
// frame the previous ones left: bg += (template > bg) * template, RG:156-158); label = any kgauss > its threshold (RG:155,166).
// targets[f][t] = (lx, ly, kernel_wr, kernel_hr, a, b, c, thr)
__global__ __launch_bounds__(256) void clutter_targets_crop_kernel(const float* __restrict__ amp, const double* __restrict__ erc,
                                                                   const float* __restrict__ targets, const float* __restrict__ snr_db,
                                                                   int n_targets, float* __restrict__ out, float* __restrict__ label,
                                                                   int frames, int H, int W, int y0, int x0) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= (int64_t)frames * H * W) return;
    const int f = (int)(i / ((int64_t)H * W)), p = (int)(i % ((int64_t)H * W));
    const int y = y0 + p / W, x = x0 + p % W;
    float v = amp[((int64_t)f * FR + y) * FR + x], lab = 0.f;
    if (n_targets > 0) {
        const float peak = (float)sqrt(pow(10.0, (double)snr_db[f] * 0.1) * erc[f]);                        // RG:89
        for (int t = 0; t < n_targets; ++t) {
            const float* tg = targets + ((int64_t)f * n_targets + t) * 8;
            const int wr = (int)tg[2], hr = (int)tg[3];
            const int kx = x - (int)tg[0] - wr, ky = y - (int)tg[1] - hr;                                    // RG:43-45,77-85
            if (kx < -wr || kx > wr || ky < -hr || ky > hr) continue;
            const float kg = expf(-(tg[4] * (float)(kx * kx) + 2.0f * tg[5] * (float)(kx * ky) + tg[6] * (float)(ky * ky)));
            const float tm = kg * peak;
            if (tm > v) v += tm;
            if (kg > tg[7]) lab = 1.f;
        }
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

int onet_clutter_fft_size(void) { return FN; }
int onet_clutter_frame_size(void) { return FR; }

int64_t onet_clutter_ws_bytes(int frames) {
    // two complex fields [frames][512][512] + amplitude [frames][400][400] + row sums [frames][400] + power [frames]
    return (int64_t)frames * FN * FN * (8 + 8) + (int64_t)frames * FR * FR * 4 + (int64_t)frames * (FR + 1) * 8 + 256;
}

int onet_clutter_generate(float* out, float* label, float* fields, const float* targets, const float* snr_db, int n_targets, int frames,
                          int H, int W, uint64_t seed, const float* filt_texture, const float* filt_speckle, void* ws, int64_t ws_bytes,
                          void* stream) {
    ONET_REQUIRE(out && ws && filt_texture && filt_speckle && (n_targets == 0 || (targets && snr_db)), "clutter_generate: null pointer");
    ONET_REQUIRE(frames > 0 && H > 0 && W > 0 && H <= FR && W <= FR && n_targets >= 0, "clutter_generate: bad shape (H, W <= %d)", FR);
    ONET_REQUIRE(ws_bytes >= onet_clutter_ws_bytes(frames), "clutter_generate: workspace too small");
    ONET_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15) == 0 && (reinterpret_cast<uintptr_t>(filt_texture) & 7) == 0 &&
                 (reinterpret_cast<uintptr_t>(filt_speckle) & 7) == 0, "clutter_generate: workspace / filters must be aligned");
    hipStream_t st = as_stream(stream);
    const int64_t n = (int64_t)FN * FN;
    char* p = static_cast<char*>(ws);
    float2* gfield = reinterpret_cast<float2*>(p);  p += frames * n * 8;
    float2* sfield = reinterpret_cast<float2*>(p);  p += frames * n * 8;
    double* row_power = reinterpret_cast<double*>(p);  p += (int64_t)frames * FR * 8;
    double* erc = reinterpret_cast<double*>(p);        p += (int64_t)frames * 8;
    float* amp = reinterpret_cast<float*>(p);
    int rc;
    // white fields (independent Philox streams 0 / 1), coloured in the frequency domain
    const int64_t quads = (n + 3) / 4 * frames;
    hipLaunchKernelGGL(clutter_white_kernel, dim3((unsigned)cdiv(quads, 256)), dim3(256), 0, st, gfield, n, frames, (uint32_t)seed, (uint32_t)(seed >> 32), 0u);
    hipLaunchKernelGGL(clutter_white_kernel, dim3((unsigned)cdiv(quads, 256)), dim3(256), 0, st, sfield, n, frames, (uint32_t)seed, (uint32_t)(seed >> 32), 1u);
    if ((rc = check_launch("clutter_white_kernel"))) return rc;
    float2* fields2[2] = {gfield, sfield};
    const float2* filts[2] = {reinterpret_cast<const float2*>(filt_texture), reinterpret_cast<const float2*>(filt_speckle)};
    for (int k = 0; k < 2; ++k) {
        if ((rc = fft2(fields2[k], frames, 0, st))) return rc;
        hipLaunchKernelGGL(clutter_filter_kernel, dim3((unsigned)cdiv(n * frames, 256)), dim3(256), 0, st, fields2[k], filts[k], n, frames);
        if ((rc = check_launch("clutter_filter_kernel"))) return rc;
        if ((rc = fft2(fields2[k], frames, 1, st))) return rc;
    }
    hipLaunchKernelGGL(clutter_combine_kernel, dim3((unsigned)(frames * FR)), dim3(256), 0, st, (const float2*)gfield, (const float2*)sfield, amp,
                       row_power, fields);
    hipLaunchKernelGGL(clutter_power_kernel, dim3((unsigned)frames), dim3(64), 0, st, (const double*)row_power, erc);
    if ((rc = check_launch("clutter_combine"))) return rc;
    // torchvision CenterCrop (RG:302): int(round((400 - H) / 2))
    const int y0 = (int)lrint((FR - H) / 2.0), x0 = (int)lrint((FR - W) / 2.0);
    hipLaunchKernelGGL(clutter_targets_crop_kernel, dim3((unsigned)cdiv((int64_t)frames * H * W, 256)), dim3(256), 0, st, (const float*)amp,
                       (const double*)erc, targets, snr_db, n_targets, out, label, frames, H, W, y0, x0);
    return check_launch("clutter_targets_crop_kernel");
}

}  // extern "C"
