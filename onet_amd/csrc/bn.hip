// K2/K3: BatchNorm2d (train: batch statistics; eval: running statistics) fused with ReLU.
// Replaces nn.BatchNorm2d + nn.ReLU(inplace=True) at OV:48-49,52-53 and their autograd.
// All kernels are HBM-bound streaming passes: float4 loads, wave-shuffle reductions,
// deterministic two-stage per-channel reductions (partials -> fp64 finalize), no atomics.
#include "common.hpp"

using namespace onet;

// partial (sum, sumsq) over one image plane chunk: part[p][c][2], p = b*chunks + chunk
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ z, int64_t z_bs,
                                                               float* __restrict__ part, int C, int HW,
                                                               int chunks, int chunk_len) {
    __shared__ float red[8];
    const int c = blockIdx.x % C;
    const int p = blockIdx.x / C;
    const int b = p / chunks, ch = p % chunks;
    const float* src = z + (int64_t)b * z_bs + (int64_t)c * HW;
    const int beg = ch * chunk_len;
    const int end = min(beg + chunk_len, HW);
    float v[2] = {0.f, 0.f};
    if (((HW & 3) == 0) && ((z_bs & 3) == 0) && ((chunk_len & 3) == 0)) {
        for (int i = beg + threadIdx.x * 4; i < end; i += 1024) {
            const float4 q = *reinterpret_cast<const float4*>(src + i);
            v[0] += (q.x + q.y) + (q.z + q.w);
            v[1] += (q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w);
        }
    } else {
        for (int i = beg + threadIdx.x; i < end; i += 256) {
            const float q = src[i];
            v[0] += q;
            v[1] += q * q;
        }
    }
    block_sum_256<float, 2>(v, red);
    if (threadIdx.x == 0) {
        part[((int64_t)p * C + c) * 2 + 0] = v[0];
        part[((int64_t)p * C + c) * 2 + 1] = v[1];
    }
}

// one wave per channel: fp64 reduction of the partials, then the BN coefficients
__global__ __launch_bounds__(64) void bn_finalize_kernel(const float* __restrict__ part, int nparts, double count,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* running_mean,
                                                         float* running_var, float momentum, float eps,
                                                         float* __restrict__ save, int C) {
    const int c = blockIdx.x;
    double s = 0.0, ss = 0.0;
    for (int p = threadIdx.x; p < nparts; p += 64) {
        s += (double)part[((int64_t)p * C + c) * 2 + 0];
        ss += (double)part[((int64_t)p * C + c) * 2 + 1];
    }
    s = wave_sum(s);
    ss = wave_sum(ss);
    if (threadIdx.x == 0) {
        const double mean = s / count;
        double var = ss / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
        const float scale = g * invstd;
        save[c] = (float)mean;
        save[C + c] = invstd;
        save[2 * C + c] = scale;
        save[3 * C + c] = bt - (float)mean * scale;
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        if (running_var) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
        }
    }
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, float* save, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = 1.f / sqrtf(rv[c] + eps);
    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    const float scale = g * invstd;
    save[c] = rm[c];
    save[C + c] = invstd;
    save[2 * C + c] = scale;
    save[3 * C + c] = bt - rm[c] * scale;
}

// a = max(0, z*scale + shift); one block per 4096-element chunk of a (b, c) plane
__global__ __launch_bounds__(256) void bn_relu_apply_kernel(const float* __restrict__ z, int64_t z_bs,
                                                            float* __restrict__ a, int64_t a_bs,
                                                            const float* __restrict__ save, int C, int HW,
                                                            int chunks) {
    const int plane = blockIdx.x / chunks, ch = blockIdx.x % chunks;
    const int b = plane / C, c = plane % C;
    const float sc = save[2 * C + c], sh = save[3 * C + c];
    const float* src = z + (int64_t)b * z_bs + (int64_t)c * HW;
    float* dst = a + (int64_t)b * a_bs + (int64_t)c * HW;
    const int beg = ch * 4096, end = min(beg + 4096, HW);
    if (((HW & 3) == 0) && ((z_bs & 3) == 0) && ((a_bs & 3) == 0)) {
        for (int i = beg + threadIdx.x * 4; i < end; i += 1024) {
            float4 q = *reinterpret_cast<const float4*>(src + i);
            q.x = fmaxf(fmaf(q.x, sc, sh), 0.f);
            q.y = fmaxf(fmaf(q.y, sc, sh), 0.f);
            q.z = fmaxf(fmaf(q.z, sc, sh), 0.f);
            q.w = fmaxf(fmaf(q.w, sc, sh), 0.f);
            *reinterpret_cast<float4*>(dst + i) = q;
        }
    } else {
        for (int i = beg + threadIdx.x; i < end; i += 256) dst[i] = fmaxf(fmaf(src[i], sc, sh), 0.f);
    }
}

// backward pass 1: part2[p][c] = (sum dy, sum dy*xhat), dy = da * [z*scale+shift > 0]
__global__ __launch_bounds__(256) void bn_relu_bwd_reduce_kernel(const float* __restrict__ da, int64_t da_bs,
                                                                 const float* __restrict__ z, int64_t z_bs,
                                                                 const float* __restrict__ save,
                                                                 float* __restrict__ part2, int C, int HW,
                                                                 int chunks, int chunk_len) {
    __shared__ float red[8];
    const int c = blockIdx.x % C;
    const int p = blockIdx.x / C;
    const int b = p / chunks, ch = p % chunks;
    const float mean = save[c], invstd = save[C + c], sc = save[2 * C + c], sh = save[3 * C + c];
    const float* zs = z + (int64_t)b * z_bs + (int64_t)c * HW;
    const float* ds = da + (int64_t)b * da_bs + (int64_t)c * HW;
    const int beg = ch * chunk_len, end = min(beg + chunk_len, HW);
    float v[2] = {0.f, 0.f};
    if (((HW & 3) == 0) && ((z_bs & 3) == 0) && ((da_bs & 3) == 0) && ((chunk_len & 3) == 0)) {
        for (int i = beg + threadIdx.x * 4; i < end; i += 1024) {
            const float4 q = *reinterpret_cast<const float4*>(zs + i);
            const float4 g = *reinterpret_cast<const float4*>(ds + i);
            const float zz[4] = {q.x, q.y, q.z, q.w}, gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dy = fmaf(zz[k], sc, sh) > 0.f ? gg[k] : 0.f;
                v[0] += dy;
                v[1] += dy * ((zz[k] - mean) * invstd);
            }
        }
    } else {
        for (int i = beg + threadIdx.x; i < end; i += 256) {
            const float dy = fmaf(zs[i], sc, sh) > 0.f ? ds[i] : 0.f;
            v[0] += dy;
            v[1] += dy * ((zs[i] - mean) * invstd);
        }
    }
    block_sum_256<float, 2>(v, red);
    if (threadIdx.x == 0) {
        part2[((int64_t)p * C + c) * 2 + 0] = v[0];
        part2[((int64_t)p * C + c) * 2 + 1] = v[1];
    }
}

__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(const float* __restrict__ part2, int nparts,
                                                             double count, float* dgamma, float* dbeta,
                                                             float* coef, int accumulate, int C) {
    const int c = blockIdx.x;
    double s = 0.0, sx = 0.0;
    for (int p = threadIdx.x; p < nparts; p += 64) {
        s += (double)part2[((int64_t)p * C + c) * 2 + 0];
        sx += (double)part2[((int64_t)p * C + c) * 2 + 1];
    }
    s = wave_sum(s);
    sx = wave_sum(sx);
    if (threadIdx.x == 0) {
        if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)s : (float)s;
        if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)sx : (float)sx;
        if (coef) {
            coef[c] = (float)(s / count);
            coef[C + c] = (float)(sx / count);
        }
    }
}

// backward pass 2: dz = scale * (dy - c1 - xhat*c2)  (train)  |  dz = scale*dy (eval, coef == NULL)
__global__ __launch_bounds__(256) void bn_relu_bwd_apply_kernel(const float* __restrict__ da, int64_t da_bs,
                                                                const float* __restrict__ z, int64_t z_bs,
                                                                const float* __restrict__ save,
                                                                const float* __restrict__ coef,
                                                                float* __restrict__ dz, int64_t dz_bs, int C,
                                                                int HW, int chunks) {
    const int plane = blockIdx.x / chunks, ch = blockIdx.x % chunks;
    const int b = plane / C, c = plane % C;
    const float mean = save[c], invstd = save[C + c], sc = save[2 * C + c], sh = save[3 * C + c];
    const float c1 = coef ? coef[c] : 0.f, c2 = coef ? coef[C + c] : 0.f;
    const float* zs = z + (int64_t)b * z_bs + (int64_t)c * HW;
    const float* ds = da + (int64_t)b * da_bs + (int64_t)c * HW;
    float* out = dz + (int64_t)b * dz_bs + (int64_t)c * HW;
    const int beg = ch * 4096, end = min(beg + 4096, HW);
    if (((HW & 3) == 0) && ((z_bs & 3) == 0) && ((da_bs & 3) == 0) && ((dz_bs & 3) == 0)) {
        for (int i = beg + threadIdx.x * 4; i < end; i += 1024) {
            const float4 q = *reinterpret_cast<const float4*>(zs + i);
            const float4 g = *reinterpret_cast<const float4*>(ds + i);
            const float zz[4] = {q.x, q.y, q.z, q.w}, gg[4] = {g.x, g.y, g.z, g.w};
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dy = fmaf(zz[k], sc, sh) > 0.f ? gg[k] : 0.f;
                o[k] = sc * (dy - c1 - ((zz[k] - mean) * invstd) * c2);
            }
            *reinterpret_cast<float4*>(out + i) = make_float4(o[0], o[1], o[2], o[3]);
        }
    } else {
        for (int i = beg + threadIdx.x; i < end; i += 256) {
            const float dy = fmaf(zs[i], sc, sh) > 0.f ? ds[i] : 0.f;
            out[i] = sc * (dy - c1 - ((zs[i] - mean) * invstd) * c2);
        }
    }
}

// nparts must be B * chunks with chunks = ceil(HW / chunk_len); we derive chunk_len from nparts
static bool split_plan(int nparts, int B, int HW, int& chunks, int& chunk_len) {
    if (nparts <= 0 || nparts % B) return false;
    chunks = nparts / B;
    chunk_len = (HW + chunks - 1) / chunks;
    chunk_len = (chunk_len + 3) & ~3;
    return (int64_t)chunks * chunk_len >= HW;
}

extern "C" {

int onet_bn_stats_partial(const float* z, int64_t z_bs, float* part, int nparts, int B, int C, int HW,
                          void* stream) {
    ONET_REQUIRE(z && part && B > 0 && C > 0 && HW > 0, "bn_stats_partial: bad args");
    int chunks, chunk_len;
    ONET_REQUIRE(split_plan(nparts, B, HW, chunks, chunk_len), "bn_stats_partial: nparts=%d must be a multiple of B=%d", nparts, B);
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3((unsigned)((int64_t)nparts * C)), dim3(256), 0,
                       as_stream(stream), z, z_bs, part, C, HW, chunks, chunk_len);
    return check_launch("bn_stats_partial_kernel");
}

int onet_bn_finalize(const float* part, int nparts, int64_t count, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps, float* save, int C,
                     void* stream) {
    ONET_REQUIRE(part && save && nparts > 0 && count > 0 && C > 0, "bn_finalize: bad args");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64), 0, as_stream(stream), part, nparts, (double)count,
                       gamma, beta, running_mean, running_var, momentum, eps, save, C);
    return check_launch("bn_finalize_kernel");
}

int onet_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float eps, float* save, int C, void* stream) {
    ONET_REQUIRE(running_mean && running_var && save && C > 0, "bn_eval_coeffs: bad args");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(cdiv(C, 256)), dim3(256), 0, as_stream(stream), gamma, beta,
                       running_mean, running_var, eps, save, C);
    return check_launch("bn_eval_coeffs_kernel");
}

int onet_bn_relu_apply(const float* z, int64_t z_bs, float* a, int64_t a_bs, const float* save, int B, int C,
                       int HW, void* stream) {
    ONET_REQUIRE(z && a && save && B > 0 && C > 0 && HW > 0, "bn_relu_apply: bad args");
    const int chunks = cdiv(HW, 4096);
    const int64_t blocks = (int64_t)B * C * chunks;
    ONET_REQUIRE(blocks < (1ll << 31), "bn_relu_apply: grid too large");
    hipLaunchKernelGGL(bn_relu_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), z, z_bs, a,
                       a_bs, save, C, HW, chunks);
    return check_launch("bn_relu_apply_kernel");
}

int onet_bn_relu_bwd_reduce(const float* da, int64_t da_bs, const float* z, int64_t z_bs, const float* save,
                            float* part2, int nparts, int B, int C, int HW, void* stream) {
    ONET_REQUIRE(da && z && save && part2 && B > 0 && C > 0 && HW > 0, "bn_relu_bwd_reduce: bad args");
    int chunks, chunk_len;
    ONET_REQUIRE(split_plan(nparts, B, HW, chunks, chunk_len), "bn_relu_bwd_reduce: nparts=%d must be a multiple of B=%d", nparts, B);
    hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel, dim3((unsigned)((int64_t)nparts * C)), dim3(256), 0,
                       as_stream(stream), da, da_bs, z, z_bs, save, part2, C, HW, chunks, chunk_len);
    return check_launch("bn_relu_bwd_reduce_kernel");
}

int onet_bn_bwd_finalize(const float* part2, int nparts, int64_t count, float* dgamma, float* dbeta, float* coef,
                         int accumulate, int C, void* stream) {
    ONET_REQUIRE(part2 && nparts > 0 && count > 0 && C > 0, "bn_bwd_finalize: bad args");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, as_stream(stream), part2, nparts,
                       (double)count, dgamma, dbeta, coef, accumulate, C);
    return check_launch("bn_bwd_finalize_kernel");
}

int onet_bn_relu_bwd_apply(const float* da, int64_t da_bs, const float* z, int64_t z_bs, const float* save,
                           const float* coef, float* dz, int64_t dz_bs, int B, int C, int HW, void* stream) {
    ONET_REQUIRE(da && z && save && dz && B > 0 && C > 0 && HW > 0, "bn_relu_bwd_apply: bad args");
    const int chunks = cdiv(HW, 4096);
    const int64_t blocks = (int64_t)B * C * chunks;
    ONET_REQUIRE(blocks < (1ll << 31), "bn_relu_bwd_apply: grid too large");
    hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), da, da_bs,
                       z, z_bs, save, coef, dz, dz_bs, C, HW, chunks);
    return check_launch("bn_relu_bwd_apply_kernel");
}

}  // extern "C"
