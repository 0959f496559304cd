// "Next" row f-1: torch.optim.Adam (TS:181-182, TZ:89) as ONE fused launch over
// the flat parameter / gradient / moment buffers (the same flat gradient buffer the RCCL
// all-reduce uses).  HBM-bound: reads p,g,m,v and writes p,m,v once, float4 vectorised.
#include "common.hpp"

using namespace onet;

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float lr_c1, float b1, float b2,
                                      float eps, float wd, float rsb2, float gs) {
    g *= gs;
    if (wd != 0.f) g = fmaf(wd, p, g);
    m = fmaf(b1, m, (1.f - b1) * g);          // exp_avg.lerp_(grad, 1-beta1)
    v = fmaf(b2, v, (1.f - b2) * g * g);      // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1-beta2)
    const float denom = sqrtf(v) * rsb2 + eps;   // sqrt(v)/sqrt(bias_c2) + eps
    p -= lr_c1 * (m / denom);                 // step_size = lr / bias_c1
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   float lr_c1, float b1, float b2, float eps, float wd, float rsb2,
                                                   float gs) {
    const int64_t n4 = n >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        adam1(pp.x, gg.x, mm.x, vv.x, lr_c1, b1, b2, eps, wd, rsb2, gs);
        adam1(pp.y, gg.y, mm.y, vv.y, lr_c1, b1, b2, eps, wd, rsb2, gs);
        adam1(pp.z, gg.z, mm.z, vv.z, lr_c1, b1, b2, eps, wd, rsb2, gs);
        adam1(pp.w, gg.w, mm.w, vv.w, lr_c1, b1, b2, eps, wd, rsb2, gs);
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    // tail
    const int64_t t = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < n) adam1(p[t], g[t], m[t], v[t], lr_c1, b1, b2, eps, wd, rsb2, gs);
}

extern "C" int onet_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, float grad_scale,
                              void* stream) {
    ONET_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adam_step: bad args");
    ONET_REQUIRE((reinterpret_cast<uintptr_t>(p) & 15) == 0 && (reinterpret_cast<uintptr_t>(g) & 15) == 0 &&
                 (reinterpret_cast<uintptr_t>(m) & 15) == 0 && (reinterpret_cast<uintptr_t>(v) & 15) == 0,
                 "adam_step: buffers must be 16-byte aligned");
    const double c1 = 1.0 - pow((double)beta1, (double)step);
    const double c2 = 1.0 - pow((double)beta2, (double)step);
    const float lr_c1 = (float)((double)lr / c1);
    const float rsb2 = (float)(1.0 / sqrt(c2));
    int64_t blocks = ((n >> 2) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), p, g, m, v, n, lr_c1, beta1,
                       beta2, eps, weight_decay, rsb2, grad_scale);
    return check_launch("adam_kernel");
}
