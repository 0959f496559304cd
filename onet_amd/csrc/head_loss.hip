// K8 head einsum('bpxy,bpxy->bxy') (OV:176,182), K9 cat + Softmax2d over 2 channels (OV:185-189),
// K10 Jensen-Shannon MI loss with the reference's order-dependent log1pexp (OV:221-267),
// K11 argmax over the 2 classes (OV:201).  HBM-bound: one thread per pixel, lanes along H*W
// (coalesced), channel loop in registers; deterministic two-stage fp64 reduction for the loss.
#include <algorithm>
#include "common.hpp"

using namespace onet;

// Effective function of Onet.log1pexp (OV:237-251) on the ORIGINAL x:
//   x <= -37 : log(1 + exp(exp(x)))  (= ln 2: step-1 outputs are re-captured by step 2)
//   x <=  18 : log(1 + exp(x))       (literally, not log1p: underflows to 0 below ~-16.6 like torch)
//   x < 33.3 : x + exp(-x)
//   else     : x
__device__ __forceinline__ float f_quirk(float x) {
    if (x <= -37.f) return logf(1.f + expf(expf(x)));
    if (x <= 18.f) return logf(1.f + expf(x));
    if (x < 33.3f) return x + expf(-x);
    return x;
}
// autograd derivative of the composed reference ops
__device__ __forceinline__ float df_quirk(float x) {
    if (x <= -37.f) {
        const float e = expf(x), ee = expf(e);
        return ee / (1.f + ee) * e;
    }
    if (x <= 18.f) {
        const float e = expf(x);
        return e / (1.f + e);
    }
    if (x < 33.3f) return 1.f - expf(-x);
    return 1.f;
}

__global__ void log1pexp_kernel(float* x, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        x[i] = f_quirk(x[i]);
}

__global__ __launch_bounds__(256) void head_softmax_fwd_kernel(const float* __restrict__ Lt, int64_t Lt_bs,
                                                               const float* __restrict__ Ht, int64_t Ht_bs,
                                                               const float* __restrict__ Ld, int64_t Ld_bs,
                                                               const float* __restrict__ Hd, int64_t Hd_bs,
                                                               float* __restrict__ Vt, float* __restrict__ Vd,
                                                               float* __restrict__ S, float* __restrict__ sLt,
                                                               float* __restrict__ sLd, const float* __restrict__ nt,
                                                               const float* __restrict__ nd, int B, int C, int HW) {
    // nt / nd (round 5, may be NULL): Ht / Hd hold the PRE-ACTIVATION z of the last Conv-BatchNorm-ReLU unit and these are its
    // coefficients [4][C] (mean, invstd, scale, shift) for the top / down statistics group: H = max(fma(z - mean, scale, shift), 0) is
    // formed here -- bit for bit what bn_relu_apply_kernel would have written -- and the activation tensor is never materialised
    const int64_t n = (int64_t)B * HW;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / HW);
        const int p = (int)(i % HW);
        const float* lt = Lt + b * Lt_bs + p;
        const float* ht = Ht + b * Ht_bs + p;
        const float* ld = Ld + b * Ld_bs + p;
        const float* hd = Hd + b * Hd_bs + p;
        float vt = 0.f, vd = 0.f, st_ = 0.f, sd_ = 0.f;   // st_/sd_: sum_c L[c] in jsd_fwd_kernel's order (bit-identical)
        for (int c = 0; c < C; ++c) {
            const float a = lt[(int64_t)c * HW], d = ld[(int64_t)c * HW];
            float h1 = ht[(int64_t)c * HW], h2 = hd[(int64_t)c * HW];
            if (nt) {
                h1 = fmaxf(fmaf(h1 - nt[c], nt[2 * C + c], nt[3 * C + c]), 0.f);
                h2 = fmaxf(fmaf(h2 - nd[c], nd[2 * C + c], nd[3 * C + c]), 0.f);
            }
            vt = fmaf(a, h1, vt);
            vd = fmaf(d, h2, vd);
            st_ += a;
            sd_ += d;
        }
        Vt[i] = vt;
        Vd[i] = vd;
        if (sLt) sLt[i] = st_;
        if (sLd) sLd[i] = sd_;
        const float m = fmaxf(vt, vd);
        const float et = expf(vt - m), ed = expf(vd - m);
        const float den = et + ed;
        S[((int64_t)b * 2 + 0) * HW + p] = et / den;
        S[((int64_t)b * 2 + 1) * HW + p] = ed / den;
    }
}

__global__ __launch_bounds__(256) void head_softmax_bwd_kernel(
    const float* __restrict__ dVt, const float* __restrict__ dVd, const float* __restrict__ dS,
    const float* __restrict__ S, const float* __restrict__ Lt, int64_t Lt_bs, const float* __restrict__ Ht,
    int64_t Ht_bs, const float* __restrict__ Ld, int64_t Ld_bs, const float* __restrict__ Hd, int64_t Hd_bs,
    float* __restrict__ dLt, float* __restrict__ dHt, float* __restrict__ dLd, float* __restrict__ dHd,
    const float* __restrict__ gsLt, const float* __restrict__ gsLd, const float* __restrict__ nt, const float* __restrict__ nd, int B,
    int C, int HW) {
    const int64_t n = (int64_t)B * HW;
    const int64_t CHW = (int64_t)C * HW;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / HW);
        const int p = (int)(i % HW);
        float gt = dVt ? dVt[i] : 0.f, gd = dVd ? dVd[i] : 0.f;
        if (dS) {
            const float st = S[((int64_t)b * 2 + 0) * HW + p], sd = S[((int64_t)b * 2 + 1) * HW + p];
            const float a = dS[((int64_t)b * 2 + 0) * HW + p], d = dS[((int64_t)b * 2 + 1) * HW + p];
            // 2-class softmax backward in its cancellation-free form: with St + Sd = 1,
            // St*(a - (a*St + d*Sd)) == St*Sd*(a - d) and the Vd component is its negative.
            // (the textbook form loses everything once St rounds to 1, which it does at init: |V| ~ 47)
            const float t = st * sd * (a - d);
            gt += t;
            gd -= t;
        }
        const float* lt = Lt + b * Lt_bs + p;
        const float* ht = Ht + b * Ht_bs + p;
        const float* ld = Ld + b * Ld_bs + p;
        const float* hd = Hd + b * Hd_bs + p;
        float* olt = dLt + b * CHW + p;
        float* oht = dHt + b * CHW + p;
        float* old_ = dLd + b * CHW + p;
        float* ohd = dHd + b * CHW + p;
        const float at = gsLt ? gsLt[i] : 0.f, ad = gsLd ? gsLd[i] : 0.f;   // d loss / d (sum_c L[c]): same for every channel
        for (int c = 0; c < C; ++c) {
            const int64_t o = (int64_t)c * HW;
            float h1 = ht[o], h2 = hd[o];
            if (nt) {                                   // (Ht / Hd are pre-activations: see head_softmax_fwd_kernel)
                h1 = fmaxf(fmaf(h1 - nt[c], nt[2 * C + c], nt[3 * C + c]), 0.f);
                h2 = fmaxf(fmaf(h2 - nd[c], nd[2 * C + c], nd[3 * C + c]), 0.f);
            }
            olt[o] = fmaf(gt, h1, at);
            oht[o] = gt * lt[o];
            old_[o] = fmaf(gd, h2, ad);
            ohd[o] = gd * ld[o];
        }
    }
}

constexpr int JSD_BLOCKS = 2048;

// one jsd term (OV:221-235):  jsd = -mean f(-Si*sL) - mean f(Sp*sL),  sL = sum_c L[c]
__global__ __launch_bounds__(256) void jsd_fwd_kernel(const float* __restrict__ L, int64_t L_bs,
                                                      const float* __restrict__ Si, int64_t Si_bs,
                                                      const float* __restrict__ Sp, int64_t Sp_bs,
                                                      float* __restrict__ sums, double* __restrict__ part, int B,
                                                      int C, int HW) {
    __shared__ double red[4];
    const int64_t n = (int64_t)B * HW;
    double acc[1] = {0.0};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / HW);
        const int p = (int)(i % HW);
        float sl;
        if (L) {
            const float* l = L + b * L_bs + p;
            sl = 0.f;
            for (int c = 0; c < C; ++c) sl += l[(int64_t)c * HW];
            sums[i] = sl;
        } else {
            sl = sums[i];                  // channel sums handed over by onet_head_softmax_fwd
        }
        const float si = Si[b * Si_bs + p], sp = Sp[b * Sp_bs + p];
        acc[0] += (double)(f_quirk(-(sl * si)) + f_quirk(sl * sp));
    }
    block_sum_256<double, 1>(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = acc[0];
}

__global__ __launch_bounds__(256) void jsd_final_kernel(const double* __restrict__ part, int nparts, double scale,
                                                        float* __restrict__ out) {
    __shared__ double red[4];
    double acc[1] = {0.0};
    for (int i = threadIdx.x; i < nparts; i += 256) acc[0] += part[i];
    block_sum_256<double, 1>(acc, red);
    if (threadIdx.x == 0) out[0] = (float)(acc[0] * scale);
}

// gL = dJ/dL (same for every channel), dSi, dSp; g = upstream gradient of the jsd scalar
__global__ __launch_bounds__(256) void jsd_bwd_kernel(const float* __restrict__ gscale,
                                                      const float* __restrict__ sums,
                                                      const float* __restrict__ Si, int64_t Si_bs,
                                                      const float* __restrict__ Sp, int64_t Sp_bs,
                                                      float* __restrict__ gL, float* __restrict__ dSi,
                                                      float* __restrict__ dSp, int B, int HW) {
    const int64_t n = (int64_t)B * HW;
    const float g = -gscale[0] * (float)(1.0 / (double)n);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / HW);
        const int p = (int)(i % HW);
        const float sl = sums[i];
        const float si = Si[b * Si_bs + p], sp = Sp[b * Sp_bs + p];
        const float d1 = df_quirk(-(sl * si));
        const float d2 = df_quirk(sl * sp);
        gL[i] = g * (-si * d1 + sp * d2);
        dSi[i] = g * (-sl * d1);
        dSp[i] = g * (sl * d2);
    }
}

__global__ void log1pexp_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ o,
                                    int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        o[i] = g[i] * df_quirk(x[i]);
}

__global__ void argmax2_kernel(const float* __restrict__ S, int64_t* __restrict__ Y, int B, int HW) {
    const int64_t n = (int64_t)B * HW;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / HW);
        const int p = (int)(i % HW);
        Y[i] = S[((int64_t)b * 2 + 1) * HW + p] > S[((int64_t)b * 2 + 0) * HW + p] ? 1 : 0;   // ties -> 0
    }
}

static inline unsigned grid_px(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    return (unsigned)(b < 1 ? 1 : b);
}

extern "C" {

int onet_head_softmax_fwd(const float* Lt, int64_t Lt_bs, const float* Ht, int64_t Ht_bs, const float* Ld, int64_t Ld_bs, const float* Hd,
                          int64_t Hd_bs, float* Vt, float* Vd, float* S, float* sLt, float* sLd, const float* h_save_t, const float* h_save_d,
                          int B, int C, int HW, void* stream) {
    ONET_REQUIRE(Lt && Ht && Ld && Hd && Vt && Vd && S && (!sLt == !sLd) && (!h_save_t == !h_save_d) && B > 0 && C > 0 && HW > 0,
                 "head_softmax_fwd: bad args");
    hipLaunchKernelGGL(head_softmax_fwd_kernel, dim3(grid_px((int64_t)B * HW)), dim3(256), 0, as_stream(stream), Lt,
                       Lt_bs, Ht, Ht_bs, Ld, Ld_bs, Hd, Hd_bs, Vt, Vd, S, sLt, sLd, h_save_t, h_save_d, B, C, HW);
    return check_launch("head_softmax_fwd_kernel");
}

int onet_head_softmax_bwd(const float* dVt, const float* dVd, const float* dS, const float* gsLt, const float* gsLd, const float* S,
                          const float* Lt, int64_t Lt_bs, const float* Ht, int64_t Ht_bs, const float* Ld, int64_t Ld_bs, const float* Hd,
                          int64_t Hd_bs, float* dLt, float* dHt, float* dLd, float* dHd, const float* h_save_t, const float* h_save_d, int B,
                          int C, int HW, void* stream) {
    ONET_REQUIRE(S && Lt && Ht && Ld && Hd && dLt && dHt && dLd && dHd && (!h_save_t == !h_save_d) && B > 0 && C > 0 && HW > 0,
                 "head_softmax_bwd: bad args");
    hipLaunchKernelGGL(head_softmax_bwd_kernel, dim3(grid_px((int64_t)B * HW)), dim3(256), 0, as_stream(stream), dVt,
                       dVd, dS, S, Lt, Lt_bs, Ht, Ht_bs, Ld, Ld_bs, Hd, Hd_bs, dLt, dHt, dLd, dHd, gsLt, gsLd, h_save_t, h_save_d, B, C, HW);
    return check_launch("head_softmax_bwd_kernel");
}

int onet_jsd_nparts(void) { return JSD_BLOCKS; }

int onet_jsd_fwd(const float* L, int64_t L_bs, const float* Si, int64_t Si_bs, const float* Sp, int64_t Sp_bs,
                 float* sums, double* part, float* jsd, int B, int C, int HW, void* stream) {
    ONET_REQUIRE(Si && Sp && sums && part && jsd && B > 0 && C > 0 && HW > 0, "jsd_fwd: bad args");
    const int64_t n = (int64_t)B * HW;
    int blocks = (int)std::min<int64_t>((n + 255) / 256, JSD_BLOCKS);
    hipLaunchKernelGGL(jsd_fwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), L, L_bs, Si, Si_bs, Sp, Sp_bs,
                       sums, part, B, C, HW);
    int rc = check_launch("jsd_fwd_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(jsd_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), (const double*)part, blocks,
                       -1.0 / (double)n, jsd);
    return check_launch("jsd_final_kernel");
}

int onet_jsd_bwd(const float* gscale, const float* sums, const float* Si, int64_t Si_bs, const float* Sp,
                 int64_t Sp_bs, float* gL, float* dSi, float* dSp, int B, int HW, void* stream) {
    ONET_REQUIRE(gscale && sums && Si && Sp && gL && dSi && dSp && B > 0 && HW > 0, "jsd_bwd: bad args");
    hipLaunchKernelGGL(jsd_bwd_kernel, dim3(grid_px((int64_t)B * HW)), dim3(256), 0, as_stream(stream), gscale, sums,
                       Si, Si_bs, Sp, Sp_bs, gL, dSi, dSp, B, HW);
    return check_launch("jsd_bwd_kernel");
}

int onet_log1pexp_bwd(const float* x, const float* g, float* out, int64_t n, void* stream) {
    ONET_REQUIRE(x && g && out && n > 0, "log1pexp_bwd: bad args");
    hipLaunchKernelGGL(log1pexp_bwd_kernel, dim3(grid_px(n)), dim3(256), 0, as_stream(stream), x, g, out, n);
    return check_launch("log1pexp_bwd_kernel");
}

int onet_log1pexp_inplace(float* x, int64_t n, void* stream) {
    ONET_REQUIRE(x && n > 0, "log1pexp: bad args");
    hipLaunchKernelGGL(log1pexp_kernel, dim3(grid_px(n)), dim3(256), 0, as_stream(stream), x, n);
    return check_launch("log1pexp_kernel");
}

int onet_argmax2(const float* S, int64_t* Y, int B, int HW, void* stream) {
    ONET_REQUIRE(S && Y && B > 0 && HW > 0, "argmax2: bad args");
    hipLaunchKernelGGL(argmax2_kernel, dim3(grid_px((int64_t)B * HW)), dim3(256), 0, as_stream(stream), S, Y, B, HW);
    return check_launch("argmax2_kernel");
}

}  // extern "C"
