/* Plain-C restatement of the Onet hot-path arithmetic.  TEST INFRASTRUCTURE ONLY (oracle/):
 * an independent, from-first-principles check of the formulas the PyTorch-CPU oracle
 * (onet_oracle.py) and the HIP kernels implement.  Scalar loops, fp32 storage, fp64 accumulation.
 * Each function cites the reference line it follows (OV = source_code/Onet_vanilla_20240606.py).
 * Never linked into the product library. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* OV:47,51  nn.Conv2d(k=3, padding=1, bias=False): x[B][Ci][H][W], w[Co][Ci][3][3] -> z[B][Co][H][W] */
void ref_conv3x3(const float* x, const float* w, float* z, int B, int Ci, int Co, int H, int W) {
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Co; ++co)
            for (int y = 0; y < H; ++y)
                for (int xx = 0; xx < W; ++xx) {
                    double s = 0.0;
                    for (int ci = 0; ci < Ci; ++ci)
                        for (int ky = 0; ky < 3; ++ky)
                            for (int kx = 0; kx < 3; ++kx) {
                                int iy = y + ky - 1, ix = xx + kx - 1;
                                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                                s += (double)w[((co * Ci + ci) * 3 + ky) * 3 + kx] *
                                     (double)x[((size_t)(b * Ci + ci) * H + iy) * W + ix];
                            }
                    z[((size_t)(b * Co + co) * H + y) * W + xx] = (float)s;
                }
}

/* OV:48-49  BatchNorm2d(train) + ReLU: batch mean / biased var normalise; running stats with momentum
 * 0.1 and UNBIASED variance (verified against the reference). */
void ref_bn_relu_train(const float* z, const float* gamma, const float* beta, float* rm, float* rv, float* a,
                       int B, int C, int HW, float momentum, float eps) {
    double n = (double)B * HW;
    for (int c = 0; c < C; ++c) {
        double s = 0.0, ss = 0.0;
        for (int b = 0; b < B; ++b)
            for (int p = 0; p < HW; ++p) s += z[((size_t)b * C + c) * HW + p];
        double mean = s / n;
        for (int b = 0; b < B; ++b)
            for (int p = 0; p < HW; ++p) {
                double d = z[((size_t)b * C + c) * HW + p] - mean;
                ss += d * d;
            }
        double var = ss / n, inv = 1.0 / sqrt(var + eps);
        for (int b = 0; b < B; ++b)
            for (int p = 0; p < HW; ++p) {
                double v = (z[((size_t)b * C + c) * HW + p] - mean) * inv * gamma[c] + beta[c];
                a[((size_t)b * C + c) * HW + p] = (float)(v > 0 ? v : 0);
            }
        rm[c] = (float)((1 - momentum) * rm[c] + momentum * mean);
        rv[c] = (float)((1 - momentum) * rv[c] + momentum * var * n / (n - 1));
    }
}

/* OV:67  MaxPool2d(2), floor mode */
void ref_maxpool2(const float* x, float* y, int BC, int H, int W) {
    int Ho = H / 2, Wo = W / 2;
    for (int c = 0; c < BC; ++c)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                const float* p = x + ((size_t)c * H + 2 * oy) * W + 2 * ox;
                float m = p[0];
                if (p[1] > m) m = p[1];
                if (p[W] > m) m = p[W];
                if (p[W + 1] > m) m = p[W + 1];
                y[((size_t)c * Ho + oy) * Wo + ox] = m;
            }
}

/* OV:86  ConvTranspose2d(Ci, Co, k=2, s=2) with bias: w[Ci][Co][2][2] */
void ref_convT2x2(const float* x, const float* w, const float* bias, float* y, int B, int Ci, int Co, int h, int wd) {
    int Ho = 2 * h, Wo = 2 * wd;
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Co; ++co)
            for (int oy = 0; oy < Ho; ++oy)
                for (int ox = 0; ox < Wo; ++ox) {
                    double s = bias[co];
                    for (int ci = 0; ci < Ci; ++ci)
                        s += (double)x[((size_t)(b * Ci + ci) * h + oy / 2) * wd + ox / 2] *
                             (double)w[((ci * Co + co) * 2 + (oy & 1)) * 2 + (ox & 1)];
                    y[((size_t)(b * Co + co) * Ho + oy) * Wo + ox] = (float)s;
                }
}

/* OV:237-251  effective log1pexp incl. the in-place re-capture quirk (x <= -37 -> ln 2) */
float ref_log1pexp(float x) {
    if (x <= -37.f) return logf(1.f + expf(expf(x)));
    if (x <= 18.f) return logf(1.f + expf(x));
    if (x < 33.3f) return x + expf(-x);
    return x;
}

/* OV:176-189 head + softmax and OV:221-267 loss, for L/H [B][C][HW]; returns the loss, writes S[B][2][HW] */
double ref_head_loss(const float* Lt, const float* Ht, const float* Ld, const float* Hd, float* S, int B, int C,
                     int HW) {
    double acc = 0.0;
    for (int b = 0; b < B; ++b)
        for (int p = 0; p < HW; ++p) {
            double vt = 0, vd = 0, slt = 0, sld = 0;
            for (int c = 0; c < C; ++c) {
                size_t i = ((size_t)b * C + c) * HW + p;
                vt += (double)Lt[i] * Ht[i];
                vd += (double)Ld[i] * Hd[i];
                slt += Lt[i];
                sld += Ld[i];
            }
            double m = vt > vd ? vt : vd, et = exp(vt - m), ed = exp(vd - m);
            float st = (float)(et / (et + ed)), sd = (float)(ed / (et + ed));
            S[((size_t)b * 2 + 0) * HW + p] = st;
            S[((size_t)b * 2 + 1) * HW + p] = sd;
            /* loss = -(jsd_top + jsd_dwn)/2, jsd(L,Si,Sp) = -mean f(-Si*sL) - mean f(Sp*sL) */
            acc += ref_log1pexp(-(float)(slt * st)) + ref_log1pexp((float)(slt * sd)) +
                   ref_log1pexp(-(float)(sld * sd)) + ref_log1pexp((float)(sld * st));
        }
    return acc / (2.0 * B * HW);
}
