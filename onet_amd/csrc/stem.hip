// The stem: the first 3x3 convolution of the U-Net (OV:47 of `inc`, Cin = n_channels = 1 or 3 -> 64) together with the batch
// statistics of its BatchNorm (OV:48), as ONE streaming pass.
//
// With 1-4 input channels the layer is 18-72 FLOP per output element: it is bound by WRITING z (B x 64 x H x W fp32), not by
// arithmetic.  The MFMA kernel of conv_mfma.hip pads K = 9 Cin up to its 16-deep chunks and ran at 2.4x the write time
// (0.48 ms at B = 32, 4.0 ms at B = 256), and the statistics were a second pass over z.  Here a 256-thread block owns a 16 x 64
// pixel tile of one image: the halo tile sits in LDS, each thread keeps its 3 x 6 window per input channel in registers and
// walks over the output channels -- weights by scalar loads (the channel index is wave-uniform), 9 Cin FMAs per pixel, one
// float4 store per channel -- and takes the tile's (n, mean, M2) record per channel on the way: pivot-shifted sums per wave by
// DPP, the four waves merged through LDS.  Records are those of the F(4x4) kernel's epilogue (onet_bn_finalize_cm merges them).
#include <algorithm>
#include "common.hpp"

using namespace onet;

namespace {

struct StemArgs {
    const float* x;
    int64_t x_bs;
    const float* w;       // [Cout][Cin][3][3], the nn.Conv2d layout as it is
    float* z;
    int64_t z_bs;
    float* part;          // [Cout][B * tilesY * tilesX][3] = (n, mean, M2) per tile, or NULL
    int B, Cout, H, W, tilesX, tilesY;
    // K7 (OV:180) on load: the batch is the TWIN batch [X ; clip(1 - X + bias, 0, 1)] of which only X (twin_B images) exists in
    // memory -- image b >= twin_B is formed from image b - twin_B while the halo tile is filled (twin_B = 0: an ordinary batch)
    int twin_B;
    float twin_bias;
};

constexpr int ST_TH = 16, ST_TW = 64, ST_LR = ST_TH + 2, ST_LC = ST_TW + 4;   // LDS tile: 18 rows x 66 columns, row stride 68
constexpr int ST_MAXC = 128;

#define ONET_ST_DPP_ADD(v, ctrl, rmask) \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false))
// sum over the 64 lanes of a wave, valid in lane 63
__device__ __forceinline__ float wave_sum_hi(float v) {
    ONET_ST_DPP_ADD(v, 0x128, 0xf);   // row_ror:8
    ONET_ST_DPP_ADD(v, 0x124, 0xf);   // row_ror:4
    ONET_ST_DPP_ADD(v, 0x122, 0xf);   // row_ror:2
    ONET_ST_DPP_ADD(v, 0x121, 0xf);   // row_ror:1
    ONET_ST_DPP_ADD(v, 0x142, 0xa);   // row_bcast:15 into rows 1 and 3
    ONET_ST_DPP_ADD(v, 0x143, 0xc);   // row_bcast:31 into rows 2 and 3
    return v;
}
#undef ONET_ST_DPP_ADD

template <int CIN>
__global__ __launch_bounds__(256) void stem_conv_stats_kernel(StemArgs a) {
    __shared__ float tile[CIN][ST_LR][ST_LC];
    __shared__ float sc[4][ST_MAXC][2];
    int bid = blockIdx.x;
    const int tx = bid % a.tilesX;
    bid /= a.tilesX;
    const int ty = bid % a.tilesY;
    const int b = bid / a.tilesY;
    const int y0 = ty * ST_TH, x0 = tx * ST_TW;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int HW = a.H * a.W;
    const bool comp = a.twin_B > 0 && b >= a.twin_B;          // the complement half of a twin batch
    const float* xb = a.x + (int64_t)(comp ? b - a.twin_B : b) * a.x_bs;

    for (int i = tid; i < CIN * ST_LR * 66; i += 256) {
        const int ci = i / (ST_LR * 66), r = (i / 66) % ST_LR, c = i % 66;
        const int yy = y0 - 1 + r, xx = x0 - 1 + c;
        const bool ok = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
        float v = ok ? xb[(int64_t)ci * HW + (int64_t)yy * a.W + xx] : 0.f;
        if (comp && ok) v = fminf(fmaxf(1.f - v + a.twin_bias, 0.f), 1.f);      // complement_clip_kernel's expression (zero padding stays zero)
        tile[ci][r][c] = v;
    }
    __syncthreads();

    const int r = tid >> 4, q = tid & 15;              // 4 pixels of row y0 + r at x0 + 4 q
    float win[CIN][3][6];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 6; ++dx) win[ci][dy][dx] = tile[ci][r + dy][4 * q + dx];

    float* zb = a.z + (int64_t)b * a.z_bs + (int64_t)(y0 + r) * a.W + x0 + 4 * q;
    for (int co = 0; co < a.Cout; ++co) {
        const float* wp = a.w + (int64_t)co * CIN * 9;  // wave-uniform: scalar loads
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float wv = wp[(ci * 3 + dy) * 3 + dx];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = fmaf(wv, win[ci][dy][dx + j], acc[j]);
                }
        *reinterpret_cast<float4*>(zb + (int64_t)co * HW) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        if (a.part) {
            const float pv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, acc[0])));
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = acc[j] - pv;
                s1 += d;
                s2 = fmaf(d, d, s2);
            }
            s1 = wave_sum_hi(s1);
            s2 = wave_sum_hi(s2);
            if (lane == 63) {
                sc[wid][co][0] = fmaf(s1, 1.f / 256.f, pv);
                sc[wid][co][1] = fmaxf(fmaf(-s1 * (1.f / 256.f), s1, s2), 0.f);
            }
        }
    }
    if (a.part) {
        __syncthreads();
        if (tid < a.Cout) {
            float mw[4], qw[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                mw[w] = sc[w][tid][0];
                qw[w] = sc[w][tid][1];
            }
            const float mean = 0.25f * ((mw[0] + mw[1]) + (mw[2] + mw[3]));
            float m2 = (qw[0] + qw[1]) + (qw[2] + qw[3]);
#pragma unroll
            for (int w = 0; w < 4; ++w) m2 = fmaf(256.f * (mw[w] - mean), mw[w] - mean, m2);
            const int64_t nblk = (int64_t)a.B * a.tilesY * a.tilesX;
            const int64_t blk = ((int64_t)b * a.tilesY + ty) * a.tilesX + tx;
            float* sp = a.part + ((int64_t)tid * nblk + blk) * 3;
            sp[0] = 1024.f;
            sp[1] = mean;
            sp[2] = m2;
        }
    }
}

int stem_nparts(int B, int Cin, int Cout, int H, int W) {
    if (B <= 0 || Cin < 1 || Cin > 4 || Cout < 1 || Cout > ST_MAXC || H <= 0 || W <= 0 || (H % ST_TH) || (W % ST_TW)) return 0;
    const int64_t n = (int64_t)B * (H / ST_TH) * (W / ST_TW);
    return n < (1 << 30) ? (int)n : 0;
}

}  // namespace

extern "C" {

int onet_conv3x3_stem_nparts(int B, int Cin, int Cout, int H, int W) { return stem_nparts(B, Cin, Cout, H, W); }

int onet_conv3x3_stem_fwd_stats(const float* x, int64_t x_bs, const float* w, float* z, int64_t z_bs, float* part, int B, int Cin,
                                int Cout, int H, int W, int twin_B, float twin_bias, void* stream) {
    ONET_REQUIRE(x && w && z, "conv3x3_stem_fwd_stats: null pointer");
    ONET_REQUIRE(stem_nparts(B, Cin, Cout, H, W) > 0,
                 "conv3x3_stem_fwd_stats: needs 1 <= Cin <= 4, Cout <= %d and a map made of full 16 x 64 tiles (onet_conv3x3_stem_nparts() == 0 elsewhere)",
                 ST_MAXC);
    ONET_REQUIRE(x_bs >= (int64_t)Cin * H * W && z_bs >= (int64_t)Cout * H * W, "conv3x3_stem_fwd_stats: batch stride too small");
    ONET_REQUIRE((z_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(z) & 15) == 0, "conv3x3_stem_fwd_stats: 16-byte aligned output rows required");
    ONET_REQUIRE(twin_B == 0 || 2 * twin_B == B, "conv3x3_stem_fwd_stats: a twin batch holds twin_B = B / 2 images in memory");
    StemArgs a{x, x_bs, w, z, z_bs, part, B, Cout, H, W, W / ST_TW, H / ST_TH, twin_B, twin_bias};
    const int64_t blocks = (int64_t)B * a.tilesX * a.tilesY;
    const dim3 g((unsigned)blocks), t(256);
    switch (Cin) {
        case 1: hipLaunchKernelGGL(stem_conv_stats_kernel<1>, g, t, 0, as_stream(stream), a); break;
        case 2: hipLaunchKernelGGL(stem_conv_stats_kernel<2>, g, t, 0, as_stream(stream), a); break;
        case 3: hipLaunchKernelGGL(stem_conv_stats_kernel<3>, g, t, 0, as_stream(stream), a); break;
        default: hipLaunchKernelGGL(stem_conv_stats_kernel<4>, g, t, 0, as_stream(stream), a); break;
    }
    return check_launch("stem_conv_stats_kernel");
}

}  // extern "C"
