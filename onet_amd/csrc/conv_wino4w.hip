// Weight gradient of a 3x3 convolution by Winograd F(3x3,4x4) on the fp32 matrix cores (v_mfma_f32_32x32x2_f32):
//   dW = A'^T [ sum_tiles (G' dY G'^T) (.) (B^T d B) ] A'
// 6x6 input tiles d, 4x4 tiles of the output gradient dY in the role of the filter, interpolation points 0, +-1, +-2, inf
// (the B^T of conv_wino4.hip): 36 multiplies per 144 direct ones = 4x fewer MFMAs than the direct algorithm, 1.78x fewer
// than an F(2x2,3x3) kernel (round 1's, since removed).  The 36 positions are 36 GEMMs M_p[co][ci] = sum_tile DY_p[co][tile] *
// X_p[ci][tile] (M = co, N = ci, K = tiles); both operands are transformed in registers from channel-major LDS strips
// (lane = channel; channel strides 4 x odd floats keep the 16-lane groups of a ds_read_b128 on distinct banks).
// Block = 8 waves = 4 position groups (3x3 positions each, as in conv_wino4.hip) x 2 co halves: 64 co x 32 ci x 36
// positions, 9 accumulators per wave.  Unit = a row of 8 tiles (4 x 32 output pixels) of one image = 4 K-steps; split-K
// over units, raw 36-position slabs reduced deterministically and folded by A'^T . A' (wino4w_fold_kernel).
// Replaces the weight-gradient half of F.conv2d's autograd at OV:47,51.  Requires W % 32 == 0, H % 4 == 0, Cin % 32 == 0.
#include <algorithm>
#include <cstdlib>
#include "common.hpp"

using namespace onet;

#ifndef ONET_W4W_ASYM
#define ONET_W4W_ASYM 1
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4w __attribute__((ext_vector_type(4)));
typedef float f32x2w __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));

constexpr unsigned OOB_W4 = 0x80000000u;
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t w4_rsrc(const void* base, int64_t bytes) {
    const int n = bytes > 0x7fffffffll ? 0x7fffffff : (int)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, n, 0x00020000);
}

// half of B^T (input transform), as in conv_wino4.hip
template <int HALF>
static __device__ __forceinline__ void w_bt3(float d0, float d1, float d2, float d3, float d4, float& o0, float& o1, float& o2) {
    if constexpr (HALF == 0) {
        o0 = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
        const float a = fmaf(-4.f, d2, d4), b = fmaf(-4.f, d1, d3);
        o1 = a + b; o2 = a - b;
    } else {
        const float c = d3 - d1, e = d2 - d0;
        o0 = fmaf(2.f, e, c); o1 = fmaf(-2.f, e, c); o2 = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
    }
}
// half of G' (6x4: rows p^j / N_p for p = 0, 1, -1, 2, -2 with N = 4, -6, -6, 24, 24; row inf = (0,0,0,1)) WITHOUT the
// 1/N_p factors: they are applied once per position by the fold kernel (W4_SCALE), not per tile here
template <int HALF>
static __device__ __forceinline__ void w_g3(float d0, float d1, float d2, float d3, float& o0, float& o1, float& o2) {
    if constexpr (HALF == 0) {
        const float e = d0 + d2, o = d1 + d3;
        o0 = d0; o1 = e + o; o2 = e - o;
    } else {
        const float e = fmaf(4.f, d2, d0), o = fmaf(8.f, d3, 2.f * d1);
        o0 = e + o; o1 = e - o; o2 = d3;
    }
}

struct W4wArgs {
    const float* x;
    int64_t x_bs;
    const float* dz;
    int64_t dz_bs;
    float* slab;          // [splitK][36][Cout][Cin]
    int B, Cin, Cout, H, W, ciTiles, coTiles, splitK, tilesY, tilesX;
};

constexpr int W4_SDZ = 132;                               // floats per co: 4 rows x 32 px (+4: stride 4 x odd)
// input strip per ci: 6 patch rows of W4C<W16>::XROW floats.  W16 = 0: one image row segment, element 0 = column x0 - 1,
// 1..32 the interior, 33 = column x0 + 32.  W16 = 1 (16-px-wide maps: two IMAGES side by side, their tile rows are the
// unit's 8 tiles): 0 | 16 px of image 0 | 0 | 2 pad | 0 | 16 px of image 1 | 0 -- image 1 starts at element 20 so that its
// patches stay 16-byte aligned; every halo column is the zero padding
template <int W16>
struct W4C {
    static constexpr int XROW = W16 ? 40 : 36, SX = W16 ? 244 : 220;          // SX = 6 * XROW + 4 (4 x odd)
    static constexpr int BUF_FLOATS = 32 * SX + 64 * W4_SDZ;                   // one unit: 62 / 65 KB
    static constexpr int LDS_FLOATS = 2 * BUF_FLOATS;                          // double-buffered: one barrier per unit
};

template <int RH, int CH, int W16>
static __device__ __forceinline__ void wino4w_body(const W4wArgs a, float* lds) {
    constexpr int W4_XROW = W4C<W16>::XROW, W4_SX = W4C<W16>::SX, W4_BUF_FLOATS = W4C<W16>::BUF_FLOATS;
    int bid;
    {
        const int n = gridDim.x, q = n >> 3, r = n & 7, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tiles = a.ciTiles * a.coTiles;
    const int ks = bid / tiles, tile = bid % tiles;
    const int ci0 = (tile % a.ciTiles) * 32, co0 = (tile / a.ciTiles) * 64;
    const int nunits = W16 ? (a.B / 2) * a.tilesY : a.B * a.tilesY * a.tilesX;
    const int per = (nunits + a.splitK - 1) / a.splitK;
    const int u0 = ks * per, u1 = min(u0 + per, nunits);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid & 1;
    const int l31 = lane & 31, kh = lane >> 5;
    const int HW = a.H * a.W;

    f32x16 acc[9];
#pragma unroll
    for (int p = 0; p < 9; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    // staging roles: threads 0..191 one x row (ci, patch row) of 34 floats, threads 256..511 one dz row (co, row) of 32
    const bool x_role = tid < 192, z_role = tid >= 256;
    const int xr_c = tid / 6, xr_r = tid % 6;
    const int zr_c = (tid - 256) >> 2, zr_r = (tid - 256) & 3;
    u32x4w q[8];
    float hl = 0.f, hr = 0.f;
    auto issue = [&](int u) __attribute__((always_inline)) {
        const bool live = u < u1;
        const int uu = live ? u : 0;
        const int tx = W16 ? 0 : uu % a.tilesX, ty = W16 ? uu % a.tilesY : (uu / a.tilesX) % a.tilesY;
        const int b = W16 ? 2 * (uu / a.tilesY) : uu / (a.tilesX * a.tilesY);          // W16: images b, b + 1
        const int y0 = ty * 4, x0 = tx * 32;
        if (x_role) {
            const __amdgpu_buffer_rsrc_t xr = w4_rsrc(a.x + (int64_t)b * a.x_bs, ((W16 ? a.x_bs : 0) + (int64_t)a.Cin * HW) * 4);
            const int yy = y0 - 1 + xr_r;
            const bool rok = live && yy >= 0 && yy < a.H && ci0 + xr_c < a.Cin;
            const unsigned base = (unsigned)(((ci0 + xr_c) * HW + yy * a.W + x0) * 4);
            const unsigned img1 = (unsigned)(a.x_bs * 4);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                q[k] = __builtin_amdgcn_raw_buffer_load_b128(xr, rok ? (W16 ? base + (k >> 2) * img1 + 16 * (k & 3) : base + 16 * k) : OOB_W4, 0, 0);
            if constexpr (!W16) {
                hl = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, (rok && x0 > 0) ? base - 4 : OOB_W4, 0, 0));
                hr = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, (rok && x0 + 32 < a.W) ? base + 128 : OOB_W4, 0, 0));
            } else {
                // (never left unassigned: with hl / hr captured but untouched on this path hipcc keeps the whole closure
                // in scratch -- 368 B/lane and 2.5x the run time)
                hl = hr = 0.f;
            }
        } else if (z_role) {
            const __amdgpu_buffer_rsrc_t dr = w4_rsrc(a.dz + (int64_t)b * a.dz_bs, ((W16 ? a.dz_bs : 0) + (int64_t)a.Cout * HW) * 4);
            const int yy = y0 + zr_r;
            const bool ok = live && yy < a.H && co0 + zr_c < a.Cout;
            const unsigned base = (unsigned)(((co0 + zr_c) * HW + yy * a.W + x0) * 4);
            const unsigned img1 = (unsigned)(a.dz_bs * 4);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                q[k] = __builtin_amdgcn_raw_buffer_load_b128(dr, ok ? (W16 ? base + (k >> 2) * img1 + 16 * (k & 3) : base + 16 * k) : OOB_W4, 0, 0);
        }
    };
    auto commit = [&](float* buf) __attribute__((always_inline)) {
        float* x_lds = buf;
        float* dz_lds = buf + 32 * W4_SX;
        if (x_role) {
            // LDS row: element 0 = column x0 - 1, 1..32 the interior, 33 = column x0 + 32: the interior sits one float off
            // the 16-byte grid, so the b128 stores are re-cut from neighbouring loads (register selection, no VALU)
            float* row = x_lds + xr_c * W4_SX + xr_r * W4_XROW;
            f32x4w f[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = __builtin_bit_cast(f32x4w, q[k]);
            if constexpr (!W16) {
                f32x4w w0 = {hl, f[0][0], f[0][1], f[0][2]};
                *reinterpret_cast<f32x4w*>(row) = w0;
#pragma unroll
                for (int k = 1; k < 8; ++k) {
                    f32x4w wk = {f[k - 1][3], f[k][0], f[k][1], f[k][2]};
                    *reinterpret_cast<f32x4w*>(row + 4 * k) = wk;
                }
                f32x2w w8 = {f[7][3], hr};
                *reinterpret_cast<f32x2w*>(row + 32) = w8;
            } else {
                // image m at elements 20 m .. 20 m + 17
                const f32x4w a0 = {0.f, f[0][0], f[0][1], f[0][2]}, a1 = {f[0][3], f[1][0], f[1][1], f[1][2]};
                const f32x4w a2 = {f[1][3], f[2][0], f[2][1], f[2][2]}, a3 = {f[2][3], f[3][0], f[3][1], f[3][2]};
                const f32x2w a4 = {f[3][3], 0.f};
                const f32x4w b0 = {0.f, f[4][0], f[4][1], f[4][2]}, b1 = {f[4][3], f[5][0], f[5][1], f[5][2]};
                const f32x4w b2 = {f[5][3], f[6][0], f[6][1], f[6][2]}, b3 = {f[6][3], f[7][0], f[7][1], f[7][2]};
                const f32x2w b4 = {f[7][3], 0.f};
                *reinterpret_cast<f32x4w*>(row) = a0;
                *reinterpret_cast<f32x4w*>(row + 4) = a1;
                *reinterpret_cast<f32x4w*>(row + 8) = a2;
                *reinterpret_cast<f32x4w*>(row + 12) = a3;
                *reinterpret_cast<f32x2w*>(row + 16) = a4;
                *reinterpret_cast<f32x4w*>(row + 20) = b0;
                *reinterpret_cast<f32x4w*>(row + 24) = b1;
                *reinterpret_cast<f32x4w*>(row + 28) = b2;
                *reinterpret_cast<f32x4w*>(row + 32) = b3;
                *reinterpret_cast<f32x2w*>(row + 36) = b4;
            }
        } else if (z_role) {
            u32x4w* row = reinterpret_cast<u32x4w*>(dz_lds + zr_c * W4_SDZ + zr_r * 32);
#pragma unroll
            for (int k = 0; k < 8; ++k) row[k] = q[k];
        }
    };

    const int xb_off = l31 * W4_SX + RH * W4_XROW + kh * 4;                   // patch rows RH .. RH+4 of tile 2s + kh
    const int zb_off = 32 * W4_SX + (wm * 32 + l31) * W4_SDZ + kh * 4;         // dY rows 0..3 of tile 2s + kh

    auto ksteps = [&](const float* buf, int s0, int s1) __attribute__((always_inline)) {
        const float* xb = buf + xb_off;
        const float* zb = buf + zb_off;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < s0 || s >= s1) continue;
            float d[5][6];
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const int so = s * 8 + ((W16 && s >= 2) ? 4 : 0);               // W16: tiles 4..7 are image 1, 4 floats on
                const f32x4w v = *reinterpret_cast<const f32x4w*>(xb + r * W4_XROW + so);
                const f32x2w h = *reinterpret_cast<const f32x2w*>(xb + r * W4_XROW + so + 4);
                d[r][0] = v[0]; d[r][1] = v[1]; d[r][2] = v[2]; d[r][3] = v[3]; d[r][4] = h[0]; d[r][5] = h[1];
            }
            float y[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x4w v = *reinterpret_cast<const f32x4w*>(zb + r * 32 + s * 8);
                y[r][0] = v[0]; y[r][1] = v[1]; y[r][2] = v[2]; y[r][3] = v[3];
            }
            float t[3][5], uu[9], gt[3][4], g[9];
#pragma unroll
            for (int c = 0; c < 5; ++c) w_bt3<RH>(d[0][CH + c], d[1][CH + c], d[2][CH + c], d[3][CH + c], d[4][CH + c], t[0][c], t[1][c], t[2][c]);
#pragma unroll
            for (int i = 0; i < 3; ++i) w_bt3<CH>(t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], uu[i * 3], uu[i * 3 + 1], uu[i * 3 + 2]);
#pragma unroll
            for (int c = 0; c < 4; ++c) w_g3<RH>(y[0][c], y[1][c], y[2][c], y[3][c], gt[0][c], gt[1][c], gt[2][c]);
#pragma unroll
            for (int i = 0; i < 3; ++i) w_g3<CH>(gt[i][0], gt[i][1], gt[i][2], gt[i][3], g[i * 3], g[i * 3 + 1], g[i * 3 + 2]);
#pragma unroll
            for (int p = 0; p < 9; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[p], uu[p], acc[p], 0, 0, 0);
        }
    };

    // unit u lives in buffer (u - u0) & 1; the registers hold unit u+1 during the first half of unit u's K-steps and are
    // committed to the other buffer between K-steps 1 and 2 (every wave left that buffer at the barrier that ended unit
    // u-1), then take unit u+2: ONE barrier per unit
    issue(u0);
    commit(lds);
    __syncthreads();
    issue(u0 + 1);
    for (int u = u0; u < u1; ++u) {
        float* cur = lds + ((u - u0) & 1) * W4_BUF_FLOATS;
        float* nxt = lds + (((u - u0) & 1) ^ 1) * W4_BUF_FLOATS;
#if ONET_W4W_ASYM
        // the x rows are staged by waves 0..2, the dz rows by waves 4..7: SIMD partners (w, w + 4), i.e. position groups with
        // RH = 0 / 1.  In lockstep both partners sit in their staging burst (8 b128 loads + 9 b128 LDS stores) at once
        // and the matrix pipe idles: the RH = 1 waves stage one K-step earlier
        ksteps(cur, 0, 1);
        if constexpr (RH == 1) { commit(nxt); issue(u + 2); }
        ksteps(cur, 1, 2);
        if constexpr (RH == 0) { commit(nxt); issue(u + 2); }
        ksteps(cur, 2, 4);
#else
        ksteps(cur, 0, 2);
        commit(nxt);
        issue(u + 2);
        ksteps(cur, 2, 4);
#endif
        __syncthreads();
    }

    const int64_t n = (int64_t)a.Cout * a.Cin;
    const int ci = ci0 + l31;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int pos = (3 * RH + i) * 6 + 3 * CH + j;
            float* o = a.slab + ((int64_t)ks * 36 + pos) * n;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (co < a.Cout && ci < a.Cin) o[(int64_t)co * a.Cin + ci] = acc[i * 3 + j][r];
            }
        }
}

template <int W16>
__global__ __launch_bounds__(512, 2) void conv_wino4_wgrad_kernel(W4wArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem_w4[];
    const int pg = (threadIdx.x >> 6) >> 1;
    switch (__builtin_amdgcn_readfirstlane(pg)) {
        case 0: wino4w_body<0, 0, W16>(a, smem_w4); break;
        case 1: wino4w_body<0, 1, W16>(a, smem_w4); break;
        case 2: wino4w_body<1, 0, W16>(a, smem_w4); break;
        default: wino4w_body<1, 1, W16>(a, smem_w4); break;
    }
}

// dw[co][ci][3][3] (+)= A'^T (sum_k slab[k]) A', A'^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 1]
__global__ __launch_bounds__(256) void wino4w_fold_kernel(const float* __restrict__ slab, float* __restrict__ dw, int splitK,
                                                          int64_t n, int accumulate) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr float W4_SCALE[6] = {0.25f, -1.f / 6.f, -1.f / 6.f, 1.f / 24.f, 1.f / 24.f, 1.f};   // 1 / N_p of G'
    float m[6][6];
#pragma unroll
    for (int p = 0; p < 36; ++p) {
        float s = 0.f;
        for (int k = 0; k < splitK; ++k) s += slab[((int64_t)k * 36 + p) * n + i];
        m[p / 6][p % 6] = s * (W4_SCALE[p / 6] * W4_SCALE[p % 6]);
    }
    float t[3][6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const float s12 = m[1][c] + m[2][c], d12 = m[1][c] - m[2][c], s34 = m[3][c] + m[4][c], d34 = m[3][c] - m[4][c];
        t[0][c] = m[0][c] + s12 + s34;
        t[1][c] = fmaf(2.f, d34, d12);
        t[2][c] = fmaf(4.f, s34, s12) + m[5][c];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float s12 = t[r][1] + t[r][2], d12 = t[r][1] - t[r][2], s34 = t[r][3] + t[r][4], d34 = t[r][3] - t[r][4];
        const float o0 = t[r][0] + s12 + s34, o1 = fmaf(2.f, d34, d12), o2 = fmaf(4.f, s34, s12) + t[r][5];
        float* o = dw + i * 9 + r * 3;
        if (accumulate) { o[0] += o0; o[1] += o1; o[2] += o2; }
        else { o[0] = o0; o[1] = o1; o[2] = o2; }
    }
}

static void wino4w_plan(int B, int Cin, int Cout, int H, int W, int& splitK, int& tilesY, int& tilesX) {
    tilesY = cdiv(H, 4);
    tilesX = cdiv(W, 32);
    const int64_t units = (W == 16) ? (int64_t)(B / 2) * tilesY : (int64_t)B * tilesY * tilesX;
    const int64_t tiles = (int64_t)cdiv(Cin, 32) * cdiv(Cout, 64);
    static int target = -1;
    if (target < 0) { const char* e = getenv("ONET_W4W_BLOCKS"); target = (e && atoi(e) > 0) ? atoi(e) : 256; }
    int64_t k = std::max<int64_t>(1, (target + tiles - 1) / tiles);              // ONE 8-wave block per CU and one round: 256 beats 512 by 5-35 % (fewer slabs)
    const int64_t per = (int64_t)36 * Cout * Cin * 4;
    k = std::min<int64_t>(k, std::max<int64_t>(1, (320ll << 20) / per));          // slab budget
    k = std::min<int64_t>(k, std::max<int64_t>(1, units / 8));
    splitK = (int)k;
}

extern "C" {

int onet_conv3x3_winograd4_wgrad_ok(int B, int Cin, int Cout, int H, int W) {
    const bool wide = (W % 32) == 0, two = (W == 16) && (B % 2) == 0;       // two: a pair of 16-px-wide images per unit
    return (B > 0 && Cin >= 32 && (Cin % 32) == 0 && Cout > 0 && (Cout % 4) == 0 && (wide || two) && (H % 4) == 0) ? 1 : 0;
}

int64_t onet_conv3x3_winograd4_wgrad_ws_bytes(int B, int Cin, int Cout, int H, int W) {
    int splitK, ty, tx;
    wino4w_plan(B, Cin, Cout, H, W, splitK, ty, tx);
    return (int64_t)splitK * 36 * Cout * Cin * 4;
}

int onet_conv3x3_winograd4_wgrad(const float* x, int64_t x_bs, const float* dz, int64_t dz_bs, float* dw, void* ws, int64_t ws_bytes,
                                 int B, int Cin, int Cout, int H, int W, int accumulate, void* stream) {
    ONET_REQUIRE(x && dz && dw && ws, "conv3x3_winograd4_wgrad: null pointer");
    ONET_REQUIRE(onet_conv3x3_winograd4_wgrad_ok(B, Cin, Cout, H, W),
                 "conv3x3_winograd4_wgrad: needs W %% 32 == 0 (or W == 16 with even B), H %% 4 == 0, Cin %% 32 == 0 (use onet_conv3x3_winograd_wgrad)");
    ONET_REQUIRE((x_bs & 3) == 0 && (dz_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(dz) & 15) == 0,
                 "conv3x3_winograd4_wgrad: 16-byte aligned image planes required");
    ONET_REQUIRE(x_bs >= (int64_t)Cin * H * W && dz_bs >= (int64_t)Cout * H * W, "conv3x3_winograd4_wgrad: batch stride too small");
    ONET_REQUIRE((int64_t)std::max(Cin, Cout) * H * W * 4 < (1ll << 31), "conv3x3_winograd4_wgrad: image exceeds the 2 GiB buffer-resource range");
    W4wArgs a{x, x_bs, dz, dz_bs, (float*)ws, B, Cin, Cout, H, W, cdiv(Cin, 32), cdiv(Cout, 64), 1, 1, 1};
    wino4w_plan(B, Cin, Cout, H, W, a.splitK, a.tilesY, a.tilesX);
    const int64_t n = (int64_t)Cout * Cin;
    ONET_REQUIRE(ws_bytes >= (int64_t)a.splitK * 36 * n * 4, "conv3x3_winograd4_wgrad: workspace too small");
    const int64_t blocks = (int64_t)a.splitK * a.ciTiles * a.coTiles;
    if (W == 16) {
        ONET_REQUIRE(((x_bs + (int64_t)Cin * H * W) * 4 < (1ll << 31)) && ((dz_bs + (int64_t)Cout * H * W) * 4 < (1ll << 31)),
                     "conv3x3_winograd4_wgrad: image pair exceeds the 2 GiB buffer-resource range");
        auto kern = conv_wino4_wgrad_kernel<1>;
        static PerDeviceOnce attr_once;
        if (attr_once.first()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, W4C<1>::LDS_FLOATS * 4);
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), W4C<1>::LDS_FLOATS * 4, as_stream(stream), a);
    } else {
        auto kern = conv_wino4_wgrad_kernel<0>;
        static PerDeviceOnce attr_once;
        if (attr_once.first()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, W4C<0>::LDS_FLOATS * 4);
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), W4C<0>::LDS_FLOATS * 4, as_stream(stream), a);
    }
    int rc = check_launch("conv_wino4_wgrad_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(wino4w_fold_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream), (const float*)ws, dw, a.splitK, n,
                       accumulate);
    return check_launch("wino4w_fold_kernel");
}

}  // extern "C"
