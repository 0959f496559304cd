// K1: 3x3 / 1x1 convolution forward + dgrad (same kernel, packed weights differ) and wgrad,
// as im2col-free implicit GEMMs on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces F.conv2d / its autograd at OV:47,51 (via OV:111-120) and, with ks=1 on sub-pixel
// channels, nn.ConvTranspose2d(k=2,s=2) at OV:86.
//
// GEMM view (forward):  D[co][pix] = sum_{tap, ci} Wp[ci][tap][co] * X[ci][pix + tap]
//   M = output channels (MFMA rows), N = 32 consecutive output pixels (MFMA columns -> the
//   accumulator's lane dimension, so stores are 128-B coalesced along W), K = (tap, ci pairs).
//   A halo input tile for CI_T input channels is staged once in LDS and re-used by all 9 taps
//   and all output channels of the block; there is no im2col buffer.
// MFMA operand maps (cdna_hip_programming.md §3): A lane l -> A[i=l&31][k=l>>5],
//   B lane l -> B[k=l>>5][j=l&31], D reg r lane l -> D[(r&3)+8*(r>>2)+4*(l>>5)][l&31].
#include <algorithm>
#include <cstdlib>
#include "common.hpp"

using namespace onet;

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------ weight packing
__global__ void pack3x3_kernel(const float* __restrict__ w, float* __restrict__ wf,
                               float* __restrict__ wd, int Cout, int Cin) {
    int64_t n = (int64_t)Cout * Cin * 9;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        int t = (int)(i % 9);
        int64_t r = i / 9;
        int ci = (int)(r % Cin), co = (int)(r / Cin);
        float v = w[i];
        if (wf) wf[((int64_t)ci * 9 + t) * Cout + co] = v;
        if (wd) wd[((int64_t)co * 9 + (8 - t)) * Cin + ci] = v;
    }
}

// wq: sub-pixel index fastest (column c*4 + q) -- the layout of the fused convT forward, whose epilogue finds
// the four sub-pixels of one output channel in four consecutive accumulator rows of one lane
__global__ void packT2x2_kernel(const float* __restrict__ w, float* __restrict__ wf,
                                float* __restrict__ wd, float* __restrict__ wq, int Cin, int Cout) {
    int64_t n = (int64_t)Cin * Cout * 4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        int q = (int)(i & 3);
        int64_t r = i >> 2;
        int co = (int)(r % Cout), ci = (int)(r / Cout);
        float v = w[i];
        if (wf) wf[(int64_t)ci * (4 * Cout) + q * Cout + co] = v;
        if (wd) wd[((int64_t)q * Cout + co) * Cin + ci] = v;
        if (wq) wq[(int64_t)ci * (4 * Cout) + co * 4 + q] = v;
    }
}

// ------------------------------------------------------------------ forward / dgrad
struct ConvArgs {
    const float* x;
    int64_t x_bs;
    const float* wp;
    float* z;
    int64_t z_bs;
    float* bn_part;
    int B, Cin, Cout, H, W, tilesX, tilesY, coTiles;
    // fused ConvTranspose2d(k=2, s=2) epilogue (SHUF kernels only): z is the [Cout/4][Ho][Wo] window of the concat
    // buffer, GEMM row n = c*4 + q holds sub-pixel q = 2*di + dj of output channel c
    const float* bias = nullptr;
    int Ho = 0, Wo = 0, pt = 0, pl = 0;
};

constexpr int CI_T = 8;  // input channels staged per K-chunk

template <int KS, int MT, int NT, int WM, int WN, int TW>
struct ConvCfg {
    static constexpr int TAPS = KS * KS, PAD = KS / 2;
    static constexpr int RPT = 32 / TW;               // image rows per 32-pixel MFMA column tile
    static constexpr int ROWS = NT * WN * RPT;        // output rows per block
    static constexpr int IN_ROWS = ROWS + 2 * PAD, IN_COLS = TW + 2 * PAD;
    static constexpr int ROW_STRIDE = IN_COLS;
    static constexpr int CH_STRIDE = IN_ROWS * ROW_STRIDE;
    static constexpr int CO_T = 32 * MT * WM;         // output channels per block
    static constexpr int W_FLOATS = CI_T * TAPS * CO_T;
    static constexpr int IN_FLOATS = CI_T * CH_STRIDE;
    static constexpr int LDS_BYTES = (W_FLOATS + IN_FLOATS) * 4;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Buffer loads with a 32-bit byte offset: the hardware range check returns 0 for any offset >=
// num_records, so zero padding (image border, channel tails) costs no branch: out-of-image
// elements simply get the offset OOB_OFF.
constexpr unsigned OOB_OFF = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int64_t bytes) {
    const int n = bytes > 0x7fffffffll ? 0x7fffffff : (int)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, n, 0x00020000);
}
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ u32x4 bload4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
}

// Software pipeline per K-chunk of CI_T input channels:
//   regs(chunk c+1) <- buffer loads issued BEFORE the MFMA block of chunk c (latency hidden under
//   288 MFMAs/wave), LDS <- regs after it; 2 blocks/CU cover the commit + barrier bubbles.
// MODE 0: plain   1: ConvTranspose2d forward (pixel-shuffle + bias epilogue)   2: ConvTranspose2d input-grad (the
// GEMM input channel q*Ct + c at pixel (i, j) is gathered from dy[c][pt + 2i + di][pl + 2j + dj], q = 2*di + dj)
template <int KS, int MT, int NT, int WM, int WN, int TW, int MODE = 0>
__global__ __launch_bounds__(256, 2) void conv_fwd_kernel(ConvArgs a) {
    constexpr bool SHUF = (MODE == 1), S2D = (MODE == 2);
    using C = ConvCfg<KS, MT, NT, WM, WN, TW>;
    static_assert(WM * WN == 4, "4 waves per block");
    constexpr int TAPS = C::TAPS, PAD = C::PAD, RPT = C::RPT, ROWS = C::ROWS;
    constexpr int ROW_STRIDE = C::ROW_STRIDE;
    constexpr int CH_STRIDE = C::CH_STRIDE, CO_T = C::CO_T;
    constexpr int IN_FLOATS = C::IN_FLOATS, W_FLOATS = C::W_FLOATS;
    constexpr int NIN = (IN_FLOATS + 255) / 256;        // input dwords staged per thread per chunk
    constexpr int NW4 = (W_FLOATS / 4 + 255) / 256;     // weight float4s per thread per chunk
    constexpr int NW1 = (W_FLOATS + 255) / 256;         // scalar fallback (Cout % 4 != 0)

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* w_lds = smem;                  // [CI_T][TAPS][CO_T]
    float* in_lds = smem + W_FLOATS;      // [CI_T][IN_ROWS][ROW_STRIDE]

    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with a private
    // L2), so give every XCD a CONTIGUOUS range of logical tiles -- horizontally adjacent tiles share
    // halo columns / 128-B lines and the co-tiles of one pixel tile share the whole input tile.
    // (bijective for any grid size; placement only affects speed, never results)
    int bid;
    {
        const int n = gridDim.x, q = n >> 3, r = n & 7, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    // pixel tile fastest, output-channel tile slowest: the ~32 blocks resident on one XCD at a time then
    // stream the SAME packed weights (the larger operand per K-chunk) through its L2 once, and
    // neighbouring pixel tiles share their halo lines.
    const int tx = bid % a.tilesX;
    bid /= a.tilesX;
    const int ty = bid % a.tilesY;
    bid /= a.tilesY;
    const int b = bid % a.B;
    const int coT = bid / a.B;
    const int co0 = coT * CO_T, y0 = ty * ROWS, x0 = tx * TW;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int l31 = lane & 31, kh = lane >> 5;
    const int px = l31 % TW, py = l31 / TW;
    const int HW = a.H * a.W;

    // Two-level accumulation (small register tiles only, i.e. the deep layers where K = 9*Cin reaches
    // 9216): an MFMA accumulator is a strictly sequential fp32 fmaf chain, whose rounding error grows
    // ~sqrt(K); every FLUSH chunks (288 terms) the chain is folded into a second accumulator set.
    constexpr bool TWO_LEVEL = (MT * NT <= 4);
    constexpr int FLUSH = 4;
    f32x16 acc[MT][NT];
    f32x16 tot[TWO_LEVEL ? MT : 1][TWO_LEVEL ? NT : 1];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[m][n][r] = 0.f;
                if constexpr (TWO_LEVEL) tot[m][n][r] = 0.f;
            }

    const float* a_ptr = w_lds + kh * TAPS * CO_T + wm * 32 * MT + l31;
    const float* b_ptr = in_lds + kh * CH_STRIDE + (wn * NT * RPT + py) * ROW_STRIDE + px;

    const int HWo = a.Ho * a.Wo, s2d_Ct = a.Cin >> 2;     // S2D only
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(a.x + (int64_t)b * a.x_bs, S2D ? (int64_t)s2d_Ct * HWo * 4 : (int64_t)a.Cin * HW * 4);
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(a.wp, (int64_t)a.Cin * TAPS * a.Cout * 4);
    const bool vec4 = (a.Cout & 3) == 0;

    // chunk-invariant byte offsets of this thread's staged elements (OOB_OFF = zero padding)
    unsigned in_off[NIN];
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
        const int i = tid + 256 * k;
        const int ci = i / CH_STRIDE, rem = i % CH_STRIDE;
        const int r = rem / ROW_STRIDE, c = rem % ROW_STRIDE;
        const int yy = y0 - PAD + r, xx = x0 - PAD + c;
        const bool ok = (i < IN_FLOATS) && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
        if constexpr (S2D) in_off[k] = ok ? (unsigned)((ci * HWo + (a.pt + 2 * yy) * a.Wo + a.pl + 2 * xx) * 4) : OOB_OFF;
        else in_off[k] = ok ? (unsigned)((ci * HW + yy * a.W + xx) * 4) : OOB_OFF;
    }
    // S2D: byte offset of GEMM input channel c0 (a multiple of CI_T; Ct % CI_T == 0 keeps a chunk inside one sub-pixel)
    auto s2d_bytes = [&](int c0) -> unsigned {
        if (c0 >= a.Cin) return OOB_OFF;
        const int q = c0 / s2d_Ct, cc = c0 % s2d_Ct;
        return (unsigned)((cc * HWo + (q >> 1) * a.Wo + (q & 1)) * 4);
    };
    unsigned w_off[NW4];
#pragma unroll
    for (int k = 0; k < NW4; ++k) {
        const int i = (tid + 256 * k) * 4;
        const int ci = i / (TAPS * CO_T), rem = i % (TAPS * CO_T);
        const int t = rem / CO_T, co = rem % CO_T;
        const bool ok = (i < W_FLOATS) && (co0 + co < a.Cout);
        w_off[k] = ok ? (unsigned)(((ci * TAPS + t) * a.Cout + co0 + co) * 4) : OOB_OFF;
    }
    const unsigned in_step = (unsigned)(CI_T * HW * 4), w_step = (unsigned)(CI_T * TAPS * a.Cout * 4);

    float xin[NIN];
    u32x4 wv[NW4];
    auto issue = [&](unsigned cin_bytes, unsigned cw_bytes) {
#pragma unroll
        for (int k = 0; k < NIN; ++k) xin[k] = bload(xr, in_off[k] + cin_bytes);
        if (vec4) {
#pragma unroll
            for (int k = 0; k < NW4; ++k) wv[k] = bload4(wr, w_off[k] + cw_bytes);
        }
    };
    auto commit = [&](int c0) {
#pragma unroll
        for (int k = 0; k < NIN; ++k) {
            const int i = tid + 256 * k;
            if (i < IN_FLOATS) in_lds[i] = xin[k];
        }
        if (vec4) {
#pragma unroll
            for (int k = 0; k < NW4; ++k) {
                const int i = (tid + 256 * k) * 4;
                if (i < W_FLOATS) *reinterpret_cast<u32x4*>(w_lds + i) = wv[k];
            }
        } else {   // rare: Cout not a multiple of 4 -> unpipelined scalar weight staging
#pragma unroll 1
            for (int k = 0; k < NW1; ++k) {
                const int i = tid + 256 * k;
                const int ci = i / (TAPS * CO_T), rem = i % (TAPS * CO_T);
                const int t = rem / CO_T, co = rem % CO_T;
                float v = 0.f;
                if (i < W_FLOATS && co0 + co < a.Cout)
                    v = bload(wr, (unsigned)((((c0 + ci) * TAPS + t) * a.Cout + co0 + co) * 4));
                if (i < W_FLOATS) w_lds[i] = v;
            }
        }
    };

    unsigned cin_bytes = 0, cw_bytes = 0;
    issue(0, 0);
    for (int c0 = 0; c0 < a.Cin; c0 += CI_T) {
        commit(c0);
        __syncthreads();
        if constexpr (S2D) cin_bytes = s2d_bytes(c0 + CI_T);
        else cin_bytes += in_step;
        cw_bytes += w_step;
        issue(cin_bytes, cw_bytes);                 // next chunk (past the end: range check -> zeros, no traffic)
        __builtin_amdgcn_sched_barrier(0);          // keep the loads ahead of the MFMA block
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const int ky = t / KS, kx = t % KS;
#pragma unroll
            for (int cp = 0; cp < CI_T / 2; ++cp) {
                float av[MT], bv[NT];
#pragma unroll
                for (int m = 0; m < MT; ++m) av[m] = a_ptr[(cp * 2 * TAPS + t) * CO_T + m * 32];
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    bv[n] = b_ptr[cp * 2 * CH_STRIDE + (n * RPT + ky) * ROW_STRIDE + kx];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[n], acc[m][n], 0, 0, 0);
            }
        }
        if constexpr (TWO_LEVEL) {
            if (((c0 / CI_T) % FLUSH) == FLUSH - 1) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        tot[m][n] += acc[m][n];
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
                    }
            }
        }
        __syncthreads();                            // everyone is done reading this chunk's LDS
    }
    if constexpr (TWO_LEVEL) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] += tot[m][n];
    }

    // ---- epilogue: D[co][pix] -> z (lanes run along x: coalesced 128-B rows)
    float* zb = a.z + (int64_t)b * a.z_bs;
    const int xo = x0 + px;
    if constexpr (SHUF) {
        // pixel shuffle + bias + F.pad offsets fused (OV:86, 91-100): accumulator rows 4g .. 4g+3 of a lane are the
        // 2x2 sub-pixels of ONE output channel at low-res pixel (yo, xo) -> two float2 row stores (lanes along x:
        // 512 contiguous bytes per wave row)
        const int64_t HWo = (int64_t)a.Ho * a.Wo;
        const bool vec2 = ((a.pl & 1) == 0) && ((a.Wo & 1) == 0) && ((a.z_bs & 1) == 0) &&
                          ((reinterpret_cast<uintptr_t>(a.z) & 7) == 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int yo = y0 + (wn * NT + n) * RPT + py;
                if (yo < a.H && xo < a.W) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c = ((co0 + (wm * MT + m) * 32) >> 2) + 2 * g + kh;
                        if (c < (a.Cout >> 2)) {
                            const float bv = a.bias ? a.bias[c] : 0.f;
                            float* o = zb + (int64_t)c * HWo + (int64_t)(a.pt + 2 * yo) * a.Wo + a.pl + 2 * xo;
                            const float v0 = acc[m][n][4 * g] + bv, v1 = acc[m][n][4 * g + 1] + bv;
                            const float v2 = acc[m][n][4 * g + 2] + bv, v3 = acc[m][n][4 * g + 3] + bv;
                            if (vec2) {
                                *reinterpret_cast<float2*>(o) = make_float2(v0, v1);
                                *reinterpret_cast<float2*>(o + a.Wo) = make_float2(v2, v3);
                            } else {
                                o[0] = v0; o[1] = v1; o[a.Wo] = v2; o[a.Wo + 1] = v3;
                            }
                        }
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int yo = y0 + (wn * NT + n) * RPT + py;
            if (yo < a.H && xo < a.W) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + (wm * MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    if (co < a.Cout) zb[(int64_t)co * HW + (int64_t)yo * a.W + xo] = acc[m][n][r];
                }
            }
        }
    }
}

template <int KS, int MT, int NT, int WM, int WN, int TW, int MODE = 0>
static int launch_fwd(ConvArgs a, hipStream_t st) {
    using C = ConvCfg<KS, MT, NT, WM, WN, TW>;
    a.tilesX = cdiv(a.W, TW);
    a.tilesY = cdiv(a.H, C::ROWS);
    a.coTiles = cdiv(a.Cout, C::CO_T);
    const int64_t blocks = (int64_t)a.B * a.tilesX * a.tilesY * a.coTiles;
    ONET_REQUIRE(blocks > 0 && blocks < (1ll << 31), "conv_fwd: grid %lld out of range", (long long)blocks);
    auto kern = conv_fwd_kernel<KS, MT, NT, WM, WN, TW, MODE>;
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  C::LDS_BYTES);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), C::LDS_BYTES, st, a);
    return check_launch("conv_fwd_kernel");
}

// Tile configurations <MT, NT, WM, WN, TW> (per-wave register tile MT x NT of 32x32 MFMA tiles,
// 4 waves as WM x WN, TW-pixel-wide column tiles):
//   P  <2,4,1,4,32>   64 co x (16 rows x 32 px)   pixel-heavy, single-level accumulation
//   C  <2,4,2,2,32>  128 co x ( 8 rows x 32 px)   channel-heavy, single-level accumulation
//   D  <2,2,1,4,TW>   64 co x ( 8 rows x 32 px | 16 rows x 16 px)   two-level accumulation
//   E  <2,2,2,2,32>  128 co x ( 4 rows x 32 px)   two-level accumulation
// ONET_CONV_CFG=P|C|D|E forces one (tuning / A-B runs); default: heuristic below.
static int conv_cfg_override() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ONET_CONV_CFG");
        v = (e && e[0]) ? e[0] : 0;
    }
    return v;
}

template <int KS>
static int dispatch_fwd(const ConvArgs& a, hipStream_t st) {
    const bool wide = a.W > 16;
    if (!wide) return launch_fwd<KS, 2, 2, 1, 4, 16>(a, st);
    switch (conv_cfg_override()) {
        case 'P': return launch_fwd<KS, 2, 4, 1, 4, 32>(a, st);
        case 'C': return launch_fwd<KS, 2, 4, 2, 2, 32>(a, st);
        case 'D': return launch_fwd<KS, 2, 2, 1, 4, 32>(a, st);
        case 'E': return launch_fwd<KS, 2, 2, 2, 2, 32>(a, st);
        default: break;
    }
    // Measured on MI355X (tools/bench_conv.py, B=32, 256^2 U-Net layer shapes, fwd+dgrad): P 138.6, D 134.6,
    // C 115-124, E 109-118 TFLOP/s.  D is the default: within 3 % of P and it carries the two-level
    // accumulation that keeps |z - exact| ~2e-7 (fewer ReLU-kink sign flips against the reference).
    return launch_fwd<KS, 2, 2, 1, 4, 32>(a, st);
}

// ------------------------------------------------------------------ wgrad
struct WgArgs {
    const float* x;
    int64_t x_bs;
    const float* dz;
    int64_t dz_bs;
    float* slab;
    int B, Cin, Cout, H, W, ciTiles, coTiles, splitK, stripsY, stripsX;
    // ConvTranspose2d weight-grad (S2D kernels): dz is the [Ct][Ho][Wo] window of dcat, GEMM row q*Ct + c at pixel
    // (i, j) is dy[c][pt + 2i + di][pl + 2j + dj]
    int s2d_Ct = 0, Ho = 0, Wo = 0, pt = 0, pl = 0;
    // stem weight gradient with the BatchNorm + ReLU backward applied on load (conv3x3_stem_wgrad_kernel<.., true>): `dz` is then da
    // (the gradient of the unit's OUTPUT), bn_z the pre-activation, bn_save / bn_coef [G][4][Cout] (bn.hip), bn_gimg images per group
    const float* bn_z = nullptr;
    int64_t bn_z_bs = 0;
    const float* bn_save = nullptr;
    const float* bn_coef = nullptr;
    int bn_gimg = 0;
    // stem kernels: x is the first half of the twin batch [X ; clip(1 - X + bias, 0, 1)] (stem.hip: StemArgs); 0: ordinary batch
    int twin_B = 0;
    float twin_bias = 0.f;
};

template <int KS, int PW>
struct WgCfg {
    static constexpr int TAPS = KS * KS, PAD = KS / 2;
    static constexpr int PR = 64 / PW;                    // rows per pixel strip (64 pixels per stage)
    static constexpr int XR = PR + 2 * PAD, XC = PW + 2 * PAD;
    static constexpr int DZ_STRIDE = PR * PW + 1;         // odd: lanes index channels -> conflict-free
    static constexpr int X_STRIDE = (XR * XC) | 1;
    static constexpr int LDS_BYTES = 64 * (DZ_STRIDE + X_STRIDE) * 4;
};

// slab[ks][tap][co][ci] = sum over this block's pixel strips of dz[co][p] * x[ci][p + tap]
// Block = 4*TG waves: a 2 x 2 grid of 32co x 32ci sub-tiles times TG tap groups (3x3: one kernel ROW
// per group, 3 accumulators = 48 VGPRs per wave -> 12 waves = 3 per SIMD hide LDS latency and the
// commit/barrier bubbles; 1x1: TG = 1).  Staging is software-pipelined like the forward kernel: the
// buffer loads of strip u+splitK are issued before the MFMA block of strip u and committed to LDS
// after it.  All staged elements of a thread sit at a constant channel stride, so one base offset
// per operand suffices.
template <int KS, int PW, bool S2D = false>
__global__ __launch_bounds__(256 * KS, KS == 3 ? 3 : 2) void conv_wgrad_kernel(WgArgs a) {
    using C = WgCfg<KS, PW>;
    constexpr int TAPS = C::TAPS, PAD = C::PAD, PR = C::PR, XR = C::XR, XC = C::XC;
    constexpr int DZ_STRIDE = C::DZ_STRIDE, X_STRIDE = C::X_STRIDE;
    constexpr int TG = KS;                                   // tap groups = kernel rows
    constexpr int NTAP = TAPS / TG;                          // taps per wave (one kernel row)
    constexpr int NTHR = 256 * TG;
    constexpr int DZ_CSTEP = NTHR / 64;                      // dz channels covered per step
    constexpr int NDZ = (64 + DZ_CSTEP - 1) / DZ_CSTEP;
    constexpr int ROWS_PER_K = NTHR / PW;                    // x rows covered by the block per step
    constexpr int CH_PER_K = ROWS_PER_K / XR;                // whole channels per step
    constexpr int NXM = (64 + CH_PER_K - 1) / CH_PER_K;      // steps for the 64 channels (main columns)
    constexpr int NXH = (KS == 3) ? (64 * XR * 2 + NTHR - 1) / NTHR : 0;   // halo-column dwords per thread

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dz_lds = smem;                    // [64 co][DZ_STRIDE]
    float* x_lds = smem + 64 * DZ_STRIDE;    // [64 ci][X_STRIDE]

    // XCD-aware order (see conv_fwd_kernel): every XCD gets a contiguous range of logical blocks,
    // ordered split-K slice major / (ci,co) tile minor, so the blocks that stream the SAME pixel strips
    // (all tiles of one slice) run back to back on one XCD and share them through its L2.
    int bid;
    {
        const int n = gridDim.x, q = n >> 3, r = n & 7, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tiles = a.ciTiles * a.coTiles;
    const int ks = bid / tiles;
    const int tile = bid % tiles;
    const int ciT = tile % a.ciTiles;
    const int coT = tile / a.ciTiles;
    const int co0 = coT * 64, ci0 = ciT * 64;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tg = wid >> 2, wm = (wid >> 1) & 1, wn = wid & 1;
    const int l31 = lane & 31, kh = lane >> 5;
    const int HW = a.H * a.W;

    // (split-K already bounds every fp32 accumulation chain to nunits/splitK strips of 64 pixels)
    f32x16 acc[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const float* a_ptr = dz_lds + (wm * 32 + l31) * DZ_STRIDE + kh;
    const float* b_ptr = x_lds + (wn * 32 + l31) * X_STRIDE + kh + tg * XC;   // this wave's kernel row

    // per-thread staging coordinates (unit-invariant parts)
    const int dz_c = tid >> 6, dz_p = tid & 63, dz_r = dz_p / PW, dz_col = dz_p % PW;
    const int xm_col = tid % PW, xm_rowid = tid / PW, xm_cp = xm_rowid / XR, xm_r = xm_rowid % XR;
    const bool xm_thread = xm_rowid < CH_PER_K * XR;
    const unsigned dz_kstep = (unsigned)(DZ_CSTEP * HW * 4), xm_kstep = (unsigned)(CH_PER_K * HW * 4);

    float dzv[NDZ], xmv[NXM], xhv[NXH > 0 ? NXH : 1];
    const int nunits = a.B * a.stripsY * a.stripsX;

    auto issue = [&](int u) {
        // past the last unit every offset is out of range: the loads return 0 without traffic
        const bool live = u < nunits;
        const int uu = live ? u : 0;
        const int sx = uu % a.stripsX;
        const int sy = (uu / a.stripsX) % a.stripsY;
        const int b = uu / (a.stripsX * a.stripsY);
        const int y0 = sy * PR, x0 = sx * PW;
        const int HWo = a.Ho * a.Wo;
        const __amdgpu_buffer_rsrc_t dr = make_rsrc(a.dz + (int64_t)b * a.dz_bs, S2D ? (int64_t)a.s2d_Ct * HWo * 4 : (int64_t)a.Cout * HW * 4);
        const __amdgpu_buffer_rsrc_t xr = make_rsrc(a.x + (int64_t)b * a.x_bs, (int64_t)a.Cin * HW * 4);
        {
            const int yy = y0 + dz_r, xx = x0 + dz_col;
            const bool ok = live && yy < a.H && xx < a.W;
            unsigned base, kstep;
            if constexpr (S2D) {          // Ct % 64 == 0: the block's 64 GEMM rows share one sub-pixel q
                const int q = co0 / a.s2d_Ct, cc = co0 % a.s2d_Ct + dz_c;
                base = ok ? (unsigned)((cc * HWo + (a.pt + 2 * yy + (q >> 1)) * a.Wo + a.pl + 2 * xx + (q & 1)) * 4) : OOB_OFF;
                kstep = (unsigned)(DZ_CSTEP * HWo * 4);
            } else {
                base = ok ? (unsigned)(((co0 + dz_c) * HW + yy * a.W + xx) * 4) : OOB_OFF;
                kstep = dz_kstep;
            }
#pragma unroll
            for (int k = 0; k < NDZ; ++k)
                dzv[k] = bload(dr, (ok && dz_c + DZ_CSTEP * k < 64) ? base + k * kstep : OOB_OFF);
        }
        {
            const int yy = y0 - PAD + xm_r, xx = x0 + xm_col;
            const bool ok = live && xm_thread && yy >= 0 && yy < a.H && xx < a.W;
            const unsigned base = ok ? (unsigned)(((ci0 + xm_cp) * HW + yy * a.W + xx) * 4) : OOB_OFF;
#pragma unroll
            for (int k = 0; k < NXM; ++k)
                xmv[k] = bload(xr, (xm_cp + CH_PER_K * k < 64) ? base + k * xm_kstep : OOB_OFF);
        }
#pragma unroll
        for (int j = 0; j < NXH; ++j) {
            const int e = tid + NTHR * j;
            const int c = e / (2 * XR), q = e % (2 * XR);
            const int r = q >> 1, side = q & 1;
            const int yy = y0 - PAD + r, xx = side ? x0 + PW : x0 - 1;
            const bool ok = live && e < 64 * XR * 2 && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
            xhv[j] = bload(xr, ok ? (unsigned)(((ci0 + c) * HW + yy * a.W + xx) * 4) : OOB_OFF);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < NDZ; ++k)
            if (dz_c + DZ_CSTEP * k < 64) dz_lds[(dz_c + DZ_CSTEP * k) * DZ_STRIDE + dz_p] = dzv[k];
        if (xm_thread) {
#pragma unroll
            for (int k = 0; k < NXM; ++k)
                if (xm_cp + CH_PER_K * k < 64)
                    x_lds[(xm_cp + CH_PER_K * k) * X_STRIDE + xm_r * XC + xm_col + PAD] = xmv[k];
        }
#pragma unroll
        for (int j = 0; j < NXH; ++j) {
            const int e = tid + NTHR * j;
            const int c = e / (2 * XR), q = e % (2 * XR);
            if (e < 64 * XR * 2) x_lds[c * X_STRIDE + (q >> 1) * XC + ((q & 1) ? XC - 1 : 0)] = xhv[j];
        }
    };

    issue(ks);
    for (int u = ks; u < nunits; u += a.splitK) {
        commit();
        __syncthreads();
        issue(u + a.splitK);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < PR; ++r) {
#pragma unroll
            for (int j = 0; j < PW / 2; ++j) {
                const float av = a_ptr[r * PW + 2 * j];
#pragma unroll
                for (int t = 0; t < NTAP; ++t) {
                    const float bv = b_ptr[r * XC + 2 * j + t];     // kx = t within this wave's kernel row
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // slab[ks][t][co][ci]; lanes run along ci (coalesced)
    float* sl = a.slab + (int64_t)ks * TAPS * a.Cout * a.Cin;
    const int ci = ci0 + wn * 32 + l31;
    if (ci < a.Cin) {
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (co < a.Cout) sl[((int64_t)(tg * NTAP + t) * a.Cout + co) * a.Cin + ci] = acc[t][r];
            }
        }
    }
}

// dw (+)= sum_ks slab[ks][t][co][ci], written in the nn.Module parameter layout.
// block = 64 consecutive (co,ci) x 4 split-K groups for one tap; LDS combine of the 4 groups.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                           int splitK, int taps, int Cout, int Cin, int out_layout,
                                                           int accumulate) {
    __shared__ float red[256];
    const int64_t n = (int64_t)Cout * Cin;
    const int t = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const int g = threadIdx.x >> 6;
    float s = 0.f;
    if (i < n)
        for (int k = g; k < splitK; k += 4) s += slab[((int64_t)k * taps + t) * n + i];
    red[threadIdx.x] = s;
    __syncthreads();
    if (g == 0 && i < n) {
        s = (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
        const int ci = (int)(i % Cin), co = (int)(i / Cin);
        int64_t o;
        if (out_layout == 0) {
            o = i * taps + t;                               // [co][ci][ky][kx]
        } else {
            const int Ct = Cout / 4, q = co / Ct, cot = co % Ct;
            o = ((int64_t)ci * Ct + cot) * 4 + q;           // convT [ci][co][dy][dx]
        }
        dw[o] = accumulate ? dw[o] + s : s;
    }
}

namespace onet {
int launch_wgrad_reduce(const float* slab, float* dw, int splitK, int taps, int Cout, int Cin, int out_layout, int accumulate,
                        hipStream_t st) {
    const int64_t n = (int64_t)Cout * Cin;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv(n, 64), (unsigned)taps), dim3(256), 0, st, slab, dw, splitK,
                       taps, Cout, Cin, out_layout, accumulate);
    return check_launch("wgrad_reduce_kernel");
}
}  // namespace onet

// Stem wgrad (Cin <= 4: the 1- or 3-channel input image, OV:111): dW[co][ci][tap] has only 64*Cin*9
// entries but reduces over every pixel, so it is an HBM-bound streaming reduction over dz, not a GEMM
// (the MFMA tile would be 63/64 padding).  One block per (image, 32-row band, group of COG output
// channels); a wave takes 8 rows of the band, a thread FOUR neighbouring pixel columns (one 16-byte load per
// row and channel plane: 1 KB per wave-instruction in flight instead of 256 B) and walks down its rows with a
// 3 x 6 sliding window of the input in registers, COG*CIN*9 accumulators, then wave-shuffle + LDS
// reduction; partial slab [image*band][tap][co][ci] for wgrad_reduce_kernel (deterministic).
// BN: dz is not read but computed per element from (da, z) and the layer's BatchNorm-backward coefficients -- the arithmetic of
// bn_relu_bwd_apply_kernel (fp64 per element, rounded once), so the result is bit for bit the weight gradient of the dz that pass
// would have written: the stem's input has no gradient, its dz no other reader, and the 12 B/element pass + this kernel's 4 B/element
// read become one 8 B/element read.
constexpr int STEM_ROWS = 32;
template <int CIN, int COG, bool BN = false>
__global__ __launch_bounds__(256) void conv3x3_stem_wgrad_kernel(WgArgs a) {
    constexpr int NV = COG * CIN * 9;
    __shared__ float red[4][NV];
    const int cogs = (a.Cout + COG - 1) / COG;
    const int bandsY = (a.H + STEM_ROWS - 1) / STEM_ROWS;
    const int cg = blockIdx.x % cogs, bb = blockIdx.x / cogs;
    const int b = bb / bandsY, yb = (bb % bandsY) * STEM_ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int HW = a.H * a.W, W = a.W, H = a.H;
    const bool comp = a.twin_B > 0 && b >= a.twin_B;          // K7 on load (OV:180): the complement half of a twin batch
    const float* xb = a.x + (int64_t)(comp ? b - a.twin_B : b) * a.x_bs;
    const float* dzb = a.dz + (int64_t)b * a.dz_bs + (int64_t)cg * COG * HW;
    const float* zb = BN ? a.bn_z + (int64_t)b * a.bn_z_bs + (int64_t)cg * COG * HW : nullptr;
    const int y0 = yb + wid * (STEM_ROWS / 4), y1 = min(y0 + STEM_ROWS / 4, H);
    // 16-byte loads of the gradient planes need aligned rows
    const bool vec = (W & 3) == 0 && (a.dz_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(a.dz) & 15) == 0 &&
                     (!BN || ((a.bn_z_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(a.bn_z) & 15) == 0));
    float bmean[COG], binv[COG], bsc[COG], bsh[COG];       // (few live registers: this streaming loop lives on occupancy)
    double bc1[COG], bc2[COG];
    if (BN) {
        const int grp = a.bn_gimg ? b / a.bn_gimg : 0;
        const float* sv = a.bn_save + (int64_t)grp * 4 * a.Cout;
        const float* cf = a.bn_coef ? a.bn_coef + (int64_t)grp * 4 * a.Cout : nullptr;
#pragma unroll
        for (int g = 0; g < COG; ++g) {
            const int c = min(cg * COG + g, a.Cout - 1);
            bmean[g] = sv[c];
            binv[g] = sv[a.Cout + c];
            bsc[g] = sv[2 * a.Cout + c];
            bsh[g] = sv[3 * a.Cout + c];
            bc1[g] = cf ? (double)cf[c] + (double)cf[a.Cout + c] : 0.0;
            bc2[g] = cf ? (double)cf[2 * a.Cout + c] + (double)cf[3 * a.Cout + c] : 0.0;
        }
    }
    float acc[COG][CIN][9];
#pragma unroll
    for (int g = 0; g < COG; ++g)
#pragma unroll
        for (int c = 0; c < CIN; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[g][c][t] = 0.f;
    auto in_row = [&](int c, int yy, int x0, float (&r)[6]) __attribute__((always_inline)) {       // input columns x0 - 1 .. x0 + 4 of row yy
        const bool rowok = yy >= 0 && yy < H;
        const float* p = xb + (int64_t)c * HW + (int64_t)yy * W;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int xx = x0 - 1 + k;
            const bool ok = rowok && xx >= 0 && xx < W;
            float v = ok ? p[xx] : 0.f;
            if (comp && ok) v = fminf(fmaxf(1.f - v + a.twin_bias, 0.f), 1.f);
            r[k] = v;
        }
    };
    for (int x0 = lane * 4; x0 < W && y0 < y1; x0 += 256) {
        float win[CIN][3][6];   // rows y-1, y, y+1 ; cols x0-1 .. x0+4
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
            in_row(c, y0 - 1, x0, win[c][1]);
            in_row(c, y0, x0, win[c][2]);
        }
        for (int y = y0; y < y1; ++y) {
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    win[c][0][k] = win[c][1][k];
                    win[c][1][k] = win[c][2][k];
                }
                in_row(c, y + 1, x0, win[c][2]);
            }
#pragma unroll
            for (int g = 0; g < COG; ++g) {
                float gz[4] = {0.f, 0.f, 0.f, 0.f}, zz[4] = {0.f, 0.f, 0.f, 0.f};
                if (cg * COG + g < a.Cout) {
                    const int64_t o = (int64_t)g * HW + (int64_t)y * W + x0;
                    if (vec) {
                        const float4 q = *reinterpret_cast<const float4*>(dzb + o);
                        gz[0] = q.x; gz[1] = q.y; gz[2] = q.z; gz[3] = q.w;
                        if (BN) {
                            const float4 r = *reinterpret_cast<const float4*>(zb + o);
                            zz[0] = r.x; zz[1] = r.y; zz[2] = r.z; zz[3] = r.w;
                        }
                    } else {
#pragma unroll
                        for (int p = 0; p < 4; ++p)
                            if (x0 + p < W) {
                                gz[p] = dzb[o + p];
                                if (BN) zz[p] = zb[o + p];
                            }
                    }
                    if (BN) {
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            const double dy = fmaf(zz[p] - bmean[g], bsc[g], bsh[g]) > 0.f ? (double)gz[p] : 0.0;
                            const float v = (float)((double)bsc[g] * (dy - bc1[g] - (((double)zz[p] - (double)bmean[g]) * (double)binv[g]) * bc2[g]));
                            gz[p] = (vec || x0 + p < W) ? v : 0.f;
                        }
                    }
                }
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int c = 0; c < CIN; ++c)
#pragma unroll
                        for (int t = 0; t < 9; ++t) acc[g][c][t] = fmaf(gz[p], win[c][t / 3][p + t % 3], acc[g][c][t]);
            }
        }
    }
#pragma unroll
    for (int g = 0; g < COG; ++g)
#pragma unroll
        for (int c = 0; c < CIN; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float s = wave_sum(acc[g][c][t]);
                if (lane == 0) red[wid][(g * CIN + c) * 9 + t] = s;
            }
    __syncthreads();
    float* sl = a.slab + (int64_t)bb * 9 * a.Cout * CIN;
    for (int i = tid; i < NV; i += 256) {
        const int g = i / (CIN * 9), c = (i / 9) % CIN, t = i % 9;
        const int co = cg * COG + g;
        if (co < a.Cout)
            sl[((int64_t)t * a.Cout + co) * CIN + c] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    }
}

template <int KS, int PW, bool S2D = false>
static void launch_wgrad(const WgArgs& a, int64_t blocks, hipStream_t st) {
    using C = WgCfg<KS, PW>;
    auto kern = conv_wgrad_kernel<KS, PW, S2D>;
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  C::LDS_BYTES);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256 * KS), C::LDS_BYTES, st, a);
}

static void wgrad_plan(int B, int Cin, int Cout, int H, int W, int ks, int& pw, int& splitK,
                       int& stripsX, int& stripsY) {
    pw = (W > 16) ? 32 : 16;
    const int pr = 64 / pw;
    stripsX = cdiv(W, pw);
    stripsY = cdiv(H, pr);
    const int64_t nunits = (int64_t)B * stripsX * stripsY;
    const int64_t tiles = (int64_t)cdiv(Cout, 64) * cdiv(Cin, 64);
    int64_t s = (1024 + tiles - 1) / tiles;
    // keep the slab under ~192 MB
    const int64_t per = (int64_t)ks * ks * Cout * Cin * 4;
    const int64_t cap = (192ll << 20) / (per > 0 ? per : 1);
    if (s > cap) s = cap;
    if (s > nunits) s = nunits;
    if (s < 1) s = 1;
    splitK = (int)s;
}

extern "C" {

int onet_conv3x3_pack_weights(const float* w, float* wp_fwd, float* wp_dgrad, int Cout, int Cin,
                              void* stream) {
    ONET_REQUIRE(w && Cout > 0 && Cin > 0, "pack3x3: bad args");
    const int64_t n = (int64_t)Cout * Cin * 9;
    hipLaunchKernelGGL(pack3x3_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n, 256), 4096)), dim3(256), 0,
                       as_stream(stream), w, wp_fwd, wp_dgrad, Cout, Cin);
    return check_launch("pack3x3_kernel");
}

int onet_convT2x2_pack_weights(const float* w, float* wp_fwd, float* wp_dgrad, int Cin, int Cout,
                               void* stream) {
    ONET_REQUIRE(w && Cout > 0 && Cin > 0, "packT2x2: bad args");
    const int64_t n = (int64_t)Cout * Cin * 4;
    hipLaunchKernelGGL(packT2x2_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n, 256), 4096)), dim3(256), 0,
                       as_stream(stream), w, wp_fwd, wp_dgrad, (float*)nullptr, Cin, Cout);
    return check_launch("packT2x2_kernel");
}

int onet_convT2x2_pack_weights_fused(const float* w, float* wq, int Cin, int Cout, void* stream) {
    ONET_REQUIRE(w && wq && Cout > 0 && Cin > 0, "packT2x2_fused: bad args");
    const int64_t n = (int64_t)Cout * Cin * 4;
    hipLaunchKernelGGL(packT2x2_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n, 256), 4096)), dim3(256), 0,
                       as_stream(stream), w, (float*)nullptr, (float*)nullptr, wq, Cin, Cout);
    return check_launch("packT2x2_kernel");
}

// ONET_CONVT_GEMM=0 keeps the 64-row direct kernels for the ConvTranspose2d GEMMs (A/B runs)
static bool convt_gemm_enabled() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("ONET_CONVT_GEMM"); on = (e && e[0] == '0') ? 0 : 1; }
    return on != 0;
}

int64_t onet_convT2x2_wgrad_ws_bytes(int B, int Cin, int Ct, int h, int w) {
    return std::max<int64_t>(onet_conv_wgrad_ws_bytes(B, Cin, 4 * Ct, h, w, 1), convt_gemm_wgrad_ws_bytes(B, Cin, Ct, h, w));
}

int onet_convT2x2_fwd(const float* x, int64_t x_bs, const float* wq, const float* bias, float* y, int64_t y_bs,
                      int B, int Cin, int Ct, int h, int w, int Ho, int Wo, int pt, int pl, int operand_bf16, void* stream) {
    ONET_REQUIRE(x && wq && y, "convT2x2_fwd: null pointer");
    ONET_REQUIRE(B > 0 && Cin > 0 && Ct > 0 && h > 0 && w > 0, "convT2x2_fwd: bad shape");
    ONET_REQUIRE(pt >= 0 && pl >= 0 && pt + 2 * h <= Ho && pl + 2 * w <= Wo, "convT2x2_fwd: window outside plane");
    ONET_REQUIRE(x_bs >= (int64_t)Cin * h * w && y_bs >= (int64_t)Ct * Ho * Wo, "convT2x2_fwd: batch stride too small");
    if (convt_gemm_enabled()) {
        const int rc = convt_gemm_fwd(x, x_bs, wq, bias, y, y_bs, nullptr, 0, B, Cin, Ct, h, w, Ho, Wo, pt, pl, operand_bf16, as_stream(stream));
        if (rc <= 0) return rc;          // 1: shape outside the 128 x 128 GEMM's fast path
    }
    ConvArgs a{x, x_bs, wq, y, y_bs, nullptr, B, Cin, 4 * Ct, h, w, 0, 0, 0, bias, Ho, Wo, pt, pl};
    return (w > 16) ? launch_fwd<1, 2, 2, 1, 4, 32, 1>(a, as_stream(stream))
                    : launch_fwd<1, 2, 2, 1, 4, 16, 1>(a, as_stream(stream));
}

// ... with the up-sampled tensor written PRE-SPLIT (fp16 hi | mid slots, conv_split.hip) into yP [B][Ct/8][Ho][2][Wo][8], e.g. the
// up-sampled channel groups of a pre-split concat buffer (yP_bs: batch stride in 4-byte units); no fp32 output.  Fast path only:
// returns 1 (nothing done) elsewhere.
int onet_convT2x2_fwd_p(const float* x, int64_t x_bs, const float* wq, const float* bias, void* yP, int64_t yP_bs, const void* y_amax,
                        int nparts, int B, int Cin, int Ct, int h, int w, int Ho, int Wo, int pt, int pl, int operand_bf16, void* stream) {
    ONET_REQUIRE(x && wq && yP && B > 0 && Cin > 0 && Ct > 0 && h > 0 && w > 0 && (nparts == 1 || nparts == 2), "convT2x2_fwd_p: bad args");
    return convt_gemm_fwd(x, x_bs, wq, bias, nullptr, 0, yP, yP_bs, B, Cin, Ct, h, w, Ho, Wo, pt, pl, operand_bf16, as_stream(stream),
                          nparts == 1 ? 2 : 1, y_amax);
}

// the bound of |ConvTranspose2d(x) + bias| from the weights (nn layout [Cin][Ct][2][2]) and the exact max |x| (x_amax), into y_amax
int onet_convT2x2_out_bound(const float* w, const float* bias, int Cin, int Ct, const void* x_amax, void* y_amax, void* stream) {
    ONET_REQUIRE(w && x_amax && y_amax && Cin > 0 && Ct > 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0, "convT2x2_out_bound: bad args");
    return convt_out_bound(w, bias, Cin, Ct, x_amax, y_amax, as_stream(stream));
}

int onet_convT2x2_dgrad(const float* dy, int64_t dy_bs, const float* wp_dgrad, float* dx, int64_t dx_bs, int B,
                        int Cin, int Ct, int h, int w, int Ho, int Wo, int pt, int pl, int operand_bf16, void* stream) {
    ONET_REQUIRE(dy && wp_dgrad && dx, "convT2x2_dgrad: null pointer");
    ONET_REQUIRE(B > 0 && Cin > 0 && Ct > 0 && h > 0 && w > 0, "convT2x2_dgrad: bad shape");
    ONET_REQUIRE(Ct % CI_T == 0, "convT2x2_dgrad: Ct must be a multiple of %d (use onet_space_to_depth2 + onet_conv_fwd)", CI_T);
    ONET_REQUIRE(pt >= 0 && pl >= 0 && pt + 2 * h <= Ho && pl + 2 * w <= Wo, "convT2x2_dgrad: window outside plane");
    ONET_REQUIRE(dy_bs >= (int64_t)Ct * Ho * Wo && dx_bs >= (int64_t)Cin * h * w, "convT2x2_dgrad: batch stride too small");
    if (convt_gemm_enabled()) {
        const int rc = convt_gemm_dgrad(dy, dy_bs, wp_dgrad, dx, dx_bs, nullptr, nullptr, B, Cin, Ct, h, w, Ho, Wo, pt, pl, operand_bf16, as_stream(stream));
        if (rc <= 0) return rc;
    }
    ConvArgs a{dy, dy_bs, wp_dgrad, dx, dx_bs, nullptr, B, 4 * Ct, Cin, h, w, 0, 0, 0, nullptr, Ho, Wo, pt, pl};
    return (w > 16) ? launch_fwd<1, 2, 2, 1, 4, 32, 2>(a, as_stream(stream))
                    : launch_fwd<1, 2, 2, 1, 4, 16, 2>(a, as_stream(stream));
}

int64_t onet_convT2x2_dgrad_dbias_ws_bytes(int B, int Ct, int h, int w) { return convt_gemm_dbias_ws_bytes(B, Ct, h, w); }

// dx AND dbias of ConvTranspose2d(k=2, s=2): on the fast path of convt_gemm.hip the bias gradient is summed from the dy rows the
// dgrad GEMM stages anyway (one read of the concat gradient's upper half less); returns 1 (and does nothing) when the shape is
// outside that path -- the caller then uses onet_convT2x2_dgrad + onet_convT2x2_dbias
int onet_convT2x2_dgrad_dbias(const float* dy, int64_t dy_bs, const float* wp_dgrad, float* dx, int64_t dx_bs, float* dbias, void* ws,
                              int64_t ws_bytes, int B, int Cin, int Ct, int h, int w, int Ho, int Wo, int pt, int pl, int operand_bf16, void* stream) {
    ONET_REQUIRE(dy && wp_dgrad && dx && dbias && ws, "convT2x2_dgrad_dbias: null pointer");
    ONET_REQUIRE(B > 0 && Cin > 0 && Ct > 0 && h > 0 && w > 0, "convT2x2_dgrad_dbias: bad shape");
    ONET_REQUIRE(dy_bs >= (int64_t)Ct * Ho * Wo && dx_bs >= (int64_t)Cin * h * w, "convT2x2_dgrad_dbias: batch stride too small");
    if (!convt_gemm_enabled() || ws_bytes < convt_gemm_dbias_ws_bytes(B, Ct, h, w)) return 1;
    return convt_gemm_dgrad(dy, dy_bs, wp_dgrad, dx, dx_bs, dbias, (float*)ws, B, Cin, Ct, h, w, Ho, Wo, pt, pl, operand_bf16, as_stream(stream));
}

int onet_conv_fwd_nparts(int B, int Cout, int H, int W) {
    (void)B; (void)Cout; (void)H; (void)W;
    return 0;  // fused BN partials not emitted by this build; use onet_bn_stats_partial
}

int onet_conv_fwd(const float* x, int64_t x_bs, const float* wp, float* z, int64_t z_bs, float* bn_part,
                  int B, int Cin, int Cout, int H, int W, int ks, void* stream) {
    ONET_REQUIRE(x && wp && z, "conv_fwd: null pointer");
    ONET_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "conv_fwd: bad shape B=%d Cin=%d Cout=%d H=%d W=%d",
                 B, Cin, Cout, H, W);
    ONET_REQUIRE(ks == 1 || ks == 3, "conv_fwd: ks must be 1 or 3 (got %d)", ks);
    ONET_REQUIRE(bn_part == nullptr, "conv_fwd: fused BN partials not available in this build");
    ONET_REQUIRE(x_bs >= (int64_t)Cin * H * W && z_bs >= (int64_t)Cout * H * W, "conv_fwd: batch stride too small");
    ConvArgs a{x, x_bs, wp, z, z_bs, bn_part, B, Cin, Cout, H, W, 0, 0, 0};
    return ks == 3 ? dispatch_fwd<3>(a, as_stream(stream)) : dispatch_fwd<1>(a, as_stream(stream));
}

static inline bool use_stem_wgrad(int Cin, int ks) { return ks == 3 && Cin <= 4; }
static inline int stem_blocks(int B, int H) { return B * ((H + STEM_ROWS - 1) / STEM_ROWS); }

int64_t onet_conv_wgrad_ws_bytes(int B, int Cin, int Cout, int H, int W, int ks) {
    if (use_stem_wgrad(Cin, ks)) return (int64_t)stem_blocks(B, H) * 9 * Cout * Cin * 4;
    int pw, splitK, sx, sy;
    wgrad_plan(B, Cin, Cout, H, W, ks, pw, splitK, sx, sy);
    return (int64_t)splitK * ks * ks * Cout * Cin * 4;
}

int onet_conv3x3_stem_wgrad_bn(const float* x, int64_t x_bs, const float* da, int64_t da_bs, const float* z, int64_t z_bs, const float* save,
                               const float* coef, int group_images, float* dw, void* ws, int64_t ws_bytes, int B, int Cin, int Cout, int H,
                               int W, int accumulate, int twin_B, float twin_bias, void* stream) {
    ONET_REQUIRE(x && da && z && save && dw && ws, "conv3x3_stem_wgrad_bn: null pointer");
    ONET_REQUIRE(B > 0 && Cin > 0 && Cin <= 4 && Cout > 0 && H > 0 && W > 0 && group_images >= 0 && (group_images == 0 || B % group_images == 0),
                 "conv3x3_stem_wgrad_bn: bad shape (Cin <= 4)");
    WgArgs a{x, x_bs, da, da_bs, (float*)ws, B, Cin, Cout, H, W, 1, cdiv(Cout, 64), 1, 1, 1};
    a.bn_z = z;
    a.bn_z_bs = z_bs;
    a.bn_save = save;
    a.bn_coef = coef;
    a.bn_gimg = group_images;
    ONET_REQUIRE(twin_B == 0 || 2 * twin_B == B, "conv3x3_stem_wgrad_bn: a twin batch holds twin_B = B / 2 images in memory");
    a.twin_B = twin_B;
    a.twin_bias = twin_bias;
    const int nb = stem_blocks(B, H);
    const int64_t need = (int64_t)nb * 9 * Cout * Cin * 4;
    ONET_REQUIRE(ws_bytes >= need, "conv3x3_stem_wgrad_bn: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)need);
    hipStream_t st = as_stream(stream);
    switch (Cin) {
        case 1: hipLaunchKernelGGL((conv3x3_stem_wgrad_kernel<1, 4, true>), dim3(nb * cdiv(Cout, 4)), dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL((conv3x3_stem_wgrad_kernel<2, 2, true>), dim3(nb * cdiv(Cout, 2)), dim3(256), 0, st, a); break;
        case 3: hipLaunchKernelGGL((conv3x3_stem_wgrad_kernel<3, 2, true>), dim3(nb * cdiv(Cout, 2)), dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL((conv3x3_stem_wgrad_kernel<4, 1, true>), dim3(nb * cdiv(Cout, 1)), dim3(256), 0, st, a); break;
    }
    int rc = check_launch("conv3x3_stem_wgrad_kernel");
    if (rc) return rc;
    const int64_t n = (int64_t)Cout * Cin;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv(n, 64), 9u), dim3(256), 0, st, (const float*)ws, dw, nb, 9, Cout, Cin, 0, accumulate);
    return check_launch("wgrad_reduce_kernel");
}

int onet_conv_wgrad(const float* x, int64_t x_bs, const float* dz, int64_t dz_bs, float* dw, void* ws,
                    int64_t ws_bytes, int B, int Cin, int Cout, int H, int W, int ks, int out_layout,
                    int accumulate, void* stream) {
    ONET_REQUIRE(x && dz && dw && ws, "conv_wgrad: null pointer");
    ONET_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "conv_wgrad: bad shape");
    ONET_REQUIRE(ks == 1 || ks == 3, "conv_wgrad: ks must be 1 or 3 (got %d)", ks);
    ONET_REQUIRE(out_layout == 0 || (out_layout == 1 && ks == 1 && Cout % 4 == 0), "conv_wgrad: bad out_layout");
    WgArgs a{x, x_bs, dz, dz_bs, (float*)ws, B, Cin, Cout, H, W, cdiv(Cin, 64), cdiv(Cout, 64), 1, 1, 1};
    if (use_stem_wgrad(Cin, ks) && out_layout == 0) {
        const int nb = stem_blocks(B, H);
        const int64_t need = (int64_t)nb * 9 * Cout * Cin * 4;
        ONET_REQUIRE(ws_bytes >= need, "conv_wgrad(stem): workspace %lld < %lld bytes", (long long)ws_bytes, (long long)need);
        hipStream_t st = as_stream(stream);
        switch (Cin) {
            case 1: hipLaunchKernelGGL((conv3x3_stem_wgrad_kernel<1, 4>), dim3(nb * cdiv(Cout, 4)), dim3(256), 0, st, a); break;
            case 2: hipLaunchKernelGGL((conv3x3_stem_wgrad_kernel<2, 2>), dim3(nb * cdiv(Cout, 2)), dim3(256), 0, st, a); break;
            case 3: hipLaunchKernelGGL((conv3x3_stem_wgrad_kernel<3, 2>), dim3(nb * cdiv(Cout, 2)), dim3(256), 0, st, a); break;
            default: hipLaunchKernelGGL((conv3x3_stem_wgrad_kernel<4, 1>), dim3(nb * cdiv(Cout, 1)), dim3(256), 0, st, a); break;
        }
        int rc = check_launch("conv3x3_stem_wgrad_kernel");
        if (rc) return rc;
        const int64_t n = (int64_t)Cout * Cin;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv(n, 64), 9u), dim3(256), 0, st,
                           (const float*)ws, dw, nb, 9, Cout, Cin, 0, accumulate);
        return check_launch("wgrad_reduce_kernel");
    }
    int pw;
    wgrad_plan(B, Cin, Cout, H, W, ks, pw, a.splitK, a.stripsX, a.stripsY);
    const int64_t need = (int64_t)a.splitK * ks * ks * Cout * Cin * 4;
    ONET_REQUIRE(ws_bytes >= need, "conv_wgrad: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)need);
    const int64_t blocks = (int64_t)a.splitK * a.ciTiles * a.coTiles;
    hipStream_t st = as_stream(stream);
    if (ks == 3) {
        if (pw == 32) launch_wgrad<3, 32>(a, blocks, st); else launch_wgrad<3, 16>(a, blocks, st);
    } else {
        if (pw == 32) launch_wgrad<1, 32>(a, blocks, st); else launch_wgrad<1, 16>(a, blocks, st);
    }
    int rc = check_launch("conv_wgrad_kernel");
    if (rc) return rc;
    const int64_t n = (int64_t)Cout * Cin;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv(n, 64), (unsigned)(ks * ks)), dim3(256), 0, st,
                       (const float*)ws, dw, a.splitK, ks * ks, Cout, Cin, out_layout, accumulate);
    return check_launch("wgrad_reduce_kernel");
}

int onet_convT2x2_wgrad(const float* x, int64_t x_bs, const float* dy, int64_t dy_bs, float* dw, void* ws,
                        int64_t ws_bytes, int B, int Cin, int Ct, int h, int w, int Ho, int Wo, int pt, int pl, int operand_bf16,
                        void* stream) {
    ONET_REQUIRE(x && dy && dw && ws, "convT2x2_wgrad: null pointer");
    ONET_REQUIRE(B > 0 && Cin > 0 && Ct > 0 && h > 0 && w > 0, "convT2x2_wgrad: bad shape");
    ONET_REQUIRE(Ct % 64 == 0, "convT2x2_wgrad: Ct must be a multiple of 64 (use onet_space_to_depth2 + onet_conv_wgrad)");
    ONET_REQUIRE(pt >= 0 && pl >= 0 && pt + 2 * h <= Ho && pl + 2 * w <= Wo, "convT2x2_wgrad: window outside plane");
    ONET_REQUIRE(dy_bs >= (int64_t)Ct * Ho * Wo && x_bs >= (int64_t)Cin * h * w, "convT2x2_wgrad: batch stride too small");
    if (convt_gemm_enabled()) {
        const int rc = convt_gemm_wgrad(x, x_bs, dy, dy_bs, dw, ws, ws_bytes, B, Cin, Ct, h, w, Ho, Wo, pt, pl, operand_bf16, as_stream(stream));
        if (rc <= 0) return rc;
    }
    const int Cout = 4 * Ct;
    WgArgs a{x, x_bs, dy, dy_bs, (float*)ws, B, Cin, Cout, h, w, cdiv(Cin, 64), cdiv(Cout, 64), 1, 1, 1, Ct, Ho, Wo, pt, pl};
    int pw;
    wgrad_plan(B, Cin, Cout, h, w, 1, pw, a.splitK, a.stripsX, a.stripsY);
    const int64_t need = (int64_t)a.splitK * Cout * Cin * 4;
    ONET_REQUIRE(ws_bytes >= need, "convT2x2_wgrad: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)need);
    const int64_t blocks = (int64_t)a.splitK * a.ciTiles * a.coTiles;
    hipStream_t st = as_stream(stream);
    if (pw == 32) launch_wgrad<1, 32, true>(a, blocks, st); else launch_wgrad<1, 16, true>(a, blocks, st);
    int rc = check_launch("conv_wgrad_kernel");
    if (rc) return rc;
    const int64_t n = (int64_t)Cout * Cin;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv(n, 64), 1u), dim3(256), 0, st, (const float*)ws, dw,
                       a.splitK, 1, Cout, Cin, 1, 0);
    return check_launch("wgrad_reduce_kernel");
}

}  // extern "C"
