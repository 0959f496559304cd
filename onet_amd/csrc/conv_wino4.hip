// K1b: 3x3 convolution forward / dgrad by Winograd F(4x4, 3x3) on the fp32 matrix cores.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A     with 6x6 input tiles, 4x4 output tiles: 36 multiplies per 16
//   outputs = 4x fewer than the direct form (F(2x2,3x3): 2.25x), Lavin & Gray 2016, interpolation points
//   {0, +-1, +-2}.  fp32 error per layer ~2.7e-6 rms of the output scale (F(2x2): 4e-7, direct: 2e-7; measured
//   against fp64 at Cin = 512), two orders below the 1e-3 parity bar.
//
// Same call sites as conv_wino.hip (F.conv2d + input-grad, OV:47,51).  The 36 Winograd positions are 36
// GEMMs  M_pos[co][tile] = sum_ci U_pos[co][ci] * V_pos[ci][tile]; the accumulators of a 64-channel x 32-tile
// block (36 x 2 MFMA tiles of 32x32 = 288 KB) fill over half of the CU's register file, so the block is
// 8 waves = 4 position groups (3x3 positions each: rows {0,1,2}|{3,4,5} x columns {0,1,2}|{3,4,5}) x 2 channel
// halves, 9 accumulators (144 VGPRs) per wave, two waves per SIMD.
// * a position group needs only a 5x5 corner of the 6x6 input patch and half of B^T: 15 LDS reads and
//   48 VALU ops per K-step of 9 MFMAs;
// * weights are pre-transformed once per optimizer step (packed [ci][36][co]) and staged by LDS-DMA;
// * K-loop: the register-level software pipeline of conv_wino.hip (operands of step s+1 are read and
//   transformed under the MFMAs of step s; the chunk barrier sits inside a step; weight DMA in slices);
//   the halo input dwords are prefetched TWO chunks ahead in two register sets (HBM latency);
// * epilogue: each wave applies its quarter of A^T . A (linear in the positions) and the four partial 4x4
//   outputs are summed through LDS, two accumulator registers per pass; float4 row stores.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "common.hpp"

using namespace onet;

// ONET_W4_ASYM: the two waves of a SIMD (position groups pg and pg + 2: RH = 0 / 1) run the same program in lockstep,
// so whatever stalls one stalls both and the matrix pipe idles.  1: weight DMA pieces of a chunk issued by the RH = 0 waves in
// step A and by the RH = 1 waves in step B (same-box +2 % on the kernel).  (Moving the RH = 1 waves' input dwords to step A as
// well -- one VMEM burst per wave and step -- was slower: 1.109 vs 1.066 ms per launch.)
#ifndef ONET_W4_ASYM
#define ONET_W4_ASYM 1
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned OOB4 = 0x80000000u;

static __device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* base, int64_t bytes) {
    const int n = bytes > 0x7fffffffll ? 0x7fffffff : (int)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, n, 0x00020000);
}
static __device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
// LDS-DMA, inline asm for the reason given in conv_wino.hip (the compiler must not see it)
static __device__ __forceinline__ void dma16(i32x4 rsrc, unsigned lds_byte_addr, unsigned voff) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(lds_byte_addr)
                 : "memory");
}
static __device__ __forceinline__ i32x4 rsrc_words(const void* base, int64_t bytes) {
    const uint64_t p = reinterpret_cast<uint64_t>(base);
    i32x4 r;
    r.x = (int)(p & 0xffffffffu);
    r.y = (int)((p >> 32) & 0xffffu);
    r.z = bytes > 0x7fffffffll ? 0x7fffffff : (int)bytes;
    r.w = 0x00020000;
    return r;
}
static __device__ __forceinline__ unsigned lds_addr(const float* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)p;
}

// ------------------------------------------------------------------ weight transform + packing
// U = G g G^T (6x6),  G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
static __device__ __forceinline__ void g3(const float a, const float b, const float c, float (&o)[6]) {
    const float s = a + c;
    o[0] = 0.25f * a;
    o[1] = (-1.f / 6.f) * (s + b);
    o[2] = (-1.f / 6.f) * (s - b);
    const float q = (1.f / 24.f) * a + (1.f / 6.f) * c;
    o[3] = q + (1.f / 12.f) * b;
    o[4] = q - (1.f / 12.f) * b;
    o[5] = c;
}
static __device__ __forceinline__ void wino4_G(const float g[3][3], float U[6][6]) {
    float r[6][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float o[6];
        g3(g[0][j], g[1][j], g[2][j], o);
#pragma unroll
        for (int i = 0; i < 6; ++i) r[i][j] = o[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) g3(r[i][0], r[i][1], r[i][2], U[i]);
}

// One thread per (co, ci) pair.  The two packed layouts want opposite thread orders for coalesced stores (wf rows run
// along co, wd rows along ci), so the kernel is launched once per layout: CO_FAST = 1 maps consecutive threads to
// consecutive co (wf), 0 to consecutive ci (wd).  (One launch for both wrote wf at a stride of 36*Cout floats:
// 1.1 TB/s.)
__global__ void pack3x3_wino4_kernel(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wd,
                                     int Cout, int Cin, int co_fast) {
    const int64_t n = (int64_t)Cout * Cin;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = co_fast ? (int)(i / Cout) : (int)(i % Cin);
        const int co = co_fast ? (int)(i % Cout) : (int)(i / Cin);
        const float* wp = w + ((int64_t)co * Cin + ci) * 9;
        float g[3][3], gr[3][3], U[6][6];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                g[a][b] = wp[a * 3 + b];
                gr[2 - a][2 - b] = g[a][b];
            }
        if (wf) {
            wino4_G(g, U);
#pragma unroll
            for (int p = 0; p < 36; ++p) wf[((int64_t)ci * 36 + p) * Cout + co] = U[p / 6][p % 6];
        }
        if (wd) {
            wino4_G(gr, U);
#pragma unroll
            for (int p = 0; p < 36; ++p) wd[((int64_t)co * 36 + p) * Cin + ci] = U[p / 6][p % 6];
        }
    }
}

// ------------------------------------------------------------------ forward / dgrad
struct Wino4Args {
    const float* x;
    int64_t x_bs;
    const float* wq;      // [Cin][36][Cout]
    float* z;
    int64_t z_bs;
    int B, Cin, Cout, H, W, tilesX, tilesY, imgGroups, coTiles;
    float* stats;         // EP 1: BatchNorm partials [Cout][B * tilesX * tilesY][3] = (n, mean, M2) per block
                          // EP 2: BatchNorm-backward partials [Cout][B * tilesX * tilesY][2] = (sum dy, sum dy * xhat)
    const float* br_z;    // EP 2: pre-activation of the layer whose activation gradient this launch produces
    int64_t br_z_bs;
    const float* br_save; // EP 2: [groups][4][Cout] = (mean, invstd, scale, shift) per statistics group
    int br_group;         // EP 2: images per statistics group
};

template <int TXB>
struct W4Cfg {
    static constexpr int TYB = 4;                          // tile rows per block
    static constexpr int IMG = 32 / (TXB * TYB);           // images per block (2 for the 16-px-wide layout)
    static constexpr int PXW = 4 * TXB, PXH = 4 * TYB;
    static constexpr int IN_ROWS = PXH + 2, IN_COLS = PXW + 2;
    static constexpr int RS = (TXB == 8) ? 40 : 24;        // LDS row stride (floats, even)
    static constexpr int IMG_STRIDE = IN_ROWS * RS;
    static constexpr int CH_STRIDE = IMG * IMG_STRIDE;
    static constexpr int CI_T = 4, CO_T = 64, NTHR = 512;
    static constexpr int W_FLOATS = CI_T * 36 * CO_T;      // 9216 floats = 36 LDS-DMA pieces of 1 KB
    static constexpr int IN_LOGICAL = CI_T * IMG * IN_ROWS * IN_COLS;
    static constexpr int NIN = (IN_LOGICAL + NTHR - 1) / NTHR;
    static constexpr int BUF_FLOATS = W_FLOATS + CI_T * CH_STRIDE;
    static constexpr int EX_LANE = 36;                     // exchange: 32 values per lane, lane stride 36 floats (144 B): 16-B
    static constexpr int EX_FLOATS = 6 * 64 * EX_LANE;     // aligned and the 16 lanes of a b128 group land on 16 distinct 4-bank sets
    static constexpr int LDS_FLOATS = (2 * BUF_FLOATS > 2 * EX_FLOATS) ? 2 * BUF_FLOATS : 2 * EX_FLOATS;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;
};

// half of B^T (input transform): HALF 0 -> rows 0,1,2 from d0..d4 ; HALF 1 -> rows 3,4,5 from d1..d5 (passed as d[0..4])
template <int HALF>
static __device__ __forceinline__ void bt3(const float d0, const float d1, const float d2, const float d3, const float d4,
                                    float& o0, float& o1, float& o2) {
    if constexpr (HALF == 0) {
        o0 = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
        const float a = fmaf(-4.f, d2, d4), b = fmaf(-4.f, d1, d3);
        o1 = a + b;
        o2 = a - b;
    } else {
        const float c = d3 - d1, e = d2 - d0;
        o0 = fmaf(2.f, e, c);
        o1 = fmaf(-2.f, e, c);
        o2 = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
    }
}
// half of A^T (output transform): HALF 0 -> columns 0,1,2 of A^T ; HALF 1 -> columns 3,4,5
template <int HALF>
static __device__ __forceinline__ void at3(const float m0, const float m1, const float m2, float (&o)[4]) {
    if constexpr (HALF == 0) {
        const float s = m1 + m2, dd = m1 - m2;
        o[0] = m0 + s; o[1] = dd; o[2] = s; o[3] = dd;
    } else {
        const float s = m0 + m1, dd = m0 - m1;
        o[0] = s; o[1] = 2.f * dd; o[2] = 4.f * s; o[3] = fmaf(8.f, dd, m2);
    }
}

struct Patch { f32x4 q[5]; f32x2 h[5]; };     // five patch rows: columns 0..3 and 4..5

// sum over the 32 lanes of a wave half, valid in lanes 16..31 / 48..63: four cyclic rotations inside the rows of 16
// (DPP row_ror), then lane 15 of the even rows added into the odd rows (DPP row_bcast:15)
#define ONET_DPP_ADD(v, ctrl, rmask) \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false))
static __device__ __forceinline__ float half_sum_hi(float v) {
    ONET_DPP_ADD(v, 0x128, 0xf);   // row_ror:8
    ONET_DPP_ADD(v, 0x124, 0xf);   // row_ror:4
    ONET_DPP_ADD(v, 0x122, 0xf);   // row_ror:2
    ONET_DPP_ADD(v, 0x121, 0xf);   // row_ror:1
    ONET_DPP_ADD(v, 0x142, 0xa);   // row_bcast:15 into rows 1 and 3
    return v;
}
#undef ONET_DPP_ADD

template <int TXB, int RH, int CH, int EP = 0>
static __device__ __forceinline__ void wino4_body(const Wino4Args& a, float* smem) {
    using C = W4Cfg<TXB>;
    constexpr int IMG = C::IMG, IN_ROWS = C::IN_ROWS, IN_COLS = C::IN_COLS, RS = C::RS;
    constexpr int CH_STRIDE = C::CH_STRIDE, CO_T = C::CO_T, W_FLOATS = C::W_FLOATS, NTHR = C::NTHR, NIN = C::NIN;
    constexpr int CI_T = C::CI_T;
    constexpr int NPIECE = W_FLOATS / 256;            // 36
    constexpr int NWK = (NPIECE + 7) / 8;             // 5 pieces for waves 0..3, 4 for the rest

    int bid;
    {   // XCD-aware tile order (see conv_mfma.hip)
        const int n = gridDim.x, q = n >> 3, r = n & 7, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tx = bid % a.tilesX;
    bid /= a.tilesX;
    const int ty = bid % a.tilesY;
    bid /= a.tilesY;
    const int bg = bid % a.imgGroups;
    const int coT = bid / a.imgGroups;
    const int co0 = coT * CO_T, y0 = ty * C::PXH, x0 = tx * C::PXW, b0 = bg * IMG;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid & 1;                           // (wid >> 1) = position group = RH * 2 + CH (compile-time here)
    const int l31 = lane & 31, kh = lane >> 5;
    const int img = l31 / (TXB * 4), trem = l31 % (TXB * 4);
    const int tr = trem / TXB, tc = trem % TXB;
    const int HW = a.H * a.W;

    f32x16 acc[9];                                    // position (3RH + i, 3CH + j) -> acc[i*3 + j]
#pragma unroll
    for (int p = 0; p < 9; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    const int a_idx = (kh * 36 + (3 * RH) * 6 + 3 * CH) * CO_T + wm * 32 + l31;
    const int p_idx = W_FLOATS + kh * CH_STRIDE + img * C::IMG_STRIDE + (4 * tr + RH) * RS + 4 * tc;   // 16-B aligned

    const i32x4 wr4 = rsrc_words(a.wq, (int64_t)a.Cin * 36 * a.Cout * 4);
    // one resource over the block's IMG images (Cin % 4 == 0, so only never-consumed trailing prefetches can
    // run past an image's channels; a missing second image is masked per lane)
    const int nimg = (a.B - b0 < IMG) ? a.B - b0 : IMG;
    const __amdgpu_buffer_rsrc_t xr = mk_rsrc(a.x + (int64_t)b0 * a.x_bs, ((int64_t)(nimg - 1) * a.x_bs + (int64_t)a.Cin * HW) * 4);

    unsigned in_off[NIN];
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
        const int i = tid + NTHR * k;
        const int ci = i / (IMG * IN_ROWS * IN_COLS), rem = i % (IMG * IN_ROWS * IN_COLS);
        const int m = rem / (IN_ROWS * IN_COLS), r = (rem % (IN_ROWS * IN_COLS)) / IN_COLS, c = rem % IN_COLS;
        const int yy = y0 - 1 + r, xx = x0 - 1 + c;
        bool ok = (i < C::IN_LOGICAL) && m < nimg && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
#if defined(ONET_W4_ABL) && ONET_W4_ABL == 1       // timing experiment: no halo COLUMNS (a row piece = one aligned 128-byte line instead of three)
        ok = ok && c >= 1 && c <= IN_COLS - 2;
#endif
        in_off[k] = ok ? (unsigned)(((int64_t)m * a.x_bs + ci * HW + yy * a.W + xx) * 4) : OOB4;
    }
    // weight DMA piece q = wid + 8k covers packed rows 4q .. 4q+3 of this block's 64-channel slice
    const int wid_u = __builtin_amdgcn_readfirstlane(wid);
    unsigned w_off0;
    {
        const int row = 4 * wid + (lane >> 4), co = (lane & 15) * 4;
        w_off0 = (co0 + co < a.Cout) ? (unsigned)((row * a.Cout + co0 + co) * 4) : OOB4;
    }
    const unsigned w_kstep = (unsigned)(32 * a.Cout * 4);                 // 8 pieces = 32 packed rows
    const unsigned in_step = (unsigned)(CI_T * HW * 4), w_step = (unsigned)(CI_T * 36 * a.Cout * 4);

    float xin[2][NIN];
    unsigned cin_bytes = 0, cw_bytes = 0;            // offsets of the NEXT chunk to issue (inputs / weights)
    auto issue_in = [&](auto setc) __attribute__((always_inline)) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int k = 0; k < NIN; ++k) xin[S][k] = bload(xr, in_off[k] + cin_bytes);
        cin_bytes += in_step;
    };
    auto issue_in_range = [&](auto setc, int k0, int k1) __attribute__((always_inline)) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int k = 0; k < NIN; ++k)
            if (k >= k0 && k < k1) xin[S][k] = bload(xr, in_off[k] + cin_bytes);
    };
    // weight pieces k in [K0, K1) of the chunk at cw_bytes
    auto issue_w = [&](float* buf, int K0, int K1) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NWK; ++k)
            if (k >= K0 && k < K1 && wid_u + 8 * k < NPIECE)
                dma16(wr4, lds_addr(buf) + (unsigned)((wid_u + 8 * k) * 1024), w_off0 + cw_bytes + k * w_kstep);
    };
    auto commit = [&](auto setc, float* buf) __attribute__((always_inline)) {
        constexpr int S = decltype(setc)::value;
        float* in_lds = buf + W_FLOATS;
#pragma unroll
        for (int k = 0; k < NIN; ++k) {
            const int i = tid + NTHR * k;
            const int ci = i / (IMG * IN_ROWS * IN_COLS), rem = i % (IMG * IN_ROWS * IN_COLS);
            const int m = rem / (IN_ROWS * IN_COLS), r = (rem % (IN_ROWS * IN_COLS)) / IN_COLS, c = rem % IN_COLS;
            if (i < C::IN_LOGICAL) in_lds[ci * CH_STRIDE + m * C::IMG_STRIDE + r * RS + c] = xin[S][k];
        }
    };

    float uC[9], avC[9];                             // operands of the current K-step
    // rows RH .. RH+4 of the 6x6 patch, all six columns, channel 2*CP + kh of buffer BUF: one ds_read_b128 +
    // one ds_read_b64 per row.  With RS == 8 (mod 16) the four 16-lane groups of the b128 read cover the 64
    // banks exactly (tile rows are 4*RS = 32 banks apart, tile columns 4 dwords); the group keeps the five
    // columns CH .. CH+4.  Inline asm: from C++ the compiler drops the unused column and re-merges the rest
    // into ds_read2_b32 pairs (banked modulo 32, twice the LDS cycles).  The reads are retired by the
    // explicit lgkmcnt(0) in front of the row pass.
    const unsigned p_addr = lds_addr(smem) + (unsigned)p_idx * 4u;
    auto lds_patch_row = [&](auto bufc, auto cpc, auto ic, Patch& pt) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value, CP = decltype(cpc)::value, I = decltype(ic)::value;
        constexpr int OFF = (BUF * C::BUF_FLOATS + CP * 2 * CH_STRIDE + I * RS) * 4;
        static_assert(OFF + 24 < 65536, "LDS immediate offset range");
        const unsigned pa = p_addr;                     // (a plain use: asm operands alone do not capture)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pt.q[I]) : "v"(pa), "n"(OFF));
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(pt.h[I]) : "v"(pa), "n"(OFF + 16));
    };
    // retire the asm reads; the patch registers pass through the statement so that no use can be scheduled above it
    auto lds_patch_wait = [&](Patch& pt, float (&d)[5][6]) __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(pt.q[0]), "+v"(pt.q[1]), "+v"(pt.q[2]), "+v"(pt.q[3]), "+v"(pt.q[4]), "+v"(pt.h[0]), "+v"(pt.h[1]),
                       "+v"(pt.h[2]), "+v"(pt.h[3]), "+v"(pt.h[4])
                     :
                     : "memory");
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            d[i][0] = pt.q[i].x; d[i][1] = pt.q[i].y; d[i][2] = pt.q[i].z; d[i][3] = pt.q[i].w;
            d[i][4] = pt.h[i].x; d[i][5] = pt.h[i].y;
        }
    };
    auto lds_a = [&](auto bufc, auto cpc, float (&av)[9]) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value, CP = decltype(cpc)::value;
        const float* a_ptr = smem + a_idx + BUF * C::BUF_FLOATS + CP * 2 * 36 * CO_T;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) av[i * 3 + j] = a_ptr[(i * 6 + j) * CO_T];
    };
    auto xform_rows = [&](const float (&d)[5][6], float (&t)[3][5]) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 5; ++c)
            bt3<RH>(d[0][CH + c], d[1][CH + c], d[2][CH + c], d[3][CH + c], d[4][CH + c], t[0][c], t[1][c], t[2][c]);
    };
    auto xform_cols = [&](const float (&t)[3][5], float (&u)[9]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 3; ++i) bt3<CH>(t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], u[i * 3 + 0], u[i * 3 + 1], u[i * 3 + 2]);
    };
    auto mfma_range = [&](int p0, int p1) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 9; ++p)
            if (p >= p0 && p < p1) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(avC[p], uC[p], acc[p], 0, 0, 0);
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // One K-step: 9 MFMAs on (avC, uC) while the operands at (NBUF, NCP) are read and transformed.
    // MODE 1 (first step of a chunk c in buffer TB^1... see chunk()): commit chunk c+1, barrier, first weight slice of c+2
    // MODE 2 (second step): remaining weight slices of c+2, input dwords of c+3
    auto kstep = [&](auto nbufc, auto ncpc, auto modec, auto curbufc) __attribute__((always_inline)) {
        constexpr int MODE = decltype(modec)::value, CUR = decltype(curbufc)::value;
        float d[5][6], t[3][5], avn[9], un[9];
        __builtin_amdgcn_sched_barrier(0);
        // R1: the ten patch reads of step s+1 under MFMAs 0..2
        using R0 = std::integral_constant<int, 0>;
        using R1 = std::integral_constant<int, 1>;
        using R2 = std::integral_constant<int, 2>;
        using R3 = std::integral_constant<int, 3>;
        using R4 = std::integral_constant<int, 4>;
        Patch pt;
        lds_patch_row(nbufc, ncpc, R0{}, pt);
        lds_patch_row(nbufc, ncpc, R1{}, pt);
        __builtin_amdgcn_sched_barrier(0);
        mfma_range(0, 1);
        __builtin_amdgcn_sched_barrier(0);
        lds_patch_row(nbufc, ncpc, R2{}, pt);
        lds_patch_row(nbufc, ncpc, R3{}, pt);
        __builtin_amdgcn_sched_barrier(0);
        mfma_range(1, 2);
        __builtin_amdgcn_sched_barrier(0);
        lds_patch_row(nbufc, ncpc, R4{}, pt);
        __builtin_amdgcn_sched_barrier(0);
        mfma_range(2, 3);
        __builtin_amdgcn_sched_barrier(0);
        lds_patch_wait(pt, d);
        __builtin_amdgcn_sched_barrier(0);
        // R2: row pass of B^T (30 VALU) under MFMAs 3..5.  Step B also issues the remaining three weight pieces of chunk
        // c+2 here, one per MFMA -- BEFORE the input dwords of chunk c+3 (R3): the vmcnt(NIN) in front of the next
        // barrier relies on those NIN loads being the youngest in the queue.
        float* cur_buf = smem + CUR * C::BUF_FLOATS;
        xform_rows(d, t);
        if constexpr (MODE != 2) {
            mfma_range(3, 6);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                mfma_range(3 + q, 4 + q);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
                __builtin_amdgcn_sched_barrier(0);
#if ONET_W4_ASYM
                // the two waves of a SIMD (position groups pg and pg + 2, i.e. RH = 0 / 1) run this program in lockstep: a
                // weight piece issued by both at once stalls both and the matrix pipe idles; with the RH = 0 waves issuing
                // ALL their pieces of the chunk in step A and the RH = 1 waves theirs in step B, one of the pair keeps
                // issuing MFMAs while the other sits in the DMA issue
                if constexpr (RH == 1) issue_w(cur_buf, 2 * q, 2 * q + 2);   // pieces (0,1), (2,3), (4)
#else
                issue_w(cur_buf, 2 + q, 3 + q);              // NWK == 5: pieces 2, 3, 4
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
#if ONET_W4_ASYM
            if constexpr (RH == 1) cw_bytes += w_step;
#else
            cw_bytes += w_step;
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
        using SETN = std::integral_constant<int, CUR ^ 1>;   // MODE 1: inputs of chunk c+1; MODE 2: set (c+3)&1, freed in MODE 1
        if constexpr (MODE == 1) {
            lds_a(nbufc, ncpc, avn);      // MUST precede the barrier: the DMA issued after it overwrites these weights
            // chunk c lives in CUR; its last operands are in registers once the LDS counter drains (the
            // compiler's lgkmcnt(0) in front of s_barrier), so after the barrier CUR may be overwritten
            commit(SETN{}, smem + (CUR ^ 1) * C::BUF_FLOATS);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIN) : "memory");        // all but the youngest input set: the DMA of c+1 landed
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
        }
        // R3: column pass (18 VALU) under MFMAs 6..8.  The staging VMEM instructions go out ONE OR TWO per MFMA (step A:
        // the first two weight pieces of chunk c+2 after the barrier; step B: the input dwords of chunk c+3): the eight
        // waves of the block share one texture-address path, and a burst of 8 VMEM per wave parks every wave's next
        // MFMA behind it.
        if constexpr (MODE != 1) lds_a(nbufc, ncpc, avn);
        xform_cols(t, un);
        if constexpr (MODE == 1) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                mfma_range(6 + q, 7 + q);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                __builtin_amdgcn_sched_barrier(0);
#if ONET_W4_ASYM
                if constexpr (RH == 0) issue_w(cur_buf, 2 * q, 2 * q + 2);
#else
                if (q < 2) issue_w(cur_buf, q, q + 1);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
#if ONET_W4_ASYM
            if constexpr (RH == 0) cw_bytes += w_step;
#endif
        } else {
            mfma_range(6, 9);
            if constexpr (MODE == 2) {
                issue_in_range(SETN{}, 0, NIN);
                cin_bytes += in_step;
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                if constexpr (MODE == 2) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
            }
        }
#pragma unroll
        for (int p = 0; p < 9; ++p) { uC[p] = un[p]; avC[p] = avn[p]; }
    };
    auto chunk = [&](auto bufc) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value;
        using BC = std::integral_constant<int, BUF>;
        using BN = std::integral_constant<int, BUF ^ 1>;
        kstep(BC{}, I1{}, I1{}, BC{});
        kstep(BN{}, I0{}, std::integral_constant<int, 2>{}, BC{});
    };

    // ---- prologue
    issue_in(I0{});                                   // inputs of chunk 0 -> set 0
    issue_w(smem, 0, NWK);                            // weights of chunk 0 -> buffer 0
    cw_bytes += w_step;
    issue_in(I1{});                                   // inputs of chunk 1 -> set 1
    commit(I0{}, smem);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIN) : "memory");
    __syncthreads();
    issue_w(smem + C::BUF_FLOATS, 0, NWK);            // weights of chunk 1 -> buffer 1
    cw_bytes += w_step;
    issue_in(I0{});                                   // inputs of chunk 2 -> set 0
    {
        float d[5][6], t[3][5];
        Patch pt;
        lds_patch_row(I0{}, I0{}, std::integral_constant<int, 0>{}, pt);
        lds_patch_row(I0{}, I0{}, std::integral_constant<int, 1>{}, pt);
        lds_patch_row(I0{}, I0{}, std::integral_constant<int, 2>{}, pt);
        lds_patch_row(I0{}, I0{}, std::integral_constant<int, 3>{}, pt);
        lds_patch_row(I0{}, I0{}, std::integral_constant<int, 4>{}, pt);
        lds_patch_wait(pt, d);
        lds_a(I0{}, I0{}, avC);
        xform_rows(d, t);
        xform_cols(t, uC);
    }
    const int nch = (a.Cin + CI_T - 1) / CI_T;
    int c = 0;
    for (; c + 2 <= nch; c += 2) {
        chunk(I0{});
        chunk(I1{});
    }
    if (c < nch) chunk(I0{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing (all-zero) staging must land before the LDS is reused
    __syncthreads();
#if defined(ONET_W4_ABL) && ONET_W4_ABL == 3           // timing experiment: no epilogue at all (accumulators kept alive)
    {
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < 9; ++p) s += acc[p][0];
        if (s == 12345.678f) a.z[tid] = s;
        return;
    }
#endif

    // ---- epilogue: partial A^T M A of this position group; the four groups are summed through LDS by the
    // RH = CH = 0 wave of each channel half, two accumulator registers (= 2 channels x 32 tiles x 16 px) per pass
    float* zb[IMG];
#pragma unroll
    for (int m = 0; m < IMG; ++m) zb[m] = a.z + (int64_t)(b0 + m) * a.z_bs;
    const int oy = y0 + 4 * tr, ox = x0 + 4 * tc;
    const bool img_ok = (b0 + img) < a.B;
    const bool vec4 = ((a.W & 3) == 0) && ((a.z_bs & 3) == 0) && (ox + 3 < a.W);
    constexpr int PG = RH * 2 + CH;
    // ST: BatchNorm statistics of this block's 64 channels x (16 x 32) pixels from the final sums, shifted by a
    // pivot (the block's first output of the channel) so that fp32 is enough; merged in fp64 by bn_finalize
    constexpr float st_n = (float)(C::PXW * C::PXH), st_inv = 1.f / st_n;   // full blocks only (host-checked)
    const int64_t st_nblk = (int64_t)a.B * a.tilesX * a.tilesY;
    const int64_t st_blk = ((int64_t)bg * a.tilesY + ty) * a.tilesX + tx;
    // EP 2: this launch is a dgrad whose output is the gradient of the ACTIVATION a = relu(bn(z)) of the layer below
    // (OV:47-49 -> 51): the first BatchNorm-backward pass (sum dy, sum dy * xhat with dy = da * [a > 0]) is taken here
    // from the final sums.  The summing waves fetch the z rows and the channel's coefficients ONE PASS AHEAD (the
    // loads of a pass would otherwise sit on its critical path: +8 % on the launch, measured).
    float4 zn[4];
    float cn[4];
    auto br_issue = [&](int ps_) __attribute__((always_inline)) {
        if constexpr (EP == 2 && PG < 2) {
            const int r_ = 2 * ps_ + PG;
            const int co_ = co0 + wm * 32 + (r_ & 3) + 8 * (r_ >> 2) + 4 * kh;
            const int cc = co_ < a.Cout ? co_ : 0;
            const float* sv = a.br_save + (int64_t)(bg / a.br_group) * 4 * a.Cout;
            cn[0] = sv[cc]; cn[1] = sv[a.Cout + cc]; cn[2] = sv[2 * a.Cout + cc]; cn[3] = sv[3 * a.Cout + cc];
            const float* zp = a.br_z + (int64_t)bg * a.br_z_bs + (int64_t)cc * HW + (int64_t)oy * a.W + ox;
#pragma unroll
            for (int y = 0; y < 4; ++y) zn[y] = *reinterpret_cast<const float4*>(zp + y * a.W);
        }
    };
    br_issue(0);
    // The eight passes run as two rolled halves of four (accumulator registers 8..15 are moved down to 0..7 between
    // them): half the straight-line code.  The kernel's four position-group bodies were 68 KB against a 64 KB
    // instruction cache shared by two CUs, most of it this epilogue.
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int ps4 = 0; ps4 < 4; ++ps4) {
        const int ps = half * 4 + ps4;
        float* ex = smem + (ps4 & 1) * C::EX_FLOATS;
        float yv[2][4][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = 2 * ps4 + h;
            float rowp[4][3];                         // A^T (row half) applied to the 3 position rows, per column j
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float o[4];
                at3<RH>(acc[0 * 3 + j][r], acc[1 * 3 + j][r], acc[2 * 3 + j][r], o);
#pragma unroll
                for (int y = 0; y < 4; ++y) rowp[y][j] = o[y];
            }
#pragma unroll
            for (int y = 0; y < 4; ++y) at3<CH>(rowp[y][0], rowp[y][1], rowp[y][2], yv[h][y]);
        }
        // summing role: group 0 finishes register h = 0 of the pass, group 1 register h = 1; exchange slot 0 holds
        // group 1's h = 0 partial and group 0's h = 1 partial, slots 1 and 2 both partials of groups 2 and 3
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (PG == h) continue;
            constexpr int SLOT = (PG < 2) ? 0 : PG - 1;
#pragma unroll
            for (int y = 0; y < 4; ++y)
                *reinterpret_cast<float4*>(ex + ((SLOT * 2 + wm) * 64 + lane) * C::EX_LANE + h * 16 + y * 4) =
                    make_float4(yv[h][y][0], yv[h][y][1], yv[h][y][2], yv[h][y][3]);
        }
        __syncthreads();
        if constexpr (PG < 2) {
            {
                constexpr int h = PG;
                const int r = 2 * ps + h;
                const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                // EP 2: the z rows and coefficients of this pass were requested one pass ahead (br_issue)
                float4 zr[4];
                float br_mean = 0.f, br_inv = 0.f, br_sc = 0.f, br_sh = 0.f;
                if constexpr (EP == 2) {
#pragma unroll
                    for (int y = 0; y < 4; ++y) zr[y] = zn[y];
                    br_mean = cn[0]; br_inv = cn[1]; br_sc = cn[2]; br_sh = cn[3];
                    if (ps + 1 < 8) br_issue(ps + 1);
                }
#pragma unroll
                for (int y = 0; y < 4; ++y)
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        const float4 v = *reinterpret_cast<const float4*>(ex + ((g * 2 + wm) * 64 + lane) * C::EX_LANE + h * 16 + y * 4);
                        yv[h][y][0] += v.x; yv[h][y][1] += v.y; yv[h][y][2] += v.z; yv[h][y][3] += v.w;
                    }
                if constexpr (EP == 1) {
                    static_assert(EP == 0 || IMG == 1, "fused statistics: one image per block");
                    const int piv = __builtin_bit_cast(int, yv[h][0][0]);
                    const float p0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(piv, 0));
                    const float p1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(piv, 32));
                    const float pv = kh ? p1 : p0;
                    // packed fp32 (v_pk_add_f32 / v_pk_fma_f32): two pixels per VALU op
                    f32x2 q1 = {0.f, 0.f}, q2 = {0.f, 0.f};
                    const f32x2 pv2 = {pv, pv};
#pragma unroll
                    for (int y = 0; y < 4; ++y)
#pragma unroll
                        for (int x = 0; x < 4; x += 2) {
                            const f32x2 v2 = {yv[h][y][x], yv[h][y][x + 1]};
                            const f32x2 d = v2 - pv2;
                            q1 += d;
                            q2 = __builtin_elementwise_fma(d, d, q2);
                        }
                    float s1 = q1.x + q1.y, s2 = q2.x + q2.y;
                    s1 = half_sum_hi(s1);
                    s2 = half_sum_hi(s2);
                    if (l31 == 31 && co < a.Cout) {
                        float* sp = a.stats + ((int64_t)co * st_nblk + st_blk) * 3;
                        sp[0] = st_n;
                        sp[1] = fmaf(s1, st_inv, pv);
                        sp[2] = fmaxf(fmaf(-s1 * st_inv, s1, s2), 0.f);
                    }
                }
                if constexpr (EP == 2) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int y = 0; y < 4; ++y) {
                        const float zz[4] = {zr[y].x, zr[y].y, zr[y].z, zr[y].w};
#pragma unroll
                        for (int x = 0; x < 4; ++x) {
                            const float zc = zz[x] - br_mean;
                            const float g = fmaf(zc, br_sc, br_sh) > 0.f ? yv[h][y][x] : 0.f;   // same mask expression as bn_relu_bwd_*
                            s1 += g;
                            s2 = fmaf(g, zc, s2);
                        }
                    }
                    s1 = half_sum_hi(s1);
                    s2 = half_sum_hi(s2) * br_inv;          // xhat = (z - mean) * invstd
                    if (l31 == 31 && co < a.Cout) {
                        float* sp = a.stats + ((int64_t)co * st_nblk + st_blk) * 2;
                        sp[0] = s1;
                        sp[1] = s2;
                    }
                }
                if (co < a.Cout && img_ok && ox < a.W) {
                    float* o = (IMG == 1 ? zb[0] : (img ? zb[IMG - 1] : zb[0])) + (int64_t)co * HW + (int64_t)oy * a.W + ox;
#pragma unroll
                    for (int y = 0; y < 4; ++y) {
                        if (oy + y < a.H) {
                            if (vec4) {
                                *reinterpret_cast<float4*>(o + y * a.W) = make_float4(yv[h][y][0], yv[h][y][1], yv[h][y][2], yv[h][y][3]);
                            } else {
#pragma unroll
                                for (int x = 0; x < 4; ++x)
                                    if (ox + x < a.W) o[y * a.W + x] = yv[h][y][x];
                            }
                        }
                    }
                }
            }
        }
    }
        if (half == 0) {
#pragma unroll
            for (int p = 0; p < 9; ++p)
#pragma unroll
                for (int r = 0; r < 8; ++r) acc[p][r] = acc[p][r + 8];
        }
    }
}

template <int TXB, int EP = 0>
__global__ __launch_bounds__(512, 2) void conv_wino4_kernel(Wino4Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem4[];
    const int pg = (threadIdx.x >> 6) >> 1;           // wave-uniform
    switch (__builtin_amdgcn_readfirstlane(pg)) {
        case 0: wino4_body<TXB, 0, 0, EP>(a, smem4); break;
        case 1: wino4_body<TXB, 0, 1, EP>(a, smem4); break;
        case 2: wino4_body<TXB, 1, 0, EP>(a, smem4); break;
        default: wino4_body<TXB, 1, 1, EP>(a, smem4); break;
    }
}

template <int TXB, int EP = 0>
static int launch_wino4(Wino4Args a, hipStream_t st) {
    using C = W4Cfg<TXB>;
    a.tilesX = cdiv(a.W, C::PXW);
    a.tilesY = cdiv(a.H, C::PXH);
    a.imgGroups = cdiv(a.B, C::IMG);
    a.coTiles = cdiv(a.Cout, C::CO_T);
    const int64_t blocks = (int64_t)a.imgGroups * a.tilesX * a.tilesY * a.coTiles;
    ONET_REQUIRE(blocks > 0 && blocks < (1ll << 31), "conv_wino4: grid %lld out of range", (long long)blocks);
    auto kern = conv_wino4_kernel<TXB, EP>;
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  C::LDS_BYTES);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(C::NTHR), C::LDS_BYTES, st, a);
    return check_launch("conv_wino4_kernel");
}

extern "C" {

int onet_conv3x3_pack_weights_winograd4(const float* w, float* wq_fwd, float* wq_dgrad, int Cout, int Cin, void* stream) {
    ONET_REQUIRE(w && (wq_fwd || wq_dgrad), "conv3x3_pack_weights_winograd4: null pointer");
    ONET_REQUIRE(Cout > 0 && Cin > 0, "conv3x3_pack_weights_winograd4: bad shape");
    const int64_t n = (int64_t)Cout * Cin;
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
    if (wq_fwd) {
        hipLaunchKernelGGL(pack3x3_wino4_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w, wq_fwd, (float*)nullptr, Cout, Cin, 1);
        int rc = check_launch("pack3x3_wino4_kernel");
        if (rc) return rc;
    }
    if (wq_dgrad)
        hipLaunchKernelGGL(pack3x3_wino4_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w, (float*)nullptr, wq_dgrad, Cout, Cin, 0);
    return check_launch("pack3x3_wino4_kernel");
}

int onet_conv3x3_winograd4_fwd(const float* x, int64_t x_bs, const float* wq, float* z, int64_t z_bs, int B, int Cin,
                               int Cout, int H, int W, void* stream) {
    ONET_REQUIRE(x && wq && z, "conv3x3_winograd4_fwd: null pointer");
    ONET_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "conv3x3_winograd4_fwd: bad shape");
    ONET_REQUIRE((Cout & 3) == 0 && (Cin & 3) == 0, "conv3x3_winograd4_fwd: Cin and Cout must be multiples of 4 (use onet_conv_fwd)");
    ONET_REQUIRE(x_bs >= (int64_t)Cin * H * W && z_bs >= (int64_t)Cout * H * W, "conv3x3_winograd4_fwd: batch stride too small");
    ONET_REQUIRE((x_bs + (int64_t)(Cin + 16) * H * W) * 4 < (1ll << 31) && (int64_t)(Cin + 16) * 36 * Cout * 4 < (1ll << 31),
                 "conv3x3_winograd4_fwd: operand exceeds the 2 GiB buffer-resource range");
    Wino4Args a{x, x_bs, wq, z, z_bs, B, Cin, Cout, H, W, 0, 0, 0, 0, nullptr, nullptr, 0, nullptr, 1};
    return (W > 16) ? launch_wino4<8>(a, as_stream(stream)) : launch_wino4<4>(a, as_stream(stream));
}

int onet_conv3x3_winograd4_nparts(int B, int H, int W) {
    // statistics are emitted by full 16 x 32-pixel blocks only (every U-Net level of a 2^k-sized input down to 32 px)
    if (B <= 0 || H <= 0 || W <= 0 || (W % W4Cfg<8>::PXW) != 0 || (H % W4Cfg<8>::PXH) != 0) return 0;
    const int64_t n = (int64_t)B * cdiv(W, W4Cfg<8>::PXW) * cdiv(H, W4Cfg<8>::PXH);
    return n < (1ll << 31) ? (int)n : 0;
}

int onet_conv3x3_winograd4_fwd_stats(const float* x, int64_t x_bs, const float* wq, float* z, int64_t z_bs, float* part,
                                     int B, int Cin, int Cout, int H, int W, void* stream) {
    ONET_REQUIRE(x && wq && z && part, "conv3x3_winograd4_fwd_stats: null pointer");
    ONET_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "conv3x3_winograd4_fwd_stats: bad shape");
    ONET_REQUIRE((W % W4Cfg<8>::PXW) == 0 && (H % W4Cfg<8>::PXH) == 0,
                 "conv3x3_winograd4_fwd_stats: W %% 32 == 0 and H %% 16 == 0 required (onet_conv3x3_winograd4_nparts() == 0 elsewhere)");
    ONET_REQUIRE((Cout & 3) == 0 && (Cin & 3) == 0, "conv3x3_winograd4_fwd_stats: Cin and Cout must be multiples of 4");
    ONET_REQUIRE(x_bs >= (int64_t)Cin * H * W && z_bs >= (int64_t)Cout * H * W, "conv3x3_winograd4_fwd_stats: batch stride too small");
    ONET_REQUIRE((x_bs + (int64_t)(Cin + 16) * H * W) * 4 < (1ll << 31) && (int64_t)(Cin + 16) * 36 * Cout * 4 < (1ll << 31),
                 "conv3x3_winograd4_fwd_stats: operand exceeds the 2 GiB buffer-resource range");
    Wino4Args a{x, x_bs, wq, z, z_bs, B, Cin, Cout, H, W, 0, 0, 0, 0, part, nullptr, 0, nullptr, 1};
    return launch_wino4<8, 1>(a, as_stream(stream));
}

int onet_conv3x3_winograd4_dgrad_bnreduce(const float* dz, int64_t dz_bs, const float* wq_dgrad, float* da, int64_t da_bs,
                                          const float* z_prev, int64_t z_prev_bs, const float* save_prev, int group_images,
                                          float* part2, int B, int Cdz, int Cda, int H, int W, void* stream) {
    ONET_REQUIRE(dz && wq_dgrad && da && z_prev && save_prev && part2, "conv3x3_winograd4_dgrad_bnreduce: null pointer");
    ONET_REQUIRE(B > 0 && Cdz > 0 && Cda > 0 && H > 0 && W > 0 && group_images > 0 && (B % group_images) == 0,
                 "conv3x3_winograd4_dgrad_bnreduce: bad shape");
    ONET_REQUIRE((W % W4Cfg<8>::PXW) == 0 && (H % W4Cfg<8>::PXH) == 0,
                 "conv3x3_winograd4_dgrad_bnreduce: W %% 32 == 0 and H %% 16 == 0 required (onet_conv3x3_winograd4_nparts() == 0 elsewhere)");
    ONET_REQUIRE((Cda & 3) == 0 && (Cdz & 3) == 0, "conv3x3_winograd4_dgrad_bnreduce: channel counts must be multiples of 4");
    ONET_REQUIRE(dz_bs >= (int64_t)Cdz * H * W && da_bs >= (int64_t)Cda * H * W && z_prev_bs >= (int64_t)Cda * H * W &&
                     (z_prev_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(z_prev) & 15) == 0,
                 "conv3x3_winograd4_dgrad_bnreduce: batch stride too small or z_prev not 16-byte aligned");
    ONET_REQUIRE((dz_bs + (int64_t)(Cdz + 16) * H * W) * 4 < (1ll << 31) && (int64_t)(Cdz + 16) * 36 * Cda * 4 < (1ll << 31),
                 "conv3x3_winograd4_dgrad_bnreduce: operand exceeds the 2 GiB buffer-resource range");
    Wino4Args a{dz, dz_bs, wq_dgrad, da, da_bs, B, Cdz, Cda, H, W, 0, 0, 0, 0, part2, z_prev, z_prev_bs, save_prev, group_images};
    return launch_wino4<8, 2>(a, as_stream(stream));
}

}  // extern "C"
