// K1b: 3x3 convolution forward / dgrad by Winograd F(4x4, 3x3) on the fp32 matrix cores.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A     with 6x6 input tiles, 4x4 output tiles: 36 multiplies per 16
//   outputs = 4x fewer than the direct form (F(2x2,3x3): 2.25x), Lavin & Gray 2016, interpolation points
//   {0, +-1, +-2}.  fp32 error per layer ~2.7e-6 rms of the output scale (F(2x2): 4e-7, direct: 2e-7; measured
//   against fp64 at Cin = 512), two orders below the 1e-3 parity bar.
//
// Call sites: F.conv2d + input-grad, OV:47,51.  The 36 Winograd positions are 36
// GEMMs  M_pos[co][tile] = sum_ci U_pos[co][ci] * V_pos[ci][tile]; the accumulators of a 64-channel x 32-tile
// block (36 x 2 MFMA tiles of 32x32 = 288 KB) fill over half of the CU's register file, so the block is
// 8 waves = 4 position groups (3x3 positions each: rows {0,1,2}|{3,4,5} x columns {0,1,2}|{3,4,5}) x 2 TILE halves,
// 36 accumulators of v_mfma_f32_16x16x4_f32 (144 VGPRs) per wave, two waves per SIMD.
// * a position group needs only a 5x5 corner of the 6x6 input patch and half of B^T: 10 patch reads, 9 ds_read_b128 of
//   weights and 48 VALU ops per K-step of 36 MFMAs (round 2, 32x32x2 MFMAs over channel halves: per 9 MFMAs of twice the cycles);
// * weights are pre-transformed once per optimizer step (packed [ci][36][co padded to 64s, permuted in 64-blocks]) and staged
//   by LDS-DMA into a ring of three buffers, two chunks ahead;
// * K-loop: the inputs of chunk c+1 are read and transformed under the MFMAs of chunk c; the chunk barrier sits after position
//   6; the halo input dwords are prefetched THREE chunks ahead in two register sets (HBM latency);
// * epilogue: each wave applies its quarter of A^T . A (linear in the positions) and the four partial 4x4
//   outputs are summed through LDS, four accumulator registers per pass, every wave finishing one; float4 row stores.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "common.hpp"

using namespace onet;

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
// LDS-DMA as inline asm: the compiler has no builtin that models a buffer load writing LDS at M0 + lane offset, and must not
// reorder LDS accesses around one it cannot see; the callers fence with explicit s_waitcnt vmcnt + barriers
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

// One thread per (row channel, padded packed channel) pair.  Packed rows are [ci][36][CoutP] (wf) / [co][36][CinP] (wd) with
// the row's channel dimension padded to a multiple of 64 (zeros) and PERMUTED inside every block of 64: channel 16 g + i of a
// block sits at float 4 i + g, so that a lane of the kernel's A operand (MFMA row i of the four channel blocks g) reads 16
// contiguous bytes.  The two layouts want opposite thread orders for coalesced reads of w, hence one launch per layout
// (fwd != 0: wf, consecutive threads = consecutive packed positions of one ci).
__global__ void pack3x3_wino4_kernel(const float* __restrict__ w, float* __restrict__ wq, int Cout, int Cin, int fwd) {
    const int R = fwd ? Cin : Cout, Cn = fwd ? Cout : Cin, CnP = (Cn + 63) & ~63;
    const int64_t n = (int64_t)R * CnP;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / CnP), q = (int)(i % CnP);
        const int cn = (q & ~63) + ((q & 63) >> 2) + 16 * (q & 3);          // the channel stored at packed position q
        float U[6][6];
        if (cn < Cn) {
            const int co = fwd ? cn : row, ci = fwd ? row : cn;
            const float* wp = w + ((int64_t)co * Cin + ci) * 9;
            float g[3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    if (fwd) g[a][b] = wp[a * 3 + b];
                    else g[2 - a][2 - b] = wp[a * 3 + b];
                }
            wino4_G(g, U);
        } else {
#pragma unroll
            for (int p = 0; p < 36; ++p) U[p / 6][p % 6] = 0.f;
        }
#pragma unroll
        for (int p = 0; p < 36; ++p) wq[((int64_t)row * 36 + p) * CnP + q] = U[p / 6][p % 6];
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
    // channel stride padded to 0 (TXB 8) / 48 (TXB 4) mod 64 dwords: with 16 tiles x 4 channels per wave the four 16-lane groups
    // of a patch row's ds_read_b128 then cover the 64 banks exactly (brute-forced over all paddings; unpadded: 2-way conflicts)
    static constexpr int CH_STRIDE = IMG * IMG_STRIDE + ((TXB == 8) ? 48 : 16);
    static constexpr int CI_T = 4, CO_T = 64, NTHR = 512;
    static constexpr int W_FLOATS = CI_T * 36 * CO_T;      // 9216 floats = 36 LDS-DMA pieces of 1 KB
    static constexpr int X_FLOATS = CI_T * CH_STRIDE;
    static constexpr int IN_LOGICAL = CI_T * IMG * IN_ROWS * IN_COLS;
    static constexpr int NIN = (IN_LOGICAL + NTHR - 1) / NTHR;
    static constexpr int X_BASE = 3 * W_FLOATS;            // LDS map: weight ring W[3], then input ring X[2]
    static constexpr int KLOOP_FLOATS = X_BASE + 2 * X_FLOATS;
    // epilogue exchange: [tile half 2][finishing group 4][sending slot 3][lane 64] x 9 values at a lane stride of 12 floats
    // (48 B: 16-byte aligned, and the 16 lanes of a b128 group land on 16 distinct 4-bank sets)
    static constexpr int EX_LANE = 12;
    static constexpr int EX_FLOATS = 2 * 4 * 3 * 64 * EX_LANE;
    static constexpr int LDS_FLOATS = (KLOOP_FLOATS > EX_FLOATS) ? KLOOP_FLOATS : EX_FLOATS;
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

// sum over the 16 lanes of a DPP row (four cyclic rotations): every lane of the row ends with the row's sum
#define ONET_DPP_ADD(v, ctrl) \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
static __device__ __forceinline__ float row16_sum(float v) {
    ONET_DPP_ADD(v, 0x128);   // row_ror:8
    ONET_DPP_ADD(v, 0x124);   // row_ror:4
    ONET_DPP_ADD(v, 0x122);   // row_ror:2
    ONET_DPP_ADD(v, 0x121);   // row_ror:1
    return v;
}
#undef ONET_DPP_ADD

// Round 3 structure.  Block = 64 channels x 32 tiles as before, 8 waves = 4 position groups x 2 TILE halves (round 2: x 2 channel
// halves), on v_mfma_f32_16x16x4_f32: a wave owns the 3x3 positions of its group for ALL 64 channels (4 MFMA row blocks of 16) of
// 16 tiles, K = 4 input channels per MFMA, 36 accumulators of 4 registers (144, as before).  A lane transforms the patch of ONE
// (tile, channel) pair per K-step and the result feeds FOUR MFMAs (one per channel block) instead of one: 1.3 transform VALU and
// 0.28 patch reads per MFMA-32-cycles instead of 2.7 and 0.55; every transform is computed once per block instead of twice.  The
// A operands (transformed weights, permuted by the pack so that a lane's four channel blocks are 16 contiguous bytes) come
// just in time, one ds_read_b128 per position two positions ahead, through a ring of three register quads.  A chunk is ONE
// K-step; LDS rings: weights W[3] (DMA, issued two chunks ahead right after the barrier), inputs X[2]; one barrier per chunk,
// placed after position 6 so that the first two A reads of the next chunk hide under positions 7 and 8.
template <int TXB, int RH, int CH, int EP = 0>
static __device__ __forceinline__ void wino4_body(const Wino4Args& a, float* smem) {
    using C = W4Cfg<TXB>;
    constexpr int IMG = C::IMG, IN_ROWS = C::IN_ROWS, IN_COLS = C::IN_COLS, RS = C::RS;
    constexpr int CH_STRIDE = C::CH_STRIDE, CO_T = C::CO_T, W_FLOATS = C::W_FLOATS, NTHR = C::NTHR, NIN = C::NIN;
    constexpr int CI_T = C::CI_T, X_FLOATS = C::X_FLOATS, X_BASE = C::X_BASE;
    constexpr int NPIECE = W_FLOATS / 256;            // 36
    constexpr int NWK = (NPIECE + 7) / 8;             // 5 pieces for waves 0..3, 4 for the rest
    constexpr int PG = RH * 2 + CH;

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
    const int CoutP = (a.Cout + 63) & ~63;            // packed row length (the pack pads the channel dimension to 64s)

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int th = wid & 1;                           // tile half; (wid >> 1) = position group (compile-time here)
    const int l15 = lane & 15, k4 = lane >> 4;
    const int tl = th * 16 + l15;                     // tile of the block this lane transforms / whose outputs it holds
    const int img = tl / (TXB * 4), trem = tl % (TXB * 4);
    const int tr = trem / TXB, tc = trem % TXB;
    const int HW = a.H * a.W;

    f32x4 acc[9][4];                                  // [position (3RH + i, 3CH + j) -> i*3 + j][channel block of 16]
#pragma unroll
    for (int p = 0; p < 9; ++p)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[p][g] = f32x4{0.f, 0.f, 0.f, 0.f};

    // A operand of position (i, j), chunk channel k4: floats [k4][pos][l15 * 4 .. + 3] of a weight buffer = channels l15 + 16 g
    const int a_idx = (k4 * 36 + (3 * RH) * 6 + 3 * CH) * CO_T + l15 * 4;
    const int p_idx = X_BASE + k4 * CH_STRIDE + img * C::IMG_STRIDE + (4 * tr + RH) * RS + 4 * tc;   // 16-B aligned

    const i32x4 wr4 = rsrc_words(a.wq, (int64_t)a.Cin * 36 * CoutP * 4);
    // one resource over the block's IMG images (Cin % 4 == 0, so only never-consumed trailing prefetches can
    // run past an image's channels; a missing second image is masked per lane)
    const int nimg = (a.B - b0 < IMG) ? a.B - b0 : IMG;
    const __amdgpu_buffer_rsrc_t xr = mk_rsrc(a.x + (int64_t)b0 * a.x_bs, ((int64_t)(nimg - 1) * a.x_bs + (int64_t)a.Cin * HW) * 4);

    unsigned in_off[NIN];
    unsigned in_lds[NIN];                             // LDS float index of staged element k inside an X buffer
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
        const int i = tid + NTHR * k;
        const int ci = i / (IMG * IN_ROWS * IN_COLS), rem = i % (IMG * IN_ROWS * IN_COLS);
        const int m = rem / (IN_ROWS * IN_COLS), r = (rem % (IN_ROWS * IN_COLS)) / IN_COLS, c = rem % IN_COLS;
        const int yy = y0 - 1 + r, xx = x0 - 1 + c;
        const bool ok = (i < C::IN_LOGICAL) && m < nimg && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
        in_off[k] = ok ? (unsigned)(((int64_t)m * a.x_bs + ci * HW + yy * a.W + xx) * 4) : OOB4;
        in_lds[k] = (unsigned)(X_BASE + ci * CH_STRIDE + m * C::IMG_STRIDE + r * RS + c);
    }
    // weight DMA piece q = wid + 8k covers packed rows 4q .. 4q+3 of this block's 64-channel slice
    const int wid_u = __builtin_amdgcn_readfirstlane(wid);
    const unsigned w_off0 = (unsigned)(((4 * wid + (lane >> 4)) * CoutP + co0 + (lane & 15) * 4) * 4);
    const unsigned w_kstep = (unsigned)(32 * CoutP * 4);                  // 8 pieces = 32 packed rows
    const unsigned in_step = (unsigned)(CI_T * HW * 4), w_step = (unsigned)(CI_T * 36 * CoutP * 4);

    float xin[2][NIN];
    unsigned cin_bytes = 0, cw_bytes = 0;            // offsets of the NEXT chunk to issue (inputs / weights)
    auto issue_in_range = [&](auto setc, int k0, int k1) __attribute__((always_inline)) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int k = 0; k < NIN; ++k)
            if (k >= k0 && k < k1) xin[S][k] = bload(xr, in_off[k] + cin_bytes);
    };
    // weight pieces k in [K0, K1) of the chunk at cw_bytes into the weight buffer at float offset wbuf (wave-uniform)
    auto issue_w = [&](unsigned wbuf, int K0, int K1) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NWK; ++k)
            if (k >= K0 && k < K1 && wid_u + 8 * k < NPIECE)
                dma16(wr4, lds_addr(smem) + wbuf * 4u + (unsigned)((wid_u + 8 * k) * 1024), w_off0 + cw_bytes + k * w_kstep);
    };
    auto commit = [&](auto setc, auto xbc) __attribute__((always_inline)) {
        constexpr int S = decltype(setc)::value, XB = decltype(xbc)::value;
#pragma unroll
        for (int k = 0; k < NIN; ++k)
            if (tid + NTHR * k < C::IN_LOGICAL) smem[in_lds[k] + XB * X_FLOATS] = xin[S][k];
    };

    float uC[9];                                     // transformed inputs of the current chunk
    f32x4 avA[3];                                    // A ring: slot p % 3 holds the four channel blocks' operands of position p
    // rows RH .. RH+4 of the 6x6 patch, all six columns, of this lane's (tile, channel) in input buffer XB: one ds_read_b128 +
    // one ds_read_b64 per row; the group keeps the five columns CH .. CH+4.  Inline asm: from C++ the compiler drops the unused
    // column and re-merges the rest into ds_read2_b32 pairs (banked modulo 32, twice the LDS cycles).  The reads are retired by
    // the explicit lgkmcnt(0) in front of the row pass.
    const unsigned p_addr = lds_addr(smem) + (unsigned)p_idx * 4u;
    auto lds_patch_row = [&](auto xbc, auto ic, Patch& pt) __attribute__((always_inline)) {
        constexpr int XB = decltype(xbc)::value, I = decltype(ic)::value;
        constexpr int OFF = (XB * X_FLOATS + I * RS) * 4;
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
    // A operands of position P from the weight buffer at float offset wbuf
    auto lds_a = [&](unsigned wbuf, int P) __attribute__((always_inline)) -> f32x4 {
        return *reinterpret_cast<const f32x4*>(smem + wbuf + a_idx + ((P / 3) * 6 + (P % 3)) * CO_T);
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
    // the four MFMAs (channel blocks) of position p with the operands in ring slot p % 3
    auto mfma_pos = [&](auto pc) __attribute__((always_inline)) {
        constexpr int P = decltype(pc)::value;
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[P][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(avA[P % 3][g], uC[P], acc[P][g], 0, 0, 0);
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    unsigned w_rd = 0, w_nx = W_FLOATS, w_dma = 2 * W_FLOATS;     // weight ring: chunk c / c+1 / c+2 (float offsets, wave-uniform)

    // One chunk (= one K-step of 36 MFMAs).  PAR = parity of the chunk index c:
    //   reads the patches of chunk c+1 from X[PAR ^ 1] and transforms them (-> un), commits the inputs of chunk c+2 (register set
    //   PAR ^ 1) to X[PAR], loads the inputs of chunk c+3 into set PAR;  positions 0..2 issue the remaining pieces of DMA(c+1),
    //   positions 7, 8 the first two of DMA(c+2) (after the barrier that retires chunk c-1's weight buffer).
    // VMEM order per wave: [DMA(c+1) 0,1 | previous step] DMA(c+1) 2..4, inputs(c+3) x NIN, << vmcnt(NIN), barrier >>, DMA(c+2) 0,1.
    auto step = [&](auto parc) __attribute__((always_inline)) {
        constexpr int PAR = decltype(parc)::value;
        using XRD = std::integral_constant<int, PAR ^ 1>;
        using XWR = std::integral_constant<int, PAR>;
        using SLD = std::integral_constant<int, PAR>;
        using SCM = std::integral_constant<int, PAR ^ 1>;
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        using P2 = std::integral_constant<int, 2>;
        using P3 = std::integral_constant<int, 3>;
        using P4 = std::integral_constant<int, 4>;
        using P5 = std::integral_constant<int, 5>;
        using P6 = std::integral_constant<int, 6>;
        using P7 = std::integral_constant<int, 7>;
        using P8 = std::integral_constant<int, 8>;
        float d[5][6], t[3][5], un[9];
        Patch pt;
        __builtin_amdgcn_sched_barrier(0);
        // positions 0..2: patch reads of chunk c+1, A reads two positions ahead, the rest of DMA(c+1)
        avA[2] = lds_a(w_rd, 2);
        lds_patch_row(XRD{}, P0{}, pt);
        lds_patch_row(XRD{}, P1{}, pt);
        __builtin_amdgcn_sched_barrier(0);
        mfma_pos(P0{});
        __builtin_amdgcn_sched_barrier(0);
        issue_w(w_nx, 2, 3);
        avA[0] = lds_a(w_rd, 3);
        lds_patch_row(XRD{}, P2{}, pt);
        lds_patch_row(XRD{}, P3{}, pt);
        __builtin_amdgcn_sched_barrier(0);
        mfma_pos(P1{});
        __builtin_amdgcn_sched_barrier(0);
        issue_w(w_nx, 3, 4);
        avA[1] = lds_a(w_rd, 4);
        lds_patch_row(XRD{}, P4{}, pt);
        __builtin_amdgcn_sched_barrier(0);
        mfma_pos(P2{});
        __builtin_amdgcn_sched_barrier(0);
        issue_w(w_nx, 4, 5);
        cw_bytes += w_step;                             // DMA(c+1) is complete: positions 7, 8 start DMA(c+2)
        lds_patch_wait(pt, d);
        __builtin_amdgcn_sched_barrier(0);
        // positions 3..5: row pass of B^T (30 VALU) and the input dwords of chunk c+3
        xform_rows(d, t);
        avA[2] = lds_a(w_rd, 5);
        mfma_pos(P3{});
        issue_in_range(SLD{}, 0, 2);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        avA[0] = lds_a(w_rd, 6);
        mfma_pos(P4{});
        issue_in_range(SLD{}, 2, 4);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        avA[1] = lds_a(w_rd, 7);
        mfma_pos(P5{});
        issue_in_range(SLD{}, 4, NIN);
        cin_bytes += in_step;
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // position 6: column pass begins; inputs of chunk c+2 -> X[PAR]; DMA(c+1) landed; barrier
        xform_cols(t, un);
        avA[2] = lds_a(w_rd, 8);
        mfma_pos(P6{});
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        commit(SCM{}, XWR{});
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIN) : "memory");        // all but the youngest input set: DMA(c+1) landed
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        // positions 7, 8: the first two A operands of chunk c+1, the first two pieces of DMA(c+2)
        avA[0] = lds_a(w_nx, 0);
        mfma_pos(P7{});
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        issue_w(w_dma, 0, 1);
        avA[1] = lds_a(w_nx, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_pos(P8{});
        __builtin_amdgcn_sched_barrier(0);
        issue_w(w_dma, 1, 2);
        __builtin_amdgcn_sched_barrier(0);
        // DMA(c+2) continues in the next step
#pragma unroll
        for (int p = 0; p < 9; ++p) uC[p] = un[p];
        const unsigned w_old = w_rd;
        w_rd = w_nx; w_nx = w_dma; w_dma = w_old;
    };

    // ---- prologue.  Chunk pointers: cw_bytes / cin_bytes = the chunk whose DMA / input loads are being issued
    issue_in_range(I0{}, 0, NIN);                     // inputs of chunk 0 -> set 0
    cin_bytes += in_step;
    issue_w(0u, 0, NWK);                              // weights of chunk 0 -> W[0]
    cw_bytes += w_step;
    issue_in_range(I1{}, 0, NIN);                     // inputs of chunk 1 -> set 1
    cin_bytes += in_step;
    commit(I0{}, I0{});                               // chunk 0 -> X[0]
    commit(I1{}, I1{});                               // chunk 1 -> X[1]
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    issue_in_range(I1{}, 0, NIN);                     // inputs of chunk 2 -> set 1 (committed by step 0)
    cin_bytes += in_step;
    issue_w(W_FLOATS, 0, 2);                          // DMA(1) pieces 0, 1 -> W[1]; step 0 issues the rest
    {
        float d[5][6], t[3][5];
        Patch pt;
        lds_patch_row(I0{}, std::integral_constant<int, 0>{}, pt);
        lds_patch_row(I0{}, std::integral_constant<int, 1>{}, pt);
        lds_patch_row(I0{}, std::integral_constant<int, 2>{}, pt);
        lds_patch_row(I0{}, std::integral_constant<int, 3>{}, pt);
        lds_patch_row(I0{}, std::integral_constant<int, 4>{}, pt);
        lds_patch_wait(pt, d);
        xform_rows(d, t);
        xform_cols(t, uC);
        avA[0] = lds_a(0u, 0);
        avA[1] = lds_a(0u, 1);
    }
    // step 0 commits the inputs of chunk 2 into X[0] at its position 6 -- the buffer every wave has just read chunk 0 from.  In
    // the steady state a chunk barrier separates those two; here this one does (without it a wave held up in its prologue read
    // chunk 2's inputs as chunk 0: sporadic wrong tiles, found by tools/w4big.py at 128 input channels)
    __syncthreads();
    const int nch = (a.Cin + CI_T - 1) / CI_T;
    int c = 0;
    for (; c + 2 <= nch; c += 2) {
        step(I0{});
        step(I1{});
    }
    if (c < nch) step(I0{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing (all-zero) staging must land before the LDS is reused
    __syncthreads();

    // ---- epilogue: every wave holds the 3x3 block of M (the 6x6 matrix of position sums) of ITS position group for 64 channels x
    // its 16 tiles; the output transform A^T M A needs all four blocks.  A pass takes one channel block g (accumulator registers
    // r = 0..3 = channels 16 g + 4 k4 + r): every wave sends the raw position values of three registers through LDS and
    // FINISHES register r = PG (gathers the other three groups' blocks, transforms, statistics, store), so transforms,
    // BatchNorm records and stores are spread over all eight waves.  Four passes, rolled (the accumulators of block g + 1 move
    // into block 0's registers), two barriers per pass around the single exchange buffer.
    float* zb[IMG];
#pragma unroll
    for (int m = 0; m < IMG; ++m) zb[m] = a.z + (int64_t)(b0 + m) * a.z_bs;
    const int oy = y0 + 4 * tr, ox = x0 + 4 * tc;
    const bool img_ok = (b0 + img) < a.B;
    const bool vec4 = ((a.W & 3) == 0) && ((a.z_bs & 3) == 0) && (ox + 3 < a.W);
    // ST: BatchNorm statistics of this wave's 16 tiles (16 x 16 = 256 pixels) per channel from the final sums, shifted by a
    // pivot (the half block's first output of the channel) so that fp32 is enough; merged in fp64 by bn_finalize_cm.  Two records
    // per channel and block (one per tile half).
    constexpr float st_n = 256.f, st_inv = 1.f / 256.f;                     // full blocks only (host-checked)
    const int64_t st_nrec = (int64_t)a.B * a.tilesX * a.tilesY * 2;
    const int64_t st_rec = (((int64_t)bg * a.tilesY + ty) * a.tilesX + tx) * 2 + th;
    // EP 2: this launch is a dgrad whose output is the gradient of the ACTIVATION a = relu(bn(z)) of the layer below
    // (OV:47-49 -> 51): the first BatchNorm-backward pass (sum dy, sum dy * xhat with dy = da * [a > 0]) is taken here
    // from the final sums.  The z rows and the channel's coefficients are fetched ONE PASS AHEAD.
    float4 zn[4];
    float cn[4];
    auto br_issue = [&](int g_) __attribute__((always_inline)) {
        if constexpr (EP == 2) {
            const int co_ = co0 + 16 * g_ + 4 * k4 + PG;
            const int cc = co_ < a.Cout ? co_ : 0;
            const float* sv = a.br_save + (int64_t)(bg / a.br_group) * 4 * a.Cout;
            cn[0] = sv[cc]; cn[1] = sv[a.Cout + cc]; cn[2] = sv[2 * a.Cout + cc]; cn[3] = sv[3 * a.Cout + cc];
            const float* zp = a.br_z + (int64_t)bg * a.br_z_bs + (int64_t)cc * HW + (int64_t)oy * a.W + ox;
#pragma unroll
            for (int y = 0; y < 4; ++y) zn[y] = *reinterpret_cast<const float4*>(zp + y * a.W);
        }
    };
    br_issue(0);
    float* ex = smem + th * (C::EX_FLOATS / 2);
#pragma unroll 1
    for (int g = 0; g < 4; ++g) {
        // send the accumulator registers this wave does not finish: RAW position values (9 per register -- the 3x3 block of the
        // 6x6 M this group owns), not partial outputs (16 per register): 27 instead of 48 floats per lane through LDS, and one
        // full A^T M A per finished register instead of four partial ones per wave (120 instead of 188 VALU per pass)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k == PG) continue;
            const int slot = (PG < k) ? PG : PG - 1;
            float* dst = ex + ((k * 3 + slot) * 64 + lane) * C::EX_LANE;
            *reinterpret_cast<float4*>(dst) = make_float4(acc[0][0][k], acc[1][0][k], acc[2][0][k], acc[3][0][k]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(acc[4][0][k], acc[5][0][k], acc[6][0][k], acc[7][0][k]);
            dst[8] = acc[8][0][k];
        }
        __syncthreads();
        float yv[1][4][4];
        {
            constexpr int k = PG;
            const int co = co0 + 16 * g + 4 * k4 + PG;    // MFMA row 4 k4 + r of channel block g, r = PG
            // EP 2: the z rows and coefficients of this pass were requested one pass ahead (br_issue)
            float4 zr[4];
            float br_mean = 0.f, br_inv = 0.f, br_sc = 0.f, br_sh = 0.f;
            if constexpr (EP == 2) {
#pragma unroll
                for (int y = 0; y < 4; ++y) zr[y] = zn[y];
                br_mean = cn[0]; br_inv = cn[1]; br_sc = cn[2]; br_sh = cn[3];
                if (g + 1 < 4) br_issue(g + 1);
            }
            // the full 6x6 M of register PG: own 3x3 block from the accumulators, the other three from LDS
            float m[6][6];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) m[3 * RH + i][3 * CH + j] = acc[i * 3 + j][0][k];
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
                constexpr int dummy = 0;
                (void)dummy;
                const int src = (s3 < PG) ? s3 : s3 + 1;          // sending group of slot s3
                const int rh = src >> 1, ch = src & 1;
                const float* sp = ex + ((k * 3 + s3) * 64 + lane) * C::EX_LANE;
                const float4 v0 = *reinterpret_cast<const float4*>(sp), v1 = *reinterpret_cast<const float4*>(sp + 4);
                const float v8 = sp[8];
                const float vv[9] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v8};
#pragma unroll
                for (int q = 0; q < 9; ++q) m[3 * rh + q / 3][3 * ch + q % 3] = vv[q];
            }
            // Y = A^T M A,  A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
            float rp[4][6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float s12 = m[1][j] + m[2][j], d12 = m[1][j] - m[2][j], s34 = m[3][j] + m[4][j], d34 = m[3][j] - m[4][j];
                rp[0][j] = m[0][j] + s12 + s34;
                rp[1][j] = fmaf(2.f, d34, d12);
                rp[2][j] = fmaf(4.f, s34, s12);
                rp[3][j] = fmaf(8.f, d34, d12) + m[5][j];
            }
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                const float s12 = rp[y][1] + rp[y][2], d12 = rp[y][1] - rp[y][2], s34 = rp[y][3] + rp[y][4], d34 = rp[y][3] - rp[y][4];
                yv[0][y][0] = rp[y][0] + s12 + s34;
                yv[0][y][1] = fmaf(2.f, d34, d12);
                yv[0][y][2] = fmaf(4.f, s34, s12);
                yv[0][y][3] = fmaf(8.f, d34, d12) + rp[y][5];
            }
            if constexpr (EP == 1) {
                static_assert(EP == 0 || IMG == 1, "fused statistics: one image per block");
                // pivot: the first tile's first pixel of this lane's channel (lane 16 k4 of the wave)
                const float pv = __shfl(yv[0][0][0], lane & 48, 64);
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int y = 0; y < 4; ++y)
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        const float dd = yv[0][y][x] - pv;
                        s1 += dd;
                        s2 = fmaf(dd, dd, s2);
                    }
                s1 = row16_sum(s1);
                s2 = row16_sum(s2);
                if (l15 == 15 && co < a.Cout) {
                    float* sp = a.stats + ((int64_t)co * st_nrec + st_rec) * 3;
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
                        const float gg = fmaf(zc, br_sc, br_sh) > 0.f ? yv[0][y][x] : 0.f;   // same mask expression as bn_relu_bwd_*
                        s1 += gg;
                        s2 = fmaf(gg, zc, s2);
                    }
                }
                s1 = row16_sum(s1);
                s2 = row16_sum(s2) * br_inv;            // xhat = (z - mean) * invstd
                if (l15 == 15 && co < a.Cout) {
                    float* sp = a.stats + ((int64_t)co * st_nrec + st_rec) * 2;
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
                            *reinterpret_cast<float4*>(o + y * a.W) = make_float4(yv[0][y][0], yv[0][y][1], yv[0][y][2], yv[0][y][3]);
                        } else {
#pragma unroll
                            for (int x = 0; x < 4; ++x)
                                if (ox + x < a.W) o[y * a.W + x] = yv[0][y][x];
                        }
                    }
                }
            }
        }
        __syncthreads();                              // the exchange buffer is rewritten by the next pass
#pragma unroll
        for (int p = 0; p < 9; ++p) {
            acc[p][0] = acc[p][1]; acc[p][1] = acc[p][2]; acc[p][2] = acc[p][3];
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
    const int64_t n = (int64_t)std::max(Cout, Cin) * ((std::max(Cout, Cin) + 63) & ~63);
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
    if (wq_fwd) {
        hipLaunchKernelGGL(pack3x3_wino4_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w, wq_fwd, Cout, Cin, 1);
        int rc = check_launch("pack3x3_wino4_kernel");
        if (rc) return rc;
    }
    if (wq_dgrad) hipLaunchKernelGGL(pack3x3_wino4_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w, wq_dgrad, Cout, Cin, 0);
    return check_launch("pack3x3_wino4_kernel");
}

int onet_conv3x3_winograd4_fwd(const float* x, int64_t x_bs, const float* wq, float* z, int64_t z_bs, int B, int Cin,
                               int Cout, int H, int W, void* stream) {
    ONET_REQUIRE(x && wq && z, "conv3x3_winograd4_fwd: null pointer");
    ONET_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "conv3x3_winograd4_fwd: bad shape");
    ONET_REQUIRE((Cout & 3) == 0 && (Cin & 3) == 0, "conv3x3_winograd4_fwd: Cin and Cout must be multiples of 4 (use onet_conv_fwd)");
    ONET_REQUIRE(x_bs >= (int64_t)Cin * H * W && z_bs >= (int64_t)Cout * H * W, "conv3x3_winograd4_fwd: batch stride too small");
    ONET_REQUIRE((x_bs + (int64_t)(Cin + 16) * H * W) * 4 < (1ll << 31) && (int64_t)(Cin + 16) * 36 * ((Cout + 63) & ~63) * 4 < (1ll << 31),
                 "conv3x3_winograd4_fwd: operand exceeds the 2 GiB buffer-resource range");
    Wino4Args a{x, x_bs, wq, z, z_bs, B, Cin, Cout, H, W, 0, 0, 0, 0, nullptr, nullptr, 0, nullptr, 1};
    return (W > 16) ? launch_wino4<8>(a, as_stream(stream)) : launch_wino4<4>(a, as_stream(stream));
}

int onet_conv3x3_winograd4_nparts(int B, int H, int W) {
    // statistics are emitted by full 16 x 32-pixel blocks only (every U-Net level of a 2^k-sized input down to 32 px): two records
    // (n, mean, M2) of 256 pixels per channel and block
    if (B <= 0 || H <= 0 || W <= 0 || (W % W4Cfg<8>::PXW) != 0 || (H % W4Cfg<8>::PXH) != 0) return 0;
    const int64_t n = (int64_t)B * cdiv(W, W4Cfg<8>::PXW) * cdiv(H, W4Cfg<8>::PXH) * 2;      // one record per tile half of a block
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
    ONET_REQUIRE((x_bs + (int64_t)(Cin + 16) * H * W) * 4 < (1ll << 31) && (int64_t)(Cin + 16) * 36 * ((Cout + 63) & ~63) * 4 < (1ll << 31),
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
    ONET_REQUIRE((dz_bs + (int64_t)(Cdz + 16) * H * W) * 4 < (1ll << 31) && (int64_t)(Cdz + 16) * 36 * ((Cda + 63) & ~63) * 4 < (1ll << 31),
                 "conv3x3_winograd4_dgrad_bnreduce: operand exceeds the 2 GiB buffer-resource range");
    Wino4Args a{dz, dz_bs, wq_dgrad, da, da_bs, B, Cdz, Cda, H, W, 0, 0, 0, 0, part2, z_prev, z_prev_bs, save_prev, group_images};
    return launch_wino4<8, 2>(a, as_stream(stream));
}

}  // extern "C"
