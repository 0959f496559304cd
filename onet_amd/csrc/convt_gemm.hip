// The three GEMMs of nn.ConvTranspose2d(Cin, Ct, kernel_size=2, stride=2) (OV:86) on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32), as DMA-fed 128 x 128 block tiles:
//
//   forward  y[b][c][2i+di][2j+dj] = bias[c] + sum_ci  x[b][ci][i][j] * W[ci][c][di][dj]        M = 4 Ct, N = pixels, K = Cin
//   dgrad    dx[b][ci][i][j]       = sum_{c,di,dj} dy[b][c][2i+di][2j+dj] * W[ci][c][di][dj]     M = Cin,  N = pixels, K = 4 Ct
//   wgrad    dW[ci][c][di][dj]     = sum_{b,i,j}   x[b][ci][i][j] * dy[b][c][2i+di][2j+dj]       M = Cin,  N = 4 Ct,   K = pixels
//
// These are plain dense contractions (arithmetic intensity 42-128 FLOP/B per layer, SURVEY 8d: MFMA-bound in fp32); with one
// tap per staged operand the 64-row direct kernel of conv_mfma.hip spent 3x the staging instructions per MFMA of its 3x3
// form and ran at 61-63 % of the fp32 MFMA peak.  Here:
// * block = 256 threads = 4 waves (2 x 2), wave tile 64 x 64 = four 32x32 accumulators (64 VGPRs): ONE LDS fragment read
//   per MFMA instead of two, two to four blocks per CU;
// * both operands arrive by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KB per wave-instruction, no staging registers), double
//   buffered, ONE barrier per K-chunk of 32 (forward, dgrad: 16) K-steps;
// * the pixel shuffle (forward) / its inverse (dgrad, wgrad) costs nothing: the forward epilogue finds the four sub-pixels
//   of a channel in four consecutive accumulator rows of one lane (float2 row stores straight into the concat buffer); the
//   backward kernels DMA the dy rows as they lie in the concat gradient and pick the sub-pixel at fragment-read time;
// * wgrad operands are pixel-major in memory (K contiguous): fragments are ds_read_b128 with lanes 0-31 / 32-63 taking the
//   two halves of an 8-pixel group (K-step t contracts pixels 8g + t and 8g + 4 + t), bank conflicts removed by an XOR
//   swizzle of the 16-byte chunks that the DMA applies on the SOURCE side (its LDS destination is lane-linear).
// Shapes outside the fast path (maps whose pixel count is not a multiple of 128, odd F.pad offsets, channel tails) keep the
// kernels of conv_mfma.hip: onet_convT2x2_{fwd,dgrad,wgrad} dispatch.
#include <algorithm>
#include <cstdlib>
#include "common.hpp"

using namespace onet;

#ifndef ONET_GEMM_SPREAD
#define ONET_GEMM_SPREAD 0
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4g __attribute__((ext_vector_type(4)));
typedef float f32x2g __attribute__((ext_vector_type(2)));
typedef int i32x4g __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8g __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2g __attribute__((ext_vector_type(2)));
typedef unsigned u32x4g __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ i32x4g g_rsrc(const void* base, int64_t bytes) {
    const uint64_t p = reinterpret_cast<uint64_t>(base);
    i32x4g r;
    r.x = (int)(p & 0xffffffffu);
    r.y = (int)((p >> 32) & 0xffffu);
    r.z = bytes > 0x7fffffffll ? 0x7fffffff : (int)bytes;
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ unsigned g_lds(const float* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)p;
}
// LDS-DMA of 64 x 16 bytes: lane l's 16 bytes at (rsrc base + voff) land at LDS byte address lds_base + 16 l.
// Inline asm: the compiler must not see a load (it would order every later ds_read behind a vmcnt(0)).
__device__ __forceinline__ void g_dma16(i32x4g rsrc, unsigned lds_base, unsigned voff) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(lds_base)
                 : "memory");
}

// BF (BASELINE configs[2], bf16 MFMA conv path): the same DMA-fed fp32 tiles, but a K-chunk of 16 is ONE
// v_mfma_f32_32x32x16_bf16 per accumulator instead of eight v_mfma_f32_32x32x2_f32: a lane gathers its 8 K-values from LDS
// (as many ds_read as before), rounds them to bf16 (v_cvt_pk_bf16_f32, nearest-even: the rounding the bf16 conv kernels apply
// to their operands) and accumulates in fp32.  The fp32 kernels spend 2.6-2.8x their data-movement time in the matrix pipe
// (timing-only build without the MFMAs: 0.90 vs 2.31 ms over the four decoder levels at B = 64).  Measured with bf16 operands:
// forward 0.97, dgrad 0.77, wgrad 0.92 ms (fp32: 2.32 / 2.13 / 2.51).  A ring of four chunk buffers with the DMA three chunks
// ahead (the bf16 chunk is only 128 cycles of MFMAs) was NOT faster: forward 1.10, dgrad 0.78 -- dropped.
__device__ __forceinline__ unsigned g_pack(float lo, float hi) {
    bf16x2g v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ bf16x8g g_pack8(const float (&v)[8]) {
    const u32x4g q = {g_pack(v[0], v[1]), g_pack(v[2], v[3]), g_pack(v[4], v[5]), g_pack(v[6], v[7])};
    return __builtin_bit_cast(bf16x8g, q);
}

// PREC 2 (round 3): fp32-level results on the bf16 pipe by operand splitting, as conv_split.hip: v = hi + mid with hi = bf16(v),
// mid = bf16(v - hi); a.b ~ a_mid.b_hi + a_hi.b_mid + a_hi.b_hi (three MFMAs per accumulator and K-chunk; every bf16 product is
// exact in the fp32 accumulator, the dropped terms are ~2^-16 relative).  The split runs on the gathered fragments (6 VALU per
// value pair): 3/16 of the fp32 kernel's matrix-pipe time.
__device__ __forceinline__ void g_split8(const float (&v)[8], bf16x8g& hi, bf16x8g& mid) {
    u32x4g qh, qm;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned h = g_pack(v[2 * p], v[2 * p + 1]);
        const float h0 = __builtin_bit_cast(float, h << 16), h1 = __builtin_bit_cast(float, h & 0xffff0000u);
        qh[p] = h;
        qm[p] = g_pack(v[2 * p] - h0, v[2 * p + 1] - h1);
    }
    hi = __builtin_bit_cast(bf16x8g, qh);
    mid = __builtin_bit_cast(bf16x8g, qm);
}
// acc += a.b at the precision PREC selects (1: bf16 operands, 2: split) for a wave's 2 x 2 accumulators
template <int PREC>
__device__ __forceinline__ void g_mma16(const float (&af)[2][8], const float (&bf)[2][8], f32x16 (&acc)[2][2]) {
    if constexpr (PREC == 1) {
        bf16x8g a8[2], b8[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) a8[t] = g_pack8(af[t]);
#pragma unroll
        for (int u = 0; u < 2; ++u) b8[u] = g_pack8(bf[u]);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[t], b8[u], acc[t][u], 0, 0, 0);
    } else {
        bf16x8g ah[2], am[2], bh[2], bm[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) g_split8(af[t], ah[t], am[t]);
#pragma unroll
        for (int u = 0; u < 2; ++u) g_split8(bf[u], bh[u], bm[u]);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[t], bh[u], acc[t][u], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bm[u], acc[t][u], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bh[u], acc[t][u], 0, 0, 0);
    }
}

struct GArgs {
    const float* a;       // forward: wq [Cin][4Ct];  dgrad: wd [4Ct][Cin] (row q*Ct + c);  wgrad: x
    const float* b;       // forward: x;              dgrad: dy window;                       wgrad: dy window
    float* out;           // forward: y window;       dgrad: dx;                              wgrad: slab [splitK][Cin][4Ct]
    const float* bias;
    float* dbias_part;    // dgrad: [nTiles][Ct] partial sums of dy over the tile's pixels (blocks with mt == 0), or NULL
    __bf16* out16;        // forward: the pre-split destination (out16_split != 0), batch stride out16_bs
    int64_t out16_bs;
    int64_t a_bs, b_bs, out_bs;
    int B, Cin, Ct, h, w, Wo, HoWo;   // y / dy plane: Ho x Wo with Ho = 2h, Wo = 2w (fast path: no F.pad offsets)
    int mTiles, nTiles, splitK, chunksPerSplit;
    int out16_split;      // forward: out16 is NOT a bf16 copy but the PRE-SPLIT destination (fp16 hi | mid slots [B][Ct/8][Ho][2][Wo][8],
                          // conv_split.hip; out16_bs in 4-byte units): the up-sampled groups of a pre-split concat buffer
    const unsigned* out_slots;   // ... whose fp16 parts are those of 2^k y, k = the guard exponent these magnitude slots select (NULL: k = 0)
};

__device__ __forceinline__ int xcd_order(int n) {
    const int q = n >> 3, r = n & 7, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

// sum over the 32 lanes of a wave half, valid in lanes 31 / 63 (DPP row rotations + row broadcast, as in conv_wino4.hip)
#define ONET_G_DPP_ADD(v, ctrl, rmask) \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false))
__device__ __forceinline__ float g_half_sum(float v) {
    ONET_G_DPP_ADD(v, 0x128, 0xf);   // row_ror:8
    ONET_G_DPP_ADD(v, 0x124, 0xf);   // row_ror:4
    ONET_G_DPP_ADD(v, 0x122, 0xf);   // row_ror:2
    ONET_G_DPP_ADD(v, 0x121, 0xf);   // row_ror:1
    ONET_G_DPP_ADD(v, 0x142, 0xa);   // row_bcast:15 into rows 1 and 3
    return v;
}
#undef ONET_G_DPP_ADD

constexpr int KC = 16;                 // forward / dgrad: K per chunk (8 K-steps)
constexpr int TILE_F = KC * 128;       // floats of one operand tile

// ------------------------------------------------------------------------------------------------ forward and dgrad
// MODE 0: forward (A rows = wq[k][m], B rows = x[b][k][pixels]);  MODE 1: dgrad (A rows = wd[q*Ct + c][ci], B rows =
// dy[b][c][2y + di][...] as they lie: 256 floats per (c, di) for the tile's 128 pixels)
template <int MODE, int PREC = 0>
__global__ __launch_bounds__(256, 2) void convt_gemm_kernel(GArgs g) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * TILE_F];      // [buf][A | B]
    int bid = xcd_order(gridDim.x);
    const int mt = bid % g.mTiles;                 // m tile fastest: the blocks of one pixel tile share its B rows in L2
    const int nt = bid / g.mTiles;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (SGPR)
    const int wr = wid >> 1, wc = wid & 1, l31 = lane & 31, kh = lane >> 5;
    const int hw = g.h * g.w;
    const int M = MODE == 0 ? 4 * g.Ct : g.Cin, K = MODE == 0 ? g.Cin : 4 * g.Ct;
    const int m0 = mt * 128;
    const int n0 = nt * 128;                       // global pixel index (b * hw + p); hw % 128 == 0: one image per tile
    const int b = n0 / hw, p0 = n0 % hw;

    const i32x4g ra = g_rsrc(g.a, (int64_t)K * M * 4);
    const i32x4g rb = MODE == 0 ? g_rsrc(g.b + (int64_t)b * g.b_bs, (int64_t)g.Cin * hw * 4)
                                : g_rsrc(g.b + (int64_t)b * g.b_bs, (int64_t)g.Ct * g.HoWo * 4);
    // DMA roles: wave w issues A pieces 2w, 2w+1 and B pieces 2w, 2w+1 of every chunk (piece = 1 KB)
    unsigned a_off[2], b_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = 2 * wid + j;
        const int kk = 2 * p + (lane >> 5);                                   // A row of the chunk held by this lane
        if (MODE == 0) {
            a_off[j] = (unsigned)(((int64_t)kk * M + m0 + (lane & 31) * 4) * 4);
            b_off[j] = (unsigned)(((int64_t)kk * hw + p0 + (lane & 31) * 4) * 4);
        } else {
            const int cl = kk >> 2, q = kk & 3;                               // chunk row kk = cl * 4 + q  <-  wd row q * Ct + c
            a_off[j] = (unsigned)((((int64_t)q * g.Ct + cl) * M + m0 + (lane & 31) * 4) * 4);
            // B piece p = row (cl, di) = (p >> 1, p & 1): 256 floats = the tile's 128 pixels x dj, 2 pixels per lane
            const int px = p0 + 2 * lane, y = px / g.w, x = px % g.w;
            b_off[j] = (unsigned)((((int64_t)(p >> 1)) * g.HoWo + (int64_t)(2 * y + (p & 1)) * g.Wo + 2 * x) * 4);
        }
    }
    const unsigned a_step = MODE == 0 ? (unsigned)((int64_t)KC * M * 4) : (unsigned)(4 * M * 4);       // dgrad: 4 channels on
    const unsigned b_step = MODE == 0 ? (unsigned)((int64_t)KC * hw * 4) : (unsigned)((int64_t)4 * g.HoWo * 4);
    // piece q of the wave's four per chunk: 0, 1 = its two A pieces, 2, 3 = its two B pieces
    auto issue1 = [&](int chunk, int buf, int q) __attribute__((always_inline)) {
        const unsigned la = g_lds(lds) + (unsigned)(buf * 2 * TILE_F * 4), lb = la + TILE_F * 4;
        const int j = q & 1;
        if (q < 2) g_dma16(ra, la + (2 * wid + j) * 1024, a_off[j] + chunk * a_step);
        else g_dma16(rb, lb + (2 * wid + j) * 1024, b_off[j] + chunk * b_step);
    };
    auto issue = [&](int chunk, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) issue1(chunk, buf, q);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    const int nch = K / KC;
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        const int buf = c & 1;
        const bool more = c + 1 < nch;
        const float* A = lds + buf * 2 * TILE_F + wr * 64 + l31;
        const float* Bt = lds + buf * 2 * TILE_F + TILE_F;
        if (MODE == 1 && g.dbias_part && mt == 0) {
            // ConvTranspose2d bias gradient (sum of dy over all pixels) from the staged dy rows: wave w holds the chunk's channel
            // 4c + w, lanes 0-31 its di = 0 row, lanes 32-63 its di = 1 row (256 floats each = the tile's 128 pixels x dj)
            const f32x4g* row = reinterpret_cast<const f32x4g*>(Bt + (2 * wid + kh) * 256 + l31 * 8);
            const f32x4g v0 = row[0], v1 = row[1];
            float sb = ((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]));
            sb = g_half_sum(sb);
            const float other = __shfl(sb, 63, 64);
            if (lane == 31) g.dbias_part[(int64_t)nt * g.Ct + 4 * c + wid] = sb + other;
        }
        if constexpr (PREC != 0) {
            if (more) issue(c + 1, buf ^ 1);
            // one K = 16 MFMA per accumulator: lane (i, kh) holds K-values 8 kh .. 8 kh + 7 of the chunk
            float af[2][8], bf[2][8];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 8; ++i) af[t][i] = A[(8 * kh + i) * 128 + t * 32];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (MODE == 0) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) bf[u][i] = Bt[(8 * kh + i) * 128 + wc * 64 + u * 32 + l31];
                } else {
                    // chunk row kk = 8 kh + i = (channel cl = 2 kh + i / 4, di = (i / 2) % 2, dj = i % 2): the (dj = 0, 1) pair is adjacent
#pragma unroll
                    for (int ip = 0; ip < 4; ++ip) {
                        const f32x2g v = *reinterpret_cast<const f32x2g*>(Bt + ((2 * kh + (ip >> 1)) * 2 + (ip & 1)) * 256 + 2 * (wc * 64 + u * 32 + l31));
                        bf[u][2 * ip] = v[0];
                        bf[u][2 * ip + 1] = v[1];
                    }
                }
            }
            g_mma16<PREC>(af, bf, acc);
        } else {
#pragma unroll
        for (int s = 0; s < KC / 2; ++s) {
            // the next chunk's four DMA pieces go out one per two K-steps, not as a burst in front of the MFMAs (a piece
            // holds the issuing wave for 60-180 cycles)
#if ONET_GEMM_SPREAD
            if ((s & 1) == 0 && more) issue1(c + 1, buf ^ 1, s >> 1);
#else
            if (s == 0 && more) issue(c + 1, buf ^ 1);
#endif
            float av[2], bv[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) av[t] = A[(2 * s + kh) * 128 + t * 32];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (MODE == 0) bv[u] = Bt[(2 * s + kh) * 128 + wc * 64 + u * 32 + l31];
                else bv[u] = Bt[((s >> 1) * 2 + (s & 1)) * 256 + 2 * (wc * 64 + u * 32 + l31) + kh];   // row (cl, di), dj = kh
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[u], acc[t][u], 0, 0, 0);
                }
        }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // chunk c+1 landed (issued a whole chunk of MFMAs ago)
        __syncthreads();                                        // ... and every wave is done with this buffer
    }

    // ---- epilogue
    if (MODE == 0) {
        float* yb = g.out + (int64_t)b * g.out_bs;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int p = p0 + wc * 64 + u * 32 + l31, y = p / g.w, x = p % g.w;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int c = (m0 + wr * 64 + t * 32 + 8 * gq + 4 * kh) >> 2;      // rows 4 gq .. 4 gq + 3 = sub-pixels of c
                    const float bs = g.bias ? g.bias[c] : 0.f;
                    const int64_t oo = (int64_t)c * g.HoWo + (int64_t)(2 * y) * g.Wo + 2 * x;
                    float* o = yb + oo;
                    const float v0 = acc[t][u][4 * gq] + bs, v1 = acc[t][u][4 * gq + 1] + bs, v2 = acc[t][u][4 * gq + 2] + bs,
                                v3 = acc[t][u][4 * gq + 3] + bs;
                    if (g.out) {
                        *reinterpret_cast<float2*>(o) = make_float2(v0, v1);
                        *reinterpret_cast<float2*>(o + g.Wo) = make_float2(v2, v3);
                    }
                }
                if (g.out16 && g.out16_split) {
                    // Pre-split output.  A lane holds, of the 8 channels c8 * 8 + {0 .. 7} of this 32-row group, those of its parity
                    // (kh = 0: even, kh = 1: odd; channel 2 gq + kh in acc[4 gq ..]) with all four sub-pixels (di, dj) of input pixel
                    // p.  v_permlane32_swap trades the half it does not keep with the partner lane: afterwards a kh = 0 lane owns
                    // the two output pixels of row di = 0 and a kh = 1 lane those of row di = 1, all 8 channels each = whole slots.
                    const int c8 = (m0 + wr * 64 + t * 32) >> 5;
                    float px[2][8];                         // [dj][channel of the group]
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int ch = ((m0 + wr * 64 + t * 32 + 8 * gq + 4 * kh) >> 2);
                        const float bs = g.bias ? g.bias[ch] : 0.f;
#pragma unroll
                        for (int dj = 0; dj < 2; ++dj) {
                            const unsigned d0 = __builtin_bit_cast(unsigned, acc[t][u][4 * gq + dj] + bs);
                            const unsigned d1 = __builtin_bit_cast(unsigned, acc[t][u][4 * gq + 2 + dj] + bs);
                            const auto sw = __builtin_amdgcn_permlane32_swap(d0, d1, false, false);
                            // (scalars first: __builtin_bit_cast applied to a vector ELEMENT yields element 0 with hipcc 7.2)
                            const unsigned even = sw[0], odd = sw[1];
                            px[dj][2 * gq] = __builtin_bit_cast(float, even);           // even channel of the lane's row
                            px[dj][2 * gq + 1] = __builtin_bit_cast(float, odd);        // odd channel
                        }
                    }
                    // out16_split == 2: plain bf16, one part (BASELINE configs[2]): [Ct/8][Ho][Wo][8]
                    const int np = g.out16_split == 2 ? 1 : 2;
                    float s_inv;
                    const float s_up = np == 1 ? 1.f : amax_scale(amax_read(g.out_slots), false, s_inv);
                    u32x4g* dst = reinterpret_cast<u32x4g*>(reinterpret_cast<unsigned*>(g.out16) + (int64_t)b * g.out16_bs) +
                                  ((int64_t)(c8 * (2 * g.h) + 2 * y + kh) * np) * g.Wo + 2 * x;
#pragma unroll
                    for (int dj = 0; dj < 2; ++dj) {
                        u32x4g hi, mid;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            unsigned hh, mm;
                            split2h_s(px[dj][2 * k], px[dj][2 * k + 1], s_up, hh, mm);
                            hi[k] = np == 1 ? g_pack(px[dj][2 * k], px[dj][2 * k + 1]) : hh;
                            mid[k] = mm;
                        }
                        dst[dj] = hi;
                        if (np == 2) dst[g.Wo + dj] = mid;
                    }
                }
            }
    } else {
        float* xb = g.out + (int64_t)b * g.out_bs;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int p = p0 + wc * 64 + u * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ci = m0 + wr * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    xb[(int64_t)ci * hw + p] = acc[t][u][r];
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------ wgrad
constexpr int KP = 32;                          // pixels per chunk (16 K-steps)
constexpr int WA_F = 128 * KP;                  // A tile: 128 ci rows x 32 px
constexpr int WB_F = 64 * 2 * KP;               // B tile: 64 (c, di) rows x (32 px x dj)

template <int PREC = 0>
__global__ __launch_bounds__(256, 2) void convt_wgrad_gemm_kernel(GArgs g) {
    __shared__ __attribute__((aligned(16))) float lds[2 * (WA_F + WB_F)];
    int bid = xcd_order(gridDim.x);
    const int tiles = g.mTiles * g.nTiles;
    const int ks = bid / tiles, tile = bid % tiles;
    const int mt = tile % g.mTiles, nt = tile / g.mTiles;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (SGPR)
    const int wr = wid >> 1, wc = wid & 1, l31 = lane & 31, kh = lane >> 5;
    const int hw = g.h * g.w;
    const int ci0 = mt * 128, c0 = nt * 32;                      // N tile = 32 channels x 4 sub-pixels
    const int ch0 = ks * g.chunksPerSplit;
    const int nchunks_all = (int)(((int64_t)g.B * hw) / KP);
    const int ch1 = min(ch0 + g.chunksPerSplit, nchunks_all);

    // DMA roles per chunk: 16 A pieces (8 rows x 128 B each) + 16 B pieces (4 rows x 256 B each), 8 per wave
    // A: lane -> row = 8 p + (l >> 3), physical 16-B chunk l & 7 holds source chunk (l & 7) ^ ((row >> 1) & 7)
    // B: lane -> row = 4 p + (l >> 4), physical chunk l & 15 holds source chunk (l & 15) ^ (row & 7)
    auto issue = [&](int chunk, int buf, int j0, int j1) __attribute__((always_inline)) {
        const int64_t pix = (int64_t)chunk * KP;
        const int b = (int)(pix / hw), p0 = (int)(pix % hw);
        const i32x4g rx = g_rsrc(g.a + (int64_t)b * g.a_bs, (int64_t)g.Cin * hw * 4);
        const i32x4g rd = g_rsrc(g.b + (int64_t)b * g.b_bs, (int64_t)g.Ct * g.HoWo * 4);
        const unsigned la = g_lds(lds) + (unsigned)(buf * (WA_F + WB_F) * 4), lb = la + WA_F * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < j0 || j >= j1) continue;
            const int p = 4 * wid + j;
            {
                const int row = 8 * p + (lane >> 3), src = (lane & 7) ^ ((row >> 1) & 7);
                g_dma16(rx, la + p * 1024, (unsigned)((((int64_t)(ci0 + row)) * hw + p0 + src * 4) * 4));
            }
            {
                const int row = 4 * p + (lane >> 4), src = (lane & 15) ^ (row & 7);      // row = cl * 2 + di
                const int px = p0 + 2 * src, y = px / g.w, x = px % g.w;                 // 16 B = 2 pixels x dj
                g_dma16(rd, lb + p * 1024,
                        (unsigned)((((int64_t)(c0 + (row >> 1))) * g.HoWo + (int64_t)(2 * y + (row & 1)) * g.Wo + 2 * x) * 4));
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    // fragment addresses (floats).  A: row = wr*64 + t*32 + l31, chunk 2 gg + kh.  B: lane j = l31 -> m = wc*64 + u*32 + j =
    // (cl, di, dj): row = m >> 1, dj = m & 1 = l31 & 1; chunks 4 gg + 2 kh, + 1
    int a_row[2], b_row[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) a_row[t] = wr * 64 + t * 32 + l31;
#pragma unroll
    for (int u = 0; u < 2; ++u) b_row[u] = (wc * 64 + u * 32 + l31) >> 1;
    const bool dj = (l31 & 1) != 0;

    if (ch0 < ch1) {
        issue(ch0, 0, 0, 4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    for (int c = ch0; c < ch1; ++c) {
        const int buf = (c - ch0) & 1;
        const bool more = c + 1 < ch1;
        const float* A = lds + buf * (WA_F + WB_F);
        const float* Bt = A + WA_F;
        if constexpr (PREC != 0) {
            if (more) issue(c + 1, buf ^ 1, 0, 4);
            // K = 16 pixels per MFMA: lane (i, kh) holds pixels 8 gg + 4 kh + 0..3 of two consecutive 8-pixel groups
#pragma unroll
            for (int g2 = 0; g2 < KP / 16; ++g2) {
                float af[2][8], bf[2][8];
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int gg = 2 * g2 + h2;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const f32x4g v = *reinterpret_cast<const f32x4g*>(A + a_row[t] * KP + (((2 * gg + kh) ^ ((a_row[t] >> 1) & 7)) << 2));
#pragma unroll
                        for (int i = 0; i < 4; ++i) af[t][4 * h2 + i] = v[i];
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const f32x4g v = *reinterpret_cast<const f32x4g*>(Bt + b_row[u] * (2 * KP) + (((4 * gg + 2 * kh + e) ^ (b_row[u] & 7)) << 2));
                            bf[u][4 * h2 + 2 * e] = dj ? v[1] : v[0];
                            bf[u][4 * h2 + 2 * e + 1] = dj ? v[3] : v[2];
                        }
                }
                g_mma16<PREC>(af, bf, acc);
            }
        } else {
#pragma unroll
        for (int gg = 0; gg < KP / 8; ++gg) {
#if ONET_GEMM_SPREAD
            if (more) issue(c + 1, buf ^ 1, gg, gg + 1);        // one A and one B piece of the next chunk per 16 MFMAs
#else
            if (gg == 0 && more) issue(c + 1, buf ^ 1, 0, 4);
#endif
            f32x4g a4[2], b4[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
                a4[t] = *reinterpret_cast<const f32x4g*>(A + a_row[t] * KP + (((2 * gg + kh) ^ ((a_row[t] >> 1) & 7)) << 2));
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    b4[u][e] = *reinterpret_cast<const f32x4g*>(Bt + b_row[u] * (2 * KP) + (((4 * gg + 2 * kh + e) ^ (b_row[u] & 7)) << 2));
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                float bv[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const f32x4g v = b4[u][tt >> 1];
                    bv[u] = dj ? v[(tt & 1) * 2 + 1] : v[(tt & 1) * 2];
                }
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[t][tt], bv[u], acc[t][u], 0, 0, 0);
            }
        }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    float* slab = g.out + (int64_t)ks * g.Cin * 4 * g.Ct;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int m = 4 * c0 + wc * 64 + u * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + wr * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                slab[(int64_t)ci * 4 * g.Ct + m] = acc[t][u][r];
            }
        }
}

// dw[i] (+)= sum_k slab[k][i], fixed order: bit-reproducible
__global__ __launch_bounds__(256) void convt_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int splitK,
                                                                 int64_t n4, int accumulate) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 s = reinterpret_cast<const float4*>(slab)[i];
    for (int k = 1; k < splitK; ++k) {
        const float4 v = reinterpret_cast<const float4*>(slab)[(int64_t)k * n4 + i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (accumulate) {
        const float4 o = reinterpret_cast<float4*>(dw)[i];
        s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    reinterpret_cast<float4*>(dw)[i] = s;
}

// db[c] = sum over the pixel tiles of part[tile][c], fp64, fixed order
__global__ __launch_bounds__(256) void convt_dbias_reduce_kernel(const float* __restrict__ part, float* __restrict__ db, int ntiles, int Ct) {
    const int c = blockIdx.x;
    double s = 0.0;
    for (int t = threadIdx.x; t < ntiles; t += 256) s += (double)part[(int64_t)t * Ct + c];
    __shared__ double sh[4];
    double v[1] = {s};
    block_sum_256<double, 1>(v, sh);
    if (threadIdx.x == 0) db[c] = (float)v[0];
}


// ================================================================================================ slot operands (round 5)
// The forward GEMM with BOTH operands in the 16-byte slot form of conv_split.hip -- 8 consecutive K-values (input channels) of one
// row / pixel per slot, fp16 (hi | mid) parts of a power-of-two-scaled value (NP = 2: fp32-level results, three MFMAs per term) or
// one part of plain bf16 (NP = 1: BASELINE configs[2]):
//   x  [B][Cin/8][h][NP][w][8]    written by the BatchNorm + ReLU pass of the block below (bn.hip: the pre-split producers)
//   wP [Cin/8][NP][4 Ct][8]       packed once per optimizer step (convt_pack_slots_kernel), column m = 4 c + sub-pixel
// A slot IS an MFMA fragment (lane (row, kh) of v_mfma_f32_32x32x16 holds k = 8 kh .. 8 kh + 7): both tiles are LDS-DMA copies and
// every fragment is ONE ds_read_b128 -- per 32-channel chunk a wave issues 8 NP ds_read_b128 for 8 (NP = 1) / 24 (NP = 2) MFMAs and
// no VALU; the fp32-operand kernel above gathers each fragment with 8 ds_read_b32 and splits it in registers (14.8 VALU per MFMA,
// 0.23-0.26 of the 16-bit peak issued).  Same 128 x 128 block tile, 2 x 2 waves, double-buffered chunks, one barrier per chunk.
typedef _Float16 f16x8g __attribute__((ext_vector_type(8)));

struct SArgs {
    const void* wP;
    const void* xP;
    int64_t x_bs;                  // batch stride of x in 4-byte units
    const float* bias;
    void* yP;                      // pre-split output [B][Ct/8][Ho][NP][Wo][8]
    int64_t y_bs;                  // ... its batch stride in 4-byte units
    const unsigned* x_slots;       // NP = 2: the magnitude slots x's producer scaled it by (guard rule; NULL: unscaled)
    const unsigned* y_slots;       // NP = 2: ... and those the output is scaled by (guard rule; NULL: unscaled)
    int B, Cin, Ct, h, w, mTiles, nTiles;
};

template <int NP>
__device__ __forceinline__ f32x16 s_mfma(u32x4g a, u32x4g b, f32x16 c) {
    if constexpr (NP == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8g, a), __builtin_bit_cast(f16x8g, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8g, a), __builtin_bit_cast(bf16x8g, b), c, 0, 0, 0);
}

#ifndef SLOT_NST
#define SLOT_NST 2       // measured (B = 64, the four decoder levels, forward): 2 stages / 4 blocks per CU 1.17 ms, 3 / 3 1.20, 4 / 2 1.24
#endif
#ifndef SLOT_OCC
#define SLOT_OCC 4
#endif
template <int N>
__device__ __forceinline__ void slot_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int NP>
__global__ __launch_bounds__(256, SLOT_OCC) void convt_slot_fwd_kernel(SArgs g) {
    constexpr int KS = 4 / NP;                     // slots (8 channels each) per chunk: 16 channels (NP = 2) / 32 (NP = 1)
    constexpr int TILE = KS * NP * 128;            // slots of one operand tile: [slot of the chunk][part][row / pixel] = 8 KB
    constexpr int NPC = TILE / 256;                // DMA pieces (64 slots) per wave, operand and chunk
    constexpr int NST = SLOT_NST;                  // ring of chunk buffers, the DMA NST - 1 chunks ahead.  A block lives for 8 .. 64 chunks of
                                                   // 0.15 us of MFMAs each and is a chain of memory latencies (first chunk, every later
                                                   // chunk, the stores): what hides them is MORE BLOCKS per CU (32 KB of LDS each), not a
                                                   // deeper ring -- timing builds without MFMAs / without stores: 0.27 / 0.37 of 0.53 ms on
                                                   // the 128-channel level, unchanged by the ring depth
    __shared__ __attribute__((aligned(16))) u32x4g lds[NST * 2 * TILE];     // [stage][A | B]
    const int bid = xcd_order(gridDim.x);
    const int mt = bid % g.mTiles;                 // m tile fastest: the blocks of one pixel tile share its x slots in L2
    const int nt = bid / g.mTiles;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1, l31 = lane & 31, kh = lane >> 5;
    const int hw = g.h * g.w, M = 4 * g.Ct;
    const int m0 = mt * 128, n0 = nt * 128;        // global pixel index b * hw + p; hw % 128 == 0: one image per tile
    const int b = n0 / hw, p0 = n0 % hw;

    const i32x4g ra = g_rsrc(g.wP, (int64_t)(g.Cin / 8) * NP * M * 16);
    const i32x4g rb = g_rsrc(reinterpret_cast<const unsigned*>(g.xP) + (int64_t)b * g.x_bs, (int64_t)(g.Cin / 8) * hw * NP * 16);
    unsigned a_off[NPC], b_off[NPC];
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
        const int s = (wid * NPC + j) * 64 + lane;                  // slot of the tile this lane copies
        const int c8 = s / (NP * 128), part = (s >> 7) % NP, r = s & 127;
        a_off[j] = (unsigned)((((int64_t)c8 * NP + part) * M + m0 + r) * 16);
        const int p = p0 + r, i = p / g.w, jx = p % g.w;
        b_off[j] = (unsigned)(((((int64_t)c8 * g.h + i) * NP + part) * g.w + jx) * 16);
    }
    const unsigned a_step = (unsigned)((int64_t)KS * NP * M * 16), b_step = (unsigned)((int64_t)KS * hw * NP * 16);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) u32x4g*)lds;
    auto issue = [&](int chunk, int buf) __attribute__((always_inline)) {
        const unsigned la = lds0 + (unsigned)(buf * 2 * TILE * 16), lb = la + TILE * 16;
#pragma unroll
        for (int j = 0; j < NPC; ++j) g_dma16(ra, la + (wid * NPC + j) * 1024, a_off[j] + chunk * a_step);
#pragma unroll
        for (int j = 0; j < NPC; ++j) g_dma16(rb, lb + (wid * NPC + j) * 1024, b_off[j] + chunk * b_step);
    };
    static_assert(NST >= 2 && NST <= 4, "the tail waits below cover rings of two to four stages");

    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    const int nch = g.Cin / (8 * KS);              // >= NST - 1 (host-checked)
#pragma unroll
    for (int c = 0; c < NST - 1; ++c) issue(c, c);
    // what the epilogue needs from memory, asked for now (a block lives for a few microseconds: a dependent load at its end is a stall)
    float undo = 1.f;
    if constexpr (NP == 2) {
        const float* meta = reinterpret_cast<const float*>(reinterpret_cast<const _Float16*>(g.wP) + (int64_t)NP * g.Cin * M);
        float x_inv;
        (void)amax_scale(amax_read(g.x_slots), false, x_inv);
        undo = meta[1] * x_inv;
    }
    float s_inv;
    const float s_up = NP == 1 ? 1.f : amax_scale(amax_read(g.y_slots), false, s_inv);
    float bsv[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) bsv[t][gq] = g.bias ? g.bias[(m0 + wr * 64 + t * 32 + 8 * gq + 4 * kh) >> 2] : 0.f;
    slot_wait<(NST - 2) * 2 * NPC>();              // chunk 0 landed
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        const int buf = c % NST;
        if (c + NST - 1 < nch) issue(c + NST - 1, (c + NST - 1) % NST);
        const u32x4g* A = lds + buf * 2 * TILE + wr * 64 + l31;
        const u32x4g* Bt = lds + buf * 2 * TILE + TILE + wc * 64 + l31;
#pragma unroll
        for (int s = 0; s < KS / 2; ++s) {                           // one K = 16 step: slots 2 s (kh = 0) and 2 s + 1 (kh = 1)
            u32x4g af[2][NP], bf[2][NP];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < NP; ++q) af[t][q] = A[((2 * s + kh) * NP + q) * 128 + t * 32];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int q = 0; q < NP; ++q) bf[u][q] = Bt[((2 * s + kh) * NP + q) * 128 + u * 32];
            if constexpr (NP == 2) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[t][u] = s_mfma<NP>(af[t][1], bf[u][0], acc[t][u]);      // w_mid . x_hi
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[t][u] = s_mfma<NP>(af[t][0], bf[u][1], acc[t][u]);      // w_hi . x_mid
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[t][u] = s_mfma<NP>(af[t][0], bf[u][0], acc[t][u]);          // w_hi . x_hi
        }
        // chunk c + 1 landed: all but the pieces of the chunks behind it (in flight: min(NST - 2, nch - 2 - c) chunks of 2 NPC pieces)
        const int behind = min(NST - 2, nch - 2 - c);
        constexpr int PPC = 2 * NPC;                             // DMA pieces per wave and chunk
        if (behind >= 2) slot_wait<2 * PPC>();
        else if (behind == 1) slot_wait<1 * PPC>();
        else slot_wait<0>();
        __syncthreads();                                        // ... and every wave is done with this buffer
    }

    // ---- epilogue: undo the operand scales, add the bias, write whole pre-split slots (as convt_gemm_kernel's out16_split form)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int p = p0 + wc * 64 + u * 32 + l31, y = p / g.w, x = p % g.w;
            // A lane holds, of the 8 channels c8 * 8 + {0 .. 7} of this 32-row group, those of its parity (kh = 0: even, kh = 1: odd;
            // channel 2 gq + kh in acc[4 gq ..]) with all four sub-pixels (di, dj) of input pixel p.  v_permlane32_swap trades the half
            // it does not keep with the partner lane: afterwards a kh = 0 lane owns the two output pixels of row di = 0 and a kh = 1
            // lane those of row di = 1, all 8 channels each = whole slots.
            const int c8 = (m0 + wr * 64 + t * 32) >> 5;
            float px[2][8];                         // [dj][channel of the group]
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float bs = bsv[t][gq];
#pragma unroll
                for (int dj = 0; dj < 2; ++dj) {
                    const unsigned d0 = __builtin_bit_cast(unsigned, fmaf(acc[t][u][4 * gq + dj], undo, bs));
                    const unsigned d1 = __builtin_bit_cast(unsigned, fmaf(acc[t][u][4 * gq + 2 + dj], undo, bs));
                    const auto sw = __builtin_amdgcn_permlane32_swap(d0, d1, false, false);
                    const unsigned even = sw[0], odd = sw[1];
                    px[dj][2 * gq] = __builtin_bit_cast(float, even);
                    px[dj][2 * gq + 1] = __builtin_bit_cast(float, odd);
                }
            }
            u32x4g* dst = reinterpret_cast<u32x4g*>(reinterpret_cast<unsigned*>(g.yP) + (int64_t)b * g.y_bs) +
                          ((int64_t)(c8 * (2 * g.h) + 2 * y + kh) * NP) * (2 * g.w) + 2 * x;
#pragma unroll
            for (int dj = 0; dj < 2; ++dj) {
                u32x4g hi, mid;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if constexpr (NP == 2) {
                        unsigned hh, mm;
                        split2h_s(px[dj][2 * k], px[dj][2 * k + 1], s_up, hh, mm);
                        hi[k] = hh;
                        mid[k] = mm;
                    } else {
                        hi[k] = g_pack(px[dj][2 * k], px[dj][2 * k + 1]);
                    }
                }
                dst[dj] = hi;
                if constexpr (NP == 2) dst[2 * g.w + dj] = mid;
            }
        }
}

// ---- input gradient on slot operands:  dx[b][ci][i][j] = sum_{c,di,dj} dy[b][c][2i+di][2j+dj] W[ci][c][di][dj]   (M = Cin, N = pixels, K = 4 Ct)
//   dyP [B][Ct/8][Ho][NP][Wo][8]   the up-sampled half of the concat gradient, written pre-split by the input gradient of the decoder
//                                  block's first convolution (conv_split.hip: SpPreArgs::zP), parts of 2^k dy (rule `always`, dy_slots)
//   wdP [(Ct/8) 4][NP][Cin][8]     K-slot (c8, q = 2 di + dj) = channels 8 c8 .. 8 c8 + 7 at sub-pixel q: the dy slot of output pixel
//                                  (2i + di, 2j + dj) IS the B fragment of input pixel (i, j) -- the inverse pixel shuffle is an address
// Same skeleton as convt_slot_fwd_kernel; the epilogue writes fp32 dx (the gradient of the block below's activation).
struct SDArgs {
    const void* wdP;
    const void* dyP;
    int64_t dy_bs;                 // 4-byte units
    float* dx;
    int64_t dx_bs;                 // elements
    const unsigned* dy_slots;
    int B, Cin, Ct, h, w, mTiles, nTiles;
};

template <int NP>
__global__ __launch_bounds__(256, SLOT_OCC) void convt_slot_dgrad_kernel(SDArgs g) {
    constexpr int KS = 4 / NP;
    constexpr int TILE = KS * NP * 128;
    constexpr int NPC = TILE / 256;
    constexpr int NST = SLOT_NST;
    __shared__ __attribute__((aligned(16))) u32x4g lds[NST * 2 * TILE];
    const int bid = xcd_order(gridDim.x);
    const int mt = bid % g.mTiles, nt = bid / g.mTiles;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1, l31 = lane & 31, kh = lane >> 5;
    const int hw = g.h * g.w, M = g.Cin, Ho = 2 * g.h, Wo = 2 * g.w;
    const int m0 = mt * 128, n0 = nt * 128;
    const int b = n0 / hw, p0 = n0 % hw;

    const i32x4g ra = g_rsrc(g.wdP, (int64_t)(g.Ct / 2) * NP * M * 16);
    const i32x4g rb = g_rsrc(reinterpret_cast<const unsigned*>(g.dyP) + (int64_t)b * g.dy_bs, (int64_t)(g.Ct / 8) * Ho * NP * Wo * 16);
    unsigned a_off[NPC], b_px[NPC];
    int b_ks[NPC];
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
        const int s = (wid * NPC + j) * 64 + lane;
        const int ks = s / (NP * 128), part = (s >> 7) % NP, r = s & 127;
        a_off[j] = (unsigned)((((int64_t)ks * NP + part) * M + m0 + r) * 16);
        const int p = p0 + r, i = p / g.w, jx = p % g.w;
        b_px[j] = (unsigned)(((((int64_t)2 * i) * NP + part) * Wo + 2 * jx) * 16);
        b_ks[j] = ks;
    }
    const unsigned a_step = (unsigned)((int64_t)KS * NP * M * 16);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) u32x4g*)lds;
    auto issue = [&](int chunk, int buf) __attribute__((always_inline)) {
        const unsigned la = lds0 + (unsigned)(buf * 2 * TILE * 16), lb = la + TILE * 16;
#pragma unroll
        for (int j = 0; j < NPC; ++j) g_dma16(ra, la + (wid * NPC + j) * 1024, a_off[j] + chunk * a_step);
#pragma unroll
        for (int j = 0; j < NPC; ++j) {
            const int kq = chunk * KS + b_ks[j], c8 = kq >> 2, di = (kq >> 1) & 1, dj = kq & 1;
            g_dma16(rb, lb + (wid * NPC + j) * 1024, b_px[j] + (unsigned)((((int64_t)c8 * Ho + di) * NP * Wo + dj) * 16));
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    const int nch = (g.Ct / 2) / KS;               // K-slots: 4 Ct / 8
#pragma unroll
    for (int c = 0; c < NST - 1; ++c) issue(c, c);
    float undo = 1.f;
    if constexpr (NP == 2) {
        const float* meta = reinterpret_cast<const float*>(reinterpret_cast<const _Float16*>(g.wdP) + (int64_t)NP * 4 * g.Ct * M);
        float dy_inv;
        (void)amax_scale(amax_read(g.dy_slots), true, dy_inv);
        undo = meta[1] * dy_inv;
    }
    slot_wait<(NST - 2) * 2 * NPC>();
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        const int buf = c % NST;
        if (c + NST - 1 < nch) issue(c + NST - 1, (c + NST - 1) % NST);
        const u32x4g* A = lds + buf * 2 * TILE + wr * 64 + l31;
        const u32x4g* Bt = lds + buf * 2 * TILE + TILE + wc * 64 + l31;
#pragma unroll
        for (int s = 0; s < KS / 2; ++s) {
            u32x4g af[2][NP], bf[2][NP];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < NP; ++q) af[t][q] = A[((2 * s + kh) * NP + q) * 128 + t * 32];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int q = 0; q < NP; ++q) bf[u][q] = Bt[((2 * s + kh) * NP + q) * 128 + u * 32];
            if constexpr (NP == 2) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[t][u] = s_mfma<NP>(af[t][1], bf[u][0], acc[t][u]);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[t][u] = s_mfma<NP>(af[t][0], bf[u][1], acc[t][u]);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[t][u] = s_mfma<NP>(af[t][0], bf[u][0], acc[t][u]);
        }
        const int behind = min(NST - 2, nch - 2 - c);
        constexpr int PPC = 2 * NPC;
        if (behind >= 2) slot_wait<2 * PPC>();
        else if (behind == 1) slot_wait<1 * PPC>();
        else slot_wait<0>();
        __syncthreads();
    }
    float* xb = g.dx + (int64_t)b * g.dx_bs;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int p = p0 + wc * 64 + u * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = m0 + wr * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                xb[(int64_t)ci * hw + p] = acc[t][u][r] * undo;
            }
        }
}

// ---- weight gradient on slot operands:  dW[ci][c][di][dj] = sum_{b,i,j} x[b][ci][i][j] dy[b][c][2i+di][2j+dj]   (M = Cin, N = 4 Ct, K = pixels)
// Both operands are pre-split tensors, channels contiguous inside a slot and the reduction index (pixels) running ACROSS slots: the
// fragments are read with the transposing LDS read ds_read_b64_tr_b16, exactly as conv3x3_split_wgrad_pre_kernel does (conv_split.hip:
// a 16-lane group fetches 4 pixel slots x 16 channels and every lane receives ITS channel's four pixels).  Block tile: 128 input
// channels (16 slot groups) x 32 output channels (4 groups) x 4 sub-pixels; a K-chunk = 32 input pixels; LDS image of a chunk =
// planes of PXP = 36 slots (32 + 4 pad: 144 dwords = 16 banks mod 64, the conflict-free plane pitch of the 3x3 kernel):
//   A planes [part][16 groups], B planes [part][sub-pixel q][4 groups]: 64 planes = 36 KB per stage, two stages.
// Waves 2 x 2: wr = which 64 input channels, wc = which output row parity (di): a wave's two 32-column tiles are dj = 0, 1.
// The ConvTranspose2d bias gradient (sum of dy over all pixels) rides along as a row of ones: the waves wr == 0 of the blocks mt == 0
// multiply their dy fragments by a constant-one A fragment (row 0 of the result = the column sums).
struct SWArgs {
    const void* xP;
    int64_t x_bs;                  // 4-byte units
    const void* dyP;
    int64_t dy_bs;
    float* slab;                   // [splitK][Cin][4 Ct]
    float* db_part;                // [splitK * 4][Ct] partial bias gradients (NULL: not wanted)
    const unsigned* x_slots;       // guard rule
    const unsigned* dy_slots;      // rule `always`
    int B, Cin, Ct, h, w, lw;      // lw = log2(w)
    int mTiles, nTiles, splitK, chunksPerSplit;
};

// Tile configurations.  BIG = 0: 128 x 128 block tile, 4 waves, 32-pixel chunks, two stages (2 blocks per CU).  BIG = 1 (Cin % 256 == 0,
// Ct % 64 == 0): 256 input channels x (64 output channels x 4 sub-pixels), EIGHT waves of 128 x 64, one block per CU: twice the MFMAs
// per staged byte and per barrier (MFMA busy 0.35 -> 0.40 on the 1024-channel level; a per-chunk 64-bit division in the staging code
// had cost 6 scalar instructions per MFMA).
#ifndef SLOT_WGRAD_BIG
#define SLOT_WGRAD_BIG 1
#endif
#ifndef SLOT_WGRAD_BIG_CP
#define SLOT_WGRAD_BIG_CP 32     // (measured, B = 64, four levels: 32-pixel chunks / 2 stages 1.28 ms, 16-pixel chunks / ring of 3 1.35; the 128 x 128 tile 1.47)
#endif
template <int BIG> struct SwCfg {
    static constexpr int NWAVE = BIG ? 8 : 4;
    static constexpr int TM = BIG ? 4 : 2;                    // 32-row accumulator tiles per wave along M (waves: 2 along M)
    static constexpr int GB = BIG ? 8 : 4;                    // output-channel groups per block (GB * 8 channels)
    static constexpr int APL = 2 * TM * 4;                    // A planes per part (input-channel groups)
    static constexpr int BPL = 4 * GB;                        // B planes per part: [sub-pixel q][group]
    static constexpr int CP = BIG ? SLOT_WGRAD_BIG_CP : 32;   // pixels per chunk
    static constexpr int PXP = CP + 4;                        // plane pitch in slots: 36 / 20 -> 144 / 80 dwords = 16 banks mod 64
    static constexpr int NST = BIG ? (SLOT_WGRAD_BIG_CP == 16 ? 3 : 2) : 2;
    static constexpr int MT = 64 * TM;                        // block tile rows (input channels)
};

__device__ __forceinline__ u32x4g sw_frag(unsigned addr) {   // 8 consecutive pixels (K) of this lane's channel: two transposed reads
    typedef short s16x4g __attribute__((ext_vector_type(4)));
    const s16x4g lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4g*)(uintptr_t)addr);
    const s16x4g hv = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4g*)(uintptr_t)(addr + 64));
    const unsigned long long a = __builtin_bit_cast(unsigned long long, lo), b = __builtin_bit_cast(unsigned long long, hv);
    return u32x4g{(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
}

template <int NP, int BIG>
__global__ __launch_bounds__(SwCfg<BIG>::NWAVE * 64, BIG ? 1 : 2) void convt_slot_wgrad_kernel(SWArgs g) {
    using C = SwCfg<BIG>;
    constexpr int PXP = C::PXP, TM = C::TM, GB = C::GB, APL = C::APL, BPL = C::BPL, CP = C::CP, NST = C::NST, NWAVE = C::NWAVE;
    constexpr int A_SLOTS = APL * PXP, B_SLOTS = BPL * PXP;                 // per part; both whole numbers of 64-slot DMA pieces
    static_assert(A_SLOTS % 64 == 0 && B_SLOTS % 64 == 0 && A_SLOTS == B_SLOTS, "a DMA piece must be all x or all dy");
    constexpr int STAGE = NP * (A_SLOTS + B_SLOTS);
    constexpr int NPIECE = STAGE / 64;
    constexpr int NPW = (NPIECE + NWAVE - 1) / NWAVE;                       // pieces per wave and chunk
    constexpr int LOG_CP = CP == 16 ? 4 : 5;
    extern __shared__ __attribute__((aligned(16))) unsigned char sw_smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)sw_smem;
    const int bid = xcd_order(gridDim.x);
    const int tiles = g.mTiles * g.nTiles;
    const int ks = bid / tiles, tile = bid % tiles;
    const int mt = tile % g.mTiles, nt = tile / g.mTiles;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // waves: 2 along M (wr); along N the row parity di (and, BIG, the channel half ch)
    const int wr = BIG ? wid >> 2 : wid >> 1, di = wid & 1, chf = BIG ? (wid >> 1) & 1 : 0;
    const int l31 = lane & 31, kh = lane >> 5;
    const int hw = g.h * g.w, Ho = 2 * g.h, Wo = 2 * g.w;
    const int ch0 = ks * g.chunksPerSplit;
    const int ch1 = min(ch0 + g.chunksPerSplit, (int)(((int64_t)g.B * hw) >> LOG_CP));

    // ---- staging: slot s = (wid * NPW + k) * 64 + lane of the stage image [part][A planes | B planes][PXP]; A plane = x group
    // (MT / 8) mt + plane; B plane = (q, group): q = plane / GB, dy group GB nt + plane % GB
    int st_kind[NPW];              // 0: x, 1: dy, -1: nothing (padding slot or beyond the image)
    unsigned st_base[NPW];
    int st_px[NPW];
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
        const int s = (wid * NPW + k) * 64 + lane;
        const int part = s / (A_SLOTS + B_SLOTS), r = s % (A_SLOTS + B_SLOTS);
        const int px = r % PXP;
        st_px[k] = px;
        if (s >= STAGE || px >= CP) {
            st_kind[k] = -1;
            st_base[k] = 0;
        } else if (r < A_SLOTS) {
            const int plane = r / PXP;
            st_kind[k] = 0;
            st_base[k] = (unsigned)(((((int64_t)((C::MT / 8) * mt + plane)) * g.h * NP + part) * g.w) * 16);
        } else {
            const int plane = (r - A_SLOTS) / PXP, q = plane / GB, c8 = GB * nt + plane % GB;
            st_kind[k] = 1;
            st_base[k] = (unsigned)((((((int64_t)c8 * Ho + (q >> 1)) * NP + part) * Wo) + (q & 1)) * 16);
        }
    }
    // the chunks of a block are consecutive: image and first pixel of the next chunk to stage are carried along (a 64-bit division per
    // chunk was 6 scalar instructions per MFMA)
    int is_b, is_p0;
    {
        const int64_t pix = (int64_t)ch0 << LOG_CP;
        is_b = (int)(pix / hw);
        is_p0 = (int)(pix - (int64_t)is_b * hw);
    }
    i32x4g rx = g_rsrc(reinterpret_cast<const unsigned*>(g.xP) + (int64_t)is_b * g.x_bs, (int64_t)(g.Cin / 8) * hw * NP * 16);
    i32x4g rd = g_rsrc(reinterpret_cast<const unsigned*>(g.dyP) + (int64_t)is_b * g.dy_bs, (int64_t)(g.Ct / 8) * Ho * NP * Wo * 16);
    auto issue = [&](int chunk, int buf) __attribute__((always_inline)) {
        (void)chunk;                                       // (called for ch0, ch0 + 1, ... in order)
        const int p0 = is_p0;
        const unsigned lb = lds0 + (unsigned)(buf * STAGE * 16);
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            const int pc = wid * NPW + k;
            if (pc >= NPIECE) continue;                                        // (wave-uniform)
            // (a part's A and B regions are whole pieces: a piece is all x or all dy, wave-uniformly -- the resource must be scalar)
            const bool is_x = ((pc / (A_SLOTS / 64)) & 1) == 0;
            const int p = p0 + st_px[k], i = p >> g.lw, j = p & (g.w - 1);
            unsigned off;
            if (st_kind[k] == 0) off = st_base[k] + (unsigned)((i * NP * g.w + j) * 16);
            else if (st_kind[k] == 1) off = st_base[k] + (unsigned)((2 * i * NP * Wo + 2 * j) * 16);
            else off = 0x80000000u;                                             // out of the resource's range: the slot reads as zero
            if (is_x) g_dma16(rx, lb + (unsigned)(pc * 1024), off);
            else g_dma16(rd, lb + (unsigned)(pc * 1024), off);
        }
        is_p0 += CP;
        if (is_p0 >= hw) {                                 // (hw % CP == 0: a chunk never straddles two images)
            is_p0 = 0;
            ++is_b;
            rx = g_rsrc(reinterpret_cast<const unsigned*>(g.xP) + (int64_t)is_b * g.x_bs, (int64_t)(g.Cin / 8) * hw * NP * 16);
            rd = g_rsrc(reinterpret_cast<const unsigned*>(g.dyP) + (int64_t)is_b * g.dy_bs, (int64_t)(g.Ct / 8) * Ho * NP * Wo * 16);
        }
    };
    // DMA pieces this wave issues per chunk (the last waves may own one fewer)
    const int my_pieces = min(NPW, max(0, NPIECE - wid * NPW));

    f32x16 acc[TM][2], accb[2];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[u][r] = 0.f;
    const bool bias_wave = g.db_part != nullptr && mt == 0 && wr == 0;       // (wave-uniform)
    u32x4g ones;
    {
        const unsigned one2 = NP == 2 ? 0x3c003c00u : 0x3f803f80u;             // (1.0, 1.0) as fp16 / bf16 pairs
        ones = u32x4g{one2, one2, one2, one2};
    }

    // transposed-read lane bases (bytes): lane = 16 g + 4 q + p supplies pixel slot q, channels 4 p .. 4 p + 3 of the channel half g & 1
    const int tq = (lane >> 2) & 3, tp = lane & 3, tg = (lane >> 4) & 1;
    const unsigned lane_off = (unsigned)((((2 * tg + (tp >> 1)) * PXP + 8 * kh + tq) * 16) + (tp & 1) * 8);
    const unsigned a_lane = lane_off + (unsigned)((wr * TM * 4) * PXP * 16);                                   // + t * 4 planes
    const unsigned b_lane = lane_off + (unsigned)((A_SLOTS + ((2 * di) * GB + chf * 4) * PXP) * 16);          // + u * GB planes (q = 2 di + u)
    constexpr unsigned PART = (A_SLOTS + B_SLOTS) * 16;                                                        // bytes between the parts

    // ring: chunks c + 1 .. c + NST - 1 in flight while chunk c is consumed
    const int n_my = ch1 - ch0;
#pragma unroll
    for (int c = 0; c < NST - 1; ++c)
        if (c < n_my) issue(ch0 + c, c);
    auto wait_behind = [&](int behind) __attribute__((always_inline)) {       // all but `behind` chunks of this wave's pieces have landed
        // (vmcnt counts this wave's own pieces; my_pieces is NPW or NPW - 1: wait for the stricter of the two -- exact for NPW waves)
        if (behind >= 1 && NST > 2) {
            if (my_pieces == NPW) slot_wait<NPW>();
            else slot_wait<(NPW > 1 ? NPW - 1 : 0)>();
        } else {
            slot_wait<0>();
        }
    };
    if (n_my > 0) {
        wait_behind(min(NST - 2, n_my - 1));
        __syncthreads();
    }
    for (int c = ch0; c < ch1; ++c) {
        const int buf = (c - ch0) % NST;
        if (c + NST - 1 < ch1) issue(c + NST - 1, (c - ch0 + NST - 1) % NST);
        const unsigned base = lds0 + (unsigned)(buf * STAGE * 16);
#pragma unroll
        for (int s = 0; s < CP / 16; ++s) {                                     // K = 16 pixels per step
            u32x4g af[TM][NP], bf[2][NP];
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int q = 0; q < NP; ++q) af[t][q] = sw_frag(base + a_lane + (unsigned)(t * 4 * PXP * 16 + s * 256) + q * PART);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int q = 0; q < NP; ++q) bf[u][q] = sw_frag(base + b_lane + (unsigned)(u * GB * PXP * 16 + s * 256) + q * PART);
            if constexpr (NP == 2) {
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[t][u] = s_mfma<NP>(af[t][1], bf[u][0], acc[t][u]);
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u) acc[t][u] = s_mfma<NP>(af[t][0], bf[u][1], acc[t][u]);
            }
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[t][u] = s_mfma<NP>(af[t][0], bf[u][0], acc[t][u]);
            if (bias_wave) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    accb[u] = s_mfma<NP>(ones, bf[u][0], accb[u]);
                    if constexpr (NP == 2) accb[u] = s_mfma<NP>(ones, bf[u][1], accb[u]);
                }
            }
        }
        // chunk c + 1 has landed: everything but the chunks behind it (min(NST - 2, ch1 - 2 - c))
        wait_behind(min(NST - 2, ch1 - 2 - c));
        __syncthreads();
    }

    float undo = 1.f, dy_inv = 1.f;
    if constexpr (NP == 2) {
        float x_inv;
        (void)amax_scale(amax_read(g.x_slots), false, x_inv);
        (void)amax_scale(amax_read(g.dy_slots), true, dy_inv);
        undo = x_inv * dy_inv;
    }
    float* slab = g.slab + (int64_t)ks * g.Cin * 4 * g.Ct;
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int col = (GB * 8 * nt + chf * 32 + l31) * 4 + 2 * di + u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = C::MT * mt + wr * (TM * 32) + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                slab[(int64_t)ci * 4 * g.Ct + col] = acc[t][u][r] * undo;
            }
        }
    if (bias_wave && kh == 0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) g.db_part[((int64_t)ks * 4 + 2 * di + u) * g.Ct + GB * 8 * nt + chf * 32 + l31] = accb[u][0] * dy_inv;
    }
}

// w [Cin][Ct][2][2] fp32 (nn.ConvTranspose2d) -> wP [Cin/8][NP][4 Ct][8]: column m = 4 c + (2 di + dj) (== the weight's own memory
// order), K-slot = 8 consecutive input channels; NP = 2: fp16 (hi | mid) parts of 2^k w with max |w| in [2^13, 2^14) (k from the
// magnitude slots wamax; (2^k, 2^-k) stored behind the pack as two floats); NP = 1: bf16(w)
__global__ void convt_pack_slots_kernel(const float* __restrict__ w, void* __restrict__ wP, void* __restrict__ wdP, int Cin, int Ct, int np,
                                        const unsigned* __restrict__ wamax) {
    const int M = 4 * Ct;
    const int64_t n = (int64_t)Cin * M;
    float winv = 1.f;
    const float wscale = np == 2 ? amax_scale(amax_read(wamax), true, winv) : 1.f;
    if (np == 2 && blockIdx.x == 0 && threadIdx.x == 0) {
        float* meta = reinterpret_cast<float*>(reinterpret_cast<_Float16*>(wP) + 2 * n);
        meta[0] = wscale;
        meta[1] = winv;
        if (wdP) {
            float* md = reinterpret_cast<float*>(reinterpret_cast<_Float16*>(wdP) + 2 * n);
            md[0] = wscale;
            md[1] = winv;
        }
    }
    // (the second half of the index range, if asked for: the K-slot pack of the input gradient -- convt_slot_dgrad_kernel)
    for (int64_t i2 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i2 < (wdP ? 2 * n : n); i2 += (int64_t)gridDim.x * blockDim.x) {
        if (i2 >= n) {
            const int64_t i = i2 - n;
            const int kc = (int)(i & 7);
            const int64_t r = i >> 3;
            const int ci = (int)(r % Cin), ks = (int)(r / Cin);        // K-slot = c8 * 4 + q
            const int c = (ks >> 2) * 8 + kc, q = ks & 3;
            const float v = w[((int64_t)ci * Ct + c) * 4 + q];
            if (np == 2) {
                _Float16* o = reinterpret_cast<_Float16*>(wdP);
                const float vs = v * wscale;
                const _Float16 hi = (_Float16)vs;
                o[(((int64_t)ks * 2 + 0) * Cin + ci) * 8 + kc] = hi;
                o[(((int64_t)ks * 2 + 1) * Cin + ci) * 8 + kc] = (_Float16)(vs - (float)hi);
            } else {
                reinterpret_cast<__bf16*>(wdP)[((int64_t)ks * Cin + ci) * 8 + kc] = (__bf16)v;
            }
            continue;
        }
        const int64_t i = i2;
        const int kc = (int)(i & 7);
        const int64_t r = i >> 3;
        const int m = (int)(r % M), c8 = (int)(r / M);
        const float v = w[(int64_t)(c8 * 8 + kc) * M + m];
        if (np == 2) {
            _Float16* o = reinterpret_cast<_Float16*>(wP);
            const float vs = v * wscale;
            const _Float16 hi = (_Float16)vs;
            o[(((int64_t)c8 * 2 + 0) * M + m) * 8 + kc] = hi;
            o[(((int64_t)c8 * 2 + 1) * M + m) * 8 + kc] = (_Float16)(vs - (float)hi);
        } else {
            reinterpret_cast<__bf16*>(wP)[((int64_t)c8 * M + m) * 8 + kc] = (__bf16)v;
        }
    }
}

__global__ void convt_absmax_kernel(const float* __restrict__ x, int64_t n, unsigned* __restrict__ slots) {
    float m = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m == m)
        atomicMax(slots + ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (AMAX_SLOTS - 1)) * AMAX_STRIDE, __builtin_bit_cast(unsigned, m));
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

void wgrad_plan(int B, int Cin, int Ct, int h, int w, int& splitK, int& per) {
    const int64_t chunks = (int64_t)B * h * w / KP;
    const int64_t tiles = (int64_t)(Cin / 128) * (Ct / 32);
    int64_t k = std::max<int64_t>(1, (512 + tiles - 1) / tiles);                  // ~2 blocks per CU
    k = std::min<int64_t>(k, std::max<int64_t>(1, chunks / 16));                  // at least 16 chunks (256 K-steps) per block
    k = std::min<int64_t>(k, std::max<int64_t>(1, (256ll << 20) / ((int64_t)Cin * 4 * Ct * 4)));
    per = (int)((chunks + k - 1) / k);
    splitK = (int)((chunks + per - 1) / per);
}

}  // namespace

namespace onet {

// `prec`: operand precision of the three GEMMs, per call (the `operand_bf16` argument of the onet_convT2x2_* entry points):
// 0 = fp32 MFMA, 1 = bf16 operands with fp32 accumulation (the bf16 conv path of BASELINE configs[2]), 2 = split bf16 operands
// (fp32-level results on the bf16 pipe)

// Fast-path predicates + launches; return ONET_NOT_TAKEN (1) when the shape is not taken (caller falls back to conv_mfma.hip)
int convt_gemm_fwd(const float* x, int64_t x_bs, const float* wq, const float* bias, float* y, int64_t y_bs, void* y16, int64_t y16_bs,
                   int B, int Cin, int Ct, int h, int w, int Ho, int Wo, int pt, int pl, int prec, hipStream_t st, int y16_split,
                   const void* out_slots) {
    const int64_t hw = (int64_t)h * w;
    if (pt || pl || Ho != 2 * h || Wo != 2 * w || (Cin % KC) || (Ct % 32) || (hw % 128) || (w & 1) || !aligned16(x) || !aligned16(wq) ||
        (x_bs & 3) || (reinterpret_cast<uintptr_t>(y) & 7) || (y_bs & 1) || (int64_t)Cin * hw * 4 >= (1ll << 31) ||
        (int64_t)Cin * 4 * Ct * 4 >= (1ll << 31))
        return 1;
    if (y16 && !y16_split) return 1;                   // (round 2's plain bf16 copy of y is gone with the kernels that read it)
    if (y16 && y16_split && ((reinterpret_cast<uintptr_t>(y16) & 15) || (y16_bs & 3) || (Ct % 32))) return 1;
    if (!y && !y16) return 1;
    GArgs g{wq, x, y, bias, nullptr, (__bf16*)y16, y16_bs, 0, x_bs, y_bs, B, Cin, Ct, h, w, Wo, Ho * Wo, (4 * Ct) / 128, (int)(B * hw / 128), 1, 0, y16_split, (const unsigned*)out_slots};
    const int64_t blocks = (int64_t)g.mTiles * g.nTiles;
    if (blocks <= 0 || blocks >= (1ll << 31)) return 1;
    if (prec == 2) hipLaunchKernelGGL((convt_gemm_kernel<0, 2>), dim3((unsigned)blocks), dim3(256), 0, st, g);
    else if (prec) hipLaunchKernelGGL((convt_gemm_kernel<0, 1>), dim3((unsigned)blocks), dim3(256), 0, st, g);
    else hipLaunchKernelGGL((convt_gemm_kernel<0, 0>), dim3((unsigned)blocks), dim3(256), 0, st, g);
    return check_launch("convt_gemm_kernel<0>");
}

// |y[co]| <= max over the four sub-pixel filters of sum_ci |W[ci][co][di][dj]| * max |x| + |bias[co]|: the bound of the up-sampled tensor
// from the exact maximum of its input (x_slots) -- one block per output channel, atomicMax into the output's magnitude slots
__global__ __launch_bounds__(256) void convt_out_bound_kernel(const float* __restrict__ w, const float* __restrict__ bias, int Cin, int Ct,
                                                              const unsigned* __restrict__ x_slots, unsigned* __restrict__ out_slots) {
    __shared__ float red[4][4];
    const float xmax = amax_read(x_slots);
    const int co = blockIdx.x;
    float sm[4] = {0.f, 0.f, 0.f, 0.f};
    for (int ci = threadIdx.x; ci < Cin; ci += 256) {
        const f32x4g v = *reinterpret_cast<const f32x4g*>(w + ((int64_t)ci * Ct + co) * 4);
        sm[0] += fabsf(v[0]);
        sm[1] += fabsf(v[1]);
        sm[2] += fabsf(v[2]);
        sm[3] += fabsf(v[3]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sm[k] += __shfl_xor(sm[k], o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = sm[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) m = fmaxf(m, (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]));
        const float bound = m * xmax * 1.00001f + (bias ? fabsf(bias[co]) : 0.f);
        if (bound == bound) atomicMax(out_slots + (co & 63) * AMAX_STRIDE, __builtin_bit_cast(unsigned, bound));
    }
}

int convt_out_bound(const float* w, const float* bias, int Cin, int Ct, const void* x_slots, void* out_slots, hipStream_t st) {
    hipLaunchKernelGGL(convt_out_bound_kernel, dim3((unsigned)Ct), dim3(256), 0, st, w, bias, Cin, Ct, (const unsigned*)x_slots,
                       (unsigned*)out_slots);
    return check_launch("convt_out_bound_kernel");
}

int64_t convt_gemm_dbias_ws_bytes(int B, int Ct, int h, int w) { return (int64_t)B * h * w / 128 * Ct * 4; }

// dbias != NULL: also the ConvTranspose2d bias gradient, taken from the dy rows the GEMM stages anyway (dbias_ws: partials)
int convt_gemm_dgrad(const float* dy, int64_t dy_bs, const float* wd, float* dx, int64_t dx_bs, float* dbias, float* dbias_ws, int B,
                     int Cin, int Ct, int h, int w, int Ho, int Wo, int pt, int pl, int prec, hipStream_t st) {
    const int64_t hw = (int64_t)h * w;
    if (pt || pl || Ho != 2 * h || Wo != 2 * w || (Cin % 128) || (Ct % 4) || (hw % 128) || (w & 1) || !aligned16(dy) || !aligned16(wd) ||
        (dy_bs & 3) || (int64_t)Ct * Ho * Wo * 4 >= (1ll << 31) || (int64_t)Cin * 4 * Ct * 4 >= (1ll << 31))
        return 1;
    if (dbias && !dbias_ws) return 1;
    GArgs g{wd, dy, dx, nullptr, dbias ? dbias_ws : nullptr, nullptr, 0, 0, dy_bs, dx_bs, B, Cin, Ct, h, w, Wo, Ho * Wo, Cin / 128, (int)(B * hw / 128), 1, 0};
    const int64_t blocks = (int64_t)g.mTiles * g.nTiles;
    if (blocks <= 0 || blocks >= (1ll << 31)) return 1;
    if (prec == 2) hipLaunchKernelGGL((convt_gemm_kernel<1, 2>), dim3((unsigned)blocks), dim3(256), 0, st, g);
    else if (prec) hipLaunchKernelGGL((convt_gemm_kernel<1, 1>), dim3((unsigned)blocks), dim3(256), 0, st, g);
    else hipLaunchKernelGGL((convt_gemm_kernel<1, 0>), dim3((unsigned)blocks), dim3(256), 0, st, g);
    int rc = check_launch("convt_gemm_kernel<1>");
    if (rc || !dbias) return rc;
    hipLaunchKernelGGL(convt_dbias_reduce_kernel, dim3((unsigned)Ct), dim3(256), 0, st, (const float*)dbias_ws, dbias, g.nTiles, Ct);
    return check_launch("convt_dbias_reduce_kernel");
}

int64_t convt_gemm_wgrad_ws_bytes(int B, int Cin, int Ct, int h, int w) {
    if ((Cin % 128) || (Ct % 32) || (((int64_t)h * w) % KP)) return 0;
    int k, per;
    wgrad_plan(B, Cin, Ct, h, w, k, per);
    return (int64_t)k * Cin * 4 * Ct * 4;
}

int convt_gemm_wgrad(const float* x, int64_t x_bs, const float* dy, int64_t dy_bs, float* dw, void* ws, int64_t ws_bytes, int B,
                     int Cin, int Ct, int h, int w, int Ho, int Wo, int pt, int pl, int prec, hipStream_t st) {
    const int64_t hw = (int64_t)h * w;
    if (pt || pl || Ho != 2 * h || Wo != 2 * w || (Cin % 128) || (Ct % 32) || (hw % KP) || (w & 1) || !aligned16(x) || !aligned16(dy) ||
        !aligned16(dw) || (x_bs & 3) || (dy_bs & 3) || (int64_t)Cin * hw * 4 >= (1ll << 31) || (int64_t)Ct * Ho * Wo * 4 >= (1ll << 31))
        return 1;
    GArgs g{x, dy, (float*)ws, nullptr, nullptr, nullptr, 0, x_bs, dy_bs, 0, B, Cin, Ct, h, w, Wo, Ho * Wo, Cin / 128, Ct / 32, 1, 0};
    wgrad_plan(B, Cin, Ct, h, w, g.splitK, g.chunksPerSplit);
    const int64_t n = (int64_t)Cin * 4 * Ct;
    if (ws_bytes < (int64_t)g.splitK * n * 4) return 1;
    const int64_t blocks = (int64_t)g.splitK * g.mTiles * g.nTiles;
    if (prec == 2) hipLaunchKernelGGL(convt_wgrad_gemm_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, g);
    else if (prec) hipLaunchKernelGGL(convt_wgrad_gemm_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, g);
    else hipLaunchKernelGGL(convt_wgrad_gemm_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, st, g);
    int rc = check_launch("convt_wgrad_gemm_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(convt_wgrad_reduce_kernel, dim3((unsigned)cdiv(n / 4, 256)), dim3(256), 0, st, (const float*)ws, dw, g.splitK,
                       n / 4, 0);
    return check_launch("convt_wgrad_reduce_kernel");
}

}  // namespace onet

// split-K plan of the slot-operand weight gradient: -> big (256 x 256 tiles, 16-pixel chunks) or not (128 x 128, 32-pixel chunks)
static bool wgrad_slots_plan(int B, int Cin, int Ct, int h, int w, int& splitK, int& per) {
    const bool big = (Cin % 256) == 0 && (Ct % 64) == 0;
    const int64_t chunks = (int64_t)B * h * w / (big ? SLOT_WGRAD_BIG_CP : 32);
    const int64_t tiles = big ? (int64_t)(Cin / 256) * (Ct / 64) : (int64_t)(Cin / 128) * (Ct / 32);
    int64_t k = std::max<int64_t>(1, ((big ? 256 : 512) + tiles - 1) / tiles);   // one (two) blocks per CU
    k = std::min<int64_t>(k, std::max<int64_t>(1, chunks / (big ? 512 / SLOT_WGRAD_BIG_CP : 16)));     // at least 512 pixels per block
    k = std::min<int64_t>(k, std::max<int64_t>(1, (256ll << 20) / ((int64_t)Cin * 4 * Ct * 4)));
    per = (int)((chunks + k - 1) / k);
    splitK = (int)((chunks + per - 1) / per);
    return big;
}

template <int NP, int BIG>
static int launch_wgrad_slots(SWArgs g, hipStream_t st) {
    using C = SwCfg<BIG>;
    const int lds = NP * (C::APL + C::BPL) * C::PXP * 16 * C::NST;
    auto kern = convt_slot_wgrad_kernel<NP, BIG>;
    static PerDeviceOnce once;
    if (once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int64_t blocks = (int64_t)g.splitK * g.mTiles * g.nTiles;
    ONET_REQUIRE(blocks > 0 && blocks < (1ll << 31), "convT2x2_wgrad_slots: grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(C::NWAVE * 64), lds, st, g);
    return check_launch("convt_slot_wgrad_kernel");
}

extern "C" {

// ---- ConvTranspose2d GEMMs on slot operands (round 5; include/onet_hip.h)
int onet_convT2x2_pack_weights_slots(const float* w, void* wP, void* wdP, void* amax_ws, int Cin, int Ct, int nparts, void* stream) {
    ONET_REQUIRE(w && wP && Cin > 0 && Ct > 0 && (Cin % 8) == 0 && (!wdP || (Ct % 8) == 0) && (nparts == 1 || nparts == 2),
                 "convT2x2_pack_weights_slots: bad args");
    ONET_REQUIRE(nparts == 1 || amax_ws, "convT2x2_pack_weights_slots: the fp16 pack needs the 8 KB magnitude workspace");
    hipStream_t st = as_stream(stream);
    const int64_t n = (int64_t)Cin * 4 * Ct;
    if (nparts == 2) {
        (void)hipMemsetAsync(amax_ws, 0, AMAX_SLOTS * AMAX_STRIDE * sizeof(unsigned), st);
        hipLaunchKernelGGL(convt_absmax_kernel, dim3((unsigned)std::min<int64_t>((n + 1023) / 1024, 1024)), dim3(256), 0, st, w, n, (unsigned*)amax_ws);
        int rc = check_launch("convt_absmax_kernel");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(convt_pack_slots_kernel, dim3((unsigned)std::min<int64_t>(((wdP ? 2 : 1) * n + 255) / 256, 8192)), dim3(256), 0, st, w, wP,
                       wdP, Cin, Ct, nparts, (const unsigned*)amax_ws);
    return check_launch("convt_pack_slots_kernel");
}

int onet_convT2x2_dgrad_slots(const void* dyP, int64_t dyP_bs, const void* dy_amax, const void* wdP, float* dx, int64_t dx_bs, int nparts, int B,
                              int Cin, int Ct, int h, int w, void* stream) {
    ONET_REQUIRE(dyP && wdP && dx && B > 0 && Cin > 0 && Ct > 0 && h > 0 && w > 0 && (nparts == 1 || nparts == 2), "convT2x2_dgrad_slots: bad args");
    const int64_t hw = (int64_t)h * w;
    if ((Cin % 128) || (Ct % 32) || (hw % 128) || (w & 1) || !aligned16(dyP) || !aligned16(wdP) || (dyP_bs & 3) ||
        (int64_t)Ct * 4 * hw * 2 * nparts >= (1ll << 31) || (int64_t)Cin * 4 * Ct * 2 * nparts >= (1ll << 31))
        return 1;
    ONET_REQUIRE(dyP_bs >= (int64_t)Ct * 4 * hw * nparts / 2 && dx_bs >= (int64_t)Cin * hw, "convT2x2_dgrad_slots: batch stride too small");
    SDArgs g{wdP, dyP, dyP_bs, dx, dx_bs, (const unsigned*)dy_amax, B, Cin, Ct, h, w, Cin / 128, (int)(B * hw / 128)};
    const int64_t blocks = (int64_t)g.mTiles * g.nTiles;
    ONET_REQUIRE(blocks > 0 && blocks < (1ll << 31), "convT2x2_dgrad_slots: grid too large");
    if (nparts == 2) hipLaunchKernelGGL(convt_slot_dgrad_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), g);
    else hipLaunchKernelGGL(convt_slot_dgrad_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), g);
    return check_launch("convt_slot_dgrad_kernel");
}

int64_t onet_convT2x2_wgrad_slots_ws_bytes(int B, int Cin, int Ct, int h, int w) {
    if ((Cin % 128) || (Ct % 32) || (((int64_t)h * w) % 128) || (w & (w - 1))) return 0;
    int k, per;
    (void)wgrad_slots_plan(B, Cin, Ct, h, w, k, per);
    return (int64_t)k * ((int64_t)Cin * 4 * Ct + 4 * Ct) * 4;
}

int onet_convT2x2_wgrad_slots(const void* xP, int64_t xP_bs, const void* x_amax, const void* dyP, int64_t dyP_bs, const void* dy_amax, float* dw,
                              float* dbias, void* ws, int64_t ws_bytes, int nparts, int B, int Cin, int Ct, int h, int w, void* stream) {
    ONET_REQUIRE(xP && dyP && dw && ws && B > 0 && Cin > 0 && Ct > 0 && h > 0 && w > 0 && (nparts == 1 || nparts == 2), "convT2x2_wgrad_slots: bad args");
    const int64_t hw = (int64_t)h * w;
    if ((Cin % 128) || (Ct % 32) || (hw % 128) || (w & (w - 1)) || w < 2 || !aligned16(xP) || !aligned16(dyP) || !aligned16(dw) || (xP_bs & 3) ||
        (dyP_bs & 3) || (int64_t)Cin * hw * 2 * nparts >= (1ll << 31) || (int64_t)Ct * 4 * hw * 2 * nparts >= (1ll << 31))
        return 1;
    ONET_REQUIRE(xP_bs >= (int64_t)Cin * hw * nparts / 2 && dyP_bs >= (int64_t)Ct * 4 * hw * nparts / 2, "convT2x2_wgrad_slots: batch stride too small");
    int lw = 0;
    while ((1 << lw) < w) ++lw;
    SWArgs g{xP, xP_bs, dyP, dyP_bs, (float*)ws, nullptr, (const unsigned*)x_amax, (const unsigned*)dy_amax, B, Cin, Ct, h, w, lw, 0, 0, 1, 0};
    const bool big = wgrad_slots_plan(B, Cin, Ct, h, w, g.splitK, g.chunksPerSplit) && SLOT_WGRAD_BIG;
    if (!big) {          // (SLOT_WGRAD_BIG = 0, A/B builds: the small tile's plan)
        int k, per;
        const int64_t chunks = (int64_t)B * hw / 32, tiles = (int64_t)(Cin / 128) * (Ct / 32);
        int64_t kk = std::max<int64_t>(1, (512 + tiles - 1) / tiles);
        kk = std::min<int64_t>(kk, std::max<int64_t>(1, chunks / 16));
        kk = std::min<int64_t>(kk, std::max<int64_t>(1, (256ll << 20) / ((int64_t)Cin * 4 * Ct * 4)));
        per = (int)((chunks + kk - 1) / kk);
        k = (int)((chunks + per - 1) / per);
        g.splitK = k;
        g.chunksPerSplit = per;
    }
    g.mTiles = big ? Cin / 256 : Cin / 128;
    g.nTiles = big ? Ct / 64 : Ct / 32;
    const int64_t n = (int64_t)Cin * 4 * Ct;
    ONET_REQUIRE(ws_bytes >= (int64_t)g.splitK * (n + 4 * Ct) * 4, "convT2x2_wgrad_slots: workspace too small (onet_convT2x2_wgrad_slots_ws_bytes)");
    if (dbias) g.db_part = (float*)ws + (int64_t)g.splitK * n;
    hipStream_t st = as_stream(stream);
    int rc;
    if (nparts == 2) rc = big ? launch_wgrad_slots<2, 1>(g, st) : launch_wgrad_slots<2, 0>(g, st);
    else rc = big ? launch_wgrad_slots<1, 1>(g, st) : launch_wgrad_slots<1, 0>(g, st);
    if (rc) return rc;
    hipLaunchKernelGGL(convt_wgrad_reduce_kernel, dim3((unsigned)cdiv(n / 4, 256)), dim3(256), 0, st, (const float*)ws, dw, g.splitK, n / 4, 0);
    rc = check_launch("convt_wgrad_reduce_kernel");
    if (rc || !dbias) return rc;
    hipLaunchKernelGGL(convt_dbias_reduce_kernel, dim3((unsigned)Ct), dim3(256), 0, st, (const float*)g.db_part, dbias, g.splitK * 4, Ct);
    return check_launch("convt_dbias_reduce_kernel");
}

int onet_convT2x2_fwd_slots(const void* xP, int64_t xP_bs, const void* x_amax, const void* wP, const float* bias, void* yP, int64_t yP_bs,
                            const void* y_amax, int nparts, int B, int Cin, int Ct, int h, int w, void* stream) {
    ONET_REQUIRE(xP && wP && yP && B > 0 && Cin > 0 && Ct > 0 && h > 0 && w > 0 && (nparts == 1 || nparts == 2), "convT2x2_fwd_slots: bad args");
    const int64_t hw = (int64_t)h * w;
    if ((Cin % 32) || Cin < 128 || (Ct % 32) || (hw % 128) || (w & 1) || !aligned16(xP) || !aligned16(wP) || !aligned16(yP) || (xP_bs & 3) || (yP_bs & 3) ||
        (int64_t)Cin * hw * 2 * nparts >= (1ll << 31) || (int64_t)Cin * 4 * Ct * 2 * nparts >= (1ll << 31))
        return 1;                                              // shape outside the fast path: nothing done
    ONET_REQUIRE(xP_bs >= (int64_t)Cin * hw * nparts / 2 && yP_bs >= (int64_t)Ct * 4 * hw * nparts / 2, "convT2x2_fwd_slots: batch stride too small");
    SArgs g{wP, xP, xP_bs, bias, yP, yP_bs, (const unsigned*)x_amax, (const unsigned*)y_amax, B, Cin, Ct, h, w, (4 * Ct) / 128, (int)(B * hw / 128)};
    const int64_t blocks = (int64_t)g.mTiles * g.nTiles;
    ONET_REQUIRE(blocks > 0 && blocks < (1ll << 31), "convT2x2_fwd_slots: grid too large");
    if (nparts == 2) hipLaunchKernelGGL(convt_slot_fwd_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), g);
    else hipLaunchKernelGGL(convt_slot_fwd_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), g);
    return check_launch("convt_slot_fwd_kernel");
}

}  // extern "C"
