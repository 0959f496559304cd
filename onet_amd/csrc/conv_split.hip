// 3x3 convolution forward + dgrad with fp32 tensors and fp32-LEVEL accuracy on the bf16 matrix cores, by OPERAND SPLITTING
// (F.conv2d / its input gradient at OV:47,51; the default fp32 algorithm of round 3 wherever the map is at least 32 pixels wide).
//
//   x = x_hi + x_mid + (x_lo),  w = w_hi + w_mid + (w_lo)   with every part a bf16 number (x_hi = bf16(x), x_mid = bf16(x - x_hi)):
//   hi + mid carry 16 significant bits, and   x * w  ~=  x_hi w_hi + x_hi w_mid + x_mid w_hi   drops only terms of relative size
//   2^-16 (mid * mid, hi * lo).  Each product of two bf16 numbers is exact in the fp32 accumulator of
//   v_mfma_f32_32x32x16_bf16, so the convolution costs THREE bf16 MFMAs per (tap, 16 channels) at 16x the fp32 MFMA rate:
//   5.3x the fp32 matrix peak, against 4x (x 0.5-0.6 pipe occupancy) for the fp32 Winograd F(4x4,3x3) kernel it replaces.
//   Measured against fp64 at Cin = 512: 9.7e-7 rms / 4.4e-6 max of the output scale from the split itself
//   (emulated on the CPU; F(4x4): 2.7e-6 / 4e-5), plus the fp32 accumulation of the MFMA chain as in every other kernel here.
//
// Structure: a direct implicit GEMM (M = output channels, N = 32 consecutive pixels of an image row,
// K = 16 input channels per MFMA; LDS tiles made of 16-byte slots of 8 channels, the two 8-channel halves apart; persistent
// blocks whose load -> LDS -> MFMA pipeline runs across tile boundaries), with:
//  * both parts of both operands in LDS: weights [part][tap][half][co], input halo tile [part][half][pixel] -- the fp32
//    input is loaded ONCE per chunk (8 channel planes per slot, coalesced along x) and split on the way into LDS (2 x 8
//    v_cvt_pk_bf16_f32 + 8 v_sub per slot); the weights are split once per optimizer step by the pack kernel;
//  * 8 waves per block (one block per CU, two waves per SIMD: the doubled tiles need 156 KB of LDS): 64 output channels x 16
//    image rows x 32 pixels, 2 rows per wave; per 16-channel chunk a wave issues 60 ds_read_b128 for 108 MFMAs of 32 cycles,
//    and the block moves 39 KB of input + 36 KB of weights per 864 MFMAs -- a third of the bytes and requests per
//    MFMA of the round-2 bf16 kernel this structure came from, which is what bound that kernel (DESIGN_HISTORY.md 4.2d).
// Requires Cin % 16 == 0 and W > 16; everything else takes the fp32 Winograd / direct kernels.
#include <algorithm>
#include <cstdlib>
#include "common.hpp"

using namespace onet;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
typedef float f32x4s __attribute__((ext_vector_type(4)));

namespace {

constexpr unsigned OOB_S = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t s_rsrc(const void* base, int64_t bytes) {
    const int n = bytes > 0x7fffffffll ? 0x7fffffff : (int)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, n, 0x00020000);
}
// (hi, mid) parts of two fp32 values, packed: hi = bf16(v) (round to nearest even), mid = bf16(v - hi)
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& mid) {
    const bf16x2 h = {(__bf16)a, (__bf16)b};
    const bf16x2 m = {(__bf16)(a - (float)h[0]), (__bf16)(b - (float)h[1])};
    hi = __builtin_bit_cast(unsigned, h);
    mid = __builtin_bit_cast(unsigned, m);
}

// The FORWARD convolution splits into fp16 parts instead (F16): hi = fp16(v), mid = fp16(v - hi) carry 22 significant bits where
// the bf16 pair carries 16, at the same three MFMAs per term (v_mfma_f32_32x32x16_f16, same rate) -- the forward error drops from
// 8e-7 rms of the output scale to the fp32 direct kernel's level, which is what the saturated B = 32 gradient check needed (the
// forward kernel alone put it at 3.3e-4; input and weight gradients contribute nothing measurable).  fp16's range is the price:
// activations (BatchNorm outputs, |x| << 65504) fit as they are, weights are packed times 2^8 so that their mid parts stay normal
// numbers (undone on the accumulators, exactly); gradients -- whose magnitude is anyone's guess -- keep the bf16 split.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {       // two bf16 (round to nearest even) in one dword, `lo` first
    const bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void split2h(float a, float b, unsigned& hi, unsigned& mid) {
    const f16x2 h = {(_Float16)a, (_Float16)b};
    const f16x2 m = {(_Float16)(a - (float)h[0]), (_Float16)(b - (float)h[1])};
    hi = __builtin_bit_cast(unsigned, h);
    mid = __builtin_bit_cast(unsigned, m);
}

__global__ void absmax_slots_kernel(const float* __restrict__ v, int64_t n, unsigned* __restrict__ slots) {
    float m = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(v[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0)
        atomicMax(slots + ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (AMAX_SLOTS - 1)) * AMAX_STRIDE, __builtin_bit_cast(unsigned, m));
}

// w [Cout][Cin][3][3] fp32 -> wq [K/16][part 2][9][2][N][8] bf16: chunk of 16 reduction channels, part (hi, mid), tap,
// 8-channel half, output channel, channel within the half.  which = 0: forward (K = Cin, N = Cout); 1: input gradient
// (K = Cout rounded up to 16 with a zero tail, N = Cin, taps rotated by 180 degrees).
// f16: fp16 parts of s w with s = 2^k chosen from the tensor's largest magnitude (wamax slots; amax lands in [2^13, 2^14): the mid
// parts stay normal numbers down to 2^-27 of the largest weight, and no weight overflows whatever a loaded checkpoint holds);
// the two floats (s, 1 / s) are stored BEHIND the pack (element offset 2 K 9 N) for the convolution to undo s on its accumulators.
// f16 == 2: ONE part, bf16(w) (BASELINE configs[2]: plain bf16 operands), same slot order -- [K/16][9][2][N][8], i.e. two consecutive
// 16-channel chunks look exactly like the (hi | mid) pair of the split packs, which is how the 32-channel chunks of the plain-bf16
// kernels read them.
__global__ void pack3x3_split_kernel(const float* __restrict__ w, __bf16* __restrict__ wq, int Cout, int Cin, int which, int f16,
                                     const unsigned* __restrict__ wamax) {
    const int K = which == 0 ? Cin : ((Cout + 15) / 16) * 16, N = which == 0 ? Cout : Cin;
    const int64_t n = (int64_t)K * 9 * N;                            // elements of ONE part
    const int nparts = f16 == 2 ? 1 : 2;
    float winv = 1.f;
    const float wscale = f16 == 1 ? amax_scale(amax_read(wamax), true, winv) : 1.f;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float* meta = reinterpret_cast<float*>(wq + nparts * n);
        meta[0] = wscale;
        meta[1] = winv;
    }
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int kc = (int)(i & 7);
        int64_t r = i >> 3;
        const int nn = (int)(r % N);
        r /= N;
        const int half = (int)(r & 1);
        r >>= 1;
        const int t = (int)(r % 9), kg = (int)(r / 9);
        const int k = kg * 16 + half * 8 + kc;
        float v = 0.f;
        if (which == 0) v = w[((int64_t)nn * Cin + k) * 9 + t];
        else if (k < Cout) v = w[((int64_t)k * Cin + nn) * 9 + (8 - t)];
        const int64_t per_chunk_part = (int64_t)9 * 2 * N * 8;
        const int64_t within = i - (int64_t)kg * per_chunk_part;     // [tap][half][n][8] inside the chunk
        if (f16 == 2) {
            wq[(int64_t)kg * per_chunk_part + within] = (__bf16)v;
        } else if (f16) {                                            // fp16 parts of 2^k w
            _Float16* wh = reinterpret_cast<_Float16*>(wq);
            const float vs = v * wscale;
            const _Float16 hi = (_Float16)vs;
            wh[((int64_t)kg * 2 + 0) * per_chunk_part + within] = hi;
            wh[((int64_t)kg * 2 + 1) * per_chunk_part + within] = (_Float16)(vs - (float)hi);
        } else {
            const __bf16 hi = (__bf16)v;
            wq[((int64_t)kg * 2 + 0) * per_chunk_part + within] = hi;
            wq[((int64_t)kg * 2 + 1) * per_chunk_part + within] = (__bf16)(v - (float)hi);
        }
    }
}

struct SpArgs {
    const float* x;       // fp32 NCHW
    int64_t x_bs;
    const __bf16* wq;     // [Cin/16][2][9][2][Cout][8]
    float* z;
    int64_t z_bs;
    int B, Cin, Cout, H, W, tilesX, tilesY, coTiles;
    float* stats;         // ST: BatchNorm partials [Cout][B * tilesY * tilesX][3] = (n, mean, M2) per tile
    const float* nsave;   // NORM: x is the PRE-activation of the Conv-BatchNorm-ReLU unit below; its coefficients [groups][4][Cin]
    int nimg;             //       (rows mean, invstd, scale, shift; bn.hip) for statistics groups of nimg consecutive images
    const unsigned* x_amax;   // F16: magnitude slots of x (NULL: no scaling)
    int x_always;             //      1: scale x so that its amax lands in [2^13, 2^14) (gradients); 0: only as an overflow guard
};

// sum over the 32 lanes of a wave half, valid in lanes 31 / 63 (DPP row rotations + row broadcast)
#define ONET_SP_DPP_ADD(v, ctrl, rmask) \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false))
__device__ __forceinline__ float sp_half_sum(float v) {
    ONET_SP_DPP_ADD(v, 0x128, 0xf);   // row_ror:8
    ONET_SP_DPP_ADD(v, 0x124, 0xf);   // row_ror:4
    ONET_SP_DPP_ADD(v, 0x122, 0xf);   // row_ror:2
    ONET_SP_DPP_ADD(v, 0x121, 0xf);   // row_ror:1
    ONET_SP_DPP_ADD(v, 0x142, 0xa);   // row_bcast:15 into rows 1 and 3
    return v;
}
#undef ONET_SP_DPP_ADD

struct SpCfg {
    static constexpr int NW = 8, NT = 2, TW = 32, CO_T = 64;
    static constexpr int ROWS = NW * NT;                                // 16 image rows per tile
    static constexpr int IN_ROWS = ROWS + 2, IN_COLS = TW + 2;
    static constexpr int NPIX = IN_ROWS * IN_COLS;                      // 612 halo pixels
    static constexpr int NROUND = (NPIX + 31) / 32;                     // 20 wave-rounds of 32 pixels x 2 halves
    static constexpr int NPIXP = NROUND * 32;                           // 640 pixel slots per half
    static constexpr int W_PART = 9 * 2 * CO_T;                         // 1152 slots per part
    static constexpr int W_SLOTS = 2 * W_PART;                          // hi | mid
    static constexpr int NWI = (W_SLOTS + 511) / 512;                   // 5
    static constexpr int IN_PART = 2 * NPIXP;                           // [half][pixel]
    static constexpr int BUF_SLOTS = W_SLOTS + 2 * IN_PART;             // 4864 slots = 76 KB
    static constexpr int LDS_BYTES = 2 * BUF_SLOTS * 16;                // two chunk buffers: 152 KB
    static constexpr int NB = (NT - 1) + 3;                             // distinct B row-fragments per horizontal tap
};

// PERSISTENT blocks: a block walks over output tiles and the chunk pipeline -- global loads two 16-channel
// chunks ahead of the MFMAs, LDS commit one ahead -- runs ACROSS tile boundaries.
// ST: the forward of a Conv-BatchNorm pair (OV:47-48, 51-52) also emits the BatchNorm statistics of its output, one (n, mean,
// M2) record per tile and channel from the final accumulators (pivot-shifted sums per wave, the eight waves merged through LDS).
// NORM (normalise on load): the input is the pre-activation z of the unit below and the staging threads apply that unit's
// BatchNorm + ReLU -- max(fma(z - mean, scale, shift), 0), the expression of bn_relu_apply_kernel, so the operand is bit for
// bit the activation that pass would have written -- before splitting; zero padding stays zero.  The second convolution of a
// DoubleConv (OV:51) then needs no materialised activation of the first (OV:49): one write and one read of the tensor less.
template <bool ST, bool NORM, bool F16>
__global__ __launch_bounds__(512, 2) void conv3x3_split_kernel(SpArgs a) {
    using C = SpCfg;
    constexpr int NT = C::NT, IN_COLS = C::IN_COLS, NWI = C::NWI, CO_T = C::CO_T, NPIXP = C::NPIXP, NB = C::NB;
    constexpr int BUF = C::BUF_SLOTS, W_PART = C::W_PART, IN_PART = C::IN_PART, ROWS = C::ROWS, TW = C::TW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_s[];
    u32x4s* lds = reinterpret_cast<u32x4s*>(smem_s);                   // [2 buffers][weights hi|mid | input hi|mid]

    // tiles of this block: every XCD (blockIdx % 8) owns a contiguous range of the tile list; its blocks stride through the range.
    // Tile list order: OUTPUT-CHANNEL tile fastest, then x, y, image -- the Cout / 64 blocks that need the same input tile run on
    // one XCD at the same time and share it in that L2 (with the channel tile slowest every one of the Cout / 64 passes over the
    // batch streamed the whole input from HBM again: 1.8 x the compulsory bytes per launch on average); the weight slices of
    // all channel tiles (<= 9 MB) then live in L2 / MALL instead of one slice at a time
    const int ntiles = a.tilesX * a.tilesY * a.B * a.coTiles;
    int t_first, t_end, t_stride;
    {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int q = ntiles >> 3, r = ntiles & 7;
        const int start = xcd * q + min(xcd, r);
        t_stride = (gridDim.x + 7 - xcd) >> 3;
        t_first = start + j;
        t_end = start + q + (xcd < r ? 1 : 0);
    }
    if (t_first >= t_end) return;

    const int tid = threadIdx.x, lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int HW = a.H * a.W;
    const int nchunks = a.Cin >> 4;
    // F16: the operand scales (powers of two): x by its magnitude slots, the weights by the pack's own (stored behind the pack)
    float xs_inv = 1.f;
    const float xs_scale = F16 ? amax_scale(amax_read(a.x_amax), a.x_always != 0, xs_inv) : 1.f;
    const float acc_scale = F16 ? xs_inv * reinterpret_cast<const float*>(a.wq + (int64_t)a.Cin * 2 * 9 * a.Cout)[1] : 1.f;

    const __amdgpu_buffer_rsrc_t wr = s_rsrc(a.wq, (int64_t)a.Cin * 2 * 9 * a.Cout * 2);
    const unsigned in_step = (unsigned)(16 * HW * 4), w_step = (unsigned)(2 * 9 * 2 * a.Cout * 16);
    const unsigned plane = (unsigned)(HW * 4);

    // ---- staging side: runs two chunks ahead of the compute side, across tiles.  Input roles (wave-uniform):
    //   waves 0..4, item it = tid < 288 = (half h, halo row r, 4-pixel group g):  8 x buffer_load_dwordx4 -- the 4 interior pixels
    //              x0 + 4g .. + 3 of row r in the 8 channel planes of half h -- give the 4 slots (8 channels x 1 pixel) of those pixels;
    //   waves 5..6, item tid - 320 < 72 = (half, row, side): the halo column x0 - 1 / x0 + 32, 8 x buffer_load_dword, one slot.
    // (Round 3's first version loaded every slot as 8 dwords, one pixel per lane: 24 VMEM instructions per thread and chunk, 232 per
    // block; a timing build without them ran 19-22 % faster.  Now 8 per staging thread, ~100 per block, the same bytes.)
    // Requires W % 4 == 0 and 16-byte aligned image rows (host-checked): a float4 is entirely inside or outside the image.
    // Weight slots i = tid + 512 k = (part, tap, half, co) -> 16 bytes of the slice [chunk][part][tap][half][co][8].
    unsigned in_off, w_off[NWI];
    const int role = wn < 5 ? 0 : (wn < 7 ? 1 : 2);                    // 0 interior, 1 halo column, 2 none
    const int it = role == 0 ? tid : tid - 320;
    const int it_h = role == 0 ? it / 144 : it / 36;
    const int it_r = role == 0 ? (it % 144) / 8 : (it % 36) / 2;
    const int it_g = role == 0 ? (it & 7) : (it & 1);                  // 4-pixel group / side
    const bool it_ok = role == 0 ? it < 288 : (role == 1 && it < 72);
    __amdgpu_buffer_rsrc_t xr;
    int st_tile = t_first, st_chunk = 0, st_b = 0;
    unsigned cin_bytes = 0, cw_bytes = 0;
    // NORM: the BatchNorm coefficients of a chunk's 16 channels travel beside its data -- 48 floats (mean, scale, shift) fetched
    // by the one wave without a staging role and parked in LDS [chunk buffer][half][channel][4], where the staging threads pick
    // them up at commit time (per-lane vector loads would double the staging VMEM count, 48 scalar loads spilled the SGPR file)
    float* const coef = reinterpret_cast<float*>(lds + 2 * BUF) + (ST ? C::NW * 64 * 2 : 0);
    float pend_cv = 0.f;
    bool pend_keep = false;                          // validity of the item whose loads are in flight
    auto setup_stage = [&]() __attribute__((always_inline)) {
        const bool live = st_tile < t_end;
        int v = live ? st_tile : t_first;
        const int co0 = (v % a.coTiles) * CO_T;      // output-channel tile fastest: see the tile order note at the top of the kernel
        v /= a.coTiles;
        const int tx = v % a.tilesX;
        v /= a.tilesX;
        const int ty = v % a.tilesY;
        const int b = v / a.tilesY;
        const int y0 = ty * ROWS, x0 = tx * TW;
        st_b = b;
        xr = s_rsrc(a.x + (int64_t)b * a.x_bs, (int64_t)a.Cin * HW * 4);
        {
            const int yy = y0 - 1 + it_r;
            const int xx = role == 0 ? x0 + 4 * it_g : (it_g ? x0 + TW : x0 - 1);
            const bool ok = live && it_ok && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
            in_off = ok ? (unsigned)(((it_h * 8) * HW + yy * a.W + xx) * 4) : OOB_S;
        }
#pragma unroll
        for (int k = 0; k < NWI; ++k) {
            const int i = tid + 512 * k;
            const int co = i & (CO_T - 1), pth = i >> 6;               // pth = (part * 9 + tap) * 2 + half
            const bool ok = live && (i < C::W_SLOTS) && (co0 + co < a.Cout);
            w_off[k] = ok ? (unsigned)((pth * a.Cout + co0 + co) * 16) : OOB_S;
        }
    };
    auto advance = [&]() __attribute__((always_inline)) {
        ++st_chunk;
        cin_bytes += in_step;
        cw_bytes += w_step;
        if (st_chunk == nchunks) {
            st_chunk = 0;
            cin_bytes = cw_bytes = 0;
            st_tile += t_stride;
            setup_stage();                           // past the last tile: every slot out of range -> zeros, no traffic
        }
    };

    u32x4s xin[8];                                   // plane c of the item: 4 pixels (interior) or 1 pixel in lane 0 (halo column)
    u32x4s wv[NWI];
    // loads of planes [c0, c1) of this thread's item; the chunk / plane part of the address is wave-uniform and rides in the
    // instruction's scalar offset, the per-lane part alone decides the range check (OOB_S -> 0)
    auto issue_in = [&](int c0, int c1) __attribute__((always_inline)) {
        if (NORM && c0 == 0) {
            pend_keep = in_off != OOB_S;
            if (wn == 7 && lane < 48) {              // lane = kind * 16 + channel; rows of `save`: 0 mean, 2 scale, 3 shift
                const int k = lane >> 4, row = k == 0 ? 0 : k + 1;
                pend_cv = st_tile < t_end ? a.nsave[(int64_t)((st_b / a.nimg) * 4 + row) * a.Cin + st_chunk * 16 + (lane & 15)] : 0.f;
            }
        }
        if (role == 0) {
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c >= c0 && c < c1) xin[c] = __builtin_amdgcn_raw_buffer_load_b128(xr, in_off, (int)(cin_bytes + c * plane), 0);
        } else if (role == 1) {
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c >= c0 && c < c1) xin[c][0] = __builtin_amdgcn_raw_buffer_load_b32(xr, in_off, (int)(cin_bytes + c * plane), 0);
        }
    };
    auto issue_w = [&](int k0, int k1) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NWI; ++k)
            if (k >= k0 && k < k1) wv[k] = __builtin_amdgcn_raw_buffer_load_b128(wr, w_off[k], (int)cw_bytes, 0);
    };
    u32x4s* const w_st = lds + tid;                                     // + 512 k          (+ buffer * BUF)
    // slot of halo pixel (row r, column c) = W_SLOTS + part * IN_PART + half * NPIXP + r * IN_COLS + c
    u32x4s* const in_st = lds + C::W_SLOTS + it_h * NPIXP + it_r * IN_COLS + (role == 0 ? 1 + 4 * it_g : (it_g ? IN_COLS - 1 : 0));
    // pixels [p0, p1) of the item (interior: 4 pixels, halo column: pixel 0 only): 8 channels -> one slot per part
    auto commit_in = [&](int buf, int p0, int p1) __attribute__((always_inline)) {
        if (role == 2 || !it_ok) return;
        f32x4s f[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) f[c] = __builtin_bit_cast(f32x4s, xin[c]);      // (re-typed as a whole: see commit_dz below)
        if constexpr (NORM) {
            // channel pairs outermost (a dword of a slot = two neighbouring channels): six coefficients live at a time
            const f32x4s* cfp = reinterpret_cast<const f32x4s*>(coef) + (buf * 2 + it_h) * 8;
            u32x4s hi[4], mid[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4s ca = cfp[2 * c], cb = cfp[2 * c + 1];
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    if (px < p0 || px >= p1 || (role == 1 && px > 0)) continue;
                    float va = fmaxf(fmaf(f[2 * c][px] - ca[0], ca[1], ca[2]), 0.f);
                    float vb = fmaxf(fmaf(f[2 * c + 1][px] - cb[0], cb[1], cb[2]), 0.f);
                    va = pend_keep ? va : 0.f;               // outside the image: the convolution's zero padding
                    vb = pend_keep ? vb : 0.f;
                    unsigned h, m;
                    if constexpr (F16) split2h_s(va, vb, xs_scale, h, m);
                    else split2(va, vb, h, m);
                    hi[px][c] = h;
                    mid[px][c] = m;
                }
            }
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                if (px < p0 || px >= p1 || (role == 1 && px > 0)) continue;
                in_st[buf * BUF + px] = hi[px];
                in_st[buf * BUF + px + IN_PART] = mid[px];
            }
            return;
        }
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            if (px < p0 || px >= p1 || (role == 1 && px > 0)) continue;
            u32x4s hi, mid;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                unsigned h, m;
                if constexpr (F16) split2h_s(f[2 * c][px], f[2 * c + 1][px], xs_scale, h, m);
                else split2(f[2 * c][px], f[2 * c + 1][px], h, m);
                hi[c] = h;
                mid[c] = m;
            }
            in_st[buf * BUF + px] = hi;
            in_st[buf * BUF + px + IN_PART] = mid;
        }
    };
    auto commit_w = [&](int buf, int k0, int k1) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NWI; ++k)
            if (k >= k0 && k < k1 && tid + 512 * k < C::W_SLOTS) w_st[buf * BUF + 512 * k] = wv[k];
    };

    auto commit_coef = [&](int cb) __attribute__((always_inline)) {
        if (NORM && wn == 7 && lane < 48) coef[((cb * 2 + ((lane >> 3) & 1)) * 8 + (lane & 7)) * 4 + (lane >> 4)] = pend_cv;
    };

    // fragments: A (weights) slot = part * W_PART + (tap * 2 + kh) * 64 + m * 32 + l31;
    //            B (input)   slot = W_SLOTS + part * IN_PART + kh * NPIXP + (2 wn + j) * IN_COLS + l31 + kx
    const u32x4s* const a_ptr = lds + kh * CO_T + l31;
    const u32x4s* const b_ptr = lds + C::W_SLOTS + kh * NPIXP + (wn * NT) * IN_COLS + l31;

    setup_stage();
    issue_in(0, 8);
    issue_w(0, NWI);
    if constexpr (NORM) {
        commit_coef(0);
        __syncthreads();
    }
    commit_in(0, 0, 4);
    commit_w(0, 0, NWI);
    advance();
    issue_in(0, 8);                                  // second chunk (or the first of the next tile)
    issue_w(0, NWI);
    commit_coef(1);
    __syncthreads();
    int buf = 0;
    for (int tile = t_first; tile < t_end; tile += t_stride) {
        f32x16 acc[2][NT];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

        for (int c = 0; c < nchunks; ++c) {
            const u32x4s* const ab = a_ptr + buf * BUF;
            const u32x4s* const bb = b_ptr + buf * BUF;
            // Fragment reads run ONE tap ahead of the MFMAs that consume them (two register sets), the taps in kx-major order
            // so that the NB row-fragments of a horizontal shift serve its three vertical taps; the row-fragments of the next
            // shift are fetched during the three taps of the current one.  [part]: 0 = hi, 1 = mid.
            u32x4s Aq[2][2][2], Bq[2][NB][2];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                Bq[0][j][0] = bb[j * IN_COLS];
                Bq[0][j][1] = bb[j * IN_COLS + IN_PART];
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                Aq[0][m][0] = ab[m * 32];
                Aq[0][m][1] = ab[m * 32 + W_PART];
            }
#pragma unroll
            for (int idx = 0; idx < 9; ++idx) {
                const int kx = idx / 3, ky = idx % 3;
                if (idx < 8) {
                    const int nt = ((idx + 1) % 3) * 3 + (idx + 1) / 3;
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        Aq[(idx + 1) & 1][m][0] = ab[nt * 2 * CO_T + m * 32];
                        Aq[(idx + 1) & 1][m][1] = ab[nt * 2 * CO_T + m * 32 + W_PART];
                    }
                }
                if (kx < 2) {
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        if (j % 3 == ky) {
                            Bq[(kx + 1) & 1][j][0] = bb[j * IN_COLS + kx + 1];
                            Bq[(kx + 1) & 1][j][1] = bb[j * IN_COLS + kx + 1 + IN_PART];
                        }
                }
                // Staging, spread over the taps so that no wave sits in a burst of VMEM issues while its SIMD partner does the same:
                // the other buffer was last read one chunk ago (barrier since), so the chunk after this one goes in piece by piece
                // -- and once an item's pixels are committed its registers take the loads of the chunk after that (every load
                // keeps most of a chunk of MFMAs between issue and use).  Inputs: pixels 0, 1 at tap 0, pixels 2, 3 at tap 1, the
                // eight plane loads two per tap at taps 2..5; weights: commit + reload of slot k at taps 4..8.
                if (idx == 0) advance();
                if (idx == 0) commit_in(buf ^ 1, 0, 2);
                if (idx == 1) commit_in(buf ^ 1, 2, 4);
                if (idx >= 2 && idx <= 5) issue_in(2 * (idx - 2), 2 * (idx - 2) + 2);
                if (idx >= 4) {
                    commit_w(buf ^ 1, idx - 4, idx - 3);
                    issue_w(idx - 4, idx - 3);
                }
                if (idx == 8) commit_coef(buf);      // for the chunk whose loads went out at taps 2..5: committed into `buf` next
                __builtin_amdgcn_sched_barrier(0);
                const int tap = ky * 3 + kx;
                (void)tap;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        // smallest terms first: the two cross terms, then hi * hi
                        if constexpr (F16) {
                            const f16x8 ah = __builtin_bit_cast(f16x8, Aq[idx & 1][m][0]), am = __builtin_bit_cast(f16x8, Aq[idx & 1][m][1]);
                            const f16x8 bh = __builtin_bit_cast(f16x8, Bq[kx & 1][n + ky][0]), bm = __builtin_bit_cast(f16x8, Bq[kx & 1][n + ky][1]);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(am, bh, acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bm, acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[m][n], 0, 0, 0);
                        } else {
                            const bf16x8 ah = __builtin_bit_cast(bf16x8, Aq[idx & 1][m][0]), am = __builtin_bit_cast(bf16x8, Aq[idx & 1][m][1]);
                            const bf16x8 bh = __builtin_bit_cast(bf16x8, Bq[kx & 1][n + ky][0]), bm = __builtin_bit_cast(bf16x8, Bq[kx & 1][n + ky][1]);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[m][n], 0, 0, 0);
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            buf ^= 1;
        }

        if constexpr (F16) {                         // undo the operands' power-of-two scales (exact)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][n][r] *= acc_scale;
        }
        int v = tile;
        const int co0 = (v % a.coTiles) * CO_T;      // output-channel tile fastest: see the tile order note at the top of the kernel
        v /= a.coTiles;
        const int tx = v % a.tilesX;
        v /= a.tilesX;
        const int ty = v % a.tilesY;
        const int b = v / a.tilesY;
        const int y0 = ty * ROWS, x0 = tx * TW;
        if constexpr (ST) {
            float* sc = reinterpret_cast<float*>(lds + 2 * BUF);       // [8 waves][64 channels][mean, M2]
            constexpr float npw = (float)(NT * 32), inv_npw = 1.f / npw;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // lanes 0-31 hold 32 pixels of channel c, lanes 32-63 of channel c + 4; pivot = the wave's first pixel
                    const int piv = __builtin_bit_cast(int, acc[m][0][r]);
                    const float p0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(piv, 0));
                    const float p1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(piv, 32));
                    const float pv = kh ? p1 : p0;
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const float d = acc[m][n][r] - pv;
                        s1 += d;
                        s2 = fmaf(d, d, s2);
                    }
                    s1 = sp_half_sum(s1);
                    s2 = sp_half_sum(s2);
                    if (l31 == 31) {
                        const int cl = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                        sc[(wn * 64 + cl) * 2] = fmaf(s1, inv_npw, pv);
                        sc[(wn * 64 + cl) * 2 + 1] = fmaxf(fmaf(-s1 * inv_npw, s1, s2), 0.f);
                    }
                }
            __syncthreads();
            if (tid < 64 && co0 + tid < a.Cout) {
                float mw[8], qw[8];
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    mw[w] = sc[(w * 64 + tid) * 2];
                    qw[w] = sc[(w * 64 + tid) * 2 + 1];
                }
                const float mean = 0.125f * (((mw[0] + mw[1]) + (mw[2] + mw[3])) + ((mw[4] + mw[5]) + (mw[6] + mw[7])));
                float m2 = ((qw[0] + qw[1]) + (qw[2] + qw[3])) + ((qw[4] + qw[5]) + (qw[6] + qw[7]));
#pragma unroll
                for (int w = 0; w < 8; ++w) m2 = fmaf(npw * (mw[w] - mean), mw[w] - mean, m2);
                const int64_t nblk = (int64_t)a.B * a.tilesY * a.tilesX;
                const int64_t blk = ((int64_t)b * a.tilesY + ty) * a.tilesX + tx;
                float* sp = a.stats + ((int64_t)(co0 + tid) * nblk + blk) * 3;
                sp[0] = 8.f * npw;
                sp[1] = mean;
                sp[2] = m2;
            }
            __syncthreads();                         // the scratch is rewritten by the next tile
        }
        float* zb = a.z + (int64_t)b * a.z_bs;
        const int xo = x0 + l31;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int yo = y0 + wn * NT + n;
                if (yo < a.H && xo < a.W) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                        if (co < a.Cout) zb[(int64_t)co * HW + (int64_t)yo * a.W + xo] = acc[m][n][r];
                    }
                }
            }
    }
}


// ------------------------------------------------------------------ pre-split operands (round 4)
// The activation arrives ALREADY split, in the slot layout the tile wants:
//     xs [B][C/8][H][part 2][W][8] 16-bit   (part 0 = hi, 1 = mid; fp16 or bf16; 4 bytes per element -- the fp32 tensor's footprint)
// written by the producers (bn.hip: BatchNorm + ReLU apply / pooling, BatchNorm backward apply; convt_gemm.hip epilogue).  The
// MFMA kernel's staging is then a COPY: every 16-byte slot of the halo tile and of the weight slice goes HBM -> LDS by LDS-DMA
// (buffer_load_dwordx4 ... lds: no staging registers, no conversion VALU, no ds_write -- the round-3 kernel spent 19-26 % of its
// time there, with 4-way bank conflicts on its ds_write_b128), one chunk ahead of the MFMAs into the other chunk buffer.
typedef int i32x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4s sp_rsrc4(const void* base, int64_t bytes) {
    const uint64_t p = reinterpret_cast<uint64_t>(base);
    i32x4s r;
    r.x = (int)(p & 0xffffffffu);
    r.y = (int)((p >> 32) & 0xffffu);
    r.z = bytes > 0x7fffffffll ? 0x7fffffff : (int)bytes;
    r.w = 0x00020000;
    return r;
}
// LDS-DMA of 64 x 16 bytes: lane l's 16 bytes at (rsrc base + voff + soff) land at LDS byte address lds_base + 16 l; a lane
// whose address is out of range (voff = OOB_S) writes zeros.  Inline asm: the compiler must not see a load (it would order every
// later ds_read behind a vmcnt(0)); m0 is restored because the compiler does not know it was touched.
__device__ __forceinline__ void sp_dma16(i32x4s rsrc, unsigned lds_base, unsigned voff, unsigned soff) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(lds_base), "s"(soff)
                 : "memory");
}

// fp32 NCHW -> the slot layout above (tests, tools, and producers that have no fused variant yet)
// f16 == 2: plain bf16, one part: xs [B][C/8][H][W][8]
// slots != NULL: the tensor's magnitude slots -- the guard scale of amax_scale(.., always = false) multiplies `scale` (what a fused
// producer with the same slots would have applied; the consumer undoes it from the same slots)
__global__ void split_pack_act_kernel(const float* __restrict__ x, int64_t x_bs, unsigned* __restrict__ xs, int64_t xs_bs, int B, int C8,
                                      int H, int W, int f16, float scale, const unsigned* __restrict__ slots) {
    if (slots) {
        float inv;
        scale *= amax_scale(amax_read(slots), false, inv);
    }
    const int64_t n = (int64_t)B * C8 * H * W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % W);
        int64_t r = i / W;
        const int y = (int)(r % H);
        r /= H;
        const int c8 = (int)(r % C8), b = (int)(r / C8);
        const float* src = x + (int64_t)b * x_bs + ((int64_t)c8 * 8 * H + y) * W + xx;
        u32x4s hi, mid;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float va = src[(int64_t)(2 * c) * H * W] * scale, vb = src[(int64_t)(2 * c + 1) * H * W] * scale;
            unsigned h, m;
            if (f16 == 1) split2h(va, vb, h, m);
            else split2(va, vb, h, m);
            hi[c] = h;
            mid[c] = m;
        }
        if (f16 == 2) {
            reinterpret_cast<u32x4s*>(xs + (int64_t)b * xs_bs)[((int64_t)c8 * H + y) * W + xx] = hi;
            continue;
        }
        u32x4s* dst = reinterpret_cast<u32x4s*>(xs + (int64_t)b * xs_bs) + (((int64_t)c8 * H + y) * 2) * W + xx;
        dst[0] = hi;
        dst[W] = mid;
    }
}

struct SpPreArgs {
    const void* xs;       // pre-split activation, slot layout
    int64_t xs_bs;        // batch stride in 4-byte units (= C * H * W for a dense tensor)
    const __bf16* wq;     // [Cin/16][2][9][2][Cout][8]
    float* z;
    int64_t z_bs;
    int B, Cin, Cout, H, W, tilesX, tilesY, coTiles;
    float* stats;         // ST: BatchNorm partials (as conv3x3_split_kernel)
    const unsigned* x_slots;   // magnitude slots the producer scaled xs by (NULL: unscaled); x_always: the rule it used (amax_scale)
    int x_always;
    const unsigned* x_slots2;  // a concat buffer has two producers: channels >= split_ch were scaled by these slots (NULL: unscaled)
    int split_ch;              // 0: one group
    // RD (conv3x3_pre16_kernel, input-gradient launches): the output z IS the gradient da of the Conv-BatchNorm-ReLU unit below; its
    // BatchNorm-backward reduce pass rides in the epilogue: rd_z = that unit's pre-activation [B][Cout][H][W] fp32, rd_save its
    // coefficients [G][4][Cout] (groups of rd_gimg images; 0: one group), rd_rec [tiles][Cout][4] receives (sum dy, 0, sum dy xhat, 0)
    // per tile (bn_relu_bwd_reduce_kernel's record format), rd_amax (may be NULL) the magnitude slots of da.
    const float* rd_z = nullptr;
    int64_t rd_z_bs = 0;
    const float* rd_save = nullptr;
    int rd_gimg = 0;
    float* rd_rec = nullptr;
    unsigned* rd_amax = nullptr;
    // conv3x3_pre16_kernel, plain bf16 operands (BASELINE configs[2]): z16 -- the output z is STORED as bf16 (rounded once, to nearest
    // even; the statistics epilogue still reads the fp32 accumulators), z_bs in elements; rd_z16 -- rd_z is such a bf16 tensor
    int z16 = 0, rd_z16 = 0;
    // conv3x3_split_pre_kernel, fp16 (hi | mid) parts (round 5): output channels >= zP_ch0 (a multiple of 64) are NOT written to z but,
    // pre-split, to zP [B][(Cout - zP_ch0) / 8][H][2][W][8] (batch stride zP_bs in 4-byte units) as parts of 2^k z, k by the `always`
    // rule from the bound in zP_slots: the input gradient of a decoder block's first convolution hands the up-sampled half of the
    // concat gradient to the ConvTranspose2d backward GEMMs in their operand form
    void* zP = nullptr;
    int64_t zP_bs = 0;
    int zP_ch0 = 0;
    const unsigned* zP_slots = nullptr;
};

#ifndef SP_PRE_LAST_TAP
#define SP_PRE_LAST_TAP 4   // the next chunk's DMA pieces go out during taps 0 .. SP_PRE_LAST_TAP
#endif
// PM: 0 = bf16 (hi | mid) parts, 1 = fp16 (hi | mid) parts -- three MFMAs per term on 16-channel chunks; 2 = PLAIN bf16 operands
// (BASELINE configs[2]'s bf16 MFMA conv path: xs [B][C/8][H][W][8] bf16, one part): the same LDS image and DMA schedule with the
// "part" index standing for the second 16 channels of a 32-channel chunk, two MFMAs per term (one per 16 channels).
// W16: maps 16 pixels wide (the U-Net's 16-pixel level): a tile's 32 pixel columns are TWO IMAGES side by side, each with its own halo
// columns in the LDS image (18 + 18 columns), so a horizontal tap never reads the neighbour image; a.B counts image PAIRS.
#ifndef SP_PRE_NW
#define SP_PRE_NW 8      // waves per block (8: two rows per wave; 4 (timing experiment): four rows per wave, one wave per SIMD)
#endif
template <bool W16> struct SpPreCfg {
    static constexpr int NW = SP_PRE_NW, NT = 16 / NW, TW = 32, CO_T = 64, ROWS = NW * NT;
    static constexpr int IN_ROWS = ROWS + 2, IN_COLS = W16 ? 36 : 34;
    static constexpr int NPIX = IN_ROWS * IN_COLS;                      // 612 / 648 halo pixels
    static constexpr int NPIXP = W16 ? 656 : 640;                       // pixel slots per half (4 NPIXP = whole 64-slot DMA pieces)
    static constexpr int W_PART = 9 * 2 * CO_T, W_SLOTS = 2 * W_PART, NWI = (W_SLOTS + NW * 64 - 1) / (NW * 64);
    static constexpr int IN_PART = 2 * NPIXP;
    static constexpr int BUF_SLOTS = W_SLOTS + 2 * IN_PART;             // 76 / 77 KB
    static constexpr int LDS_BYTES = 2 * BUF_SLOTS * 16;
    static constexpr int NB = (NT - 1) + 3;
};
template <bool ST, int PM, bool W16>
__global__ __launch_bounds__(SP_PRE_NW * 64, SP_PRE_NW / 4) void conv3x3_split_pre_kernel(SpPreArgs a) {
    constexpr bool F16 = PM == 1;
    using C = SpPreCfg<W16>;
    constexpr int NT = C::NT, IN_COLS = C::IN_COLS, NWI = C::NWI, CO_T = C::CO_T, NPIXP = C::NPIXP, NB = C::NB;
    constexpr int BUF = C::BUF_SLOTS, W_PART = C::W_PART, IN_PART = C::IN_PART, ROWS = C::ROWS, TW = C::TW;
    constexpr int NWV = C::NW, NTH = NWV * 64;
    constexpr int NII = (2 * IN_PART + NTH - 1) / NTH;                     // 5 (6) input DMA rounds per wave and chunk (2560 / 2624 slot positions)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_s[];
    u32x4s* lds = reinterpret_cast<u32x4s*>(smem_s);                   // [2 buffers][weights hi|mid | input hi|mid]

    const int ntiles = a.tilesX * a.tilesY * a.B * a.coTiles;          // tile list order: see conv3x3_split_kernel
    int t_first, t_end, t_stride;
    {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int q = ntiles >> 3, r = ntiles & 7;
        const int start = xcd * q + min(xcd, r);
        t_stride = (gridDim.x + 7 - xcd) >> 3;
        t_first = start + j;
        t_end = start + q + (xcd < r ? 1 : 0);
    }
    if (t_first >= t_end) return;

    const int tid = threadIdx.x, lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kh = lane >> 5;
    const int HW = a.H * a.W;
    const int nchunks = PM == 2 ? a.Cin >> 5 : a.Cin >> 4;

    const i32x4s wr = sp_rsrc4(a.wq, (int64_t)a.Cin * (PM == 2 ? 1 : 2) * 9 * a.Cout * 2);
    float xs_inv = 1.f, xs_inv2 = 1.f;
    (void)amax_scale(amax_read(a.x_slots), a.x_always != 0, xs_inv);
    (void)amax_scale(amax_read(a.x_slots2), a.x_always != 0, xs_inv2);
    // two groups: sum = inv1 A1 + inv2 A2 = inv2 ((inv1 / inv2) A1 + A2): the accumulators are rescaled once, at the first chunk of
    // group 2, by a power of two (1 in every ordinary network: nothing is done then)
    const int split_chunk = a.split_ch ? (PM == 2 ? a.split_ch >> 5 : a.split_ch >> 4) : 0;
    const float grp_ratio = split_chunk ? xs_inv / xs_inv2 : 1.f;
    const float acc_scale = (split_chunk ? xs_inv2 : xs_inv) * (F16 ? reinterpret_cast<const float*>(a.wq + (int64_t)a.Cin * 2 * 9 * a.Cout)[1] : 1.f);
    const unsigned in_step = (unsigned)(16 * HW * 4), w_step = (unsigned)(2 * 9 * 2 * a.Cout * 16);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_s;

    // ---- staging: slot position i = (wn + 8 k) * 64 + lane of the chunk image.  Input k < NII: i = (part * 2 + half) * NPIXP + pix,
    // pix = r * IN_COLS + c (pix >= 612: padding of the image, zeros).  Weights k < NWI: i = (part * 9 + tap) * 2 + half) * 64 + co.
    unsigned in_off[NII], w_off[NWI];
    i32x4s xr;
    int st_tile = t_first, st_chunk = 0;
    unsigned cin_bytes = 0, cw_bytes = 0;
    auto setup_stage = [&]() __attribute__((always_inline)) {
        const bool live = st_tile < t_end;
        int v = live ? st_tile : t_first;
        const int co0 = (v % a.coTiles) * CO_T;
        v /= a.coTiles;
        const int tx = v % a.tilesX;
        v /= a.tilesX;
        const int ty = v % a.tilesY;
        const int b = v / a.tilesY;
        const int y0 = ty * ROWS, x0 = tx * TW;
        // (W16: b is the image PAIR; the resource spans both images, the second one's offset rides in the lane's byte offset)
        xr = sp_rsrc4(reinterpret_cast<const unsigned*>(a.xs) + (int64_t)(W16 ? 2 * b : b) * a.xs_bs,
                      (W16 ? a.xs_bs * 4 : 0) + (int64_t)a.Cin * HW * (PM == 2 ? 2 : 4));
#pragma unroll
        for (int k = 0; k < NII; ++k) {
            const int i = (wn + NWV * k) * 64 + lane;
            const int ph = i / NPIXP, pix = i % NPIXP;                 // ph = part * 2 + half (PM 2: the chunk's channel group 0 .. 3)
            const int r = pix / IN_COLS, c = pix % IN_COLS;
            const int img = W16 ? c / 18 : 0;
            const int yy = y0 - 1 + r, xx = W16 ? c % 18 - 1 : x0 - 1 + c;
            const bool ok = live && i < 2 * IN_PART && pix < C::NPIX && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
            const unsigned io = W16 ? (unsigned)(img * a.xs_bs * 4) : 0u;
            in_off[k] = !ok ? OOB_S : PM == 2 ? io + (unsigned)(((ph * a.H + yy) * a.W + xx) * 16)
                                              : io + (unsigned)(((((ph & 1) * a.H + yy) * 2 + (ph >> 1)) * a.W + xx) * 16);
        }
#pragma unroll
        for (int k = 0; k < NWI; ++k) {
            const int i = tid + NTH * k;
            const int co = i & (CO_T - 1), pth = i >> 6;
            const bool ok = live && (i < C::W_SLOTS) && (co0 + co < a.Cout);
            w_off[k] = ok ? (unsigned)((pth * a.Cout + co0 + co) * 16) : OOB_S;
        }
    };
    auto advance = [&]() __attribute__((always_inline)) {
        ++st_chunk;
        cin_bytes += in_step;
        cw_bytes += w_step;
        if (st_chunk == nchunks) {
            st_chunk = 0;
            cin_bytes = cw_bytes = 0;
            st_tile += t_stride;
            setup_stage();
        }
    };
    // piece k of the chunk the staging state points at, into chunk buffer `buf`
    auto dma_in = [&](int buf, int k) __attribute__((always_inline)) {
        if ((wn + NWV * k) * 64 < 2 * IN_PART)
            sp_dma16(xr, lds0 + (unsigned)((buf * BUF + C::W_SLOTS + (wn + NWV * k) * 64) * 16), in_off[k], cin_bytes);
    };
    auto dma_w = [&](int buf, int k) __attribute__((always_inline)) {
        if (k < NWI - 1 || wn < (C::W_SLOTS - NTH * (NWI - 1)) / 64)
            sp_dma16(wr, lds0 + (unsigned)((buf * BUF + (wn + NWV * k) * 64) * 16), w_off[k], cw_bytes);
    };

    const u32x4s* const a_ptr = lds + kh * CO_T + l31;
    const u32x4s* const b_ptr = lds + C::W_SLOTS + kh * NPIXP + (wn * NT) * IN_COLS + (W16 ? (l31 >> 4) * 18 + (l31 & 15) : l31);

    setup_stage();
#pragma unroll
    for (int k = 0; k < NII; ++k) dma_in(0, k);
#pragma unroll
    for (int k = 0; k < NWI; ++k) dma_w(0, k);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    float zP_scale = 1.f;
    if constexpr (PM == 1 && !ST && !W16) {
        float zinv;
        if (a.zP) zP_scale = amax_scale(amax_read(a.zP_slots), true, zinv);
    }
    for (int tile = t_first; tile < t_end; tile += t_stride) {
        f32x16 acc[2][NT];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

        for (int c = 0; c < nchunks; ++c) {
            if (split_chunk && c == split_chunk && grp_ratio != 1.f) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[m][n][r] *= grp_ratio;
            }
            const u32x4s* const ab = a_ptr + buf * BUF;
            const u32x4s* const bb = b_ptr + buf * BUF;
            u32x4s Aq[2][2][2], Bq[2][NB][2];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                Bq[0][j][0] = bb[j * IN_COLS];
                Bq[0][j][1] = bb[j * IN_COLS + IN_PART];
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                Aq[0][m][0] = ab[m * 32];
                Aq[0][m][1] = ab[m * 32 + W_PART];
            }
#pragma unroll
            for (int idx = 0; idx < 9; ++idx) {
                const int kx = idx / 3, ky = idx % 3;
                if (idx < 8) {
                    const int nt = ((idx + 1) % 3) * 3 + (idx + 1) / 3;
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        Aq[(idx + 1) & 1][m][0] = ab[nt * 2 * CO_T + m * 32];
                        Aq[(idx + 1) & 1][m][1] = ab[nt * 2 * CO_T + m * 32 + W_PART];
                    }
                }
                if (kx < 2) {
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        if (j % 3 == ky) {
                            Bq[(kx + 1) & 1][j][0] = bb[j * IN_COLS + kx + 1];
                            Bq[(kx + 1) & 1][j][1] = bb[j * IN_COLS + kx + 1 + IN_PART];
                        }
                }
                // the other chunk buffer was last read one chunk ago (barrier since): the next chunk's slots go in by DMA, a piece
                // or two per tap during the first taps, so that they have the rest of the chunk's MFMAs to land
                if (idx == 0) advance();
                {
                    constexpr int NPIECE = NII + NWI, PER = (NPIECE + SP_PRE_LAST_TAP) / (SP_PRE_LAST_TAP + 1);
#pragma unroll
                    for (int q = 0; q < PER; ++q) {
                        const int pc = idx * PER + q;
                        if (idx <= SP_PRE_LAST_TAP && pc < NPIECE) {
                            if (pc & 1) { if (pc / 2 < NWI) dma_w(buf ^ 1, pc / 2); else dma_in(buf ^ 1, pc - NWI); }
                            else { if (pc / 2 < NII) dma_in(buf ^ 1, pc / 2); else dma_w(buf ^ 1, pc - NII); }
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        if constexpr (PM == 2) {         // plain bf16: channels 0-15 of the chunk, then 16-31
                            const bf16x8 a0 = __builtin_bit_cast(bf16x8, Aq[idx & 1][m][0]), a1 = __builtin_bit_cast(bf16x8, Aq[idx & 1][m][1]);
                            const bf16x8 b0 = __builtin_bit_cast(bf16x8, Bq[kx & 1][n + ky][0]), b1 = __builtin_bit_cast(bf16x8, Bq[kx & 1][n + ky][1]);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[m][n], 0, 0, 0);
                        } else if constexpr (F16) {
                            const f16x8 ah = __builtin_bit_cast(f16x8, Aq[idx & 1][m][0]), am = __builtin_bit_cast(f16x8, Aq[idx & 1][m][1]);
                            const f16x8 bh = __builtin_bit_cast(f16x8, Bq[kx & 1][n + ky][0]), bm = __builtin_bit_cast(f16x8, Bq[kx & 1][n + ky][1]);
#ifdef SP_PRE_TIMING_MFMA16      // TIMING ONLY (wrong results): the same matrix work as six v_mfma_f32_16x16x32_f16 on accumulator quarters
                            {
                                f32x4s q0 = __builtin_shufflevector(acc[m][n], acc[m][n], 0, 1, 2, 3), q1 = __builtin_shufflevector(acc[m][n], acc[m][n], 4, 5, 6, 7);
                                f32x4s q2 = __builtin_shufflevector(acc[m][n], acc[m][n], 8, 9, 10, 11), q3 = __builtin_shufflevector(acc[m][n], acc[m][n], 12, 13, 14, 15);
                                q0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(am, bh, q0, 0, 0, 0);
                                q1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bm, q1, 0, 0, 0);
                                q2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, q2, 0, 0, 0);
                                q3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(am, bh, q3, 0, 0, 0);
                                q0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bm, q0, 0, 0, 0);
                                q1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, q1, 0, 0, 0);
                                const auto lo = __builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7), hi = __builtin_shufflevector(q2, q3, 0, 1, 2, 3, 4, 5, 6, 7);
                                acc[m][n] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
                            }
#ifdef SP_PRE_TIMING_EXTRA        // ... plus what the real K-packed kernel adds: one wasted half MFMA per tile and chunk, 36 more fragment reads
                            if (idx == 8 && n == 0) {
                                f32x4s q0 = __builtin_shufflevector(acc[m][1], acc[m][1], 0, 1, 2, 3), q1 = __builtin_shufflevector(acc[m][1], acc[m][1], 4, 5, 6, 7);
                                f32x4s q2 = __builtin_shufflevector(acc[m][1], acc[m][1], 8, 9, 10, 11), q3 = __builtin_shufflevector(acc[m][1], acc[m][1], 12, 13, 14, 15);
                                q0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(am, bm, q0, 0, 0, 0);
                                q1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(am, bm, q1, 0, 0, 0);
                                q2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(am, bm, q2, 0, 0, 0);
                                q3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(am, bm, q3, 0, 0, 0);
                                const auto lo = __builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7), hi = __builtin_shufflevector(q2, q3, 0, 1, 2, 3, 4, 5, 6, 7);
                                acc[m][1] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
                            }
                            if (n == 0) {
                                const u32x4s e0 = ab[(idx * 2 + m) * 64 + 16], e1 = bb[(idx + m) * IN_COLS + 2 + IN_PART];
                                asm volatile("" :: "v"(e0), "v"(e1));
                            }
#endif
                            continue;
#endif
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(am, bh, acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bm, acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[m][n], 0, 0, 0);
                        } else {
                            const bf16x8 ah = __builtin_bit_cast(bf16x8, Aq[idx & 1][m][0]), am = __builtin_bit_cast(bf16x8, Aq[idx & 1][m][1]);
                            const bf16x8 bh = __builtin_bit_cast(bf16x8, Bq[kx & 1][n + ky][0]), bm = __builtin_bit_cast(bf16x8, Bq[kx & 1][n + ky][1]);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[m][n], 0, 0, 0);
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next chunk's slots have landed ...
            __syncthreads();                                        // ... and every wave is done with this buffer
            buf ^= 1;
        }

#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] *= acc_scale;
        int v = tile;
        const int co0 = (v % a.coTiles) * CO_T;
        v /= a.coTiles;
        const int tx = v % a.tilesX;
        v /= a.tilesX;
        const int ty = v % a.tilesY;
        const int b = v / a.tilesY;
        const int y0 = ty * ROWS, x0 = tx * TW;
        if constexpr (ST) {
            float* sc = reinterpret_cast<float*>(lds + 2 * BUF);       // [8 waves][64 channels][mean, M2]
            constexpr float npw = (float)(NT * 32), inv_npw = 1.f / npw;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int piv = __builtin_bit_cast(int, acc[m][0][r]);
                    const float p0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(piv, 0));
                    const float p1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(piv, 32));
                    const float pv = kh ? p1 : p0;
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const float d = acc[m][n][r] - pv;
                        s1 += d;
                        s2 = fmaf(d, d, s2);
                    }
                    s1 = sp_half_sum(s1);
                    s2 = sp_half_sum(s2);
                    if (l31 == 31) {
                        const int cl = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                        sc[(wn * 64 + cl) * 2] = fmaf(s1, inv_npw, pv);
                        sc[(wn * 64 + cl) * 2 + 1] = fmaxf(fmaf(-s1 * inv_npw, s1, s2), 0.f);
                    }
                }
            __syncthreads();
            if (tid < 64 && co0 + tid < a.Cout) {
                float mw[8], qw[8];
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    mw[w] = sc[(w * 64 + tid) * 2];
                    qw[w] = sc[(w * 64 + tid) * 2 + 1];
                }
                const float mean = 0.125f * (((mw[0] + mw[1]) + (mw[2] + mw[3])) + ((mw[4] + mw[5]) + (mw[6] + mw[7])));
                float m2 = ((qw[0] + qw[1]) + (qw[2] + qw[3])) + ((qw[4] + qw[5]) + (qw[6] + qw[7]));
#pragma unroll
                for (int w = 0; w < 8; ++w) m2 = fmaf(npw * (mw[w] - mean), mw[w] - mean, m2);
                const int64_t nblk = (int64_t)a.B * a.tilesY * a.tilesX;
                const int64_t blk = ((int64_t)b * a.tilesY + ty) * a.tilesX + tx;
                float* sp = a.stats + ((int64_t)(co0 + tid) * nblk + blk) * 3;
                sp[0] = 8.f * npw;
                sp[1] = mean;
                sp[2] = m2;
            }
            __syncthreads();
        }
        float* zb = a.z + (int64_t)(W16 ? 2 * b + (l31 >> 4) : b) * a.z_bs;      // (W16: b is the image pair, lanes 16-31 hold the second image)
        const int xo = W16 ? (l31 & 15) : x0 + l31;
        if constexpr (PM == 1 && !ST && !W16) {
            if (a.zP && co0 >= a.zP_ch0) {
                // Pre-split output (SpPreArgs::zP).  A lane holds channels 8 g + 4 kh + {0 .. 3} of its pixel for the four groups g of a
                // 32-row accumulator; v_permlane32_swap trades halves with the partner lane (kh ^ 1): afterwards a kh = 0 lane owns all
                // 8 channels of groups 0 and 2, a kh = 1 lane those of groups 1 and 3 -- whole slots, 512 contiguous bytes per half-wave
                const float zs = zP_scale;
                u32x4s* zp = reinterpret_cast<u32x4s*>(reinterpret_cast<unsigned*>(a.zP) + (int64_t)b * a.zP_bs);
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const int yo = y0 + wn * NT + n;
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            float v[8];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                // (scalars first: __builtin_bit_cast applied to a vector ELEMENT yields element 0 with hipcc 7.2)
                                const float f0 = acc[m][n][8 * e + j], f1 = acc[m][n][8 * e + 4 + j];             // groups 2 e, 2 e + 1
                                const unsigned d0 = __builtin_bit_cast(unsigned, f0);
                                const unsigned d1 = __builtin_bit_cast(unsigned, f1);
                                const auto sw = __builtin_amdgcn_permlane32_swap(d0, d1, false, false);
                                const unsigned lo = sw[0], hi4 = sw[1];
                                v[j] = __builtin_bit_cast(float, lo);
                                v[4 + j] = __builtin_bit_cast(float, hi4);
                            }
                            u32x4s hi, mid;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                unsigned hh, mm;
                                split2h_s(v[2 * k], v[2 * k + 1], zs, hh, mm);
                                hi[k] = hh;
                                mid[k] = mm;
                            }
                            if (yo < a.H && xo < a.W) {
                                const int c8 = ((co0 - a.zP_ch0 + m * 32) >> 3) + 2 * e + kh;
                                u32x4s* d = zp + ((int64_t)(c8 * a.H + yo) * 2) * a.W + xo;
                                d[0] = hi;
                                d[a.W] = mid;
                            }
                        }
                    }
                continue;
            }
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int yo = y0 + wn * NT + n;
                if (yo < a.H && xo < a.W) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                        if (co < a.Cout) zb[(int64_t)co * HW + (int64_t)yo * a.W + xo] = acc[m][n][r];
                    }
                }
            }
    }
}


// ------------------------------------------------------------------ the same tile on v_mfma_f32_16x16x32 (round 5)
// MI355X_MICROARCH.md, 'DVFS give-back' (7): in clock-limited MFMA loops the chip holds a higher clock on the 16x16x32 shape than on
// 32x32x16 at equal cycles per FLOP (half the accumulator traffic per FLOP).  Measured in this kernel (profiles/r05_mfma_shape_ab.md):
// the same matrix work issued as 16x16x32 runs 8-9 % faster on every layer shape.  So: the block, its LDS image, the DMA schedule and
// the persistent tile walk of conv3x3_split_pre_kernel, with the wave's 64 channels x 2 rows x 32 pixels held as SIXTEEN 16 x 16
// accumulators, M = 16 pixels of a row (A = input fragment), N = 16 output channels (B = weight fragment): a lane then owns four
// CONSECUTIVE pixels of one channel -- the epilogue stores float4s (16 store instructions per wave and tile instead of 64) and the
// BatchNorm statistics need in-lane sums plus two cross-lane steps per channel tile (the 32x32 form: 5 DPP steps per register).
// K = 32 per instruction:
//  * PM == 2 (plain bf16, 32-channel chunks): K = the chunk's 32 channels -- lane group q = lane / 16 reads channel group q.  Nine
//    steps of 16 MFMAs per chunk, the fragment reads of the 32x32 form (36 weight + 24 input per wave and chunk).
//  * split operands (16-channel chunks, three product terms per tap): K-packing.
//      'a' step (tap):          A = [x_hi | x_hi]            B = [w_hi(tap) | w_mid(tap)]      -> x_hi w_hi + x_hi w_mid
//      'H' step (ky), taps (ky,0), (ky,1):  A = [x_mid(kx=0) | x_mid(kx=1)]  B = [w_hi(ky,0) | w_hi(ky,1)]
//      'V' step, taps (0,2), (1,2):         A = [x_mid(row n) | x_mid(row n+1)] at kx = 2,  B = [w_hi(0,2) | w_hi(1,2)]
//      'S' step, tap (2,2):                 A = [x_mid(row n+2) | same],  B = [w_hi(2,2) | 0]   (27 K16 products are an odd number:
//    one half instruction per tile and chunk is padding -- 14 steps for 13.5 steps' worth, +3.7 %; the zero half reads the zero padding
//    behind the halo image).  Lanes pick their K group by address: every fragment is still ONE conflict-free ds_read_b128.
template <bool ST, int PM, bool W16, bool RD = false>
__global__ __launch_bounds__(512, 2) void conv3x3_pre16_kernel(SpPreArgs a) {
    constexpr bool F16 = PM == 1;
    using C = SpPreCfg<W16>;
    static_assert(C::NW == 8 && C::NT == 2, "conv3x3_pre16_kernel: 8 waves of 2 rows");
    constexpr int NT = 2, IN_COLS = C::IN_COLS, NWI = C::NWI, CO_T = C::CO_T, NPIXP = C::NPIXP;
    constexpr int BUF = C::BUF_SLOTS, W_PART = C::W_PART, IN_PART = C::IN_PART, ROWS = C::ROWS, TW = C::TW;
    constexpr int NII = (2 * IN_PART + 511) / 512;
    constexpr int CH_OFF = W16 ? 18 : 16;                              // column of the second 16-pixel tile in the halo image
    constexpr int NSTEP = PM == 2 ? 9 : 14;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_s[];
    u32x4s* lds = reinterpret_cast<u32x4s*>(smem_s);

    const int ntiles = a.tilesX * a.tilesY * a.B * a.coTiles;
    int t_first, t_end, t_stride;
    {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int q = ntiles >> 3, r = ntiles & 7;
        const int start = xcd * q + min(xcd, r);
        t_stride = (gridDim.x + 7 - xcd) >> 3;
        t_first = start + j;
        t_end = start + q + (xcd < r ? 1 : 0);
    }
    if (t_first >= t_end) return;

    const int tid = threadIdx.x, lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, lq = lane >> 4, hq = lq & 1, gq = lq >> 1;
    const int HW = a.H * a.W;
    const int nchunks = PM == 2 ? a.Cin >> 5 : a.Cin >> 4;

    const i32x4s wr = sp_rsrc4(a.wq, (int64_t)a.Cin * (PM == 2 ? 1 : 2) * 9 * a.Cout * 2);
    float xs_inv = 1.f, xs_inv2 = 1.f;
    (void)amax_scale(amax_read(a.x_slots), a.x_always != 0, xs_inv);
    (void)amax_scale(amax_read(a.x_slots2), a.x_always != 0, xs_inv2);
    const int split_chunk = a.split_ch ? (PM == 2 ? a.split_ch >> 5 : a.split_ch >> 4) : 0;
    const float grp_ratio = split_chunk ? xs_inv / xs_inv2 : 1.f;
    const float acc_scale = (split_chunk ? xs_inv2 : xs_inv) * (F16 ? reinterpret_cast<const float*>(a.wq + (int64_t)a.Cin * 2 * 9 * a.Cout)[1] : 1.f);
    const unsigned in_step = (unsigned)(16 * HW * 4), w_step = (unsigned)(2 * 9 * 2 * a.Cout * 16);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_s;

    // ---- staging (conv3x3_split_pre_kernel's): slot position i = (wn + 8 k) * 64 + lane of the chunk image
    unsigned in_off[NII], w_off[NWI];
    i32x4s xr;
    int st_tile = t_first, st_chunk = 0;
    unsigned cin_bytes = 0, cw_bytes = 0;
    auto setup_stage = [&]() __attribute__((always_inline)) {
        const bool live = st_tile < t_end;
        int v = live ? st_tile : t_first;
        const int co0 = (v % a.coTiles) * CO_T;
        v /= a.coTiles;
        const int tx = v % a.tilesX;
        v /= a.tilesX;
        const int ty = v % a.tilesY;
        const int b = v / a.tilesY;
        const int y0 = ty * ROWS, x0 = tx * TW;
        xr = sp_rsrc4(reinterpret_cast<const unsigned*>(a.xs) + (int64_t)(W16 ? 2 * b : b) * a.xs_bs,
                      (W16 ? a.xs_bs * 4 : 0) + (int64_t)a.Cin * HW * (PM == 2 ? 2 : 4));
#pragma unroll
        for (int k = 0; k < NII; ++k) {
            const int i = (wn + 8 * k) * 64 + lane;
            const int ph = i / NPIXP, pix = i % NPIXP;
            const int r = pix / IN_COLS, c = pix % IN_COLS;
            const int img = W16 ? c / 18 : 0;
            const int yy = y0 - 1 + r, xx = W16 ? c % 18 - 1 : x0 - 1 + c;
            const bool ok = live && i < 2 * IN_PART && pix < C::NPIX && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
            const unsigned io = W16 ? (unsigned)(img * a.xs_bs * 4) : 0u;
            in_off[k] = !ok ? OOB_S : PM == 2 ? io + (unsigned)(((ph * a.H + yy) * a.W + xx) * 16)
                                              : io + (unsigned)(((((ph & 1) * a.H + yy) * 2 + (ph >> 1)) * a.W + xx) * 16);
        }
#pragma unroll
        for (int k = 0; k < NWI; ++k) {
            const int i = tid + 512 * k;
            const int co = i & (CO_T - 1), pth = i >> 6;
            const bool ok = live && (i < C::W_SLOTS) && (co0 + co < a.Cout);
            w_off[k] = ok ? (unsigned)((pth * a.Cout + co0 + co) * 16) : OOB_S;
        }
    };
    auto advance = [&]() __attribute__((always_inline)) {
        ++st_chunk;
        cin_bytes += in_step;
        cw_bytes += w_step;
        if (st_chunk == nchunks) {
            st_chunk = 0;
            cin_bytes = cw_bytes = 0;
            st_tile += t_stride;
            setup_stage();
        }
    };
    auto dma_in = [&](int buf, int k) __attribute__((always_inline)) {
        if ((wn + 8 * k) * 64 < 2 * IN_PART)
            sp_dma16(xr, lds0 + (unsigned)((buf * BUF + C::W_SLOTS + (wn + 8 * k) * 64) * 16), in_off[k], cin_bytes);
    };
    auto dma_w = [&](int buf, int k) __attribute__((always_inline)) {
        if (k < NWI - 1 || wn < (C::W_SLOTS - 512 * (NWI - 1)) / 64)
            sp_dma16(wr, lds0 + (unsigned)((buf * BUF + (wn + 8 * k) * 64) * 16), w_off[k], cw_bytes);
    };

    // ---- fragment bases (slots inside a chunk buffer).  Lane (r16, lq): row / column r16 of the 16 x 16 tile, K group lq (8 values)
    const int xrow0 = (wn * NT) * IN_COLS + r16;
    const u32x4s* const xa_p = lds + C::W_SLOTS + (PM == 2 ? lq : hq) * NPIXP + xrow0;        // + row * IN_COLS + ch * CH_OFF + kx
    const u32x4s* const xh_p = lds + C::W_SLOTS + (2 + hq) * NPIXP + xrow0 + gq;                // 'H': x_mid, kx = gq
    const u32x4s* const xv_p = lds + C::W_SLOTS + (2 + hq) * NPIXP + xrow0 + gq * IN_COLS + 2;  // 'V': x_mid, rows n + gq, kx = 2
    const u32x4s* const xs_p = lds + C::W_SLOTS + (2 + hq) * NPIXP + xrow0 + 2;                 // 'S': x_mid, kx = 2 (both K halves)
    const u32x4s* const wa_p = lds + (gq * 9 * 2 + hq) * CO_T + r16;                            // + tap * 2 CO_T + ct * 16: [w_hi | w_mid] (PM 2: 32 channels)
    const u32x4s* const wh_p = lds + (gq * 2 + hq) * CO_T + r16;                                // 'H': + ky * 6 CO_T + ct * 16: taps (ky, gq)
    const u32x4s* const wv_p = lds + ((3 * gq + 2) * 2 + hq) * CO_T + r16;                      // 'V': taps (gq, 2)
    const u32x4s* const ws_p = gq == 0 ? lds + (8 * 2 + hq) * CO_T + r16 : lds + C::W_SLOTS + C::NPIX;   // 'S': tap (2,2) | zeros
    const int ws_ct = gq == 0 ? 16 : 0;

    setup_stage();
#pragma unroll
    for (int k = 0; k < NII; ++k) dma_in(0, k);
#pragma unroll
    for (int k = 0; k < NWI; ++k) dma_w(0, k);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (int tile = t_first; tile < t_end; tile += t_stride) {
        f32x4s acc[4][NT][2];                       // [channel tile][row][16-pixel tile]
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) acc[ct][n][ch] = f32x4s{0.f, 0.f, 0.f, 0.f};

        for (int c = 0; c < nchunks; ++c) {
            if (split_chunk && c == split_chunk && grp_ratio != 1.f) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int ch = 0; ch < 2; ++ch) acc[ct][n][ch] *= grp_ratio;
            }
            const int bo = buf * BUF;
            u32x4s Xp[2][NT + 2][2], Wf[2][4];
            // the input fragments of pool p for phase ph (0..2: 'a' at kx = ph; 3: 'H'; 4: 'V' rows 0..1 + 'S' rows 2..3), fragment f
            auto load_x = [&](int ph, int f) __attribute__((always_inline)) {
                const int j = f >> 1, ch = f & 1, p = ph & 1;
                if (ph < 3) Xp[p][j][ch] = xa_p[bo + j * IN_COLS + ch * CH_OFF + ph];
                else if (ph == 3) Xp[p][j][ch] = xh_p[bo + j * IN_COLS + ch * CH_OFF];
                else if (j < NT) Xp[p][j][ch] = xv_p[bo + j * IN_COLS + ch * CH_OFF];
                else Xp[p][j][ch] = xs_p[bo + j * IN_COLS + ch * CH_OFF];
            };
            auto load_w = [&](int s) __attribute__((always_inline)) {       // the weight fragments of step s
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    if (s < 9) Wf[s & 1][ct] = wa_p[bo + ((s % 3) * 3 + s / 3) * 2 * CO_T + ct * 16];
                    else if (s < 12) Wf[s & 1][ct] = wh_p[bo + (s - 9) * 6 * CO_T + ct * 16];
                    else if (s == 12) Wf[s & 1][ct] = wv_p[bo + ct * 16];
                    else Wf[s & 1][ct] = ws_p[bo + ct * ws_ct];
                }
            };
#pragma unroll
            for (int f = 0; f < 2 * (NT + 2); ++f) load_x(0, f);
            load_w(0);
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                const int ph = s < 9 ? s / 3 : (s < 12 ? 3 : 4);           // phase of this step; its fragments sit in pool ph & 1
                const int ro = s < 9 ? s % 3 : (s < 12 ? s - 9 : (s == 12 ? 0 : NT));   // first fragment row of output row 0
                if (s + 1 < NSTEP) load_w(s + 1);
                if (s < 12 && ph + 1 < (PM == 2 ? 3 : 5)) {                // the next phase's input fragments, a third per step
#pragma unroll
                    for (int f = 0; f < 2 * (NT + 2); ++f)
                        if (f % 3 == s % 3) load_x(ph + 1, f);
                }
                if (s == 0) advance();
                {
                    constexpr int NPIECE = NII + NWI, PER = (NPIECE + SP_PRE_LAST_TAP) / (SP_PRE_LAST_TAP + 1);
#pragma unroll
                    for (int q = 0; q < PER; ++q) {
                        const int pc = s * PER + q;
                        if (s <= SP_PRE_LAST_TAP && pc < NPIECE) {
                            if (pc & 1) { if (pc / 2 < NWI) dma_w(buf ^ 1, pc / 2); else dma_in(buf ^ 1, pc - NWI); }
                            else { if (pc / 2 < NII) dma_in(buf ^ 1, pc / 2); else dma_w(buf ^ 1, pc - NII); }
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int ch = 0; ch < 2; ++ch) {
                            if constexpr (F16)
                                acc[ct][n][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, Xp[ph & 1][ro + n][ch]),
                                                                                        __builtin_bit_cast(f16x8, Wf[s & 1][ct]), acc[ct][n][ch], 0, 0, 0);
                            else
                                acc[ct][n][ch] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Xp[ph & 1][ro + n][ch]),
                                                                                         __builtin_bit_cast(bf16x8, Wf[s & 1][ct]), acc[ct][n][ch], 0, 0, 0);
                        }
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            buf ^= 1;
        }

#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) acc[ct][n][ch] *= acc_scale;
        int v = tile;
        const int co0 = (v % a.coTiles) * CO_T;
        v /= a.coTiles;
        const int tx = v % a.tilesX;
        v /= a.tilesX;
        const int ty = v % a.tilesY;
        const int b = v / a.tilesY;
        const int y0 = ty * ROWS, x0 = tx * TW;
        if constexpr (ST) {
            // BatchNorm (n, mean, M2) record of the tile per channel.  Lane = channel r16 of tile ct, 16 pixel values in registers:
            // own pivot-shifted sums -> (mean, M2) of 16; the four lane groups merged pairwise (equal counts: Chan's formula); the
            // eight waves through LDS -- in the chunk buffer nobody reads until the next chunk's DMA (issued behind the barriers below)
            float* sc = reinterpret_cast<float*>(lds + (buf ^ 1) * BUF);       // [8 waves][64 channels][mean, M2]
            constexpr float npw = (float)(NT * 32);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const float pv = acc[ct][0][0][0];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float d = acc[ct][n][ch][r] - pv;
                            s1 += d;
                            s2 = fmaf(d, d, s2);
                        }
                float mean = fmaf(s1, 1.f / 16.f, pv), m2 = fmaxf(fmaf(-s1 * (1.f / 16.f), s1, s2), 0.f), cnt = 16.f;
#pragma unroll
                for (int o = 16; o <= 32; o <<= 1) {
                    const float mo = __shfl_xor(mean, o, 64), qo = __shfl_xor(m2, o, 64);
                    const float dlt = mo - mean;
                    m2 = (m2 + qo) + (0.5f * cnt) * dlt * dlt;
                    mean = 0.5f * (mean + mo);
                    cnt *= 2.f;
                }
                if (lq == 0) {
                    sc[(wn * 64 + ct * 16 + r16) * 2] = mean;
                    sc[(wn * 64 + ct * 16 + r16) * 2 + 1] = m2;
                }
            }
            __syncthreads();
            if (tid < 64 && co0 + tid < a.Cout) {
                float mw[8], qw[8];
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    mw[w] = sc[(w * 64 + tid) * 2];
                    qw[w] = sc[(w * 64 + tid) * 2 + 1];
                }
                const float mean = 0.125f * (((mw[0] + mw[1]) + (mw[2] + mw[3])) + ((mw[4] + mw[5]) + (mw[6] + mw[7])));
                float m2 = ((qw[0] + qw[1]) + (qw[2] + qw[3])) + ((qw[4] + qw[5]) + (qw[6] + qw[7]));
#pragma unroll
                for (int w = 0; w < 8; ++w) m2 = fmaf(npw * (mw[w] - mean), mw[w] - mean, m2);
                const int64_t nblk = (int64_t)a.B * a.tilesY * a.tilesX;
                const int64_t blk = ((int64_t)b * a.tilesY + ty) * a.tilesX + tx;
                float* sp = a.stats + ((int64_t)(co0 + tid) * nblk + blk) * 3;
                sp[0] = 8.f * npw;
                sp[1] = mean;
                sp[2] = m2;
            }
            __syncthreads();
        }
        // RD, part 1: the z tile of the unit below goes into registers FIRST (16 loads in flight; the fragment registers of the main loop
        // are dead here), the da stores below run while they travel, and the sums are taken after the stores
        f32x4s zr[4][NT][2];
        const float* rd_sv = nullptr;
        if constexpr (RD) {
            const int img0 = W16 ? 2 * b : b;
            rd_sv = a.rd_save + (int64_t)(a.rd_gimg ? img0 / a.rd_gimg : 0) * 4 * a.Cout;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch) {
                        const int co = co0 + ct * 16 + r16, yo = y0 + wn * NT + n, xo = (W16 ? 0 : x0 + 16 * ch) + 4 * lq;
                        f32x4s zz = {0.f, 0.f, 0.f, 0.f};
                        if (co < a.Cout && yo < a.H && xo < a.W) {
                            const int64_t zo = (int64_t)(W16 ? img0 + ch : img0) * a.rd_z_bs + (int64_t)co * HW + (int64_t)yo * a.W + xo;
                            if (a.rd_z16) {
                                const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const __bf16*>(a.rd_z) + zo);
                                zz = f32x4s{__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                                            __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u)};
                            } else {
                                zz = *reinterpret_cast<const f32x4s*>(a.rd_z + zo);
                            }
                        }
                        zr[ct][n][ch] = zz;
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
        // A lane holds, per (channel tile, row), channel r16's pixels 4 lq .. + 3 of BOTH 16-pixel tiles: stored as they stand, an
        // instruction would write 64-byte half lines (16 channels x 4 lanes x 16 B).  Lanes r16 >= 8 of the first tile and lanes
        // r16 < 8 of the second trade places (two masked row_ror:8 DPP moves per register) so that each store instruction writes
        // eight channels' WHOLE 128-byte lines: instruction 1 channels 0-7 (pixels 0-15 from lanes r16 < 8, 16-31 from lanes r16 >= 8),
        // instruction 2 channels 8-15.
        {
            const int half = r16 >> 3, c8 = r16 & 7;
            float* zb = a.z + (int64_t)(W16 ? 2 * b + half : b) * a.z_bs;      // (W16: the second 16-pixel tile is the pair's second image)
            const int xo = (W16 ? 0 : x0 + 16 * half) + 4 * lq;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int yo = y0 + wn * NT + n;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    const i32x4s va = __builtin_bit_cast(i32x4s, acc[ct][n][0]), vb = __builtin_bit_cast(i32x4s, acc[ct][n][1]);
                    // d1: lanes 8-15 <- tile 1 of channel r16 - 8;  d2: lanes 0-7 <- tile 0 of channel r16 + 8
                    const i32x4s e1 = {__builtin_amdgcn_update_dpp(va.x, vb.x, 0x128, 0xf, 0xc, false), __builtin_amdgcn_update_dpp(va.y, vb.y, 0x128, 0xf, 0xc, false),
                                       __builtin_amdgcn_update_dpp(va.z, vb.z, 0x128, 0xf, 0xc, false), __builtin_amdgcn_update_dpp(va.w, vb.w, 0x128, 0xf, 0xc, false)};
                    const i32x4s e2 = {__builtin_amdgcn_update_dpp(vb.x, va.x, 0x128, 0xf, 0x3, false), __builtin_amdgcn_update_dpp(vb.y, va.y, 0x128, 0xf, 0x3, false),
                                       __builtin_amdgcn_update_dpp(vb.z, va.z, 0x128, 0xf, 0x3, false), __builtin_amdgcn_update_dpp(vb.w, va.w, 0x128, 0xf, 0x3, false)};
                    const f32x4s d1 = __builtin_bit_cast(f32x4s, e1), d2 = __builtin_bit_cast(f32x4s, e2);
                    if constexpr (PM == 2 && !ST && !W16 && !RD) {
                        if (a.zP && co0 >= a.zP_ch0) {
                            // Pre-split output, plain bf16 (SpPreArgs::zP, one part): the eight lanes c8 = 0 .. 7 of a (pixel quad, tile half)
                            // hold 8 channels x 4 pixels of channel group A (d1: channels 16 ct + c8) and of group B (d2: + 8).  Three
                            // butterfly exchanges (lane ^ 1, ^ 2, ^ 4) transpose them: afterwards lane c8 owns one whole slot -- the 8
                            // channels of group (c8 >> 2 ? B : A) at pixel c8 & 3 of the quad.
                            const unsigned a0 = pack2bf(d1[0], d1[1]), a1 = pack2bf(d1[2], d1[3]);
                            const unsigned b0 = pack2bf(d2[0], d2[1]), b1 = pack2bf(d2[2], d2[3]);
                            const bool o1 = (c8 & 1) != 0, o2 = (c8 & 2) != 0, o4 = (c8 & 4) != 0;
                            // stage 1: (pixel pair of one channel) x 2 lanes -> (channel pair) of pixel (c8 & 1) [and + 2]
                            auto st1 = [&](unsigned x) {
                                const unsigned y = (unsigned)__shfl_xor((int)x, 1, 64);
                                return o1 ? ((y >> 16) | (x & 0xffff0000u)) : ((x & 0xffffu) | (y << 16));
                            };
                            const unsigned A0 = st1(a0), A1 = st1(a1), B0 = st1(b0), B1 = st1(b1);
                            // stage 2: -> the channel quad (c8 >> 2) of pixel c8 & 3: lanes with bit 1 clear keep pixel (c8 & 1), the others + 2
                            const unsigned ra = (unsigned)__shfl_xor((int)(o2 ? A0 : A1), 2, 64), rb = (unsigned)__shfl_xor((int)(o2 ? B0 : B1), 2, 64);
                            const unsigned ka = o2 ? A1 : A0, kb = o2 ? B1 : B0;
                            const unsigned Alo = o2 ? ra : ka, Ahi = o2 ? ka : ra, Blo = o2 ? rb : kb, Bhi = o2 ? kb : rb;
                            // stage 3: lanes with bit 2 clear collect group A (their quad = channels 0 .. 3, the partner's 4 .. 7), the others B
                            const unsigned r0 = (unsigned)__shfl_xor((int)(o4 ? Alo : Blo), 4, 64), r1 = (unsigned)__shfl_xor((int)(o4 ? Ahi : Bhi), 4, 64);
                            const u32x4s slot = o4 ? u32x4s{r0, r1, Blo, Bhi} : u32x4s{Alo, Ahi, r0, r1};
                            const int xs = x0 + 16 * half + 4 * lq + (c8 & 3);
                            if (yo < a.H && xs < a.W) {
                                const int g8 = ((co0 - a.zP_ch0) >> 3) + 2 * ct + (o4 ? 1 : 0);
                                u32x4s* zp = reinterpret_cast<u32x4s*>(reinterpret_cast<unsigned*>(a.zP) + (int64_t)b * a.zP_bs);
                                zp[((int64_t)g8 * a.H + yo) * a.W + xs] = slot;
                            }
                            continue;
                        }
                    }
                    if (yo < a.H && xo < a.W) {
                        const int co = co0 + ct * 16 + c8;
                        const int64_t off = (int64_t)co * HW + (int64_t)yo * a.W + xo;
                        if (a.z16) {          // z stored as bf16: 8 bytes per lane, 64 contiguous bytes per channel row and instruction
                            __bf16* o = reinterpret_cast<__bf16*>(a.z) + (int64_t)(W16 ? 2 * b + half : b) * a.z_bs + off;
                            const bf16x2 p0 = {(__bf16)d1[0], (__bf16)d1[1]}, p1 = {(__bf16)d1[2], (__bf16)d1[3]};
                            const bf16x2 q0 = {(__bf16)d2[0], (__bf16)d2[1]}, q1 = {(__bf16)d2[2], (__bf16)d2[3]};
                            if (co < a.Cout) *reinterpret_cast<uint2*>(o) = uint2{__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1)};
                            if (co + 8 < a.Cout) *reinterpret_cast<uint2*>(o + (int64_t)8 * HW) = uint2{__builtin_bit_cast(unsigned, q0), __builtin_bit_cast(unsigned, q1)};
                        } else {
                            float* o = zb + off;
                            if (co < a.Cout) *reinterpret_cast<f32x4s*>(o) = d1;
                            if (co + 8 < a.Cout) *reinterpret_cast<f32x4s*>(o + (int64_t)8 * HW) = d2;
                        }
                    }
                }
            }
        }
        if constexpr (RD) {
            // RD, part 2: BatchNorm-backward reduce of the unit below (bn_relu_bwd_reduce_kernel's sums) from the tile in registers: dy = da
            // where that unit's output was positive, (sum dy, sum dy xhat) per channel.  Lane = channel, 16 pixel values: in-lane fp32
            // sums, the four lane groups by two shuffles, the eight waves through LDS in fp64 -> one record per tile and channel.  The
            // pass that read (da, z) -- 8 bytes per element -- becomes a 4-byte (2-byte: stored bf16) read of z here.
            __builtin_amdgcn_sched_barrier(0);
            float* sc = reinterpret_cast<float*>(lds + (buf ^ 1) * BUF);       // [8 waves][64 channels][s, sx]
            float vmax = 0.f;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int co = co0 + ct * 16 + r16, cc = min(co, a.Cout - 1);
                const float mean = rd_sv[cc], invstd = rd_sv[a.Cout + cc], scl = rd_sv[2 * a.Cout + cc], sh = rd_sv[3 * a.Cout + cc];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch) {
                        const int yo = y0 + wn * NT + n, xo = (W16 ? 0 : x0 + 16 * ch) + 4 * lq;
                        if (co < a.Cout && yo < a.H && xo < a.W) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float da = acc[ct][n][ch][r], d = zr[ct][n][ch][r] - mean;
                                const float dy = fmaf(d, scl, sh) > 0.f ? da : 0.f;
                                s1 += dy;
                                s2 = fmaf(dy, d * invstd, s2);
                                vmax = fmaxf(vmax, fabsf(da));
                            }
                        }
                    }
                s1 += __shfl_xor(s1, 16, 64);
                s2 += __shfl_xor(s2, 16, 64);
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (lq == 0) {
                    sc[(wn * 64 + ct * 16 + r16) * 2] = s1;
                    sc[(wn * 64 + ct * 16 + r16) * 2 + 1] = s2;
                }
            }
            if (a.rd_amax) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
                if (lane == 0 && vmax == vmax) atomicMax(a.rd_amax + ((blockIdx.x * 8 + wn) & (AMAX_SLOTS - 1)) * AMAX_STRIDE, __builtin_bit_cast(unsigned, vmax));
            }
            __syncthreads();
            if (tid < 64 && co0 + tid < a.Cout) {
                double t1 = 0.0, t2 = 0.0;
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    t1 += (double)sc[(w * 64 + tid) * 2];
                    t2 += (double)sc[(w * 64 + tid) * 2 + 1];
                }
                const int64_t blk = ((int64_t)b * a.tilesY + ty) * a.tilesX + tx;
                float* o = a.rd_rec + (blk * a.Cout + co0 + tid) * 4;
                o[0] = (float)t1;
                o[1] = (float)(t1 - (double)o[0]);
                o[2] = (float)t2;
                o[3] = (float)(t2 - (double)o[2]);
            }
            __syncthreads();
        }
    }
}

#ifndef SP_PRE16
#define SP_PRE16 1      // 1: conv3x3_pre16_kernel (v_mfma_f32_16x16x32); 0: conv3x3_split_pre_kernel (32x32x16) -- same-box A/B builds
#endif
template <bool ST, int PM, bool W16, bool RD = false>
int launch_split_pre(SpPreArgs a, hipStream_t st) {
    using C = SpPreCfg<W16>;
    const int LDS_BYTES = C::LDS_BYTES + (ST ? C::NW * 64 * 2 * 4 : 0);
    a.tilesX = W16 ? 1 : cdiv(a.W, C::TW);
    a.tilesY = cdiv(a.H, C::ROWS);
    a.coTiles = cdiv(a.Cout, C::CO_T);
    if (W16) a.B /= 2;                               // tiles hold image pairs
    const int64_t tiles = (int64_t)a.B * a.tilesX * a.tilesY * a.coTiles;
    ONET_REQUIRE(tiles > 0 && tiles < (1ll << 31), "conv3x3_split_pre: tile count %lld out of range", (long long)tiles);
    // which launches take the 16x16x32 kernel (profiles/r05_mfma_shape_ab.md): plain bf16 operands (K = 32 channels: no packing) always; the
    // (hi | mid) forward WITH the statistics epilogue and the input gradients that carry the fused reduce (the lane layout's cheap
    // epilogues: -5 %); the plain (hi | mid) input gradient keeps the 32x32x16 kernel (equal speed), and so does the (hi | mid) 16-pixel
    // level (that instance of the new kernel exceeds the register budget).  SP_PRE16 = 2 / 0: everything / nothing (A-B builds).
    constexpr bool P16 = RD || (SP_PRE16 > 1) || (SP_PRE16 == 1 && (PM == 2 || (ST && !W16)));
    void (*kern)(SpPreArgs);
    if constexpr (P16) kern = conv3x3_pre16_kernel<ST, PM, W16, RD>;      // (constexpr: the instances not dispatched are not built)
    else kern = conv3x3_split_pre_kernel<ST, PM, W16>;
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    }
    const int64_t resident = (int64_t)device_cu_count();
    const int64_t blocks = std::min<int64_t>((tiles + 7) / 8 * 8, std::max<int64_t>(8, resident / 8 * 8));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(C::NW * 64), LDS_BYTES, st, a);
    return check_launch("conv3x3_split_pre_kernel");
}

int split_nparts(int B, int H, int W) {
    if (B <= 0 || W < 32 || (W % SpCfg::TW) || (H % SpCfg::ROWS)) return 0;
    const int64_t n = (int64_t)B * (H / SpCfg::ROWS) * (W / SpCfg::TW);
    return n < (1 << 30) ? (int)n : 0;
}

template <bool ST, bool NORM, bool F16>
int launch_split(SpArgs a, hipStream_t st) {
    using C = SpCfg;
    const int LDS_BYTES = C::LDS_BYTES + (ST ? C::NW * 64 * 2 * 4 : 0) + (NORM ? 2 * 2 * 8 * 4 * 4 : 0);
    a.tilesX = cdiv(a.W, C::TW);
    a.tilesY = cdiv(a.H, C::ROWS);
    a.coTiles = cdiv(a.Cout, C::CO_T);
    const int64_t tiles = (int64_t)a.B * a.tilesX * a.tilesY * a.coTiles;
    ONET_REQUIRE(tiles > 0 && tiles < (1ll << 31), "conv3x3_split: tile count %lld out of range", (long long)tiles);
    auto kern = conv3x3_split_kernel<ST, NORM, F16>;
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    }
    // persistent grid: one 8-wave block per CU, a multiple of the 8 XCDs
    const int64_t resident = (int64_t)device_cu_count();
    const int64_t blocks = std::min<int64_t>((tiles + 7) / 8 * 8, std::max<int64_t>(8, resident / 8 * 8));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), LDS_BYTES, st, a);
    return check_launch("conv3x3_split_kernel");
}

int split_fwd(const float* x, int64_t x_bs, const void* wq, float* z, int64_t z_bs, int B, int Cin, int Cout, int H, int W,
              void* stream, float* stats, int wq_f16, const float* nsave = nullptr, int n_groups = 0, const unsigned* x_amax = nullptr,
              int x_always = 0) {
    ONET_REQUIRE(x && wq && z, "conv3x3_split_fwd: null pointer");
    ONET_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 16, "conv3x3_split_fwd: bad shape (maps wider than 16 pixels)");
    ONET_REQUIRE((Cin % 16) == 0, "conv3x3_split_fwd: Cin must be a multiple of 16 (use onet_conv_fwd)");
    ONET_REQUIRE((W & 3) == 0 && (x_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0,
                 "conv3x3_split_fwd: W %% 4 == 0 and 16-byte aligned image rows required (use onet_conv3x3_winograd4_fwd)");
    ONET_REQUIRE(x_bs >= (int64_t)Cin * H * W && z_bs >= (int64_t)Cout * H * W, "conv3x3_split_fwd: batch stride too small");
    ONET_REQUIRE((int64_t)(Cin + 32) * H * W * 4 < (1ll << 31) && (int64_t)(Cin + 32) * 2 * 9 * Cout * 2 < (1ll << 31),
                 "conv3x3_split_fwd: operand exceeds the 2 GiB buffer-resource range");
    SpArgs a{x, x_bs, (const __bf16*)wq, z, z_bs, B, Cin, Cout, H, W, 0, 0, 0, stats, nsave, 1, x_amax, x_always};
    ONET_REQUIRE(!x_amax || wq_f16, "conv3x3_split: magnitude slots go with the fp16 pack");
    if (stats) ONET_REQUIRE(split_nparts(B, H, W) > 0, "conv3x3_split_fwd_stats: the map must be made of full 16 x 32 tiles");
    if (nsave) {
        ONET_REQUIRE(n_groups > 0 && B % n_groups == 0, "conv3x3_split_fwd_norm: the batch must hold n_groups equal statistics groups");
        a.nimg = B / n_groups;
        if (wq_f16) return stats ? launch_split<true, true, true>(a, as_stream(stream)) : launch_split<false, true, true>(a, as_stream(stream));
        return stats ? launch_split<true, true, false>(a, as_stream(stream)) : launch_split<false, true, false>(a, as_stream(stream));
    }
    if (wq_f16) return stats ? launch_split<true, false, true>(a, as_stream(stream)) : launch_split<false, false, true>(a, as_stream(stream));
    return stats ? launch_split<true, false, false>(a, as_stream(stream)) : launch_split<false, false, false>(a, as_stream(stream));
}


// ------------------------------------------------------------------ weight gradient, split operands
//   dW[co][ci][ky][kx] = sum_{b,y,x} dz[b][co][y][x] * X[b][ci][y+ky-1][x+kx-1]:  M = co, N = ci, K = pixels, one accumulator per
//   tap, v_mfma_f32_32x32x16_bf16 with the 8 K-values of a lane = 8 consecutive pixels of one channel row.  The row-streaming
//   structure (first used by round 2's bf16 weight gradient, DESIGN_HISTORY.md 4.2d): a unit is ONE image row of a 64-pixel strip -- dz [64 co][64 px],
//   x rows y-1, y, y+1 [64 ci][66 px] in a 4-slot ring, a block walks down its strip and loads one new row of each operand per
//   unit as whole 128-byte lines -- with both operands split on the way into LDS (hi | mid parts of dz and of x) and three MFMAs
//   per (16-pixel segment, tap): dz_mid x_hi + dz_hi x_mid + dz_hi x_hi.  110 KB of LDS: one block per CU, of EIGHT waves -- two
//   groups of four that split a unit's four segments between them and keep their own accumulators (a split over K inside the
//   block: 4 waves at one per SIMD had nothing to cover the barrier, the commit and the fragment reads, 0.33 of the bf16 peak);
//   54 MFMAs per wave and unit.  Split-K over (image, strip, row) units, raw slabs [split][tap][co][ci] reduced deterministically by
//   wgrad_reduce_kernel (conv_mfma.hip).  Requires W >= 64 with W % 4 == 0, or W = 32 / 16 (below).
struct SwArgs {
    const float* x;
    int64_t x_bs;
    const float* dz;
    int64_t dz_bs;
    float* slab;
    int B, Cin, Cout, H, W, ciTiles, coTiles, splitK, tilesX;
    const float* nsave;   // NORM: x is the pre-activation of the unit below, normalised on load (see conv3x3_split_kernel): its
    int nimg;             //       coefficients [groups <= 2][4][Cin], statistics groups of nimg consecutive images
    const unsigned* x_amax;    // F16: magnitude slots of x (overflow guard only; NULL: unscaled) and of dz (always scaled: amax
    const unsigned* dz_amax;   //      lands in [2^13, 2^14)); the slab store undoes both scales
};

// MFMA groups of a unit (6 with 64-channel tiles, 12 with 128) before the next unit's rows are committed
#ifndef SW_CAT64
#define SW_CAT64 2
#endif
#ifndef SW_CAT128
#define SW_CAT128 5
#endif
// Maps narrower than a strip (G = 64 / W = 2 or 4: the 32- and 16-pixel levels): a unit is row y of G IMAGES side by side -- the dz
// row is their 64 pixels back to back, the x row keeps each image's own halo (sub-row pitch SR_P dwords: W / 2 data dwords + the
// right-halo dword, rounded to 16 bytes), so a horizontal shift never reads a neighbouring image's pixel.  Everything else -- the
// ring over rows, the two pixel groups, the commit inside the MFMA block -- is the 64-pixel strip's.
constexpr int SR_SDZ = 36;                                  // dword stride per co
template <int G> struct SwCfg {
    static constexpr int SPG = 8 / G;                       // 8-pixel segments per sub-row
    static constexpr int P = G == 1 ? 36 : (G == 2 ? 20 : 12);
    static constexpr int SLOT = G * P;                      // 36 / 40 / 48 dwords per ring slot
    static constexpr int SX = 4 * SLOT + 4;                 // per ci: 148 / 164 / 196 = 4 x odd
    static constexpr int X_PART = 64 * SX;
    static constexpr int GOFF = G == 1 ? 16 : (G == 2 ? 20 : 24);   // fragment offset of pixel group 1 / of segment 1
    static constexpr int SOFF = G == 4 ? 12 : 8;
};

// COT = output channels per block.  64: the 8 waves are two K-groups x (2 x 2) quadrants, group g takes the 16-pixel segments 2g,
// 2g + 1 of a unit and writes its own slab (54 MFMAs per wave between barriers).  128 (Cout % 128 == 0, G < 4: the LDS holds
// it): 4 x 2 quadrants, every wave takes all four segments -- 108 MFMAs per wave between barriers, the x rows staged once per
// 128 output channels, one slab per block.
// F16: fp16 parts (22 significant bits per operand instead of bf16's 16) of 2^kx x and 2^kd dz -- see amax_scale
template <int G, int COT, bool NORM, bool F16>
__global__ __launch_bounds__(512, 2) void conv3x3_split_wgrad_kernel(SwArgs a) {
    using C = SwCfg<G>;
    constexpr int SR_SX = C::SX, SR_SLOT = C::SLOT, SR_X_PART = C::X_PART;
    constexpr int SR_DZ_PART = COT * SR_SDZ;
    constexpr int NSEG = COT == 64 ? 2 : 4;                  // 16-pixel segments per wave and unit
    constexpr int DZI = COT / 64;                            // dz staging items (8 pixels of one channel) per thread
    __shared__ __attribute__((aligned(16))) unsigned dz_lds[2 * 2 * SR_DZ_PART];     // [buf][part][COT co][SR_SDZ]
    __shared__ __attribute__((aligned(16))) unsigned x_lds[2 * SR_X_PART];           // [part][64 ci][SR_SX]

    int bid;
    {
        const int n = gridDim.x, q = n >> 3, r = n & 7, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tiles = a.ciTiles * a.coTiles;
    const int ks = bid / tiles, tile = bid % tiles;
    const int ci0 = (tile % a.ciTiles) * 64, co0 = (tile / a.ciTiles) * COT;
    const int nunits = (a.B / G) * a.tilesX * a.H;             // unit = (image [group of G], 64-pixel strip, row), rows fastest
    const int per = (nunits + a.splitK - 1) / a.splitK;
    const int u0 = ks * per, u1 = min(u0 + per, nunits);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int grp = COT == 64 ? wid >> 2 : 0, wm = COT == 64 ? (wid >> 1) & 1 : wid >> 1, wn = wid & 1;
    const int l31 = lane & 31, kh = lane >> 5;
    const int HW = a.H * a.W;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float x_inv = 1.f, dz_inv = 1.f;
    const float x_scale = F16 ? amax_scale(amax_read(a.x_amax), false, x_inv) : 1.f;
    const float dz_scale = F16 ? amax_scale(amax_read(a.dz_amax), true, dz_inv) : 1.f;

    // staging roles: thread = (channel = tid / 8, 8-pixel segment = tid % 8) of the dz row and of the x row
    const int st_c = tid >> 3, st_s = tid & 7;
    const int st_sub = st_s / C::SPG, st_w = st_s % C::SPG;     // image of the group, 8-pixel segment of its row
    u32x4s dzv[2 * DZI], xq[2];
    float xh[2];
    // NORM: BatchNorm coefficients of this thread's x channel for the (at most two) statistics groups; the group and the
    // validity of the row whose loads are in flight
    float n_mean[2] = {0.f, 0.f}, n_sc[2] = {0.f, 0.f}, n_sh[2] = {0.f, 0.f};
    bool pend_v[4] = {false, false, false, false}, pend_g1 = false;      // left halo | pixels 0-3 | pixels 4-7 | right halo
    if constexpr (NORM) {
        const int ch = min(ci0 + st_c, a.Cin - 1), last = a.B / a.nimg - 1;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const float* sv = a.nsave + (int64_t)min(g, last) * 4 * a.Cin + ch;
            n_mean[g] = sv[0];
            n_sc[g] = sv[2 * a.Cin];
            n_sh[g] = sv[3 * a.Cin];
        }
    }
    // loads of one dz row and / or one x row of strip (b, x0): row < 0 or >= H -> zeros.  G == 1: one buffer resource per image
    // (a 256 x 256 level's batch exceeds the 2 GiB range); G > 1: the lanes of a wave address different images, so the resource
    // spans the whole (small) tensor and the image offset goes into the lane's byte offset (range checked by the host)
    auto issue_dz = [&](int b, int x0, int y, int p0 = 0, int p1 = 64) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t dr = G == 1 ? s_rsrc(a.dz + (int64_t)b * a.dz_bs, (int64_t)a.Cout * HW * 4)
                                                 : s_rsrc(a.dz, ((int64_t)(a.B - 1) * a.dz_bs + (int64_t)a.Cout * HW) * 4);
        const int xs = x0 + 8 * st_w;
#pragma unroll
        for (int i = 0; i < DZI; ++i) {
            const int co = co0 + st_c + 64 * i;
            const bool ok = y >= 0 && y < a.H && co < a.Cout;
            const unsigned base = (unsigned)((co * HW + y * a.W + xs) * 4) + (G == 1 ? 0u : (unsigned)((int64_t)(b * G + st_sub) * a.dz_bs * 4));
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (2 * i + k >= p0 && 2 * i + k < p1)
                    dzv[2 * i + k] = __builtin_amdgcn_raw_buffer_load_b128(dr, (ok && xs + 4 * k < a.W) ? base + 16 * k : OOB_S, 0, 0);
        }
    };
    auto issue_x = [&](int b, int x0, int y, int p0 = 0, int p1 = 64) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t xr = G == 1 ? s_rsrc(a.x + (int64_t)b * a.x_bs, (int64_t)a.Cin * HW * 4)
                                                 : s_rsrc(a.x, ((int64_t)(a.B - 1) * a.x_bs + (int64_t)a.Cin * HW) * 4);
        const int xs = x0 + 8 * st_w;
        const bool ok = y >= 0 && y < a.H && ci0 + st_c < a.Cin;
        const unsigned base = (unsigned)(((ci0 + st_c) * HW + y * a.W + xs) * 4) + (G == 1 ? 0u : (unsigned)((int64_t)(b * G + st_sub) * a.x_bs * 4));
        if (NORM && p0 == 0) {
            pend_v[0] = ok && xs > 0 && xs - 1 < a.W;
            pend_v[1] = ok && xs < a.W;
            pend_v[2] = ok && xs + 4 < a.W;
            pend_v[3] = ok && st_w == C::SPG - 1 && xs + 8 < a.W;
            pend_g1 = (G == 1 ? b : b * G + st_sub) >= a.nimg;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (k >= p0 && k < p1) xq[k] = __builtin_amdgcn_raw_buffer_load_b128(xr, (ok && xs + 4 * k < a.W) ? base + 16 * k : OOB_S, 0, 0);
        if (2 >= p0 && 2 < p1)
            xh[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, (ok && xs > 0 && xs - 1 < a.W) ? base - 4 : OOB_S, 0, 0));
        if (3 >= p0 && 3 < p1)
            xh[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, (ok && st_w == C::SPG - 1 && xs + 8 < a.W) ? base + 32 : OOB_S, 0, 0));
    };
    auto commit_dz = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < DZI; ++i) {
            unsigned* d = dz_lds + buf * 2 * SR_DZ_PART + (st_c + 64 * i) * SR_SDZ + st_s * 4;
            // (the whole vector is re-typed, then indexed: element-wise bit casts of the loaded vector's lanes came out as four
            // copies of lane 0 with hipcc 7.2 -- found with delta-function inputs)
            const f32x4s lo = __builtin_bit_cast(f32x4s, dzv[2 * i]), hv = __builtin_bit_cast(f32x4s, dzv[2 * i + 1]);
            const float f[8] = {lo[0], lo[1], lo[2], lo[3], hv[0], hv[1], hv[2], hv[3]};
            u32x4s hi, mid;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                unsigned hh, mm;
                if constexpr (F16) split2h_s(f[2 * c], f[2 * c + 1], dz_scale, hh, mm);
                else split2(f[2 * c], f[2 * c + 1], hh, mm);
                hi[c] = hh;
                mid[c] = mm;
            }
            *reinterpret_cast<u32x4s*>(d) = hi;
            *reinterpret_cast<u32x4s*>(d + SR_DZ_PART) = mid;
        }
    };
    // row element e = column x0 - 1 + e; dword p = (e[2p], e[2p+1]) = (f[2p-1], f[2p]) of the interior row f: an 8-pixel segment's
    // four dwords start with (pixel left of the segment, its first pixel); its last pixel opens the next segment's first dword
    // (loaded there as that segment's left neighbour); the strip's last dword (f[63], right halo) is segment 7's
    auto commit_x = [&](int slot) __attribute__((always_inline)) {
        unsigned* row = x_lds + st_c * SR_SX + slot * SR_SLOT + st_sub * C::P + st_w * 4;
        const f32x4s f0 = __builtin_bit_cast(f32x4s, xq[0]), f1 = __builtin_bit_cast(f32x4s, xq[1]);
        float e[10] = {xh[0], f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3], xh[1]};
        if constexpr (NORM) {
            // the activation bn_relu_apply_kernel would have written, element for element; what lies outside the image (rows
            // above / below, the strip's halo columns at the image border) is the convolution's zero padding and stays zero
            const float mean = pend_g1 ? n_mean[1] : n_mean[0], sc = pend_g1 ? n_sc[1] : n_sc[0], sh = pend_g1 ? n_sh[1] : n_sh[0];
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                const float v = fmaxf(fmaf(e[k] - mean, sc, sh), 0.f);
                e[k] = pend_v[k == 0 ? 0 : (k <= 4 ? 1 : (k <= 8 ? 2 : 3))] ? v : 0.f;
            }
        }
        unsigned hi[5], mid[5];
#pragma unroll
        for (int p2 = 0; p2 < 5; ++p2) {
            if constexpr (F16) split2h_s(e[2 * p2], e[2 * p2 + 1], x_scale, hi[p2], mid[p2]);
            else split2(e[2 * p2], e[2 * p2 + 1], hi[p2], mid[p2]);
        }
        *reinterpret_cast<u32x4s*>(row) = u32x4s{hi[0], hi[1], hi[2], hi[3]};
        *reinterpret_cast<u32x4s*>(row + SR_X_PART) = u32x4s{mid[0], mid[1], mid[2], mid[3]};
        if (st_w == C::SPG - 1) {
            row[4] = hi[4];
            row[SR_X_PART + 4] = mid[4];
        }
    };

    const unsigned* a_ptr = dz_lds + (wm * 32 + l31) * SR_SDZ + kh * 4 + grp * 16;
    const unsigned* b_ptr = x_lds + (wn * 32 + l31) * SR_SX + kh * 4 + grp * C::GOFF;
    // x-row dword offset of segment s of a wave: pixel groups 2 s, 2 s + 1 -> sub-row (2 s) / SPG, position (2 s) % SPG
    auto seg_off = [](int sg) { return G == 1 ? sg * 8 : (G == 2 ? (sg >> 1) * 20 + (sg & 1) * 8 : sg * 12); };

    int u = u0;
    while (u < u1) {
        // a run of rows inside one strip: y = yb .. ye - 1
        const int yb = u % a.H, sb = u / a.H;
        const int tx = sb % a.tilesX, b = sb / a.tilesX;
        const int x0 = G == 1 ? tx * 64 : 0;
        const int ye = min(a.H, yb + (u1 - u));
        // run prologue: rows yb - 1 and yb of x into the ring (synchronously), then the dz row and the next x row of the first unit
        __syncthreads();                              // every wave is done with the previous run's ring and dz buffers
        issue_x(b, x0, yb - 1);
        commit_x((yb - 1) & 3);
        issue_x(b, x0, yb);
        commit_x(yb & 3);
        issue_dz(b, x0, yb);
        issue_x(b, x0, yb + 1);
        commit_dz(yb & 1);
        commit_x((yb + 1) & 3);
        __syncthreads();
        if (yb + 1 < ye) {
            issue_dz(b, x0, yb + 1);
            issue_x(b, x0, yb + 2);
        }
        __builtin_amdgcn_sched_barrier(0);
        for (int y = yb; y < ye; ++y) {
            // unit y: MFMAs on dz buffer y & 1 and ring rows y - 1 .. y + 1.  Staging rides inside the MFMA block: the rows of unit
            // y + 1 (in registers since the previous unit) are split and committed after the first MFMA group -- into the other dz
            // buffer and the ring slot of row y - 2, both last read in unit y - 1 -- and the loads of unit y + 2's rows then go
            // out ONE PER MFMA GROUP.  (Issued as a burst behind the barrier, by all eight waves at once, they held every wave's
            // issue port while the matrix pipe idled: a timing build without them ran 16 % faster, one without the barrier 4 %.)
            const int buf = y & 1;
            const bool more = y + 1 < ye, more2 = y + 2 < ye;
            const unsigned* ab = a_ptr + buf * 2 * SR_DZ_PART;
            u32x4s ah, am;
            constexpr int CAT = COT == 64 ? SW_CAT64 : SW_CAT128;
            // Segments run LAST TO FIRST: a lane's 8-pixel fragment is one aligned ds_read_b128 (4 dwords); the fifth dword its
            // shifted copies need is the first dword of the NEXT 8-pixel group -- for the kh = 0 half that of its kh = 1 partner in
            // the same read, for the kh = 1 half that of the kh = 0 lane of the segment read one group earlier: both arrive by
            // v_permlane32_swap instead of a ds_read_b32, which on a row stride that is a multiple of 4 dwords is a 4-way bank
            // conflict (20 l mod 32 repeats every 8 lanes; SQ_LDS_BANK_CONFLICT was 63 % of the kernel's LDS cycles, the LDS 70 %
            // busy).  Only where the next group is another image's or the strip's halo does the dword still come from LDS.
            unsigned lo_prev[3][2] = {};
#pragma unroll
            for (int gk = 0; gk < 3 * NSEG; ++gk) {
                const int sg = NSEG - 1 - gk / 3, ky = gk % 3;
                const bool from_prev = (sg + 1 < NSEG) && (G == 1 || (G == 2 && (sg & 1) == 0));
                if (gk == CAT && more) {
                    commit_dz(buf ^ 1);
                    commit_x((y + 2) & 3);
                }
                if (more2) {
                    // pieces: 2 DZI dz loads, then the four x loads; groups CAT + 1 ... (two or more pieces each where
                    // they do not fit one per group)
                    constexpr int NP = 2 * DZI + 4, NG = 3 * NSEG - CAT - 1, PER = (NP + NG - 1) / NG;
                    const int g0 = gk - CAT - 1;
                    if (g0 >= 0) {
#pragma unroll
                        for (int q = 0; q < PER; ++q) {
                            const int pc = g0 * PER + q;
                            if (pc < 2 * DZI) issue_dz(b, x0, y + 2, pc, pc + 1);
                            else if (pc < NP) issue_x(b, x0, y + 3, pc - 2 * DZI, pc - 2 * DZI + 1);
                        }
                    }
                }
                if (ky == 0) {
                    ah = *reinterpret_cast<const u32x4s*>(ab + sg * 8);
                    am = *reinterpret_cast<const u32x4s*>(ab + SR_DZ_PART + sg * 8);
                }
                const unsigned* br = b_ptr + ((y - 1 + ky) & 3) * SR_SLOT + seg_off(sg);
                u32x4s sh[2][3];                  // [part][horizontal shift]
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    const u32x4s q = *reinterpret_cast<const u32x4s*>(br + pt * SR_X_PART);
                    unsigned d4;
                    if (G == 4) {
                        d4 = br[pt * SR_X_PART + 4];
                    } else {
                        const auto sw = __builtin_amdgcn_permlane32_swap(q[0], q[0], false, false);   // [0]: the kh = 0 value in
                        const unsigned nxt = from_prev ? lo_prev[ky][pt] : br[pt * SR_X_PART + 4];       // both halves, [1]: kh = 1's
                        d4 = kh ? nxt : sw[1];
                        lo_prev[ky][pt] = sw[0];
                    }
                    sh[pt][0] = q;
                    sh[pt][1] = u32x4s{__builtin_amdgcn_alignbit(q[1], q[0], 16), __builtin_amdgcn_alignbit(q[2], q[1], 16),
                                       __builtin_amdgcn_alignbit(q[3], q[2], 16), __builtin_amdgcn_alignbit(d4, q[3], 16)};
                    sh[pt][2] = u32x4s{q[1], q[2], q[3], d4};
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if constexpr (F16) {
                        const f16x8 a_h = __builtin_bit_cast(f16x8, ah), a_m = __builtin_bit_cast(f16x8, am);
                        const f16x8 bh = __builtin_bit_cast(f16x8, sh[0][j]), bm = __builtin_bit_cast(f16x8, sh[1][j]);
                        acc[ky * 3 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_m, bh, acc[ky * 3 + j], 0, 0, 0);
                        acc[ky * 3 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, bm, acc[ky * 3 + j], 0, 0, 0);
                        acc[ky * 3 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, bh, acc[ky * 3 + j], 0, 0, 0);
                    } else {
                        const bf16x8 a_h = __builtin_bit_cast(bf16x8, ah), a_m = __builtin_bit_cast(bf16x8, am);
                        const bf16x8 bh = __builtin_bit_cast(bf16x8, sh[0][j]), bm = __builtin_bit_cast(bf16x8, sh[1][j]);
                        acc[ky * 3 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_m, bh, acc[ky * 3 + j], 0, 0, 0);
                        acc[ky * 3 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, bm, acc[ky * 3 + j], 0, 0, 0);
                        acc[ky * 3 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, bh, acc[ky * 3 + j], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
        u += ye - yb;
    }

    const int64_t n = (int64_t)a.Cout * a.Cin;
    const int ci = ci0 + wn * 32 + l31;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float* o = a.slab + ((int64_t)(COT == 64 ? ks * 2 + grp : ks) * 9 + t) * n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (co < a.Cout && ci < a.Cin) o[(int64_t)co * a.Cin + ci] = F16 ? acc[t][r] * (x_inv * dz_inv) : acc[t][r];
        }
    }
}

// ------------------------------------------------------------------ weight gradient from PRE-SPLIT operands (round 4)
// Both operands arrive in the slot layout [B][C/8][H][part][W][8] (16 bytes = 8 channels of one pixel), i.e. with the reduction
// index (pixels) running ACROSS slots.  The MFMA wants 8 consecutive pixels of one channel per lane, so the fragments are read
// with the gfx950 transposing LDS read ds_read_b64_tr_b16 (a 16-lane group fetches 4 pixel rows x 16 channels and every lane
// receives ITS channel's four pixels): no in-kernel split, no transposition VALU, and a tap's horizontal shift is a pixel-row
// offset of the read address (the round-3 kernel rebuilt shifted fragments with v_alignbit + v_permlane32_swap).  Staging is a
// lane-linear LDS-DMA copy as in conv3x3_split_pre_kernel.  Unit, ring over rows, split-K and slabs as conv3x3_split_wgrad_kernel.
//   x ring : [4 row slots][part][8 channel groups][PXP = 68 pixel slots]   (G = 1: x0-1 .. x0+64 + 2 pad; G = 2: two images' 34)
//   dz     : [2 buffers] [part][COT / 8 groups]  [PXP = 68]                (64 pixels + 4 pad)
// A channel group's plane of 68 slots = 272 dwords = 16 banks (mod 64): the 4 pixel rows x (2 groups x 2 half-slots) a 32-lane
// half reads fall on 64 different banks -- conflict-free without a swizzle.
typedef short s16x4w __attribute__((ext_vector_type(4)));
#ifndef SWP_SHARE_B
#define SWP_SHARE_B 1
#endif
struct SwPreArgs {
    const void* xs;
    int64_t xs_bs;        // batch strides in 4-byte units
    const void* dzs;
    int64_t dzs_bs;
    float* slab;
    int B, Cin, Cout, H, W, ciTiles, coTiles, splitK, tilesX;
    const unsigned* x_slots;    // magnitude slots the producers scaled the operands by (NULL: unscaled); x: guard rule, dz: always
    const unsigned* dz_slots;
    const unsigned* x_slots2;   // concat buffer: input channels >= split_ch (a multiple of 64) were scaled by these (NULL: unscaled)
    int split_ch;
};
constexpr int SWP_PXP = 68;          // G = 4 (16-pixel maps: four images of 18 slots each): 76 (304 dwords = 48 banks mod 64: conflict-free too)

__device__ __forceinline__ u32x4s swp_frag(unsigned addr) {      // 8 consecutive pixels (K) of this lane's channel: two transposed reads
    const s16x4w lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4w*)(uintptr_t)addr);
    const s16x4w hv = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4w*)(uintptr_t)(addr + 64));
    const unsigned long long a = __builtin_bit_cast(unsigned long long, lo), b = __builtin_bit_cast(unsigned long long, hv);
    return u32x4s{(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
}

// PM: 0 bf16 (hi | mid), 1 fp16 (hi | mid) -- three MFMAs per term; 2: PLAIN bf16 operands, one part (BASELINE configs[2]), one MFMA
template <int G, int COT, int PM>
__global__ __launch_bounds__(512, 2) void conv3x3_split_wgrad_pre_kernel(SwPreArgs a) {
    constexpr bool F16 = PM == 1;
    constexpr int NP = PM == 2 ? 1 : 2;                        // parts per operand
    constexpr int PXP = G == 4 ? 76 : SWP_PXP;
    constexpr int XS = 8, DS = COT / 8;                        // channel groups per image
    // (row images padded to whole 64-slot DMA pieces: the lanes of a piece beyond the image write zeros, which must not land in the
    // next ring slot / buffer)
    constexpr int X_PART = XS * PXP, X_ROW = (NP * X_PART + 63) / 64 * 64;      // slots
    constexpr int DZ_PART = DS * PXP, DZ_BUF = (NP * DZ_PART + 63) / 64 * 64;
    constexpr int NXI = (X_ROW + 511) / 512, NDI = (DZ_BUF + 511) / 512;      // DMA rounds per wave (64 slots each, 8 waves)
    constexpr int NKS = COT == 64 ? 2 : 4;                     // k-steps (16 pixels) per wave and unit
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_w;
    const unsigned x_base = lds0, dz_base = lds0 + 4 * X_ROW * 16;

    int bid;
    {
        const int n = gridDim.x, q = n >> 3, r = n & 7, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tiles = a.ciTiles * a.coTiles;
    const int ks = bid / tiles, tile = bid % tiles;
    const int ci0 = (tile % a.ciTiles) * 64, co0 = (tile / a.ciTiles) * COT;
    const int nunits = (a.B / G) * a.tilesX * a.H;
    const int per = (nunits + a.splitK - 1) / a.splitK;
    const int u0 = ks * per, u1 = min(u0 + per, nunits);

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = COT == 64 ? wid >> 2 : 0, wm = COT == 64 ? (wid >> 1) & 1 : wid >> 1, wn = wid & 1;
    const int l31 = lane & 31, kh = lane >> 5;
    const int HW = a.H * a.W;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // transposed-read lane bases (bytes): lane = 16 g + 4 q + p supplies pixel row q, channels 4 p .. 4 p + 3 of the channel half g & 1
    const int tq = (lane >> 2) & 3, tp = lane & 3, tg = (lane >> 4) & 1;
    const unsigned a_lane = (unsigned)((((wm * 4 + 2 * tg + (tp >> 1)) * PXP + 8 * kh + tq) * 16) + (tp & 1) * 8);
    const unsigned b_lane = (unsigned)((((wn * 4 + 2 * tg + (tp >> 1)) * PXP + 8 * kh + tq) * 16) + (tp & 1) * 8);

    // ---- staging: slot position i = (wid + 8 k) * 64 + lane of a row image [part][group][PXP]
    i32x4s xr, dr;
    unsigned x_off[NXI], d_off[NDI];
    auto setup_strip = [&](int b, int x0) __attribute__((always_inline)) {
        if (G == 1) {
            xr = sp_rsrc4(reinterpret_cast<const unsigned*>(a.xs) + (int64_t)b * a.xs_bs, (int64_t)a.Cin * HW * 2 * NP);
            dr = sp_rsrc4(reinterpret_cast<const unsigned*>(a.dzs) + (int64_t)b * a.dzs_bs, (int64_t)a.Cout * HW * 2 * NP);
        } else {
            xr = sp_rsrc4(a.xs, (int64_t)(a.B - 1) * a.xs_bs * 4 + (int64_t)a.Cin * HW * 2 * NP);
            dr = sp_rsrc4(a.dzs, (int64_t)(a.B - 1) * a.dzs_bs * 4 + (int64_t)a.Cout * HW * 2 * NP);
        }
#pragma unroll
        for (int k = 0; k < NXI; ++k) {
            const int i = (wid + 8 * k) * 64 + lane;
            const int part = i / X_PART, s = (i % X_PART) / PXP, pi = i % PXP;
            const int img = G == 1 ? 0 : pi / (a.W + 2), xx = G == 1 ? x0 - 1 + pi : pi % (a.W + 2) - 1;
            const bool ok = i < NP * X_PART && pi < (G == 1 ? 66 : G * (a.W + 2)) && xx >= 0 && xx < a.W && ci0 + 8 * s < a.Cin;
            const int64_t img_off = G == 1 ? 0 : (int64_t)(b * G + img) * a.xs_bs * 4;
            x_off[k] = ok ? (unsigned)(img_off + ((int64_t)((ci0 / 8 + s) * a.H) * NP + part) * a.W * 16 + xx * 16) : OOB_S;
        }
#pragma unroll
        for (int k = 0; k < NDI; ++k) {
            const int i = (wid + 8 * k) * 64 + lane;
            const int part = i / DZ_PART, s = (i % DZ_PART) / PXP, pi = i % PXP;
            const int img = G == 1 ? 0 : pi / a.W, xx = G == 1 ? x0 + pi : pi % a.W;
            const bool ok = i < NP * DZ_PART && pi < 64 && xx < a.W && co0 + 8 * s < a.Cout;
            const int64_t img_off = G == 1 ? 0 : (int64_t)(b * G + img) * a.dzs_bs * 4;
            d_off[k] = ok ? (unsigned)(img_off + ((int64_t)((co0 / 8 + s) * a.H) * NP + part) * a.W * 16 + xx * 16) : OOB_S;
        }
    };
    // row y of the strip into ring slot / dz buffer; rows outside the image are zeros (every lane out of range)
    auto dma_x = [&](int y, int k) __attribute__((always_inline)) {
        if ((wid + 8 * k) * 64 < X_ROW) {
            const bool ok = y >= 0 && y < a.H;
            sp_dma16(xr, x_base + (unsigned)(((y & 3) * X_ROW + (wid + 8 * k) * 64) * 16), ok ? x_off[k] : OOB_S, ok ? (unsigned)(y * NP * a.W * 16) : 0u);
        }
    };
    auto dma_dz = [&](int y, int k) __attribute__((always_inline)) {
        if ((wid + 8 * k) * 64 < DZ_BUF) {
            const bool ok = y >= 0 && y < a.H;
            sp_dma16(dr, dz_base + (unsigned)(((y & 1) * DZ_BUF + (wid + 8 * k) * 64) * 16), ok ? d_off[k] : OOB_S, ok ? (unsigned)(y * NP * a.W * 16) : 0u);
        }
    };

    int u = u0;
    while (u < u1) {
        const int yb = u % a.H, sb = u / a.H;
        const int tx = sb % a.tilesX, b = sb / a.tilesX;
        const int x0 = G == 1 ? tx * 64 : 0;
        const int ye = min(a.H, yb + (u1 - u));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                              // every wave is done with the previous run's ring and dz buffers
        setup_strip(b, x0);
#pragma unroll
        for (int k = 0; k < NXI; ++k) {
            dma_x(yb - 1, k);
            dma_x(yb, k);
            dma_x(yb + 1, k);
        }
#pragma unroll
        for (int k = 0; k < NDI; ++k) dma_dz(yb, k);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int y = yb; y < ye; ++y) {
            // unit y: MFMAs on dz buffer y & 1 and ring rows y - 1 .. y + 1; the rows of unit y + 1 (dz row y + 1, x row y + 2) go
            // by DMA into the other dz buffer and the ring slot of row y - 2 (both last read in unit y - 1), spread over the k-steps
            const bool more = y + 1 < ye;
            const unsigned ab = dz_base + (unsigned)((y & 1) * DZ_BUF * 16) + a_lane;
            unsigned bb[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) bb[ky] = x_base + (unsigned)((((y - 1 + ky) & 3) * X_ROW) * 16) + b_lane;
#pragma unroll
            for (int kk = 0; kk < NKS; ++kk) {
                const int kst = grp * 2 + kk;                                  // 16-pixel k-step of the unit
                const int dpx = 16 * kst;                                      // dz pixel slot
                const int xpx = G == 1 ? 16 * kst : (G == 2 ? (kst >> 1) * 34 + (kst & 1) * 16 : kst * 18);      // x pixel slot of tap kx = 0
                if (more) {
                    constexpr int NP = NXI + NDI, PER = (NP + NKS - 1) / NKS;
#pragma unroll
                    for (int q = 0; q < PER; ++q) {
                        const int pc = kk * PER + q;
                        if (pc < NDI) dma_dz(y + 1, pc);
                        else if (pc < NP) dma_x(y + 2, pc - NDI);
                    }
                }
                const u32x4s Ah = swp_frag(ab + dpx * 16), Am = NP == 2 ? swp_frag(ab + (DZ_PART + dpx) * 16) : Ah;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
#if SWP_SHARE_B
                    // the three horizontal taps of a row read pixels 8 kh + kx .. + 7: ONE set of three transposed reads (12 pixels, six
                    // dwords of two pixels each) serves all three -- kx = 0: dwords 0-3, kx = 2: dwords 1-4, kx = 1: the 16-bit-shifted
                    // pairs (v_alignbit) -- instead of two reads per tap and part (40 -> 22 LDS reads per 27 MFMAs)
                    u32x4s Bs[2][3];
#pragma unroll
                    for (int pt = 0; pt < NP; ++pt) {
                        const unsigned ba = bb[ky] + (pt * X_PART + xpx) * 16;
                        const u32x4s lo = swp_frag(ba);                                       // pixels 0 .. 7 of the lane's window
                        const s16x4w r2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4w*)(uintptr_t)(ba + 128));
                        const unsigned d4 = (unsigned)__builtin_bit_cast(unsigned long long, r2);   // pixels 8, 9
                        Bs[pt][0] = lo;
                        Bs[pt][1] = u32x4s{__builtin_amdgcn_alignbit(lo[1], lo[0], 16), __builtin_amdgcn_alignbit(lo[2], lo[1], 16),
                                           __builtin_amdgcn_alignbit(lo[3], lo[2], 16), __builtin_amdgcn_alignbit(d4, lo[3], 16)};
                        Bs[pt][2] = u32x4s{lo[1], lo[2], lo[3], d4};
                    }
#endif
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
#if SWP_SHARE_B
                        const u32x4s Bh = Bs[0][kx], Bm = NP == 2 ? Bs[1][kx] : Bs[0][kx];
#else
                        const u32x4s Bh = swp_frag(bb[ky] + (xpx + kx) * 16), Bm = NP == 2 ? swp_frag(bb[ky] + (X_PART + xpx + kx) * 16) : Bh;
#endif
                        const int t = ky * 3 + kx;
                        if constexpr (PM == 2) {
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Ah), __builtin_bit_cast(bf16x8, Bh), acc[t], 0, 0, 0);
                        } else if constexpr (F16) {
                            const f16x8 ah = __builtin_bit_cast(f16x8, Ah), am = __builtin_bit_cast(f16x8, Am);
                            const f16x8 bh = __builtin_bit_cast(f16x8, Bh), bm = __builtin_bit_cast(f16x8, Bm);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(am, bh, acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bm, acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
                        } else {
                            const bf16x8 ah = __builtin_bit_cast(bf16x8, Ah), am = __builtin_bit_cast(bf16x8, Am);
                            const bf16x8 bh = __builtin_bit_cast(bf16x8, Bh), bm = __builtin_bit_cast(bf16x8, Bm);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
                        }
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        u += ye - yb;
    }

    float x_inv = 1.f, dz_inv = 1.f;
    const unsigned* xsl = (a.split_ch && ci0 >= a.split_ch) ? a.x_slots2 : a.x_slots;      // this block's 64 input channels
    (void)amax_scale(amax_read(xsl), false, x_inv);
    (void)amax_scale(amax_read(a.dz_slots), true, dz_inv);
    const float out_scale = (xsl ? x_inv : 1.f) * (a.dz_slots ? dz_inv : 1.f);
    const int64_t n = (int64_t)a.Cout * a.Cin;
    const int ci = ci0 + wn * 32 + l31;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float* o = a.slab + ((int64_t)(COT == 64 ? ks * 2 + grp : ks) * 9 + t) * n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (co < a.Cout && ci < a.Cin) o[(int64_t)co * a.Cin + ci] = acc[t][r] * out_scale;
        }
    }
}

// Round 5: the same kernel on v_mfma_f32_16x16x32 (profiles/r05_mfma_shape_ab.md: the shape the chip clocks higher on).  K = pixels here,
// so K = 32 is natural -- one instruction takes 32 consecutive pixels of a unit's row (two of the 32x32x16 form's 16-pixel k-steps),
// no K-packing, the same LDS image, DMA schedule and fragment bytes.  A wave's 32 x 32 channel tile per tap is four 16 x 16
// accumulators; M = input channels (A = x fragment), N = output channels (B = dz fragment), so a lane owns four CONSECUTIVE input
// channels of one output channel and the slab leaves as float4 stores.  Lane group lg = lane / 16 takes pixels 8 lg .. 8 lg + 7 of the
// step (G = 4, four 16-pixel images per unit: lane groups 0-1 / 2-3 sit in two different images' sub-rows).
template <int G, int COT, int PM>
__global__ __launch_bounds__(512, 2) void conv3x3_wgrad_pre16_kernel(SwPreArgs a) {
    constexpr bool F16 = PM == 1;
    constexpr int NP = PM == 2 ? 1 : 2;                        // parts per operand
    constexpr int PXP = G == 4 ? 76 : SWP_PXP;
    constexpr int XS = 8, DS = COT / 8;                        // channel groups per image
    // (row images padded to whole 64-slot DMA pieces: the lanes of a piece beyond the image write zeros, which must not land in the
    // next ring slot / buffer)
    constexpr int X_PART = XS * PXP, X_ROW = (NP * X_PART + 63) / 64 * 64;      // slots
    constexpr int DZ_PART = DS * PXP, DZ_BUF = (NP * DZ_PART + 63) / 64 * 64;
    constexpr int NXI = (X_ROW + 511) / 512, NDI = (DZ_BUF + 511) / 512;      // DMA rounds per wave (64 slots each, 8 waves)
    constexpr int NKS = COT == 64 ? 1 : 2;                     // k-steps (32 pixels) per wave and unit
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_w;
    const unsigned x_base = lds0, dz_base = lds0 + 4 * X_ROW * 16;

    int bid;
    {
        const int n = gridDim.x, q = n >> 3, r = n & 7, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tiles = a.ciTiles * a.coTiles;
    const int ks = bid / tiles, tile = bid % tiles;
    const int ci0 = (tile % a.ciTiles) * 64, co0 = (tile / a.ciTiles) * COT;
    const int nunits = (a.B / G) * a.tilesX * a.H;
    const int per = (nunits + a.splitK - 1) / a.splitK;
    const int u0 = ks * per, u1 = min(u0 + per, nunits);

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = COT == 64 ? wid >> 2 : 0, wm = COT == 64 ? (wid >> 1) & 1 : wid >> 1, wn = wid & 1;
    const int r16 = lane & 15, lg = lane >> 4;
    const int HW = a.H * a.W;

    f32x4s acc[9][2][2];                                        // [tap][input-channel half (M)][output-channel half (N)]
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i >> 1][i & 1] = f32x4s{0.f, 0.f, 0.f, 0.f};

    // transposed-read lane bases (bytes): lane = 16 lg + 4 q + p supplies pixel row q (of the group's 8 pixels), channels 4 p .. 4 p + 3
    // of the 16-channel half; the wave's second half sits two channel groups (2 PXP slots) further
    const int tq = (lane >> 2) & 3, tp = lane & 3;
    const int xg = G == 4 ? (lg >> 1) * 18 + 8 * (lg & 1) : 8 * lg;              // x pixel slot of the lane group inside a 32-pixel step
    const unsigned a_lane = (unsigned)((((wm * 4 + (tp >> 1)) * PXP + 8 * lg + tq) * 16) + (tp & 1) * 8);       // dz
    const unsigned b_lane = (unsigned)((((wn * 4 + (tp >> 1)) * PXP + xg + tq) * 16) + (tp & 1) * 8);           // x
    constexpr unsigned HALF = 2 * PXP * 16;

    // ---- staging: slot position i = (wid + 8 k) * 64 + lane of a row image [part][group][PXP]
    i32x4s xr, dr;
    unsigned x_off[NXI], d_off[NDI];
    auto setup_strip = [&](int b, int x0) __attribute__((always_inline)) {
        if (G == 1) {
            xr = sp_rsrc4(reinterpret_cast<const unsigned*>(a.xs) + (int64_t)b * a.xs_bs, (int64_t)a.Cin * HW * 2 * NP);
            dr = sp_rsrc4(reinterpret_cast<const unsigned*>(a.dzs) + (int64_t)b * a.dzs_bs, (int64_t)a.Cout * HW * 2 * NP);
        } else {
            xr = sp_rsrc4(a.xs, (int64_t)(a.B - 1) * a.xs_bs * 4 + (int64_t)a.Cin * HW * 2 * NP);
            dr = sp_rsrc4(a.dzs, (int64_t)(a.B - 1) * a.dzs_bs * 4 + (int64_t)a.Cout * HW * 2 * NP);
        }
#pragma unroll
        for (int k = 0; k < NXI; ++k) {
            const int i = (wid + 8 * k) * 64 + lane;
            const int part = i / X_PART, s = (i % X_PART) / PXP, pi = i % PXP;
            const int img = G == 1 ? 0 : pi / (a.W + 2), xx = G == 1 ? x0 - 1 + pi : pi % (a.W + 2) - 1;
            const bool ok = i < NP * X_PART && pi < (G == 1 ? 66 : G * (a.W + 2)) && xx >= 0 && xx < a.W && ci0 + 8 * s < a.Cin;
            const int64_t img_off = G == 1 ? 0 : (int64_t)(b * G + img) * a.xs_bs * 4;
            x_off[k] = ok ? (unsigned)(img_off + ((int64_t)((ci0 / 8 + s) * a.H) * NP + part) * a.W * 16 + xx * 16) : OOB_S;
        }
#pragma unroll
        for (int k = 0; k < NDI; ++k) {
            const int i = (wid + 8 * k) * 64 + lane;
            const int part = i / DZ_PART, s = (i % DZ_PART) / PXP, pi = i % PXP;
            const int img = G == 1 ? 0 : pi / a.W, xx = G == 1 ? x0 + pi : pi % a.W;
            const bool ok = i < NP * DZ_PART && pi < 64 && xx < a.W && co0 + 8 * s < a.Cout;
            const int64_t img_off = G == 1 ? 0 : (int64_t)(b * G + img) * a.dzs_bs * 4;
            d_off[k] = ok ? (unsigned)(img_off + ((int64_t)((co0 / 8 + s) * a.H) * NP + part) * a.W * 16 + xx * 16) : OOB_S;
        }
    };
    // row y of the strip into ring slot / dz buffer; rows outside the image are zeros (every lane out of range)
    auto dma_x = [&](int y, int k) __attribute__((always_inline)) {
        if ((wid + 8 * k) * 64 < X_ROW) {
            const bool ok = y >= 0 && y < a.H;
            sp_dma16(xr, x_base + (unsigned)(((y & 3) * X_ROW + (wid + 8 * k) * 64) * 16), ok ? x_off[k] : OOB_S, ok ? (unsigned)(y * NP * a.W * 16) : 0u);
        }
    };
    auto dma_dz = [&](int y, int k) __attribute__((always_inline)) {
        if ((wid + 8 * k) * 64 < DZ_BUF) {
            const bool ok = y >= 0 && y < a.H;
            sp_dma16(dr, dz_base + (unsigned)(((y & 1) * DZ_BUF + (wid + 8 * k) * 64) * 16), ok ? d_off[k] : OOB_S, ok ? (unsigned)(y * NP * a.W * 16) : 0u);
        }
    };

    int u = u0;
    while (u < u1) {
        const int yb = u % a.H, sb = u / a.H;
        const int tx = sb % a.tilesX, b = sb / a.tilesX;
        const int x0 = G == 1 ? tx * 64 : 0;
        const int ye = min(a.H, yb + (u1 - u));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                              // every wave is done with the previous run's ring and dz buffers
        setup_strip(b, x0);
#pragma unroll
        for (int k = 0; k < NXI; ++k) {
            dma_x(yb - 1, k);
            dma_x(yb, k);
            dma_x(yb + 1, k);
        }
#pragma unroll
        for (int k = 0; k < NDI; ++k) dma_dz(yb, k);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int y = yb; y < ye; ++y) {
            // unit y: MFMAs on dz buffer y & 1 and ring rows y - 1 .. y + 1; the rows of unit y + 1 (dz row y + 1, x row y + 2) go
            // by DMA into the other dz buffer and the ring slot of row y - 2 (both last read in unit y - 1), spread over the k-steps
            const bool more = y + 1 < ye;
            const unsigned ab = dz_base + (unsigned)((y & 1) * DZ_BUF * 16) + a_lane;
            unsigned bb[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) bb[ky] = x_base + (unsigned)((((y - 1 + ky) & 3) * X_ROW) * 16) + b_lane;
#pragma unroll
            for (int kk = 0; kk < NKS; ++kk) {
                const int kst = COT == 64 ? grp : kk;                          // 32-pixel k-step of the unit
                const int dpx = 32 * kst;                                      // dz pixel slot
                const int xpx = G == 1 ? 32 * kst : (G == 2 ? kst * 34 : kst * 36);      // x pixel slot of tap kx = 0 (G = 4: image pair)
                if (more) {
                    constexpr int NP = NXI + NDI, PER = (NP + NKS - 1) / NKS;
#pragma unroll
                    for (int q = 0; q < PER; ++q) {
                        const int pc = kk * PER + q;
                        if (pc < NDI) dma_dz(y + 1, pc);
                        else if (pc < NP) dma_x(y + 2, pc - NDI);
                    }
                }
                u32x4s Dh[2], Dm[2];                                           // dz fragments of the two output-channel halves
#pragma unroll
                for (int hd = 0; hd < 2; ++hd) {
                    Dh[hd] = swp_frag(ab + hd * HALF + dpx * 16);
                    Dm[hd] = NP == 2 ? swp_frag(ab + hd * HALF + (DZ_PART + dpx) * 16) : Dh[hd];
                }
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    // the three horizontal taps of a row read pixels 8 lg + kx .. + 7: ONE set of three transposed reads (12 pixels, six
                    // dwords of two pixels each) serves all three -- kx = 0: dwords 0-3, kx = 2: dwords 1-4, kx = 1: the 16-bit-shifted
                    // pairs (v_alignbit)
#pragma unroll
                    for (int hx = 0; hx < 2; ++hx) {
                        u32x4s Bs[2][3];
#pragma unroll
                        for (int pt = 0; pt < NP; ++pt) {
                            const unsigned ba = bb[ky] + hx * HALF + (pt * X_PART + xpx) * 16;
                            const u32x4s lo = swp_frag(ba);                                       // pixels 0 .. 7 of the lane's window
                            const s16x4w r2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4w*)(uintptr_t)(ba + 128));
                            const unsigned d4 = (unsigned)__builtin_bit_cast(unsigned long long, r2);   // pixels 8, 9
                            Bs[pt][0] = lo;
                            Bs[pt][1] = u32x4s{__builtin_amdgcn_alignbit(lo[1], lo[0], 16), __builtin_amdgcn_alignbit(lo[2], lo[1], 16),
                                               __builtin_amdgcn_alignbit(lo[3], lo[2], 16), __builtin_amdgcn_alignbit(d4, lo[3], 16)};
                            Bs[pt][2] = u32x4s{lo[1], lo[2], lo[3], d4};
                        }
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const u32x4s Xh = Bs[0][kx], Xm = NP == 2 ? Bs[1][kx] : Bs[0][kx];
                            const int t = ky * 3 + kx;
#pragma unroll
                            for (int hd = 0; hd < 2; ++hd) {
                                if constexpr (PM == 2) {
                                    acc[t][hx][hd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Xh), __builtin_bit_cast(bf16x8, Dh[hd]), acc[t][hx][hd], 0, 0, 0);
                                } else if constexpr (F16) {
                                    const f16x8 xh = __builtin_bit_cast(f16x8, Xh), xm = __builtin_bit_cast(f16x8, Xm);
                                    const f16x8 dh = __builtin_bit_cast(f16x8, Dh[hd]), dm = __builtin_bit_cast(f16x8, Dm[hd]);
                                    acc[t][hx][hd] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, dm, acc[t][hx][hd], 0, 0, 0);
                                    acc[t][hx][hd] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xm, dh, acc[t][hx][hd], 0, 0, 0);
                                    acc[t][hx][hd] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, dh, acc[t][hx][hd], 0, 0, 0);
                                } else {
                                    const bf16x8 xh = __builtin_bit_cast(bf16x8, Xh), xm = __builtin_bit_cast(bf16x8, Xm);
                                    const bf16x8 dh = __builtin_bit_cast(bf16x8, Dh[hd]), dm = __builtin_bit_cast(bf16x8, Dm[hd]);
                                    acc[t][hx][hd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, dm, acc[t][hx][hd], 0, 0, 0);
                                    acc[t][hx][hd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm, dh, acc[t][hx][hd], 0, 0, 0);
                                    acc[t][hx][hd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, dh, acc[t][hx][hd], 0, 0, 0);
                                }
                            }
                        }
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        u += ye - yb;
    }

    float x_inv = 1.f, dz_inv = 1.f;
    const unsigned* xsl = (a.split_ch && ci0 >= a.split_ch) ? a.x_slots2 : a.x_slots;      // this block's 64 input channels
    (void)amax_scale(amax_read(xsl), false, x_inv);
    (void)amax_scale(amax_read(a.dz_slots), true, dz_inv);
    const float out_scale = (xsl ? x_inv : 1.f) * (a.dz_slots ? dz_inv : 1.f);
    const int64_t n = (int64_t)a.Cout * a.Cin;
    // lane: output channel co0 + 32 wm + 16 hd + r16, input channels ci0 + 32 wn + 16 hx + 4 lg .. + 3: one float4 (Cin % 8 == 0)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float* o = a.slab + ((int64_t)(COT == 64 ? ks * 2 + grp : ks) * 9 + t) * n;
#pragma unroll
        for (int hx = 0; hx < 2; ++hx)
#pragma unroll
            for (int hd = 0; hd < 2; ++hd) {
                const int co = co0 + wm * 32 + 16 * hd + r16, ci = ci0 + wn * 32 + 16 * hx + 4 * lg;
                if (co < a.Cout && ci < a.Cin) *reinterpret_cast<f32x4s*>(o + (int64_t)co * a.Cin + ci) = acc[t][hx][hd] * out_scale;
            }
    }
}

#ifndef SP_WGRAD16
#define SP_WGRAD16 1     // 1: conv3x3_wgrad_pre16_kernel (v_mfma_f32_16x16x32); 0: conv3x3_split_wgrad_pre_kernel -- same-box A/B builds
#endif

int split_wgrad_group(int W) { return W >= 64 ? 1 : 64 / W; }      // images per unit: 1, 2 (W = 32), 4 (W = 16)

#ifndef SW_COT128
#define SW_COT128 1
#endif
int split_wgrad_cot(int Cout, int W) { return (SW_COT128 && Cout % 128 == 0 && W >= 32) ? 128 : 64; }    // output channels per block

void split_wgrad_plan(int B, int Cin, int Cout, int H, int W, int& splitK, int& tilesX) {
    tilesX = W >= 64 ? cdiv(W, 64) : 1;
    const int64_t units = (int64_t)(B / split_wgrad_group(W)) * H * tilesX;
    const int tiles = cdiv(Cin, 64) * cdiv(Cout, split_wgrad_cot(Cout, W));
    int64_t k = std::max<int64_t>(1, device_cu_count() / tiles);    // one 4-wave block per CU (110 KB of LDS), one round
    k = std::min<int64_t>(k, std::max<int64_t>(1, units / 16));      // at least 16 rows per block
    splitK = (int)k;
}

}  // namespace

extern "C" {

int onet_conv3x3_split_wgrad_ok(int B, int Cin, int Cout, int H, int W) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0) return 0;
    if (W >= 64) return (W & 3) == 0 ? 1 : 0;
    return ((W == 32 || W == 16) && B % (64 / W) == 0) ? 1 : 0;         // narrower maps: 64 / W images side by side per unit
}

int64_t onet_conv3x3_split_wgrad_ws_bytes(int B, int Cin, int Cout, int H, int W) {
    int splitK, tx;
    split_wgrad_plan(B, Cin, Cout, H, W, splitK, tx);
    return (int64_t)splitK * (split_wgrad_cot(Cout, W) == 64 ? 2 : 1) * 9 * Cout * Cin * 4;    // 64-channel tiles: two K-groups per block, a slab each
}

static int split_wgrad_impl(const float* x, int64_t x_bs, const float* dz, int64_t dz_bs, float* dw, void* ws, int64_t ws_bytes,
                            int B, int Cin, int Cout, int H, int W, int accumulate, void* stream, const float* nsave, int n_groups,
                            int f16 = 0, const unsigned* x_amax = nullptr, const unsigned* dz_amax = nullptr) {
    ONET_REQUIRE(x && dz && dw && ws, "conv3x3_split_wgrad: null pointer");
    ONET_REQUIRE(!nsave || ((n_groups == 1 || n_groups == 2) && B % n_groups == 0),
                 "conv3x3_split_wgrad_norm: one or two statistics groups of equal size");
    ONET_REQUIRE(onet_conv3x3_split_wgrad_ok(B, Cin, Cout, H, W),
                 "conv3x3_split_wgrad: needs W >= 64 with W %% 4 == 0, or W = 32 / 16 with B %% (64 / W) == 0 (use onet_conv3x3_winograd_wgrad)");
    ONET_REQUIRE((x_bs & 3) == 0 && (dz_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(dz) & 15) == 0,
                 "conv3x3_split_wgrad: 16-byte aligned image rows required");
    ONET_REQUIRE(x_bs >= (int64_t)Cin * H * W && dz_bs >= (int64_t)Cout * H * W, "conv3x3_split_wgrad: batch stride too small");
    ONET_REQUIRE((int64_t)std::max(Cin, Cout) * H * W * 4 < (1ll << 31), "conv3x3_split_wgrad: image exceeds the 2 GiB buffer-resource range");
    const int G = split_wgrad_group(W);
    ONET_REQUIRE(G == 1 || (((int64_t)(B - 1) * x_bs + (int64_t)Cin * H * W) * 4 < (1ll << 31) &&
                            ((int64_t)(B - 1) * dz_bs + (int64_t)Cout * H * W) * 4 < (1ll << 31)),
                 "conv3x3_split_wgrad: on maps narrower than 64 pixels the whole batch must lie within the 2 GiB buffer-resource range");
    const int COT = split_wgrad_cot(Cout, W), slabs = COT == 64 ? 2 : 1;
    ONET_REQUIRE(!f16 || dz_amax, "conv3x3_split_wgrad_f16: the fp16 parts need the magnitude slots of dz");
    SwArgs a{x, x_bs, dz, dz_bs, (float*)ws, B, Cin, Cout, H, W, cdiv(Cin, 64), cdiv(Cout, COT), 1, 1, nsave, nsave ? B / n_groups : B,
             x_amax, dz_amax};
    split_wgrad_plan(B, Cin, Cout, H, W, a.splitK, a.tilesX);
    const int64_t need = (int64_t)a.splitK * slabs * 9 * Cout * Cin * 4;
    ONET_REQUIRE(ws_bytes >= need, "conv3x3_split_wgrad: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)need);
    const int64_t blocks = (int64_t)a.splitK * a.ciTiles * a.coTiles;
    const dim3 grid((unsigned)blocks), blk(512);
    hipStream_t st = as_stream(stream);
#define ONET_SW_LAUNCH(G_, COT_)                                                                                      \
    do {                                                                                                              \
        if (nsave && f16) hipLaunchKernelGGL((conv3x3_split_wgrad_kernel<G_, COT_, true, true>), grid, blk, 0, st, a);    \
        else if (nsave) hipLaunchKernelGGL((conv3x3_split_wgrad_kernel<G_, COT_, true, false>), grid, blk, 0, st, a);     \
        else if (f16) hipLaunchKernelGGL((conv3x3_split_wgrad_kernel<G_, COT_, false, true>), grid, blk, 0, st, a);       \
        else hipLaunchKernelGGL((conv3x3_split_wgrad_kernel<G_, COT_, false, false>), grid, blk, 0, st, a);               \
    } while (0)
    if (COT == 128) {
        if (G == 1) ONET_SW_LAUNCH(1, 128);
        else ONET_SW_LAUNCH(2, 128);
    } else if (G == 1) ONET_SW_LAUNCH(1, 64);
    else if (G == 2) ONET_SW_LAUNCH(2, 64);
    else ONET_SW_LAUNCH(4, 64);
#undef ONET_SW_LAUNCH
    int rc = check_launch("conv3x3_split_wgrad_kernel");
    if (rc) return rc;
    return launch_wgrad_reduce((const float*)ws, dw, a.splitK * slabs, 9, Cout, Cin, 0, accumulate, as_stream(stream));
}

int onet_conv3x3_split_wgrad(const float* x, int64_t x_bs, const float* dz, int64_t dz_bs, float* dw, void* ws, int64_t ws_bytes,
                             int B, int Cin, int Cout, int H, int W, int accumulate, void* stream) {
    return split_wgrad_impl(x, x_bs, dz, dz_bs, dw, ws, ws_bytes, B, Cin, Cout, H, W, accumulate, stream, nullptr, 0);
}

int onet_conv3x3_split_wgrad_norm(const float* z_prev, int64_t z_bs, const float* save, int n_groups, const float* dz, int64_t dz_bs,
                                  float* dw, void* ws, int64_t ws_bytes, int B, int Cin, int Cout, int H, int W, int accumulate,
                                  void* stream) {
    ONET_REQUIRE(save, "conv3x3_split_wgrad_norm: null pointer");
    return split_wgrad_impl(z_prev, z_bs, dz, dz_bs, dw, ws, ws_bytes, B, Cin, Cout, H, W, accumulate, stream, save, n_groups);
}

int onet_conv3x3_split_wgrad_pre_ok(int B, int Cin, int Cout, int H, int W) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || (Cin % 8) || (Cout % 8)) return 0;
    if (W >= 64) return 1;
    return ((W == 32 && B % 2 == 0) || (W == 16 && B % 4 == 0)) ? 1 : 0;
}

int onet_conv3x3_split_wgrad_pre(const void* xs, int64_t xs_bs, const void* x_amax, const void* x_amax2, int split_ch, const void* dzs,
                                 int64_t dzs_bs, const void* dz_amax, int f16, float* dw, void* ws, int64_t ws_bytes, int B, int Cin,
                                 int Cout, int H, int W, int accumulate, void* stream) {
    ONET_REQUIRE(split_ch >= 0 && split_ch < Cin && (split_ch % 64) == 0, "conv3x3_split_wgrad_pre: split_ch must be a multiple of 64 inside Cin");
    ONET_REQUIRE(xs && dzs && dw && ws, "conv3x3_split_wgrad_pre: null pointer");
    ONET_REQUIRE(onet_conv3x3_split_wgrad_pre_ok(B, Cin, Cout, H, W),
                 "conv3x3_split_wgrad_pre: needs Cin, Cout %% 8 == 0 and W >= 64, or W = 32 / 16 with a batch multiple of 2 / 4");
    ONET_REQUIRE((xs_bs & 3) == 0 && (dzs_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(xs) & 15) == 0 && (reinterpret_cast<uintptr_t>(dzs) & 15) == 0,
                 "conv3x3_split_wgrad_pre: 16-byte aligned slots required");
    ONET_REQUIRE(xs_bs >= (int64_t)Cin * H * W / (f16 == 2 ? 2 : 1) && dzs_bs >= (int64_t)Cout * H * W / (f16 == 2 ? 2 : 1),
                 "conv3x3_split_wgrad_pre: batch stride too small");
    ONET_REQUIRE((int64_t)std::max(Cin, Cout) * H * W * 4 < (1ll << 31), "conv3x3_split_wgrad_pre: image exceeds the 2 GiB buffer-resource range");
    const int G = split_wgrad_group(W);
    ONET_REQUIRE(G == 1 || (((int64_t)(B - 1) * xs_bs + (int64_t)Cin * H * W) * 4 < (1ll << 31) &&
                            ((int64_t)(B - 1) * dzs_bs + (int64_t)Cout * H * W) * 4 < (1ll << 31)),
                 "conv3x3_split_wgrad_pre: on 32-pixel maps the whole batch must lie within the 2 GiB buffer-resource range");
    const int COT = split_wgrad_cot(Cout, W), slabs = COT == 64 ? 2 : 1;
    SwPreArgs a{xs, xs_bs, dzs, dzs_bs, (float*)ws, B, Cin, Cout, H, W, cdiv(Cin, 64), cdiv(Cout, COT), 1, 1, (const unsigned*)x_amax,
                (const unsigned*)dz_amax, (const unsigned*)x_amax2, split_ch};
    split_wgrad_plan(B, Cin, Cout, H, W, a.splitK, a.tilesX);
    const int64_t need = (int64_t)a.splitK * slabs * 9 * Cout * Cin * 4;
    ONET_REQUIRE(ws_bytes >= need, "conv3x3_split_wgrad_pre: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)need);
    const int64_t blocks = (int64_t)a.splitK * a.ciTiles * a.coTiles;
    const dim3 grid((unsigned)blocks), blk(512);
    hipStream_t st = as_stream(stream);
    const int np = f16 == 2 ? 1 : 2, pxp = G == 4 ? 76 : SWP_PXP;
    const int lds = (4 * ((np * 8 * pxp + 63) / 64 * 64) + 2 * ((np * (COT / 8) * pxp + 63) / 64 * 64)) * 16;
#define ONET_SWP_LAUNCH(G_, COT_, F_)                                                                              \
    do {                                                                                                           \
        auto kern = SP_WGRAD16 ? conv3x3_wgrad_pre16_kernel<G_, COT_, F_> : conv3x3_split_wgrad_pre_kernel<G_, COT_, F_>; \
        static PerDeviceOnce once;                                                                                 \
        if (once.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        hipLaunchKernelGGL(kern, grid, blk, lds, st, a);                                                           \
    } while (0)
#define ONET_SWP_F(G_, COT_) do { if (f16 == 2) ONET_SWP_LAUNCH(G_, COT_, 2); else if (f16) ONET_SWP_LAUNCH(G_, COT_, 1); else ONET_SWP_LAUNCH(G_, COT_, 0); } while (0)
    if (COT == 128) {
        if (G == 1) ONET_SWP_F(1, 128);
        else if (G == 2) ONET_SWP_F(2, 128);
        else ONET_SWP_F(4, 128);
    } else if (G == 1) ONET_SWP_F(1, 64);
    else if (G == 2) ONET_SWP_F(2, 64);
    else ONET_SWP_F(4, 64);
#undef ONET_SWP_F
#undef ONET_SWP_LAUNCH
    int rc = check_launch("conv3x3_split_wgrad_pre_kernel");
    if (rc) return rc;
    return launch_wgrad_reduce((const float*)ws, dw, a.splitK * slabs, 9, Cout, Cin, 0, accumulate, as_stream(stream));
}

int onet_conv3x3_split_pack_weights(const float* w, void* wq_fwd, void* wq_dgrad, void* amax_ws, int Cout, int Cin, int fwd_f16,
                                    int dgrad_f16, void* stream) {
    ONET_REQUIRE(w && (wq_fwd || wq_dgrad), "conv3x3_split_pack_weights: null pointer");
    ONET_REQUIRE(amax_ws || !(fwd_f16 == 1 || dgrad_f16 == 1), "conv3x3_split_pack_weights: the fp16 packs need the 8 KB magnitude workspace");
    if (fwd_f16 == 1 || dgrad_f16 == 1) {
        (void)hipMemsetAsync(amax_ws, 0, AMAX_SLOTS * AMAX_STRIDE * sizeof(unsigned), as_stream(stream));
        const int64_t nw = (int64_t)Cout * Cin * 9;
        hipLaunchKernelGGL(absmax_slots_kernel, dim3((unsigned)std::min<int64_t>((nw + 1023) / 1024, 1024)), dim3(256), 0, as_stream(stream), w,
                           nw, (unsigned*)amax_ws);
        int rc0 = check_launch("absmax_slots_kernel");
        if (rc0) return rc0;
    }
    ONET_REQUIRE(Cout > 0 && Cin > 0, "conv3x3_split_pack_weights: bad shape");
    ONET_REQUIRE(!wq_fwd || (Cin % 16) == 0, "conv3x3_split_pack_weights: the forward pack needs Cin %% 16 == 0");
    const int64_t n = (int64_t)std::max(Cin, ((Cout + 15) / 16) * 16) * 9 * std::max(Cin, Cout);
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 8192);
    if (wq_fwd) {
        hipLaunchKernelGGL(pack3x3_split_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w, (__bf16*)wq_fwd, Cout, Cin, 0, fwd_f16,
                           (const unsigned*)amax_ws);
        int rc = check_launch("pack3x3_split_kernel");
        if (rc) return rc;
    }
    if (wq_dgrad) hipLaunchKernelGGL(pack3x3_split_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w, (__bf16*)wq_dgrad, Cout, Cin, 1,
                                     dgrad_f16, (const unsigned*)amax_ws);
    return check_launch("pack3x3_split_kernel");
}

int onet_conv3x3_split_fwd(const float* x, int64_t x_bs, const void* wq, int wq_f16, float* z, int64_t z_bs, int B, int Cin, int Cout,
                           int H, int W, void* stream) {
    return split_fwd(x, x_bs, wq, z, z_bs, B, Cin, Cout, H, W, stream, nullptr, wq_f16);
}

int onet_absmax_slots(const float* x, int64_t n, void* slots, void* stream) {
    ONET_REQUIRE(x && slots && n > 0, "absmax_slots: bad args");
    hipLaunchKernelGGL(absmax_slots_kernel, dim3((unsigned)std::min<int64_t>((n + 1023) / 1024, 4096)), dim3(256), 0, as_stream(stream), x, n,
                       (unsigned*)slots);
    return check_launch("absmax_slots_kernel");
}

int onet_conv3x3_split_conv_amax(const float* x, int64_t x_bs, const void* x_amax, int scale_always, const void* wq, float* z, int64_t z_bs,
                                 float* part, int B, int Cin, int Cout, int H, int W, void* stream) {
    return split_fwd(x, x_bs, wq, z, z_bs, B, Cin, Cout, H, W, stream, part, 1, nullptr, 0, (const unsigned*)x_amax, scale_always);
}

int onet_conv3x3_split_wgrad_f16(const float* x, int64_t x_bs, const void* x_amax, const float* save, int n_groups, const float* dz,
                                 int64_t dz_bs, const void* dz_amax, float* dw, void* ws, int64_t ws_bytes, int B, int Cin, int Cout,
                                 int H, int W, int accumulate, void* stream) {
    return split_wgrad_impl(x, x_bs, dz, dz_bs, dw, ws, ws_bytes, B, Cin, Cout, H, W, accumulate, stream, save, save ? n_groups : 0, 1,
                            (const unsigned*)x_amax, (const unsigned*)dz_amax);
}

int onet_conv3x3_split_fwd_norm(const float* z_prev, int64_t z_bs, const float* save, int n_groups, const void* wq, int wq_f16,
                                float* z, int64_t zo_bs, float* part, int B, int Cin, int Cout, int H, int W, void* stream) {
    ONET_REQUIRE(save, "conv3x3_split_fwd_norm: null pointer");
    return split_fwd(z_prev, z_bs, wq, z, zo_bs, B, Cin, Cout, H, W, stream, part, wq_f16, save, n_groups);
}

int onet_split_pack_act(const float* x, int64_t x_bs, void* xs, int64_t xs_bs, int B, int C, int H, int W, int f16, float scale,
                        const void* slots, void* stream) {
    ONET_REQUIRE(x && xs, "split_pack_act: null pointer");
    ONET_REQUIRE(B > 0 && C > 0 && (C % 8) == 0 && H > 0 && W > 0, "split_pack_act: C must be a multiple of 8");
    ONET_REQUIRE((reinterpret_cast<uintptr_t>(xs) & 15) == 0 && (xs_bs & 3) == 0, "split_pack_act: 16-byte aligned slots required");
    const int64_t n = (int64_t)B * (C / 8) * H * W;
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 1 << 20);
    hipLaunchKernelGGL(split_pack_act_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, x_bs, (unsigned*)xs, xs_bs, B, C / 8, H, W,
                       f16, scale, (const unsigned*)slots);
    return check_launch("split_pack_act_kernel");
}

// BatchNorm statistics records of onet_conv3x3_split_fwd_pre: one per 16 x 32 tile; 16-pixel-wide maps: one per image PAIR and 16 rows
int onet_conv3x3_split_pre_nparts(int B, int H, int W) {
    if (W == 16) return (B > 0 && (B % 2) == 0 && (H % 16) == 0) ? (B / 2) * (H / 16) : 0;
    return split_nparts(B, H, W);
}

int onet_conv3x3_split_fwd_pre(const void* xs, int64_t xs_bs, const void* x_amax, int scale_always, const void* x_amax2, int split_ch,
                               const void* wq, int wq_f16, void* z, int z_bf16, int64_t z_bs, float* part, int B, int Cin, int Cout, int H, int W,
                               void* stream) {
    ONET_REQUIRE(!z_bf16 || (wq_f16 == 2 && SP_PRE16 && (z_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(z) & 7) == 0),
                 "conv3x3_split_fwd_pre: a bf16 output goes with plain bf16 operands (wq_f16 == 2), 8-byte aligned rows");
    ONET_REQUIRE(split_ch >= 0 && split_ch < Cin && (split_ch % 32) == 0, "conv3x3_split_fwd_pre: split_ch must be a multiple of 32 inside Cin");
    ONET_REQUIRE(xs && wq && z, "conv3x3_split_fwd_pre: null pointer");
    ONET_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && (W > 16 || (W == 16 && (B % 2) == 0 && (H % 16) == 0)),
                 "conv3x3_split_fwd_pre: bad shape (maps wider than 16 pixels, or exactly 16 wide with an even batch and H %% 16 == 0)");
    ONET_REQUIRE((Cin % (wq_f16 == 2 ? 32 : 16)) == 0, "conv3x3_split_fwd_pre: Cin must be a multiple of 16 (32 for plain bf16 operands)");
    ONET_REQUIRE((xs_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(xs) & 15) == 0, "conv3x3_split_fwd_pre: 16-byte aligned slots required");
    ONET_REQUIRE(xs_bs >= (int64_t)Cin * H * W / (wq_f16 == 2 ? 2 : 1) && z_bs >= (int64_t)Cout * H * W, "conv3x3_split_fwd_pre: batch stride too small");
    ONET_REQUIRE((int64_t)(Cin + 32) * H * W * 4 < (1ll << 31) && (int64_t)(Cin + 32) * 2 * 9 * Cout * 2 < (1ll << 31),
                 "conv3x3_split_fwd_pre: operand exceeds the 2 GiB buffer-resource range");
    SpPreArgs a{xs, xs_bs, (const __bf16*)wq, (float*)z, z_bs, B, Cin, Cout, H, W, 0, 0, 0, part, (const unsigned*)x_amax, scale_always,
                (const unsigned*)x_amax2, split_ch};
    a.z16 = z_bf16 ? 1 : 0;
    if (part) ONET_REQUIRE(onet_conv3x3_split_pre_nparts(B, H, W) > 0, "conv3x3_split_fwd_pre: statistics need a map made of full 16 x 32 tiles");
    hipStream_t st = as_stream(stream);
    if (W == 16) {
        ONET_REQUIRE(xs_bs * 4 + (int64_t)Cin * H * W * 4 < (1ll << 31), "conv3x3_split_fwd_pre: image pair exceeds the buffer-resource range");
        if (wq_f16 == 2) return part ? launch_split_pre<true, 2, true>(a, st) : launch_split_pre<false, 2, true>(a, st);
        if (wq_f16) return part ? launch_split_pre<true, 1, true>(a, st) : launch_split_pre<false, 1, true>(a, st);
        return part ? launch_split_pre<true, 0, true>(a, st) : launch_split_pre<false, 0, true>(a, st);
    }
    if (wq_f16 == 2) return part ? launch_split_pre<true, 2, false>(a, st) : launch_split_pre<false, 2, false>(a, st);
    if (wq_f16) return part ? launch_split_pre<true, 1, false>(a, st) : launch_split_pre<false, 1, false>(a, st);
    return part ? launch_split_pre<true, 0, false>(a, st) : launch_split_pre<false, 0, false>(a, st);
}

// Input gradient of the SECOND convolution of a DoubleConv from pre-split dz, with the BatchNorm-backward reduce pass of the FIRST unit
// (whose output gradient this launch produces) in the epilogue: see SpPreArgs::rd_*.  Returns 1 (nothing launched) where the map is not
// made of full tiles.  rec4: [onet_conv3x3_split_pre_nparts(B, H, W)][Cout][4] floats.
int onet_conv3x3_split_dgrad_pre_bnreduce(const void* dzs, int64_t dzs_bs, const void* dz_amax, int scale_always, const void* wq, int wq_f16,
                                          float* da, int64_t da_bs, const void* z_prev, int z_bf16, int64_t z_bs, const float* save, int group_images,
                                          float* rec4, void* da_amax, int B, int Cin, int Cout, int H, int W, void* stream) {
    ONET_REQUIRE(dzs && wq && da && z_prev && save && rec4, "conv3x3_split_dgrad_pre_bnreduce: null pointer");
    if (onet_conv3x3_split_pre_nparts(B, H, W) <= 0 || (W != 16 && (W % 32)) || (Cout % 16)) return 1;
    if (W == 16 && ((group_images % 2) || wq_f16 != 2)) return 1;    // a tile's image pair must lie in one statistics group; (hi | mid)
                                                                     // parts on 16-pixel maps: not built (register budget of that instance)
    // a tile of fewer than four chunks is too short to cover the epilogue's z loads (plain bf16 operands, 64 channels of dz: two chunks --
    // measured at BASELINE configs[2]: the fused launches cost more than the reduce pass they replace)
    if (Cin / (wq_f16 == 2 ? 32 : 16) < 4) return 1;
    ONET_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && (Cin % (wq_f16 == 2 ? 32 : 16)) == 0, "conv3x3_split_dgrad_pre_bnreduce: bad shape");
    ONET_REQUIRE(group_images >= 0 && (group_images == 0 || B % group_images == 0), "conv3x3_split_dgrad_pre_bnreduce: bad statistics groups");
    ONET_REQUIRE((dzs_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(dzs) & 15) == 0 && (z_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(z_prev) & (z_bf16 ? 7 : 15)) == 0 &&
                 (da_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(da) & 15) == 0, "conv3x3_split_dgrad_pre_bnreduce: 16-byte aligned rows required");
    ONET_REQUIRE(dzs_bs >= (int64_t)Cin * H * W / (wq_f16 == 2 ? 2 : 1) && da_bs >= (int64_t)Cout * H * W && z_bs >= (int64_t)Cout * H * W,
                 "conv3x3_split_dgrad_pre_bnreduce: batch stride too small");
    ONET_REQUIRE((int64_t)(Cin + 32) * H * W * 4 < (1ll << 31) && (int64_t)(Cin + 32) * 2 * 9 * Cout * 2 < (1ll << 31),
                 "conv3x3_split_dgrad_pre_bnreduce: operand exceeds the 2 GiB buffer-resource range");
    SpPreArgs a{dzs, dzs_bs, (const __bf16*)wq, da, da_bs, B, Cin, Cout, H, W, 0, 0, 0, nullptr, (const unsigned*)dz_amax, scale_always, nullptr, 0};
    a.rd_z = (const float*)z_prev;
    a.rd_z16 = z_bf16 ? 1 : 0;
    a.rd_z_bs = z_bs;
    a.rd_save = save;
    a.rd_gimg = group_images;
    a.rd_rec = rec4;
    a.rd_amax = (unsigned*)da_amax;
    hipStream_t st = as_stream(stream);
    if (W == 16) {
        ONET_REQUIRE(dzs_bs * 4 + (int64_t)Cin * H * W * 4 < (1ll << 31), "conv3x3_split_dgrad_pre_bnreduce: image pair exceeds the buffer-resource range");
        return launch_split_pre<false, 2, true, true>(a, st);
    }
    if (wq_f16 == 2) return launch_split_pre<false, 2, false, true>(a, st);
    return wq_f16 ? launch_split_pre<false, 1, false, true>(a, st) : launch_split_pre<false, 0, false, true>(a, st);
}

// Input gradient of a decoder block's FIRST convolution (fp16 hi | mid parts): channels < ch0 of da as fp32, channels >= ch0 -- the
// up-sampled half of the concat gradient, read only by the ConvTranspose2d backward GEMMs -- pre-split into daP (SpPreArgs::zP).
int onet_conv3x3_split_dgrad_pre_slots(const void* dzs, int64_t dzs_bs, const void* dz_amax, int scale_always, const void* wq, int wq_f16,
                                       float* da, int64_t da_bs, void* daP, int64_t daP_bs, int ch0, const void* daP_amax, int B, int Cin,
                                       int Cout, int H, int W, void* stream) {
    ONET_REQUIRE(dzs && wq && da && daP && (daP_amax || wq_f16 == 2) && (wq_f16 == 1 || wq_f16 == 2), "conv3x3_split_dgrad_pre_slots: bad args");
    ONET_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W >= 32 && (W % 32) == 0 && (H % 16) == 0 && (Cin % (wq_f16 == 2 ? 32 : 16)) == 0,
                 "conv3x3_split_dgrad_pre_slots: maps made of full 16 x 32 tiles, Cin %% 16 == 0 (32: plain bf16)");
    ONET_REQUIRE(ch0 > 0 && ch0 < Cout && (ch0 % 64) == 0 && (Cout % 64) == 0, "conv3x3_split_dgrad_pre_slots: ch0 and Cout must be multiples of 64");
    ONET_REQUIRE((dzs_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(dzs) & 15) == 0 && (daP_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(daP) & 15) == 0,
                 "conv3x3_split_dgrad_pre_slots: 16-byte aligned slots required");
    ONET_REQUIRE(dzs_bs >= (int64_t)Cin * H * W / (wq_f16 == 2 ? 2 : 1) && da_bs >= (int64_t)ch0 * H * W &&
                     daP_bs >= (int64_t)(Cout - ch0) * H * W / (wq_f16 == 2 ? 2 : 1),
                 "conv3x3_split_dgrad_pre_slots: batch stride too small");
    ONET_REQUIRE((int64_t)(Cin + 32) * H * W * 4 < (1ll << 31) && (int64_t)(Cin + 32) * 2 * 9 * Cout * 2 < (1ll << 31),
                 "conv3x3_split_dgrad_pre_slots: operand exceeds the 2 GiB buffer-resource range");
    SpPreArgs a{dzs, dzs_bs, (const __bf16*)wq, da, da_bs, B, Cin, Cout, H, W, 0, 0, 0, nullptr, (const unsigned*)dz_amax, scale_always, nullptr, 0};
    a.zP = daP;
    a.zP_bs = daP_bs;
    a.zP_ch0 = ch0;
    a.zP_slots = (const unsigned*)daP_amax;
    if (wq_f16 == 2) return launch_split_pre<false, 2, false>(a, as_stream(stream));      // plain bf16: one part, unscaled (conv3x3_pre16_kernel)
    return launch_split_pre<false, 1, false>(a, as_stream(stream));
}

// |da[ci]| <= max |dz| * sum over (co, tap) of |w[co][ci][tap]|: the bound of the input gradient's channels >= ci0 from the bound of dz
// (dz_amax) and the weights, into out_slots (zeroed by the caller) -- what onet_conv3x3_split_dgrad_pre_slots scales its slots by
__global__ __launch_bounds__(256) void conv3x3_dgrad_bound_kernel(const float* __restrict__ w, int Cout, int Cin, int ci0,
                                                                  const unsigned* __restrict__ dz_slots, unsigned* __restrict__ out_slots) {
    __shared__ float red[4];
    const float dzmax = amax_read(dz_slots);
    const int ci = ci0 + blockIdx.x;
    float sm = 0.f;
    for (int i = threadIdx.x; i < Cout * 9; i += 256) sm += fabsf(w[((int64_t)(i / 9) * Cin + ci) * 9 + i % 9]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sm;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float bound = ((red[0] + red[1]) + (red[2] + red[3])) * dzmax * 1.00001f;
        if (bound == bound) atomicMax(out_slots + (blockIdx.x & (AMAX_SLOTS - 1)) * AMAX_STRIDE, __builtin_bit_cast(unsigned, bound));
    }
}

int onet_conv3x3_dgrad_bound(const float* w, int Cout, int Cin, int ci0, const void* dz_amax, void* out_slots, void* stream) {
    ONET_REQUIRE(w && dz_amax && out_slots && Cout > 0 && ci0 >= 0 && ci0 < Cin, "conv3x3_dgrad_bound: bad args");
    hipLaunchKernelGGL(conv3x3_dgrad_bound_kernel, dim3((unsigned)(Cin - ci0)), dim3(256), 0, as_stream(stream), w, Cout, Cin, ci0,
                       (const unsigned*)dz_amax, (unsigned*)out_slots);
    return check_launch("conv3x3_dgrad_bound_kernel");
}

int onet_conv3x3_split_nparts(int B, int H, int W) { return split_nparts(B, H, W); }

int onet_conv3x3_split_fwd_stats(const float* x, int64_t x_bs, const void* wq, int wq_f16, float* z, int64_t z_bs, float* part, int B,
                                 int Cin, int Cout, int H, int W, void* stream) {
    ONET_REQUIRE(part, "conv3x3_split_fwd_stats: null pointer");
    return split_fwd(x, x_bs, wq, z, z_bs, B, Cin, Cout, H, W, stream, part, wq_f16);
}

}  // extern "C"
