// 3x3 convolution forward + dgrad with bf16 operands on the gfx950 matrix cores (v_mfma_f32_32x32x16_bf16),
// fp32 tensors in HBM, fp32 accumulation: BASELINE config 3's "bf16 MFMA conv path" for F.conv2d / its input
// gradient at OV:47,51.  Selected only by ONET_CONV_ALGO=bf16 (the fp32 kernels stay the default: the headline metric
// is an fp32 one).
//
// GEMM view as in conv_mfma.hip: D[co][pix] = sum_{tap, ci} W[ci][tap][co] * X[ci][pix + tap], M = output channels,
// N = 32 consecutive pixels of an image row, K = 16 input channels per MFMA.  What bf16 changes:
//  * an MFMA operand is 8 bf16 = 16 bytes per lane (A lane l -> A[i = l&31][k = 8*(l>>5) .. +7], B likewise), i.e. the
//    8 K-values of a lane are 8 CHANNELS of one pixel / one output channel.  The LDS tiles are therefore made of 16-byte
//    slots of 8 channels, the two 8-channel halves of a 16-channel chunk kept apart: input halo tile [half][row][col],
//    weights [tap][half][co]; a fragment is ONE ds_read_b128 per lane and every 16-lane group of it reads 256 contiguous
//    bytes (no bank conflicts: SQ_LDS_BANK_CONFLICT = 0);
//  * the fp32 -> bf16 conversion (v_cvt_pk_bf16_f32, round-to-nearest-even) happens on the way into LDS: a staging
//    thread gathers the 8 channel planes of ONE pixel (8 loads, each coalesced along x across the wave), packs them and
//    stores 16 bytes; with bf16 STORAGE of the operand (XB) the loads are 2-byte loads of the producer's bf16 copy --
//    the same values; weights are packed to bf16 once per optimizer step (onet_conv3x3_pack_weights_bf16);
//  * per 16-channel chunk a wave issues 18 A + 3 x (NT + 2) B ds_read_b128 for 9 x 2 x NT MFMAs of 32 cycles
//    (NT = 2 image rows per wave: 64 accumulator registers, two 4-wave blocks per CU), the reads one tap ahead of the MFMAs;
//  * blocks are persistent and the load -> LDS -> MFMA pipeline runs across tile boundaries (see the kernel).
//  Measured limits (timing-only ablation builds of round 2, B = 64; table in DESIGN.md 4.2d): the cost that does not
//  hide is the request rate of the NCHW input tile's cache lines (~240 half-used 128-byte lines per chunk): loading the input
//  once saves 25 %, the weights once 5 %; a quarter of the planes is as good as none; a quarter of the load INSTRUCTIONS for the
//  same lines (8-byte quad loads + in-register transpose) changes nothing, and neither does a longer prefetch distance (two
//  register sets, loads three chunks ahead: slower) nor fewer VALU adds.  One block per CU keeps 80 % of the throughput.
// Requires Cin % 16 == 0 (K never straddles a chunk) and W > 16; everything else takes the fp32 kernels.
#include <algorithm>
#include <cstdlib>
#include "common.hpp"

using namespace onet;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4b __attribute__((ext_vector_type(4)));
typedef float f32x4b __attribute__((ext_vector_type(4)));

constexpr unsigned OOB_B = 0x80000000u;

static __device__ __forceinline__ __amdgpu_buffer_rsrc_t b_rsrc(const void* base, int64_t bytes) {
    const int n = bytes > 0x7fffffffll ? 0x7fffffff : (int)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, n, 0x00020000);
}
static __device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}

// w [Cout][Cin][3][3] fp32 -> wf [Cin/16][9][2][Cout][8] bf16 (forward: chunk of 16 input channels, tap, 8-channel half,
// output channel, channel within the half), wd [Cout16/16][9][2][Cin][8] bf16 with the taps rotated by 180 degrees (input
// gradient = the same kernel with the roles of Cin and Cout swapped); the channel tail of wd (Cout % 16 != 0) is zero-filled.
// The 8-channel halves are the two K-halves of an MFMA operand (lanes 0-31 / 32-63): keeping each half contiguous over the
// output channels makes both the global load and the LDS fragment read of a 16-lane group one contiguous 256-byte run.
__global__ void pack3x3_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ wf, __bf16* __restrict__ wd,
                                    int Cout, int Cin, int which) {
    const int K = which == 0 ? Cin : ((Cout + 15) / 16) * 16, N = which == 0 ? Cout : Cin;
    const int64_t n = (int64_t)K * 9 * N;
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
        (which == 0 ? wf : wd)[i] = (__bf16)v;
    }
}

struct BfArgs {
    const void* x;        // fp32 NCHW, or bf16 NCHW (XB kernels: bf16 STORAGE of the operand, written by the BatchNorm / pooling /
    int64_t x_bs;         // ConvTranspose2d kernels next to their fp32 outputs; strides in elements)
    const __bf16* wq;     // [Cin/16][9][2][Cout][8]
    float* z;
    int64_t z_bs;
    int B, Cin, Cout, H, W, tilesX, tilesY, coTiles;
    float* stats;         // ST kernels: BatchNorm partials [Cout][B * tilesY * tilesX][3] = (n, mean, M2) per tile
};

// sum over the 32 lanes of a wave half, valid in lanes 31 / 63 (DPP row rotations + row broadcast, as in conv_wino4.hip)
#define ONET_BF_DPP_ADD(v, ctrl, rmask) \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false))
static __device__ __forceinline__ float bf_half_sum(float v) {
    ONET_BF_DPP_ADD(v, 0x128, 0xf);   // row_ror:8
    ONET_BF_DPP_ADD(v, 0x124, 0xf);   // row_ror:4
    ONET_BF_DPP_ADD(v, 0x122, 0xf);   // row_ror:2
    ONET_BF_DPP_ADD(v, 0x121, 0xf);   // row_ror:1
    ONET_BF_DPP_ADD(v, 0x142, 0xa);   // row_bcast:15 into rows 1 and 3
    return v;
}
#undef ONET_BF_DPP_ADD

// LDS, per 16-channel chunk (two such buffers: the next chunk is committed while the current one is read):
//   weights [9 taps][2 halves][64 co] and the input halo tile [2 halves][NPIXP pixels], one 16-byte slot (8 bf16 channels) each.
// A fragment read (ds_read_b128) of a 16-lane group is then 16 consecutive slots = 256 contiguous bytes = all 64 banks once;
// the interleaved [pixel][half] order of the first version of this kernel put the 16 lanes 32 bytes apart (2-way conflict on
// every read: SQ_LDS_BANK_CONFLICT = 40 % of SQ_LDS_IDX_ACTIVE).
template <int NT, int TW_ = 32>
struct BfCfg {
    static constexpr int TW = TW_, RPT = 32 / TW;                      // image rows per 32-pixel MFMA column tile
    static constexpr int CO_T = 64, ROWS = 4 * NT * RPT;               // 4 waves x NT column tiles (TW = 16: 2 rows each)
    static constexpr int IN_ROWS = ROWS + 2, IN_COLS = TW + 2;
    static constexpr int NPIX = IN_ROWS * IN_COLS;
    static constexpr int NIT = (((NPIX + 31) / 32) * 64 + 255) / 256;  // staging rounds: 64 slots = 32 pixels x 2 halves per wave
    static constexpr int NPIXP = NIT * 128;                            // pixel slots per half (padded: every staging slot exists)
    static constexpr int W_ITEMS = 9 * 2 * CO_T;
    static constexpr int NWI = (W_ITEMS + 255) / 256;
    static constexpr int W_SLOTS = NWI * 256;
    static constexpr int BUF_SLOTS = W_SLOTS + 2 * NPIXP;
    static constexpr int LDS_BYTES = 2 * BUF_SLOTS * 16;
    static constexpr int NB = (NT - 1) * RPT + 3;                      // distinct B row-fragments per horizontal tap
};

// PERSISTENT blocks: a block walks over output tiles (tile = 64 output channels x ROWS x TW pixels of one image), and the chunk
// pipeline -- global loads two 16-channel chunks ahead of the MFMAs, LDS commit one ahead -- runs ACROSS tile boundaries: while
// a tile's accumulators are stored, the first two chunks of the next tile are already in LDS / in flight.  With one tile per
// block (the first version) the 64- and 128-channel layers of the 256x256 level spent two thirds of their time in the
// un-overlapped prologue (first loads) and epilogue (64 KB of stores per tile): 0.46 ms of 0.68 ms at 64 channels.
// ST: the forward of a Conv-BatchNorm pair (OV:47-48, 51-52) also emits the BatchNorm statistics of its output, one (n, mean,
// M2) record per tile and channel from the final accumulators (pivot-shifted sums per wave, the four waves of the tile merged
// through LDS; full tiles only, host-checked) -- the separate statistics pass re-read every z once (12.8 ms of a 355 ms step
// at B = 256).
// XB: 0 fp32 NCHW input; 1 bf16 NCHW copy; 2 bf16 CHANNEL-BLOCKED copy [C/8][H][W][8] (a staging slot = 16 contiguous bytes: one
// b128 load instead of eight 2-byte loads, and a halo row of a tile = 34 x 16 B of fully used cache lines)
template <int NT, int WPS, int TW = 32, int XB = 0, bool ST = false>
__global__ __launch_bounds__(256, WPS) void conv3x3_bf16_kernel(BfArgs a) {
    using C = BfCfg<NT, TW>;
    constexpr int ROWS = C::ROWS, IN_COLS = C::IN_COLS, NIT = C::NIT, NWI = C::NWI, CO_T = C::CO_T, RPT = C::RPT;
    constexpr int NPIXP = C::NPIXP, NB = C::NB, BUF = C::BUF_SLOTS;
    constexpr int XE = XB ? 2 : 4;                                     // bytes per element of x
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    u32x4b* lds = reinterpret_cast<u32x4b*>(smem_b);                   // [2 buffers][weights W_SLOTS | input 2 x NPIXP]

    // tiles of this block: every XCD (blockIdx % 8) owns a contiguous range of the tile list (neighbouring tiles share halo
    // rows and the weight slice in that XCD's L2); its blocks stride through the range
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
    const int px = l31 % TW, py = l31 / TW;
    const int HW = a.H * a.W;
    const int nchunks = a.Cin >> 4;

    const __amdgpu_buffer_rsrc_t wr = b_rsrc(a.wq, (int64_t)a.Cin * 9 * a.Cout * 2);
    const unsigned in_step = (unsigned)(16 * HW * XE), w_step = (unsigned)(9 * a.Cout * 16 * 2);
    const unsigned plane = (unsigned)(HW * XE);

    // ---- staging side: runs two chunks ahead of the compute side, across tiles
    // slots of this thread: round k, wave wn, lane -> (8-channel half = lane / 32, pixel = 32 * (wn + 4 k) + lane % 32): a wave
    // loads 32 consecutive halo pixels of 8 channel planes (coalesced along x), and its ds_write_b128 puts every 16-lane group on
    // 256 contiguous bytes; weight slots i = tid + 256 k = (tap, half, co) -> 16 bytes of the slice [chunk][tap][half][co][8]
    unsigned in_off[NIT], w_off[NWI];
    __amdgpu_buffer_rsrc_t xr;
    int st_tile = t_first, st_chunk = 0;
    unsigned cin_bytes = 0, cw_bytes = 0;
    auto setup_stage = [&]() __attribute__((always_inline)) {
        const bool live = st_tile < t_end;
        int v = live ? st_tile : t_first;
        // output-channel tile fastest in the tile list (as conv_split.hip): the blocks that need the same input tile run on one XCD
        // at the same time and share it in that L2 instead of every channel-tile pass streaming the input from HBM again
        const int co0 = (v % a.coTiles) * CO_T;
        v /= a.coTiles;
        const int tx = v % a.tilesX;
        v /= a.tilesX;
        const int ty = v % a.tilesY;
        const int b = v / a.tilesY;
        const int y0 = ty * ROWS, x0 = tx * TW;
        xr = b_rsrc(static_cast<const char*>(a.x) + (int64_t)b * a.x_bs * XE, (int64_t)a.Cin * HW * XE);
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int p = (wn + 4 * k) * 32 + l31;
            const int r = p / IN_COLS, c = p % IN_COLS;
            const int yy = y0 - 1 + r, xx = x0 - 1 + c;
            const bool ok = live && p < C::NPIX && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
            if constexpr (XB == 2) in_off[k] = ok ? (unsigned)((kh * HW + yy * a.W + xx) * 16) : OOB_B;
            else in_off[k] = ok ? (unsigned)(((kh * 8) * HW + yy * a.W + xx) * XE) : OOB_B;
        }
#pragma unroll
        for (int k = 0; k < NWI; ++k) {
            const int i = tid + 256 * k;
            const int co = i & (CO_T - 1), th = i >> 6;                // th = tap * 2 + half
            const bool ok = live && (i < C::W_ITEMS) && (co0 + co < a.Cout);
            w_off[k] = ok ? (unsigned)((th * a.Cout + co0 + co) * 16) : OOB_B;
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

    float xin[XB == 2 ? 1 : NIT][8];                 // XB = 1: the low 16 bits hold the bf16 value
    u32x4b xblk[XB == 2 ? NIT : 1];                  // XB = 2: the slot as it lies in memory
    u32x4b wv[NWI];
    auto issue = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            if constexpr (XB == 2) {
                xblk[k] = __builtin_amdgcn_raw_buffer_load_b128(xr, in_off[k], (int)cin_bytes, 0);
                continue;
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                // the chunk / plane part of the address is wave-uniform: it rides in the instruction's scalar offset (no VALU add
                // per load); the per-lane part alone decides the range check (OOB_B -> 0)
                if constexpr (XB == 2) {
                } else if constexpr (XB == 1)
                    xin[k][c] = __builtin_bit_cast(float, (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(
                                                              xr, in_off[k], (int)(cin_bytes + c * plane), 0));
                else
                    xin[k][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                              xr, in_off[k], (int)(cin_bytes + c * plane), 0));
            }
        }
#pragma unroll
        for (int k = 0; k < NWI; ++k)
            wv[k] = __builtin_amdgcn_raw_buffer_load_b128(wr, w_off[k], (int)cw_bytes, 0);
    };
    u32x4b* const w_st = lds + tid;                                    // + 256 k          (+ buffer * BUF)
    u32x4b* const in_st = lds + C::W_SLOTS + kh * NPIXP + wn * 32 + l31;   // + 128 k
    auto commit = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            u32x4b v;
            if constexpr (XB == 2) {
                v = xblk[k];
            } else if constexpr (XB == 1) {
                auto pk = [](float lo, float hi) { return __builtin_bit_cast(unsigned, lo) | (__builtin_bit_cast(unsigned, hi) << 16); };
                v = u32x4b{pk(xin[k][0], xin[k][1]), pk(xin[k][2], xin[k][3]), pk(xin[k][4], xin[k][5]), pk(xin[k][6], xin[k][7])};
            } else {
                v = u32x4b{pack_bf16(xin[k][0], xin[k][1]), pack_bf16(xin[k][2], xin[k][3]),
                           pack_bf16(xin[k][4], xin[k][5]), pack_bf16(xin[k][6], xin[k][7])};
            }
            in_st[buf * BUF + 128 * k] = v;          // padding slots (pixel >= NPIX) exist and receive zeros
        }
#pragma unroll
        for (int k = 0; k < NWI; ++k) w_st[buf * BUF + 256 * k] = wv[k];
    };

    const u32x4b* const a_ptr = lds + kh * CO_T + l31;                                        // + (tap * 2) * 64 + m * 32
    const u32x4b* const b_ptr = lds + C::W_SLOTS + kh * NPIXP + (wn * NT * RPT + py) * IN_COLS + px;   // + j * IN_COLS + kx

    setup_stage();
    issue();
    commit(0);
    advance();
    issue();                                         // second chunk (or the first of the next tile)
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
            const u32x4b* const ab = a_ptr + buf * BUF;
            const u32x4b* const bb = b_ptr + buf * BUF;
            // Fragment reads run ONE tap ahead of the MFMAs that consume them (two register sets), the taps in kx-major order
            // so that the NB row-fragments of a horizontal shift serve its three vertical taps; the row-fragments of the next
            // shift are fetched during the three taps of the current one.
            u32x4b Aq[2][2], Bq[2][NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) Bq[0][j] = bb[j * IN_COLS];
#pragma unroll
            for (int m = 0; m < 2; ++m) Aq[0][m] = ab[m * 32];
#pragma unroll
            for (int idx = 0; idx < 9; ++idx) {
                const int kx = idx / 3, ky = idx % 3;
                if (idx < 8) {
                    const int nt = ((idx + 1) % 3) * 3 + (idx + 1) / 3;
#pragma unroll
                    for (int m = 0; m < 2; ++m) Aq[(idx + 1) & 1][m] = ab[nt * 2 * CO_T + m * 32];
                }
                if (kx < 2) {
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        if (j % 3 == ky) Bq[(kx + 1) & 1][j] = bb[j * IN_COLS + kx + 1];
                }
                if (idx == 4) {
                    // the other buffer was last read one chunk ago (barrier since): the chunk after this one goes in, and the
                    // loads of the one after that take over the staging registers
                    commit(buf ^ 1);
                    advance();
                    issue();
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Aq[idx & 1][m]),
                                                                            __builtin_bit_cast(bf16x8, Bq[kx & 1][n * RPT + ky]), acc[m][n], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            buf ^= 1;
        }

        int v = tile;
        const int co0 = (v % a.coTiles) * CO_T;
        v /= a.coTiles;
        const int tx = v % a.tilesX;
        v /= a.tilesX;
        const int ty = v % a.tilesY;
        const int b = v / a.tilesY;
        const int y0 = ty * ROWS, x0 = tx * TW;
        if constexpr (ST) {
            float* sc = reinterpret_cast<float*>(lds + 2 * BUF);       // [4 waves][64 channels][mean, M2]
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
                    s1 = bf_half_sum(s1);
                    s2 = bf_half_sum(s2);
                    if (l31 == 31) {
                        const int cl = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                        sc[(wn * 64 + cl) * 2] = fmaf(s1, inv_npw, pv);
                        sc[(wn * 64 + cl) * 2 + 1] = fmaxf(fmaf(-s1 * inv_npw, s1, s2), 0.f);
                    }
                }
            __syncthreads();
            if (tid < 64 && co0 + tid < a.Cout) {
                float mw[4], qw[4];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    mw[w] = sc[(w * 64 + tid) * 2];
                    qw[w] = sc[(w * 64 + tid) * 2 + 1];
                }
                const float mean = 0.25f * ((mw[0] + mw[1]) + (mw[2] + mw[3]));
                float m2 = (qw[0] + qw[1]) + (qw[2] + qw[3]);
#pragma unroll
                for (int w = 0; w < 4; ++w) m2 = fmaf(npw * (mw[w] - mean), mw[w] - mean, m2);
                const int64_t nblk = (int64_t)a.B * a.tilesY * a.tilesX;
                const int64_t blk = ((int64_t)b * a.tilesY + ty) * a.tilesX + tx;
                float* sp = a.stats + ((int64_t)(co0 + tid) * nblk + blk) * 3;
                sp[0] = 4.f * npw;
                sp[1] = mean;
                sp[2] = m2;
            }
        }
        float* zb = a.z + (int64_t)b * a.z_bs;
        const int xo = x0 + px;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int yo = y0 + (wn * NT + n) * RPT + py;
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

template <int NT, int WPS, int TW = 32, int XB = 0, bool ST = false>
static int launch_bf16(BfArgs a, hipStream_t st) {
    using C = BfCfg<NT, TW>;
    static int lds_pad = -1;      // experiment: ONET_BF16_LDS_PAD=bytes of unused dynamic LDS (forces fewer blocks per CU)
    if (lds_pad < 0) { const char* e = getenv("ONET_BF16_LDS_PAD"); lds_pad = e ? atoi(e) : 0; }
    const int LDS_BYTES = C::LDS_BYTES + (ST ? 4 * 64 * 2 * 4 : 0) + lds_pad;
    a.tilesX = cdiv(a.W, TW);
    a.tilesY = cdiv(a.H, C::ROWS);
    a.coTiles = cdiv(a.Cout, C::CO_T);
    const int64_t tiles = (int64_t)a.B * a.tilesX * a.tilesY * a.coTiles;
    ONET_REQUIRE(tiles > 0 && tiles < (1ll << 31), "conv3x3_bf16: tile count %lld out of range", (long long)tiles);
    auto kern = conv3x3_bf16_kernel<NT, WPS, TW, XB, ST>;
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    }
    // persistent grid: as many blocks as are resident at once (WPS per CU), a multiple of the 8 XCDs
    const int64_t resident = (int64_t)device_cu_count() * (lds_pad > 30000 ? 1 : WPS);
    const int64_t blocks = std::min<int64_t>((tiles + 7) / 8 * 8, std::max<int64_t>(8, resident / 8 * 8));
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), LDS_BYTES, st, a);
    return check_launch("conv3x3_bf16_kernel");
}


// ------------------------------------------------------------------ weight gradient, bf16 operands
//   dW[co][ci][ky][kx] = sum_{b,y,x} dz[b][co][y][x] * X[b][ci][y+ky-1][x+kx-1]:  M = co, N = ci, K = pixels, one
//   accumulator per tap (9 x 16 registers per wave), v_mfma_f32_32x32x16_bf16 with the 8 K-values of a lane = 8
//   CONSECUTIVE PIXELS of one channel row -- the natural NCHW order, so the tiles stay pixel-innermost: a unit is a
//   4-row x 16-pixel patch of one image (4 K-steps = its 4 rows), dz [64 co][4][16] and the halo patch of x
//   [64 ci][6][18] rounded to bf16 on the way into LDS.  The A fragment is one ds_read_b128 per K-step; the three
//   horizontally shifted B fragments of a patch row come from ONE ds_read_b128 + ds_read_b32 (10 bf16): shift 0 and 2
//   are dword-aligned, shift 1 is four v_alignbit_b32.  Channel strides 36 / 76 dwords (4 x odd) keep the 16-lane
//   groups of a b128 read on 64 distinct banks.  Split-K over (image, patch) units, raw slabs [split][tap][co][ci]
//   reduced deterministically by wgrad_reduce_kernel (conv_mfma.hip).
struct BwArgs {
    const void* x;        // fp32 or bf16 (XB) NCHW, strides in elements
    int64_t x_bs;
    const void* dz;       // fp32 or bf16 (ZB)
    int64_t dz_bs;
    float* slab;
    int B, Cin, Cout, H, W, ciTiles, coTiles, splitK, tilesY, tilesX;
};

constexpr int BW_SDZ = 36, BW_SX = 76, BW_XROW = 12;        // dword strides: per co, per ci, per patch row

template <bool XB = false, bool ZB = false>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_bf16_kernel(BwArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned dz_lds[64 * BW_SDZ];
    __shared__ __attribute__((aligned(16))) unsigned x_lds[64 * BW_SX];

    int bid;
    {
        const int n = gridDim.x, q = n >> 3, r = n & 7, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tiles = a.ciTiles * a.coTiles;
    const int ks = bid / tiles, tile = bid % tiles;
    const int ci0 = (tile % a.ciTiles) * 64, co0 = (tile / a.ciTiles) * 64;
    const int nunits = a.B * a.tilesY * a.tilesX;
    const int per = (nunits + a.splitK - 1) / a.splitK;
    const int u0 = ks * per, u1 = min(u0 + per, nunits);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int l31 = lane & 31, kh = lane >> 5;
    const int HW = a.H * a.W;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // staging roles: dz -- thread = (co, patch row): 16 floats as 4 b128 loads; x -- thread = one or two (ci, patch row)
    // rows of 18 floats: the 16 interior columns as 4 b128 loads (x0 % 16 == 0, W % 4 == 0: aligned, and a float4 is
    // entirely inside or outside the image) + the two halo columns
    const int dz_c = tid >> 2, dz_r = tid & 3;
    u32x4b dzv[4];
    u32x4b xq[2][4];
    float xh[2][2];
    auto issue = [&](int u) __attribute__((always_inline)) {
        const bool live = u < u1;
        const int uu = live ? u : 0;
        const int tx = uu % a.tilesX, ty = (uu / a.tilesX) % a.tilesY, b = uu / (a.tilesX * a.tilesY);
        const int y0 = ty * 4, x0 = tx * 16;
        constexpr int XE = XB ? 2 : 4, ZE = ZB ? 2 : 4;
        const __amdgpu_buffer_rsrc_t dr = b_rsrc(static_cast<const char*>(a.dz) + (int64_t)b * a.dz_bs * ZE, (int64_t)a.Cout * HW * ZE);
        const __amdgpu_buffer_rsrc_t xr = b_rsrc(static_cast<const char*>(a.x) + (int64_t)b * a.x_bs * XE, (int64_t)a.Cin * HW * XE);
        {
            const int yy = y0 + dz_r;
            const bool ok = live && yy < a.H && co0 + dz_c < a.Cout;
            const unsigned base = (unsigned)(((co0 + dz_c) * HW + yy * a.W + x0) * ZE);
            if constexpr (ZB) {       // 16 bf16 = two b128 loads, already in the LDS format
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    dzv[k] = __builtin_amdgcn_raw_buffer_load_b128(dr, (ok && x0 + 8 * k < a.W) ? base + 16 * k : OOB_B, 0, 0);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    dzv[k] = __builtin_amdgcn_raw_buffer_load_b128(dr, (ok && x0 + 4 * k < a.W) ? base + 16 * k : OOB_B, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = tid + 256 * j;                 // row item: ci = e / 6, patch row = e % 6
            const int c = e / 6, r = e % 6;
            const int yy = y0 - 1 + r;
            const bool rok = live && e < 64 * 6 && yy >= 0 && yy < a.H && ci0 + c < a.Cin;
            const unsigned base = (unsigned)(((ci0 + c) * HW + yy * a.W + x0) * XE);
            if constexpr (XB) {       // 16 interior bf16 = two b128 loads (W % 8 == 0: a load is entirely inside or outside the row)
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    xq[j][k] = __builtin_amdgcn_raw_buffer_load_b128(xr, (rok && x0 + 8 * k < a.W) ? base + 16 * k : OOB_B, 0, 0);
                xh[j][0] = __builtin_bit_cast(float, (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(xr, (rok && x0 > 0) ? base - 2 : OOB_B, 0, 0));
                xh[j][1] = __builtin_bit_cast(float, (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(xr, (rok && x0 + 16 < a.W) ? base + 32 : OOB_B, 0, 0));
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    xq[j][k] = __builtin_amdgcn_raw_buffer_load_b128(xr, (rok && x0 + 4 * k < a.W) ? base + 16 * k : OOB_B, 0, 0);
                xh[j][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, (rok && x0 > 0) ? base - 4 : OOB_B, 0, 0));
                xh[j][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, (rok && x0 + 16 < a.W) ? base + 64 : OOB_B, 0, 0));
            }
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
        u32x4b* d = reinterpret_cast<u32x4b*>(dz_lds + dz_c * BW_SDZ + dz_r * 8);
        if constexpr (ZB) {
            d[0] = dzv[0];
            d[1] = dzv[1];
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // (bit-cast the whole vector: __builtin_bit_cast(float, vec[i]) on an element of an ext-vector reads element 0)
                const f32x4b lo = __builtin_bit_cast(f32x4b, dzv[2 * h]), hi = __builtin_bit_cast(f32x4b, dzv[2 * h + 1]);
                u32x4b v = {pack_bf16(lo[0], lo[1]), pack_bf16(lo[2], lo[3]), pack_bf16(hi[0], hi[1]), pack_bf16(hi[2], hi[3])};
                d[h] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = tid + 256 * j;
            if (e < 64 * 6 && XB) {
                // bf16 source: interior dwords q[p] = (f[2p], f[2p+1]); patch dword p = (f[2p-1], f[2p]) = the upper half of q[p-1]
                // and the lower half of q[p] (v_alignbit), dword 0 = (left halo, f[0]), dword 8 = (f[15], right halo)
                unsigned* row = x_lds + (e / 6) * BW_SX + (e % 6) * BW_XROW;
                const u32x4b q0 = xq[j][0], q1 = xq[j][1];
                const unsigned hl = __builtin_bit_cast(unsigned, xh[j][0]), hr = __builtin_bit_cast(unsigned, xh[j][1]);
                const u32x4b w0 = {hl | (q0[0] << 16), __builtin_amdgcn_alignbit(q0[1], q0[0], 16), __builtin_amdgcn_alignbit(q0[2], q0[1], 16),
                                   __builtin_amdgcn_alignbit(q0[3], q0[2], 16)};
                const u32x4b w1 = {__builtin_amdgcn_alignbit(q1[0], q0[3], 16), __builtin_amdgcn_alignbit(q1[1], q1[0], 16),
                                   __builtin_amdgcn_alignbit(q1[2], q1[1], 16), __builtin_amdgcn_alignbit(q1[3], q1[2], 16)};
                *reinterpret_cast<u32x4b*>(row) = w0;
                *reinterpret_cast<u32x4b*>(row + 4) = w1;
                row[8] = (q1[3] >> 16) | (hr << 16);
            } else if (e < 64 * 6) {
                // patch element 0 = column x0 - 1: dword p = (element 2p, 2p + 1) = (f[2p - 1], f[2p]) of the interior row f
                const f32x4b f0 = __builtin_bit_cast(f32x4b, xq[j][0]), f1 = __builtin_bit_cast(f32x4b, xq[j][1]);
                const f32x4b f2 = __builtin_bit_cast(f32x4b, xq[j][2]), f3 = __builtin_bit_cast(f32x4b, xq[j][3]);
                unsigned* row = x_lds + (e / 6) * BW_SX + (e % 6) * BW_XROW;
                const u32x4b w0 = {pack_bf16(xh[j][0], f0[0]), pack_bf16(f0[1], f0[2]), pack_bf16(f0[3], f1[0]), pack_bf16(f1[1], f1[2])};
                const u32x4b w1 = {pack_bf16(f1[3], f2[0]), pack_bf16(f2[1], f2[2]), pack_bf16(f2[3], f3[0]), pack_bf16(f3[1], f3[2])};
                *reinterpret_cast<u32x4b*>(row) = w0;
                *reinterpret_cast<u32x4b*>(row + 4) = w1;
                row[8] = pack_bf16(f3[3], xh[j][1]);
            }
        }
    };

    const unsigned* a_ptr = dz_lds + (wm * 32 + l31) * BW_SDZ + kh * 4;
    const unsigned* b_ptr = x_lds + (wn * 32 + l31) * BW_SX + kh * 4;

    issue(u0);
    for (int u = u0; u < u1; ++u) {
        commit();
        __syncthreads();
        issue(u + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 av = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4b*>(a_ptr + s * 8));
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const u32x4b q = *reinterpret_cast<const u32x4b*>(b_ptr + (s + ky) * BW_XROW);
                const unsigned d4 = b_ptr[(s + ky) * BW_XROW + 4];
                const u32x4b s1 = {__builtin_amdgcn_alignbit(q[1], q[0], 16), __builtin_amdgcn_alignbit(q[2], q[1], 16),
                                   __builtin_amdgcn_alignbit(q[3], q[2], 16), __builtin_amdgcn_alignbit(d4, q[3], 16)};
                const u32x4b s2 = {q[1], q[2], q[3], d4};
                acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, q), acc[ky * 3 + 0], 0, 0, 0);
                acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, s1), acc[ky * 3 + 1], 0, 0, 0);
                acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, s2), acc[ky * 3 + 2], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    const int64_t n = (int64_t)a.Cout * a.Cin;
    const int ci = ci0 + wn * 32 + l31;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float* o = a.slab + ((int64_t)ks * 9 + t) * n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (co < a.Cout && ci < a.Cin) o[(int64_t)co * a.Cin + ci] = acc[t][r];
        }
    }
}

// ---- weight gradient on maps at least 64 pixels wide: ROW units.  The 4-row x 16-pixel patches above fetch 32-byte pieces of
// 128-byte lines (a quarter of every line per unit, six halo rows for four): on the 256x256 and 128x128 levels the kernel was
// bound by those requests, not by the matrix pipe (without the global loads 0.31 instead of 0.77 ms at 64 channels; without
// the MFMAs 0.75).  Here a unit is ONE image row of a 64-pixel strip: dz [64 co][64 px] and x rows y-1, y, y+1 [64 ci][66 px];
// a block walks down the strip, so the x rows live in a 4-slot ring in LDS and every unit loads exactly one new row of each
// operand as whole 128-byte lines.  The four K-steps of a unit are its four 16-pixel segments; fragments, shifts and the MFMA
// mapping are those of the patch kernel.
constexpr int BR_SDZ = 36, BR_SX = 148, BR_SLOT = 36;      // dword strides: per co; per ci (4 x odd); per ring slot

template <bool XB = false, bool ZB = false>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_bf16_row_kernel(BwArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned dz_lds[2 * 64 * BR_SDZ];
    __shared__ __attribute__((aligned(16))) unsigned x_lds[64 * BR_SX];

    int bid;
    {
        const int n = gridDim.x, q = n >> 3, r = n & 7, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tiles = a.ciTiles * a.coTiles;
    const int ks = bid / tiles, tile = bid % tiles;
    const int ci0 = (tile % a.ciTiles) * 64, co0 = (tile / a.ciTiles) * 64;
    const int nunits = a.B * a.tilesX * a.H;                   // unit = (image, 64-pixel strip, row), rows fastest
    const int per = (nunits + a.splitK - 1) / a.splitK;
    const int u0 = ks * per, u1 = min(u0 + per, nunits);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int l31 = lane & 31, kh = lane >> 5;
    const int HW = a.H * a.W;
    constexpr int XE = XB ? 2 : 4, ZE = ZB ? 2 : 4;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // staging roles: thread = (channel = tid / 4, 16-pixel segment = tid % 4) of the dz row and of the x row.  NS register sets:
    // with both operands in bf16 (18 registers a set) the loads run TWO units ahead of the MFMAs -- a unit is 36 MFMAs = 0.5 us
    // of matrix time, a miss to HBM several times that
    constexpr int NS = (XB && ZB) ? 2 : 1;
    const int st_c = tid >> 2, st_s = tid & 3;
    u32x4b dzv[NS][4], xq[NS][4];
    float xh[NS][2];
    // loads of one dz row and / or one x row of strip (b, x0): row < 0 or >= H -> zeros
    auto issue_dz = [&](int set, int b, int x0, int y) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t dr = b_rsrc(static_cast<const char*>(a.dz) + (int64_t)b * a.dz_bs * ZE, (int64_t)a.Cout * HW * ZE);
        const int xs = x0 + 16 * st_s;
        const bool ok = y >= 0 && y < a.H && co0 + st_c < a.Cout;
        const unsigned base = (unsigned)(((co0 + st_c) * HW + y * a.W + xs) * ZE);
#pragma unroll
        for (int k = 0; k < (ZB ? 2 : 4); ++k)
            dzv[set][k] = __builtin_amdgcn_raw_buffer_load_b128(dr, (ok && xs + (ZB ? 8 : 4) * k < a.W) ? base + 16 * k : OOB_B, 0, 0);
    };
    auto issue_x = [&](int set, int b, int x0, int y) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t xr = b_rsrc(static_cast<const char*>(a.x) + (int64_t)b * a.x_bs * XE, (int64_t)a.Cin * HW * XE);
        const int xs = x0 + 16 * st_s;
        const bool ok = y >= 0 && y < a.H && ci0 + st_c < a.Cin;
        const unsigned base = (unsigned)(((ci0 + st_c) * HW + y * a.W + xs) * XE);
#pragma unroll
        for (int k = 0; k < (XB ? 2 : 4); ++k)
            xq[set][k] = __builtin_amdgcn_raw_buffer_load_b128(xr, (ok && xs + (XB ? 8 : 4) * k < a.W) ? base + 16 * k : OOB_B, 0, 0);
        if constexpr (XB) {
            xh[set][0] = __builtin_bit_cast(float, (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(xr, (ok && xs > 0 && xs - 1 < a.W) ? base - 2 : OOB_B, 0, 0));
            xh[set][1] = __builtin_bit_cast(float, (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(xr, (ok && st_s == 3 && xs + 16 < a.W) ? base + 32 : OOB_B, 0, 0));
        } else {
            xh[set][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, (ok && xs > 0 && xs - 1 < a.W) ? base - 4 : OOB_B, 0, 0));
            xh[set][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, (ok && st_s == 3 && xs + 16 < a.W) ? base + 64 : OOB_B, 0, 0));
        }
    };
    auto commit_dz = [&](int set, int buf) __attribute__((always_inline)) {
        u32x4b* d = reinterpret_cast<u32x4b*>(dz_lds + buf * 64 * BR_SDZ + st_c * BR_SDZ + st_s * 8);
        if constexpr (ZB) {
            d[0] = dzv[set][0];
            d[1] = dzv[set][1];
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4b lo = __builtin_bit_cast(f32x4b, dzv[set][2 * h]), hi = __builtin_bit_cast(f32x4b, dzv[set][2 * h + 1]);
                d[h] = u32x4b{pack_bf16(lo[0], lo[1]), pack_bf16(lo[2], lo[3]), pack_bf16(hi[0], hi[1]), pack_bf16(hi[2], hi[3])};
            }
        }
    };
    // row element e = column x0 - 1 + e; dword p = (e[2p], e[2p+1]) = (f[2p-1], f[2p]) of the interior row f: the segment's eight
    // dwords start with (pixel left of the segment, its first pixel); the strip's last dword (f[63], right halo) is segment 3's
    auto commit_x = [&](int set, int slot) __attribute__((always_inline)) {
        unsigned* row = x_lds + st_c * BR_SX + slot * BR_SLOT + st_s * 8;
        if constexpr (XB) {
            const u32x4b q0 = xq[set][0], q1 = xq[set][1];
            const unsigned hl = __builtin_bit_cast(unsigned, xh[set][0]), hr = __builtin_bit_cast(unsigned, xh[set][1]);
            const u32x4b w0 = {hl | (q0[0] << 16), __builtin_amdgcn_alignbit(q0[1], q0[0], 16), __builtin_amdgcn_alignbit(q0[2], q0[1], 16),
                               __builtin_amdgcn_alignbit(q0[3], q0[2], 16)};
            const u32x4b w1 = {__builtin_amdgcn_alignbit(q1[0], q0[3], 16), __builtin_amdgcn_alignbit(q1[1], q1[0], 16),
                               __builtin_amdgcn_alignbit(q1[2], q1[1], 16), __builtin_amdgcn_alignbit(q1[3], q1[2], 16)};
            *reinterpret_cast<u32x4b*>(row) = w0;
            *reinterpret_cast<u32x4b*>(row + 4) = w1;
            if (st_s == 3) row[8] = (q1[3] >> 16) | (hr << 16);
        } else {
            const f32x4b f0 = __builtin_bit_cast(f32x4b, xq[set][0]), f1 = __builtin_bit_cast(f32x4b, xq[set][1]);
            const f32x4b f2 = __builtin_bit_cast(f32x4b, xq[set][2]), f3 = __builtin_bit_cast(f32x4b, xq[set][3]);
            const u32x4b w0 = {pack_bf16(xh[set][0], f0[0]), pack_bf16(f0[1], f0[2]), pack_bf16(f0[3], f1[0]), pack_bf16(f1[1], f1[2])};
            const u32x4b w1 = {pack_bf16(f1[3], f2[0]), pack_bf16(f2[1], f2[2]), pack_bf16(f2[3], f3[0]), pack_bf16(f3[1], f3[2])};
            *reinterpret_cast<u32x4b*>(row) = w0;
            *reinterpret_cast<u32x4b*>(row + 4) = w1;
            if (st_s == 3) row[8] = pack_bf16(f3[3], xh[set][1]);
        }
    };

    const unsigned* a_ptr = dz_lds + (wm * 32 + l31) * BR_SDZ + kh * 4;
    const unsigned* b_ptr = x_lds + (wn * 32 + l31) * BR_SX + kh * 4;

    int u = u0;
    while (u < u1) {
        // a run of rows inside one strip: y = yb .. ye - 1
        const int yb = u % a.H, sb = u / a.H;
        const int tx = sb % a.tilesX, b = sb / a.tilesX;
        const int x0 = tx * 64;
        const int ye = min(a.H, yb + (u1 - u));
        // run prologue: rows yb - 1 and yb of x into the ring (synchronously), then the dz row and the next x row of the first NS
        // units in flight
        __syncthreads();                              // every wave is done with the previous run's ring and dz buffers
        issue_x(0, b, x0, yb - 1);
        commit_x(0, (yb - 1) & 3);
        issue_x(0, b, x0, yb);
        commit_x(0, yb & 3);
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            issue_dz(k, b, x0, yb + k);               // (rows past the run's end are loaded and never used)
            issue_x(k, b, x0, yb + k + 1);
        }
        for (int yq = yb; yq < ye; yq += NS) {
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int y = yq + k;
                if (y >= ye) break;
                const int buf = y & 1;
                commit_dz(k, buf);
                commit_x(k, (y + 1) & 3);             // slot of row y - 3: last read two barriers ago
                __syncthreads();
                if (y + NS < ye) {
                    issue_dz(k, b, x0, y + NS);
                    issue_x(k, b, x0, y + NS + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
                const unsigned* ab = a_ptr + buf * 64 * BR_SDZ;
                // segments last to first: the fifth dword of a lane's shifted fragments is the first dword of the next 8-pixel
                // group -- the kh = 1 partner's of the same read for the kh = 0 half, the kh = 0 lane's of the segment read one
                // step earlier for the kh = 1 half: v_permlane32_swap instead of a ds_read_b32 that is a 4-way bank conflict on
                // a row stride of 4 x odd dwords (conv_split.hip's weight gradient: LDS cycles / 5); only the strip's last
                // dword still comes from LDS
                unsigned lo_prev[3] = {0, 0, 0};
#pragma unroll
                for (int sgi = 0; sgi < 4; ++sgi) {
                    const int sg = 3 - sgi;
                    const bf16x8 av = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4b*>(ab + sg * 8));
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const unsigned* br = b_ptr + ((y - 1 + ky) & 3) * BR_SLOT + sg * 8;
                        const u32x4b q = *reinterpret_cast<const u32x4b*>(br);
                        const auto sw = __builtin_amdgcn_permlane32_swap(q[0], q[0], false, false);
                        const unsigned nxt = sg == 3 ? br[4] : lo_prev[ky];
                        const unsigned d4 = kh ? nxt : sw[1];
                        lo_prev[ky] = sw[0];
                        const u32x4b s1 = {__builtin_amdgcn_alignbit(q[1], q[0], 16), __builtin_amdgcn_alignbit(q[2], q[1], 16),
                                           __builtin_amdgcn_alignbit(q[3], q[2], 16), __builtin_amdgcn_alignbit(d4, q[3], 16)};
                        const u32x4b s2 = {q[1], q[2], q[3], d4};
                        acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, q), acc[ky * 3 + 0], 0, 0, 0);
                        acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, s1), acc[ky * 3 + 1], 0, 0, 0);
                        acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, s2), acc[ky * 3 + 2], 0, 0, 0);
                    }
                }
            }
        }
        u += ye - yb;
    }

    const int64_t n = (int64_t)a.Cout * a.Cin;
    const int ci = ci0 + wn * 32 + l31;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float* o = a.slab + ((int64_t)ks * 9 + t) * n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (co < a.Cout && ci < a.Cin) o[(int64_t)co * a.Cin + ci] = acc[t][r];
        }
    }
}

static bool wgrad_bf16_rows(int W) {
    static int mode = -1;
    if (mode < 0) { const char* e = getenv("ONET_BF16_WG_ROWS"); mode = (e && e[0] == '0') ? 0 : 1; }
    return mode == 1 && W >= 64;
}

static void wgrad_bf16_plan(int B, int Cin, int Cout, int H, int W, int& splitK, int& tilesY, int& tilesX) {
    const bool rows = wgrad_bf16_rows(W);
    tilesY = rows ? H : cdiv(H, 4);
    tilesX = rows ? cdiv(W, 64) : cdiv(W, 16);
    const int64_t units = (int64_t)B * tilesY * tilesX;
    const int tiles = cdiv(Cin, 64) * cdiv(Cout, 64);
    static int target = -1;
    if (target < 0) { const char* e = getenv("ONET_BF16_WG_BLOCKS"); target = (e && atoi(e) > 0) ? atoi(e) : 512; }
    int64_t k = std::max<int64_t>(1, target / tiles);        // blocks over the launch (2 resident per CU)
    k = std::min<int64_t>(k, std::max<int64_t>(1, units / 16)); // at least 16 units (64 K-steps) per block
    splitK = (int)k;
}

extern "C" {

int onet_conv3x3_pack_weights_bf16(const float* w, void* wq_fwd, void* wq_dgrad, int Cout, int Cin, void* stream) {
    ONET_REQUIRE(w && (wq_fwd || wq_dgrad) && Cout > 0 && Cin > 0, "conv3x3_pack_weights_bf16: bad args");
    ONET_REQUIRE(!wq_fwd || (Cin % 16) == 0, "conv3x3_pack_weights_bf16: Cin must be a multiple of 16 for the forward pack");
    const int64_t n = (int64_t)std::max(Cin, ((Cout + 15) / 16) * 16) * 9 * std::max(Cout, Cin);
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 8192);
    if (wq_fwd) {
        hipLaunchKernelGGL(pack3x3_bf16_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w, (__bf16*)wq_fwd, (__bf16*)nullptr, Cout, Cin, 0);
        int rc = check_launch("pack3x3_bf16_kernel");
        if (rc) return rc;
    }
    if (wq_dgrad)
        hipLaunchKernelGGL(pack3x3_bf16_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w, (__bf16*)nullptr, (__bf16*)wq_dgrad, Cout, Cin, 1);
    return check_launch("pack3x3_bf16_kernel");
}

// records per channel of the fused statistics (0: the map is not made of full tiles -- separate statistics pass)
static int bf16_nparts(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const int tw = W > 16 ? 32 : 16, rows = W > 16 ? 8 : 16;
    if (W % tw || H % rows) return 0;
    const int64_t n = (int64_t)B * (H / rows) * (W / tw);
    return n < (1 << 30) ? (int)n : 0;
}

static int bf16_fwd(const void* x, int x_bf16, int64_t x_bs, const void* wq, float* z, int64_t z_bs, int B, int Cin, int Cout,
                    int H, int W, void* stream, float* stats = nullptr) {
    ONET_REQUIRE(x && wq && z, "conv3x3_bf16_fwd: null pointer");
    ONET_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "conv3x3_bf16_fwd: bad shape");
    ONET_REQUIRE((Cin % 16) == 0 && (Cout % 4) == 0, "conv3x3_bf16_fwd: Cin must be a multiple of 16, Cout of 4 (use onet_conv_fwd)");
    ONET_REQUIRE(x_bs >= (int64_t)Cin * H * W && z_bs >= (int64_t)Cout * H * W, "conv3x3_bf16_fwd: batch stride too small");
    ONET_REQUIRE((int64_t)(Cin + 32) * H * W * 4 < (1ll << 31) && (int64_t)(Cin + 32) * 9 * Cout * 2 < (1ll << 31),
                 "conv3x3_bf16_fwd: operand exceeds the 2 GiB buffer-resource range");
    BfArgs a{x, x_bs, (const __bf16*)wq, z, z_bs, B, Cin, Cout, H, W, 0, 0, 0, stats};
    if (stats) {
        ONET_REQUIRE(bf16_nparts(B, H, W) > 0, "conv3x3_bf16_fwd_stats: the map must be made of full tiles (onet_conv3x3_bf16_nparts() == 0 elsewhere)");
        if (x_bf16) return (W > 16) ? launch_bf16<2, 2, 32, 1, true>(a, as_stream(stream)) : launch_bf16<2, 2, 16, 1, true>(a, as_stream(stream));
        return (W > 16) ? launch_bf16<2, 2, 32, 0, true>(a, as_stream(stream)) : launch_bf16<2, 2, 16, 0, true>(a, as_stream(stream));
    }
    // 4 waves x 2 rows x 32 px, two blocks (8 waves) per CU: 373-851 TF on the U-Net's layers against 278-527 for
    // 4-row waves at one wave per SIMD and ~100 for 4-row waves squeezed into 256 VGPRs (700 B/lane of scratch)
    // (three blocks per CU for the bf16-input kernel -- 168 VGPRs -- changed nothing: 0.412 vs 0.410 ms per launch)
    if (x_bf16 == 2) return (W > 16) ? launch_bf16<2, 2, 32, 2>(a, as_stream(stream)) : launch_bf16<2, 2, 16, 2>(a, as_stream(stream));
    if (x_bf16) return (W > 16) ? launch_bf16<2, 2, 32, 1>(a, as_stream(stream)) : launch_bf16<2, 2, 16, 1>(a, as_stream(stream));
    return (W > 16) ? launch_bf16<2, 2>(a, as_stream(stream)) : launch_bf16<2, 2, 16>(a, as_stream(stream));
}

int onet_conv3x3_bf16_fwd(const float* x, int64_t x_bs, const void* wq, float* z, int64_t z_bs, int B, int Cin, int Cout,
                          int H, int W, void* stream) {
    return bf16_fwd(x, 0, x_bs, wq, z, z_bs, B, Cin, Cout, H, W, stream);
}

int onet_conv3x3_bf16_fwd_b(const void* x_bf16, int64_t x_bs, const void* wq, float* z, int64_t z_bs, int B, int Cin, int Cout,
                            int H, int W, void* stream) {
    return bf16_fwd(x_bf16, 1, x_bs, wq, z, z_bs, B, Cin, Cout, H, W, stream);
}

// EXPERIMENT (round 3 groundwork): the same forward with the bf16 copy in the channel-blocked layout [C/8][H][W][8]
int onet_conv3x3_bf16_fwd_blk(const void* x_blk, int64_t x_bs, const void* wq, float* z, int64_t z_bs, int B, int Cin, int Cout,
                              int H, int W, void* stream) {
    ONET_REQUIRE((reinterpret_cast<uintptr_t>(x_blk) & 15) == 0 && (x_bs & 7) == 0, "conv3x3_bf16_fwd_blk: 16-byte aligned slots required");
    return bf16_fwd(x_blk, 2, x_bs, wq, z, z_bs, B, Cin, Cout, H, W, stream);
}

int onet_conv3x3_bf16_nparts(int B, int H, int W) { return bf16_nparts(B, H, W); }

int onet_conv3x3_bf16_fwd_stats(const void* x, int x_is_bf16, int64_t x_bs, const void* wq, float* z, int64_t z_bs, float* part, int B,
                                int Cin, int Cout, int H, int W, void* stream) {
    ONET_REQUIRE(part, "conv3x3_bf16_fwd_stats: null pointer");
    return bf16_fwd(x, x_is_bf16 != 0 ? 1 : 0, x_bs, wq, z, z_bs, B, Cin, Cout, H, W, stream, part);
}

int64_t onet_conv3x3_wgrad_bf16_ws_bytes(int B, int Cin, int Cout, int H, int W) {
    int splitK, ty, tx;
    wgrad_bf16_plan(B, Cin, Cout, H, W, splitK, ty, tx);
    return (int64_t)splitK * 9 * Cout * Cin * 4;
}

static int bf16_wgrad(const void* x, bool xb, int64_t x_bs, const void* dz, bool zb, int64_t dz_bs, float* dw, void* ws, int64_t ws_bytes,
                      int B, int Cin, int Cout, int H, int W, int accumulate, void* stream) {
    ONET_REQUIRE(x && dz && dw && ws, "conv3x3_wgrad_bf16: null pointer");
    ONET_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "conv3x3_wgrad_bf16: bad shape");
    ONET_REQUIRE((W & 3) == 0 && (dz_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(dz) & 15) == 0,
                 "conv3x3_wgrad_bf16: W %% 4 == 0 and 16-byte aligned dz rows required (use onet_conv_wgrad)");
    ONET_REQUIRE(!(xb || zb) || ((W & 7) == 0 && (x_bs & 7) == 0 && (dz_bs & 7) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0),
                 "conv3x3_wgrad_bf16: bf16 operands need W %% 8 == 0 and 16-byte aligned rows");
    ONET_REQUIRE(x_bs >= (int64_t)Cin * H * W && dz_bs >= (int64_t)Cout * H * W, "conv3x3_wgrad_bf16: batch stride too small");
    ONET_REQUIRE((int64_t)std::max(Cin, Cout) * H * W * 4 < (1ll << 31), "conv3x3_wgrad_bf16: image exceeds the 2 GiB buffer-resource range");
    BwArgs a{x, x_bs, dz, dz_bs, (float*)ws, B, Cin, Cout, H, W, cdiv(Cin, 64), cdiv(Cout, 64), 1, 1, 1};
    wgrad_bf16_plan(B, Cin, Cout, H, W, a.splitK, a.tilesY, a.tilesX);
    const int64_t need = (int64_t)a.splitK * 9 * Cout * Cin * 4;
    ONET_REQUIRE(ws_bytes >= need, "conv3x3_wgrad_bf16: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)need);
    const int64_t blocks = (int64_t)a.splitK * a.ciTiles * a.coTiles;
    const dim3 g((unsigned)blocks), t(256);
    if (wgrad_bf16_rows(W)) {
        if (xb && zb) hipLaunchKernelGGL((conv3x3_wgrad_bf16_row_kernel<true, true>), g, t, 0, as_stream(stream), a);
        else if (xb) hipLaunchKernelGGL((conv3x3_wgrad_bf16_row_kernel<true, false>), g, t, 0, as_stream(stream), a);
        else if (zb) hipLaunchKernelGGL((conv3x3_wgrad_bf16_row_kernel<false, true>), g, t, 0, as_stream(stream), a);
        else hipLaunchKernelGGL((conv3x3_wgrad_bf16_row_kernel<false, false>), g, t, 0, as_stream(stream), a);
    } else
    if (xb && zb) hipLaunchKernelGGL((conv3x3_wgrad_bf16_kernel<true, true>), g, t, 0, as_stream(stream), a);
    else if (xb) hipLaunchKernelGGL((conv3x3_wgrad_bf16_kernel<true, false>), g, t, 0, as_stream(stream), a);
    else if (zb) hipLaunchKernelGGL((conv3x3_wgrad_bf16_kernel<false, true>), g, t, 0, as_stream(stream), a);
    else hipLaunchKernelGGL((conv3x3_wgrad_bf16_kernel<false, false>), g, t, 0, as_stream(stream), a);
    int rc = check_launch("conv3x3_wgrad_bf16_kernel");
    if (rc) return rc;
    return launch_wgrad_reduce((const float*)ws, dw, a.splitK, 9, Cout, Cin, 0, accumulate, as_stream(stream));
}

int onet_conv3x3_wgrad_bf16(const float* x, int64_t x_bs, const float* dz, int64_t dz_bs, float* dw, void* ws, int64_t ws_bytes,
                            int B, int Cin, int Cout, int H, int W, int accumulate, void* stream) {
    return bf16_wgrad(x, false, x_bs, dz, false, dz_bs, dw, ws, ws_bytes, B, Cin, Cout, H, W, accumulate, stream);
}

int onet_conv3x3_wgrad_bf16_b(const void* x, int x_is_bf16, int64_t x_bs, const void* dz, int dz_is_bf16, int64_t dz_bs, float* dw,
                              void* ws, int64_t ws_bytes, int B, int Cin, int Cout, int H, int W, int accumulate, void* stream) {
    return bf16_wgrad(x, x_is_bf16 != 0, x_bs, dz, dz_is_bf16 != 0, dz_bs, dw, ws, ws_bytes, B, Cin, Cout, H, W, accumulate, stream);
}

}  // extern "C"
