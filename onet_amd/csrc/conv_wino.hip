// K1 (fast path): 3x3 convolution forward / dgrad by Winograd F(2x2, 3x3) on the fp32 matrix cores.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        (Lavin & Gray 2016; 2.25x fewer multiplies)
//
// Replaces the same call sites as conv_mfma.hip (F.conv2d + input-grad, OV:47,51).  The 16 Winograd
// positions become 16 independent GEMMs  M_pos[co][tile] = sum_ci V_pos[co][ci] * U_pos[ci][tile],
// i.e. per wave SIXTEEN 32x32 MFMA accumulators (256 VGPRs, one wave per SIMD):
//   rows  = 32 output channels, columns = 32 Winograd tiles (2x2 output pixels each), K = channels.
// * weights are pre-transformed once per optimizer step (V = G g G^T, packed [ci][16][co]);
// * the input transform B^T d B is done in registers, per lane, from the SAME zero-padded LDS halo
//   tile the direct kernel stages (8 ds_read_b64 + 32 adds per channel pair), so no transformed
//   tensor ever touches HBM;
// * the output transform A^T M A runs on the accumulators in the epilogue (24 adds per tile) and
//   stores float2 pixel pairs (lanes = neighbouring tiles -> 128-B coalesced rows).
// Same software pipeline as the direct kernel (buffer-load prefetch of chunk c+1 under the MFMAs of
// chunk c, hardware range check = zero padding) and the same XCD-aware tile order.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "common.hpp"

using namespace onet;

#ifndef ONET_WW_ASYM
#define ONET_WW_ASYM 1
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB_OFF_W = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wmake_rsrc(const void* base, int64_t bytes) {
    const int n = bytes > 0x7fffffffll ? 0x7fffffff : (int)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, n, 0x00020000);
}
__device__ __forceinline__ float wbload(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ u32x4 wbload4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
}
// LDS-DMA: 64 lanes x 16 B land at lds_byte_addr + lane*16 (wave-uniform base in M0, lane-linear image).
// Inline asm on purpose: after the builtin form (__builtin_amdgcn_raw_ptr_buffer_load_lds) hipcc puts
// s_waitcnt vmcnt(0) in front of the next ds_read -- it cannot tell the two LDS buffers apart -- which
// exposes the whole DMA latency every chunk (measured: 215 -> 187 TFLOP/s).  The asm load is invisible
// to its bookkeeping; it is retired by the explicit vmcnt(0) in front of the chunk barrier
// (cdna_hip_programming.md §5.7).  M0 is saved and restored inside the statement.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16_to_lds(i32x4 rsrc, unsigned lds_byte_addr, unsigned voff) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(lds_byte_addr)
                 : "memory");
}
__device__ __forceinline__ i32x4 make_rsrc_words(const void* base, int64_t bytes) {
    const uint64_t p = reinterpret_cast<uint64_t>(base);
    i32x4 r;
    r.x = (int)(p & 0xffffffffu);
    r.y = (int)((p >> 32) & 0xffffu);                     // stride 0
    r.z = bytes > 0x7fffffffll ? 0x7fffffff : (int)bytes; // num_records (bytes)
    r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ unsigned lds_addr_of(const float* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)p;
}

// ------------------------------------------------------------------ weight transform + packing
// w [Cout][Cin][3][3] -> wq_fwd [Cin][16][Cout] = G g G^T ;  wq_dgrad [Cout][16][Cin] = G g' G^T with
// g' = g rotated by 180 degrees (dgrad = convolution of dZ with the flipped, transposed filter)
__device__ __forceinline__ void wino_G(const float g[3][3], float V[4][4]) {
    float r[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        r[0][j] = g[0][j];
        r[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
        r[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
        r[3][j] = g[2][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        V[i][0] = r[i][0];
        V[i][1] = 0.5f * (r[i][0] + r[i][1] + r[i][2]);
        V[i][2] = 0.5f * (r[i][0] - r[i][1] + r[i][2]);
        V[i][3] = r[i][2];
    }
}

__global__ void pack3x3_wino_kernel(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wd,
                                    int Cout, int Cin) {
    const int64_t n = (int64_t)Cout * Cin;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin), co = (int)(i / Cin);
        float g[3][3], gr[3][3], V[4][4];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                g[a][b] = w[i * 9 + a * 3 + b];
                gr[2 - a][2 - b] = g[a][b];
            }
        if (wf) {
            wino_G(g, V);
#pragma unroll
            for (int p = 0; p < 16; ++p) wf[((int64_t)ci * 16 + p) * Cout + co] = V[p >> 2][p & 3];
        }
        if (wd) {
            wino_G(gr, V);
#pragma unroll
            for (int p = 0; p < 16; ++p) wd[((int64_t)co * 16 + p) * Cin + ci] = V[p >> 2][p & 3];
        }
    }
}

// ------------------------------------------------------------------ forward / dgrad
struct WinoArgs {
    const float* x;
    int64_t x_bs;
    const float* wq;      // [Cin][16][Cout]
    float* z;
    int64_t z_bs;
    int B, Cin, Cout, H, W, tilesX, tilesY, coTiles;
};

template <int TW, int CI_T, int WM>
struct WinoCfg {
    static constexpr int TC = TW / 2;                 // Winograd tile columns per block (= per wave)
    static constexpr int TRW = 32 / TC;               // tile rows per wave (32 tiles per wave)
    static constexpr int ROWS = 2 * TRW * 2;          // output pixel rows per block (2 waves along rows)
    static constexpr int IN_ROWS = ROWS + 2, IN_COLS = TW + 2;
    static constexpr int RS = (TW == 32) ? 48 : 24;   // LDS row stride: 2*RS*tr spreads the tile rows over
                                                      // disjoint bank ranges for the ds_read_b64 patches
    static constexpr int CH_STRIDE = IN_ROWS * RS;
    static constexpr int CO_T = 32 * WM;              // WM = 2: 8-wave block, 1 per CU;  WM = 1: 4-wave block, 2 per CU
    static constexpr int NTHR = 256 * WM;
    static constexpr int W_FLOATS = CI_T * 16 * CO_T;
    static constexpr int IN_LOGICAL = CI_T * IN_ROWS * IN_COLS;
    static constexpr int BUF_FLOATS = W_FLOATS + CI_T * CH_STRIDE;       // one staging buffer (weights + input tile)
    static constexpr int LDS_BYTES = 2 * BUF_FLOATS * 4;                 // double-buffered
};

// Block = 8 waves = 2 position halves x 2 channel halves x 2 tile halves: a wave owns the Winograd rows
// xi in {2*ph, 2*ph+1} (8 of the 16 positions, 128 accumulator VGPRs), so two waves share a SIMD and
// hide each other's LDS / transform latency, and every wave keeps registers for operand prefetch.
// The output transform is linear in the positions: each half applies A^T . A to its own rows and the
// two partial 2x2 outputs are added through LDS in the epilogue.
// DBG != 0: timing-only ablation builds (tools/ablate_wino.py); outputs are garbage.
//   bit0: no global staging   bit1: no input transform   bit2: no A-operand LDS reads   bit3: no patch LDS reads
template <int TW, int CI_T, int WM, int DBG = 0, bool PIPE = true>
__global__ __launch_bounds__(256 * WM, 2) void conv_wino_kernel(WinoArgs a) {
    using C = WinoCfg<TW, CI_T, WM>;
    constexpr int TC = C::TC, TRW = C::TRW, ROWS = C::ROWS, IN_ROWS = C::IN_ROWS, IN_COLS = C::IN_COLS;
    constexpr int RS = C::RS, CH_STRIDE = C::CH_STRIDE, CO_T = C::CO_T, W_FLOATS = C::W_FLOATS;
    constexpr int NTHR = C::NTHR;
    constexpr int NIN = (C::IN_LOGICAL + NTHR - 1) / NTHR;
    constexpr int NW4 = (W_FLOATS / 4 + NTHR - 1) / NTHR;
    static_assert(C::LDS_BYTES >= 2 * WM * 64 * 64 * 4, "epilogue exchange buffer must fit the staging LDS");
    static_assert(W_FLOATS % (64 * 4) == 0, "weights are staged in whole 1-KB LDS-DMA pieces");

    extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 x { w [CI_T][16][CO_T], in [CI_T][IN_ROWS][RS] }

    int bid;
    {   // XCD-aware tile order (see conv_mfma.hip)
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
    const int ph = wid / (2 * WM), wm = (wid >> 1) & (WM - 1), wn = wid & 1;
    const int l31 = lane & 31, kh = lane >> 5;
    const int tc = l31 % TC, tr = l31 / TC;           // this lane's Winograd tile within the wave
    const int HW = a.H * a.W;

    f32x16 acc[8];                                    // positions (2*ph + i)*4 + j  ->  acc[i*4 + j]
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    const int a_idx = (kh * 16 + ph * 8) * CO_T + wm * 32 + l31;
    // rows of the 4x4 patch this half needs: xi = 0,1 use d0,d1,d2 ; xi = 2,3 use d1,d2,d3
    const int p_idx = W_FLOATS + kh * CH_STRIDE + ((wn * TRW + tr) * 2 + ph) * RS + tc * 2;

    const __amdgpu_buffer_rsrc_t xr = wmake_rsrc(a.x + (int64_t)b * a.x_bs, (int64_t)a.Cin * HW * 4);
    const i32x4 wr4 = make_rsrc_words(a.wq, (int64_t)a.Cin * 16 * a.Cout * 4);

    unsigned in_off[NIN];
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
        const int i = tid + NTHR * k;
        const int ci = i / (IN_ROWS * IN_COLS), rem = i % (IN_ROWS * IN_COLS);
        const int r = rem / IN_COLS, c = rem % IN_COLS;
        const int yy = y0 - 1 + r, xx = x0 - 1 + c;
        const bool ok = (i < C::IN_LOGICAL) && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
        in_off[k] = ok ? (unsigned)((ci * HW + yy * a.W + xx) * 4) : OOB_OFF_W;
    }
    unsigned w_off[NW4];
#pragma unroll
    for (int k = 0; k < NW4; ++k) {
        const int i = (tid + NTHR * k) * 4;
        const int ci = i / (16 * CO_T), rem = i % (16 * CO_T);
        const int p = rem / CO_T, co = rem % CO_T;
        const bool ok = (i < W_FLOATS) && (co0 + co < a.Cout);
        w_off[k] = ok ? (unsigned)(((ci * 16 + p) * a.Cout + co0 + co) * 4) : OOB_OFF_W;
    }
    const unsigned in_step = (unsigned)(CI_T * HW * 4), w_step = (unsigned)(CI_T * 16 * a.Cout * 4);

    // Staging: the packed weight slice (32 KB per chunk, the larger operand) goes global -> LDS by
    // LDS-DMA (buffer_load ... lds: no VGPRs, no ds_write pass; the LDS image is lane-linear, which is
    // exactly the [ci][pos][co] copy order); the halo input tile (irregular, zero-padded) goes
    // through 6 prefetch VGPRs.  Out-of-range lanes of either kind read 0 by the buffer range check.
    const int wid_u = __builtin_amdgcn_readfirstlane(wid);
    float xin[NIN];
    auto issue = [&](unsigned cin_bytes, unsigned cw_bytes, float* buf) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NIN; ++k) xin[k] = wbload(xr, in_off[k] + cin_bytes);
#pragma unroll
        for (int k = 0; k < NW4; ++k) {
            if ((wid_u * 64 + NTHR * k) * 4 < W_FLOATS)      // wave-uniform: whole 1-KB pieces only
                dma16_to_lds(wr4, lds_addr_of(buf) + (unsigned)((wid_u * 64 + NTHR * k) * 16), w_off[k] + cw_bytes);
        }
    };
    auto commit = [&](float* buf) __attribute__((always_inline)) {
        float* in_lds = buf + W_FLOATS;
#pragma unroll
        for (int k = 0; k < NIN; ++k) {
            const int i = tid + NTHR * k;
            const int ci = i / (IN_ROWS * IN_COLS), rem = i % (IN_ROWS * IN_COLS);
            const int r = rem / IN_COLS, c = rem % IN_COLS;
            if (i < C::IN_LOGICAL) in_lds[ci * CH_STRIDE + r * RS + c] = xin[k];
        }
    };
    // One K-step = one channel pair: 3 patch rows (6 ds_read_b64), 16 adds, 8 A reads, 8 MFMAs.
    // PH and the LDS buffer are compile-time (specialised code per position half, immediate LDS
    // offsets): with two waves per SIMD the VALU issue slots between MFMAs are the scarce resource
    // (an fp32 MFMA leaves room for ~14 VALU issues per 64-cycle slot), so nothing may be spent on
    // selects or address arithmetic.
    auto mfma_steps = [&](auto phc, auto bufc, auto cpb, auto cpe) __attribute__((always_inline)) {
        constexpr int PH = decltype(phc)::value, BUF = decltype(bufc)::value;
        constexpr int CP0 = decltype(cpb)::value, CP1 = decltype(cpe)::value;
        const float* a_ptr = smem + a_idx + BUF * C::BUF_FLOATS;
        const float* p_ptr = smem + p_idx + BUF * C::BUF_FLOATS;
#pragma unroll
        for (int cp = CP0; cp < CP1; ++cp) {
            float d[3][4];                         // rows PH .. PH+2 of the 4x4 patch, channel 2*cp + kh
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if constexpr (DBG & 8) {
                    d[i][0] = d[i][1] = d[i][2] = d[i][3] = __builtin_bit_cast(float, (unsigned)(lane + i + cp) | 0x3f000000u);
                    asm volatile("" : "+v"(d[i][0]), "+v"(d[i][1]), "+v"(d[i][2]), "+v"(d[i][3]));
                } else {
                    const float2 lo = *reinterpret_cast<const float2*>(p_ptr + cp * 2 * CH_STRIDE + i * RS);
                    const float2 hi = *reinterpret_cast<const float2*>(p_ptr + cp * 2 * CH_STRIDE + i * RS + 2);
                    d[i][0] = lo.x; d[i][1] = lo.y; d[i][2] = hi.x; d[i][3] = hi.y;
                }
            }
            // rows of B^T d:  PH=0: t0 = d0 - d2, t1 = d1 + d2 ;  PH=1 (d[] = d1,d2,d3): t2 = d2 - d1, t3 = d1 - d3
            float t[2][4], u[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (PH == 0) {
                    t[0][j] = d[0][j] - d[2][j];
                    t[1][j] = d[1][j] + d[2][j];
                } else {
                    t[0][j] = d[1][j] - d[0][j];
                    t[1][j] = d[0][j] - d[2][j];
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                u[i * 4 + 0] = t[i][0] - t[i][2];
                u[i * 4 + 1] = t[i][1] + t[i][2];
                u[i * 4 + 2] = t[i][2] - t[i][1];
                u[i * 4 + 3] = t[i][1] - t[i][3];
            }
            if constexpr (DBG & 2) {
#pragma unroll
                for (int p = 0; p < 8; ++p) u[p] = d[p % 3][p & 3];
            }
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                float av;
                if constexpr (DBG & 4) {
                    av = d[0][p & 3];
                } else {
                    av = a_ptr[(cp * 2 * 16 + p) * CO_T];
                }
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, u[p], acc[p], 0, 0, 0);
            }
        }
    };

    // Double-buffered LDS, ONE barrier per chunk: the loads of chunk c+1 are issued before the MFMAs of
    // chunk c and committed to the other buffer half-way through them, so neither the global latency
    // nor the LDS write pass is exposed (the partner wave on the SIMD keeps the matrix pipe busy).
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using H0 = std::integral_constant<int, 0>;
    using H2 = std::integral_constant<int, CI_T / 2>;
    unsigned cin_bytes = 0, cw_bytes = 0;
    auto chunk = [&](auto phc, auto bufc) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value;
        float* bufn = smem + (BUF ^ 1) * C::BUF_FLOATS;
        cin_bytes += in_step;
        cw_bytes += w_step;
        if constexpr (!(DBG & 1)) issue(cin_bytes, cw_bytes, bufn);   // past the end: range check -> zeros, no traffic
        __builtin_amdgcn_sched_barrier(0);
        mfma_steps(phc, bufc, H0{}, H2{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(DBG & 1)) commit(bufn);      // 6 dwords per thread; the weight DMA lands on its own
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // retire this wave's LDS-DMA pieces
        __syncthreads();
    };
    auto run = [&](auto phc) __attribute__((always_inline)) {
        const int nch = (a.Cin + CI_T - 1) / CI_T;
        int c = 0;
        for (; c + 2 <= nch; c += 2) {
            chunk(phc, I0{});
            chunk(phc, I1{});
        }
        if (c < nch) chunk(phc, I0{});
    };
    issue(0, 0, smem);
    commit(smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // (tried: a static s_setprio for waves 4-7 to break SIMD-partner lockstep -- 205 vs 210 TFLOP/s, not kept)
    if constexpr (!PIPE) {
        if (ph == 0) run(I0{}); else run(I1{});
    } else {
        // Register-level software pipeline across K-steps AND chunks.  During the 8 MFMAs of K-step s the
        // wave (1) issues the 7 LDS reads of step s+1 up front, (2) transforms them (16 adds) in the issue
        // shadow of MFMAs 4..7, so every MFMA finds its operands in registers and a single wave can keep
        // the matrix pipe busy.  The chunk barrier sits in the MIDDLE of the second-to-last K-step: by
        // then every wave has pulled the last step's operands of chunk c into registers, so the barrier
        // both publishes chunk c+1 (committed just before it) and frees buffer c&1 for the staging of
        // chunk c+2, which is issued right after it -- the prefetch of step 0 of chunk c+1 then runs
        // under the last K-step of chunk c and no pipeline fill is exposed per chunk.
        constexpr int STEPS = CI_T / 2;
        float uC[8], avC[8];
        auto lds_patch = [&](auto bufc, auto cpc, float (&d)[3][4]) __attribute__((always_inline)) {
            constexpr int BUF = decltype(bufc)::value, CP = decltype(cpc)::value;
            const float* p_ptr = smem + p_idx + BUF * C::BUF_FLOATS + CP * 2 * CH_STRIDE;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if constexpr (DBG & 8) {
                    d[i][0] = d[i][1] = d[i][2] = d[i][3] = __builtin_bit_cast(float, (unsigned)(lane + i + CP) | 0x3f000000u);
                    asm volatile("" : "+v"(d[i][0]), "+v"(d[i][1]), "+v"(d[i][2]), "+v"(d[i][3]));
                } else {
                    const float2 lo = *reinterpret_cast<const float2*>(p_ptr + i * RS);
                    const float2 hi = *reinterpret_cast<const float2*>(p_ptr + i * RS + 2);
                    d[i][0] = lo.x; d[i][1] = lo.y; d[i][2] = hi.x; d[i][3] = hi.y;
                }
            }
        };
        auto lds_a = [&](auto bufc, auto cpc, float (&av)[8]) __attribute__((always_inline)) {
            constexpr int BUF = decltype(bufc)::value, CP = decltype(cpc)::value;
            const float* a_ptr = smem + a_idx + BUF * C::BUF_FLOATS + CP * 2 * 16 * CO_T;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                if constexpr (DBG & 4) av[p] = __builtin_bit_cast(float, (unsigned)(lane + p) | 0x3f000000u);
                else av[p] = a_ptr[p * CO_T];
            }
        };
        auto rows = [&](auto phc, const float (&d)[3][4], float (&t)[2][4], int i) __attribute__((always_inline)) {
            constexpr int PH = decltype(phc)::value;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (PH == 0) t[i][j] = i == 0 ? d[0][j] - d[2][j] : d[1][j] + d[2][j];
                else t[i][j] = i == 0 ? d[1][j] - d[0][j] : d[0][j] - d[2][j];
            }
        };
        auto cols = [&](const float (&t)[2][4], float (&u)[8], int i) __attribute__((always_inline)) {
            u[i * 4 + 0] = t[i][0] - t[i][2];
            u[i * 4 + 1] = t[i][1] + t[i][2];
            u[i * 4 + 2] = t[i][2] - t[i][1];
            u[i * 4 + 3] = t[i][1] - t[i][3];
        };
        auto mfma4 = [&](int p0) __attribute__((always_inline)) {
#pragma unroll
            for (int p = p0; p < p0 + 4; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(avC[p], uC[p], acc[p], 0, 0, 0);
        };
        // one K-step: MFMAs on (avC, uC); fetch + transform the operands that live at (nbuf, ncp)
        // staging of chunk c+2 in two slices (half of the input dwords, then half of the weight DMA pieces
        // each), spread over the two K-steps after the barrier so that the eight waves do not queue their
        // ten VMEM instructions on the texture-address path in front of the next MFMA
        auto issue_in = [&](int part) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < NIN; ++k)
                if (k * 2 / NIN == part) xin[k] = wbload(xr, in_off[k] + cin_bytes);
        };
        auto issue_w = [&](int part, float* buf) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < NW4; ++k)
                if (k * 2 / NW4 == part && (wid_u * 64 + NTHR * k) * 4 < W_FLOATS)
                    dma16_to_lds(wr4, lds_addr_of(buf) + (unsigned)((wid_u * 64 + NTHR * k) * 16), w_off[k] + cw_bytes);
        };
        // one K-step: MFMAs on (avC, uC); fetch + transform the operands that live at (nbuf, ncp).
        // MODE 0: plain   1: chunk barrier (commit the staged chunk, publish)   2 / 3: staging slice 0 / 1 into buffer TB
        auto kstep = [&](auto phc, auto nbufc, auto ncpc, auto modec, auto tbc) __attribute__((always_inline)) {
            constexpr int MODE = decltype(modec)::value, TB = decltype(tbc)::value;
            float d[3][4], avn[8], t[2][4], un[8];
            constexpr bool LD = !(DBG & 8), LA = !(DBG & 4);
            __builtin_amdgcn_sched_barrier(0);
            // R1: the three patch reads of step s+1, one per MFMA (a burst of 7 reads from all eight waves
            // would queue on the LDS and hold the MFMAs behind it)
            lds_patch(nbufc, ncpc, d);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(avC[0], uC[0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(avC[1], uC[1], acc[1], 0, 0, 0);
            if constexpr (LD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (LD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (LD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            // R2: the A operands of step s+1
            lds_a(nbufc, ncpc, avn);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(avC[2], uC[2], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(avC[3], uC[3], acc[3], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (LA) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (LA) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            if constexpr (MODE >= 2) {
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!(DBG & 1)) issue_w(MODE - 2, smem + TB * C::BUF_FLOATS);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (MODE == 1) {
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!(DBG & 1)) commit(smem + TB * C::BUF_FLOATS);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                // the input dwords of chunk c+2 (HBM latency, six loads) go out at once; its weight DMA
                // (L2 hits, 4 KB per wave) follows in two slices over the next two K-steps
                cin_bytes += in_step;
                cw_bytes += w_step;
                if constexpr (!(DBG & 1)) { issue_in(0); issue_in(1); }
                __builtin_amdgcn_sched_barrier(0);
            }
            // R3: the transform adds of step s+1, four at a time in the shadow of MFMAs 4..7
            mfma4(4);
            rows(phc, d, t, 0); rows(phc, d, t, 1);
            cols(t, un, 0); cols(t, un, 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
            if constexpr (DBG & 2) {
#pragma unroll
                for (int p = 0; p < 8; ++p) un[p] = d[p % 3][p & 3];
            }
#pragma unroll
            for (int p = 0; p < 8; ++p) { uC[p] = un[p]; avC[p] = avn[p]; }
        };
        auto pchunk = [&](auto phc, auto bufc) __attribute__((always_inline)) {
            constexpr int BUF = decltype(bufc)::value;
            using BC = std::integral_constant<int, BUF>;
            using BN = std::integral_constant<int, BUF ^ 1>;
            using K = std::integral_constant<int, 0>;
            static_assert(STEPS == 4, "pipelined chunk is written for 4 K-steps");
            // buffer BUF holds chunk c, BUF^1 receives chunk c+1 (slice 0 went out in the previous chunk's last step)
            kstep(phc, BC{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 3>{}, BN{});   // slice 1 of c+1
            kstep(phc, BC{}, std::integral_constant<int, 2>{}, K{}, BC{});
            kstep(phc, BC{}, std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{}, BN{});   // commit c+1, barrier
            kstep(phc, BN{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, BC{});   // slice 0 of c+2
        };
        auto prun = [&](auto phc) __attribute__((always_inline)) {
            {   // prologue: slice 0 of chunk 1 in flight, operands of (chunk 0, step 0) in registers
                cin_bytes += in_step;
                cw_bytes += w_step;
                if constexpr (!(DBG & 1)) { issue_in(0); issue_in(1); issue_w(0, smem + C::BUF_FLOATS); }
                float d[3][4], t[2][4];
                lds_patch(I0{}, I0{}, d);
                lds_a(I0{}, I0{}, avC);
                rows(phc, d, t, 0); rows(phc, d, t, 1);
                cols(t, uC, 0); cols(t, uC, 1);
            }
            const int nch = (a.Cin + CI_T - 1) / CI_T;
            int c = 0;
            for (; c + 2 <= nch; c += 2) {
                pchunk(phc, I0{});
                pchunk(phc, I1{});
            }
            if (c < nch) pchunk(phc, I0{});
        };
        if (ph == 0) prun(I0{}); else prun(I1{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing (all-zero) staging of a non-existent
        __syncthreads();                                   // chunk must land before the epilogue reuses the LDS
    }

    // ---- epilogue: partial output transform of this half's rows, halves added through LDS
    //   s0 = m0 + m1 + m2 , s1 = m1 - m2 - m3  (xi index)  ->  ph=0 contributes (m0+m1, m1), ph=1 (m2, -m2-m3)
    constexpr int EXL = 2 * WM * 64;                   // lanes of the ph=0 waves
    float* ex = smem;                                  // [64 values][EXL lanes], reused LDS
    const int slot = (wid % (2 * WM)) * 64 + lane;     // matching lane of the partner wave (same wm, wn)
    float* zb = a.z + (int64_t)b * a.z_bs;
    const int oy = y0 + (wn * TRW + tr) * 2, ox = x0 + tc * 2;
    const bool r0ok = oy < a.H, r1ok = oy + 1 < a.H, c0ok = ox < a.W, c1ok = ox + 1 < a.W;
    const bool vec2 = c1ok && ((a.W & 1) == 0) && ((a.z_bs & 1) == 0);
    float y[16][4];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float s0[4], s1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float mA = acc[j][r], mB = acc[4 + j][r];      // rows xi = 2ph, 2ph+1
            s0[j] = ph ? mA : mA + mB;
            s1[j] = ph ? -mA - mB : mB;
        }
        y[r][0] = s0[0] + s0[1] + s0[2];
        y[r][1] = s0[1] - s0[2] - s0[3];
        y[r][2] = s1[0] + s1[1] + s1[2];
        y[r][3] = s1[1] - s1[2] - s1[3];
    }
    if (ph == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) ex[(r * 4 + q) * EXL + slot] = y[r][q];
    }
    __syncthreads();
    if (ph == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            const float y00 = y[r][0] + ex[(r * 4 + 0) * EXL + slot], y01 = y[r][1] + ex[(r * 4 + 1) * EXL + slot];
            const float y10 = y[r][2] + ex[(r * 4 + 2) * EXL + slot], y11 = y[r][3] + ex[(r * 4 + 3) * EXL + slot];
            if (co < a.Cout && c0ok) {
                float* o = zb + (int64_t)co * HW + (int64_t)oy * a.W + ox;
                if (vec2) {
                    if (r0ok) *reinterpret_cast<float2*>(o) = make_float2(y00, y01);
                    if (r1ok) *reinterpret_cast<float2*>(o + a.W) = make_float2(y10, y11);
                } else {
                    if (r0ok) { o[0] = y00; if (c1ok) o[1] = y01; }
                    if (r1ok) { o[a.W] = y10; if (c1ok) o[a.W + 1] = y11; }
                }
            }
        }
    }
}

template <int TW, int CI_T, int WM, int DBG = 0, bool PIPE = true>
static int launch_wino(WinoArgs a, hipStream_t st) {
    using C = WinoCfg<TW, CI_T, WM>;
    a.tilesX = cdiv(a.W, TW);
    a.tilesY = cdiv(a.H, C::ROWS);
    a.coTiles = cdiv(a.Cout, C::CO_T);
    const int64_t blocks = (int64_t)a.B * a.tilesX * a.tilesY * a.coTiles;
    ONET_REQUIRE(blocks > 0 && blocks < (1ll << 31), "conv_wino: grid %lld out of range", (long long)blocks);
    auto kern = conv_wino_kernel<TW, CI_T, WM, DBG, PIPE>;
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  C::LDS_BYTES);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(C::NTHR), C::LDS_BYTES, st, a);
    return check_launch("conv_wino_kernel");
}

// ------------------------------------------------------------------ wgrad
//   dW = G^T [ sum_tiles (A dY A^T) (.) (B^T d B) ] G          (Winograd F(2x2,3x3) weight gradient)
// 16 GEMMs  M_pos[co][ci] = sum_tile Z_pos[co][tile] * U_pos[ci][tile]: M = co, N = ci, K = tiles.
// Block = 8 waves = 2 position halves x 2 co halves x 2 ci halves (64 co x 64 ci x 16 positions),
// 8 accumulators per wave; both transforms run in registers from channel-major LDS strips whose
// channel stride is == 2 (mod 64) floats, so the 32 lanes (= 32 channels) of a ds_read_b64 hit 64
// distinct banks.  Split-K over (image, strip) units; raw per-position slabs are reduced
// deterministically and then folded by G^T . G into the nn.Conv2d layout.
struct WwArgs {
    const float* x;
    int64_t x_bs;
    const float* dz;
    int64_t dz_bs;
    float* slab;
    int B, Cin, Cout, H, W, ciTiles, coTiles, splitK, stripsY, stripsX;
};

template <int PW>
struct WwCfg {
    static constexpr int PR = 64 / PW;                      // pixel rows per strip (64 px = 16 tiles)
    static constexpr int XR = PR + 2, XC = PW + 2;
    static constexpr int TCS = PW / 2;                      // tiles per tile row
    static constexpr int DZS = 66;                          // 64 px + 2   (== 2 mod 64)
    static constexpr int XS = (XR * XC <= 130) ? 130 : 194; // >= XR*XC and == 2 mod 64
    static constexpr int BUF_FLOATS = 64 * (DZS + XS);
    static constexpr int LDS_BYTES = 2 * BUF_FLOATS * 4;
};

template <int PW>
__global__ __launch_bounds__(512, 2) void conv_wino_wgrad_kernel(WwArgs a) {
    using C = WwCfg<PW>;
    constexpr int PR = C::PR, XR = C::XR, XC = C::XC, TCS = C::TCS, DZS = C::DZS, XS = C::XS;
    constexpr int NTHR = 512;
    constexpr int NDZ = 8;                                   // 64 ch * 64 px / 512
    constexpr int ROWS_PER_K = NTHR / PW, CH_PER_K = ROWS_PER_K / XR;
    constexpr int NXM = (64 + CH_PER_K - 1) / CH_PER_K;
    constexpr int NXH = (64 * XR * 2 + NTHR - 1) / NTHR;

    extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 x { dz [64][DZS], x [64][XS] }

    int bid;
    {
        const int n = gridDim.x, q = n >> 3, r = n & 7, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int tiles = a.ciTiles * a.coTiles;
    const int ks = bid / tiles, tile = bid % tiles;
    const int ciT = tile % a.ciTiles, coT = tile / a.ciTiles;
    const int co0 = coT * 64, ci0 = ciT * 64;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ph = wid >> 2, wm = (wid >> 1) & 1, wn = wid & 1;
    const int l31 = lane & 31, kh = lane >> 5;
    const int HW = a.H * a.W;

    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    const int a_idx = (wm * 32 + l31) * DZS + kh * 2;                        // dz tile of lane parity kh
    const int b_idx = 64 * DZS + (wn * 32 + l31) * XS + ph * XC + kh * 2;    // patch rows ph..ph+2

    const int dz_c = tid >> 6, dz_p = tid & 63, dz_r = dz_p / PW, dz_col = dz_p % PW;
    const int xm_col = tid % PW, xm_rowid = tid / PW, xm_cp = xm_rowid / XR, xm_r = xm_rowid % XR;
    const bool xm_thread = xm_rowid < CH_PER_K * XR;
    const unsigned dz_kstep = (unsigned)(8 * HW * 4), xm_kstep = (unsigned)(CH_PER_K * HW * 4);

    float dzv[NDZ], xmv[NXM], xhv[NXH];
    const int nunits = a.B * a.stripsY * a.stripsX;

    auto issue = [&](int u) __attribute__((always_inline)) {
        const bool live = u < nunits;
        const int uu = live ? u : 0;
        const int sx = uu % a.stripsX;
        const int sy = (uu / a.stripsX) % a.stripsY;
        const int b = uu / (a.stripsX * a.stripsY);
        const int y0 = sy * PR, x0 = sx * PW;
        const __amdgpu_buffer_rsrc_t dr = wmake_rsrc(a.dz + (int64_t)b * a.dz_bs, (int64_t)a.Cout * HW * 4);
        const __amdgpu_buffer_rsrc_t xr = wmake_rsrc(a.x + (int64_t)b * a.x_bs, (int64_t)a.Cin * HW * 4);
        {
            const int yy = y0 + dz_r, xx = x0 + dz_col;
            const bool ok = live && yy < a.H && xx < a.W;
            const unsigned base = ok ? (unsigned)(((co0 + dz_c) * HW + yy * a.W + xx) * 4) : OOB_OFF_W;
#pragma unroll
            for (int k = 0; k < NDZ; ++k) dzv[k] = wbload(dr, base + k * dz_kstep);
        }
        {
            const int yy = y0 - 1 + xm_r, xx = x0 + xm_col;
            const bool ok = live && xm_thread && yy >= 0 && yy < a.H && xx < a.W;
            const unsigned base = ok ? (unsigned)(((ci0 + xm_cp) * HW + yy * a.W + xx) * 4) : OOB_OFF_W;
#pragma unroll
            for (int k = 0; k < NXM; ++k)
                xmv[k] = wbload(xr, (xm_cp + CH_PER_K * k < 64) ? base + k * xm_kstep : OOB_OFF_W);
        }
#pragma unroll
        for (int j = 0; j < NXH; ++j) {
            const int e = tid + NTHR * j;
            const int c = e / (2 * XR), q = e % (2 * XR);
            const int r = q >> 1, side = q & 1;
            const int yy = y0 - 1 + r, xx = side ? x0 + PW : x0 - 1;
            const bool ok = live && e < 64 * XR * 2 && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
            xhv[j] = wbload(xr, ok ? (unsigned)(((ci0 + c) * HW + yy * a.W + xx) * 4) : OOB_OFF_W);
        }
    };
    auto commit = [&](float* buf) __attribute__((always_inline)) {
        float* dz_lds = buf;
        float* x_lds = buf + 64 * DZS;
#pragma unroll
        for (int k = 0; k < NDZ; ++k) dz_lds[(dz_c + 8 * k) * DZS + dz_p] = dzv[k];
        if (xm_thread) {
#pragma unroll
            for (int k = 0; k < NXM; ++k)
                if (xm_cp + CH_PER_K * k < 64) x_lds[(xm_cp + CH_PER_K * k) * XS + xm_r * XC + xm_col + 1] = xmv[k];
        }
#pragma unroll
        for (int j = 0; j < NXH; ++j) {
            const int e = tid + NTHR * j;
            const int c = e / (2 * XR), q = e % (2 * XR);
            if (e < 64 * XR * 2) x_lds[c * XS + (q >> 1) * XC + ((q & 1) ? XC - 1 : 0)] = xhv[j];
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // Register-level software pipeline across K-steps AND units (same idea as conv_wino_kernel): the operands of
    // step s+1 are read from LDS and transformed under the MFMAs of step s; the unit barrier sits in the
    // second-to-last step (by then every wave holds the unit's last operands in registers), so the first step of
    // the next unit never waits for LDS behind a barrier.  Timing-only ablation without the barrier: +6.6 %.
    float zC[8], uC[8];
    auto operands = [&](auto phc, auto bufc, auto stc, float (&z)[8], float (&u)[8]) __attribute__((always_inline)) {
        constexpr int PH = decltype(phc)::value, BUF = decltype(bufc)::value, st = decltype(stc)::value;
        const float* zp = smem + BUF * C::BUF_FLOATS + a_idx;
        const float* xp = smem + BUF * C::BUF_FLOATS + b_idx;
        constexpr int trow = (2 * st) / TCS, tcol2 = ((2 * st) % TCS) * 2;   // tile 2*st (+kh via the lane base)
        const float2 y0v = *reinterpret_cast<const float2*>(zp + (2 * trow) * PW + tcol2);
        const float2 y1v = *reinterpret_cast<const float2*>(zp + (2 * trow + 1) * PW + tcol2);
        float d[3][4];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float2 lo = *reinterpret_cast<const float2*>(xp + (2 * trow + i) * XC + tcol2);
            const float2 hi = *reinterpret_cast<const float2*>(xp + (2 * trow + i) * XC + tcol2 + 2);
            d[i][0] = lo.x; d[i][1] = lo.y; d[i][2] = hi.x; d[i][3] = hi.y;
        }
        float zr[2][2];
        if constexpr (PH == 0) {
            zr[0][0] = y0v.x; zr[0][1] = y0v.y;
            zr[1][0] = y0v.x + y1v.x; zr[1][1] = y0v.y + y1v.y;
        } else {
            zr[0][0] = y0v.x - y1v.x; zr[0][1] = y0v.y - y1v.y;
            zr[1][0] = -y1v.x; zr[1][1] = -y1v.y;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            z[i * 4 + 0] = zr[i][0];
            z[i * 4 + 1] = zr[i][0] + zr[i][1];
            z[i * 4 + 2] = zr[i][0] - zr[i][1];
            z[i * 4 + 3] = -zr[i][1];
        }
        float t[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (PH == 0) {
                t[0][j] = d[0][j] - d[2][j];
                t[1][j] = d[1][j] + d[2][j];
            } else {
                t[0][j] = d[1][j] - d[0][j];
                t[1][j] = d[0][j] - d[2][j];
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            u[i * 4 + 0] = t[i][0] - t[i][2];
            u[i * 4 + 1] = t[i][1] + t[i][2];
            u[i * 4 + 2] = t[i][2] - t[i][1];
            u[i * 4 + 3] = t[i][1] - t[i][3];
        }
    };
    int unext = ks;
    // MODE 0 plain; 1: commit the staged unit after the MFMAs (mid-unit); 2: barrier after the next operands are in
    auto wstep = [&](auto phc, auto nbufc, auto nstc, auto modec, auto curbufc) __attribute__((always_inline)) {
        constexpr int MODE = decltype(modec)::value, CUR = decltype(curbufc)::value;
        float zN[8], uN[8];
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE == 3) {              // first step of a unit: the ~25 staging loads of the next unit, spread
            unext += a.splitK;                  // over this step's MFMAs instead of issued in one burst in front of them
            issue(unext);
        }
        operands(phc, nbufc, nstc, zN, uN);
#pragma unroll
        for (int p = 0; p < 8; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(zC[p], uC[p], acc[p], 0, 0, 0);
        // pin the order: the five LDS reads first, then the 22 transform adds spread under the MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, MODE == 3 ? 12 : 3, 0);
            if constexpr (MODE == 3) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE == 1) commit(smem + (CUR ^ 1) * C::BUF_FLOATS);
        if constexpr (MODE == 2) __syncthreads();
#pragma unroll
        for (int p = 0; p < 8; ++p) { zC[p] = zN[p]; uC[p] = uN[p]; }
    };
    auto stage = [&](auto phc, auto bufc) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value;
        using BC = std::integral_constant<int, BUF>;
        using BN = std::integral_constant<int, BUF ^ 1>;
        using P = std::integral_constant<int, 0>;
        // SIMD partners are the waves (w, w + 4), i.e. ph = 0 / 1, and run this program in lockstep: staging bursts that
        // hit both at once idle the matrix pipe.  ONET_WW_ASYM: the ph = 1 waves issue the next unit's loads and commit
        // the staged unit two K-steps later than the ph = 0 waves (both before the unit barrier in step 6).
        constexpr int SH = (ONET_WW_ASYM && decltype(phc)::value == 1) ? 2 : 0;
        using MI = std::integral_constant<int, 3>;    // issue
        using MC = std::integral_constant<int, 1>;    // commit
        wstep(phc, BC{}, std::integral_constant<int, 1>{}, std::conditional_t<SH == 0, MI, P>{}, BC{});
        wstep(phc, BC{}, std::integral_constant<int, 2>{}, P{}, BC{});
        wstep(phc, BC{}, std::integral_constant<int, 3>{}, std::conditional_t<SH == 2, MI, P>{}, BC{});
        wstep(phc, BC{}, std::integral_constant<int, 4>{}, std::conditional_t<SH == 0, MC, P>{}, BC{});   // commit unit u+1
        wstep(phc, BC{}, std::integral_constant<int, 5>{}, P{}, BC{});
        wstep(phc, BC{}, std::integral_constant<int, 6>{}, std::conditional_t<SH == 2, MC, P>{}, BC{});
        wstep(phc, BC{}, std::integral_constant<int, 7>{}, std::integral_constant<int, 2>{}, BC{});   // barrier
        wstep(phc, BN{}, std::integral_constant<int, 0>{}, P{}, BC{});
    };
    auto run = [&](auto phc) __attribute__((always_inline)) {
        operands(phc, I0{}, I0{}, zC, uC);
        const int nst = (nunits - ks + a.splitK - 1) / a.splitK;
        int c = 0;
        for (; c + 2 <= nst; c += 2) {
            stage(phc, I0{});
            stage(phc, I1{});
        }
        if (c < nst) stage(phc, I0{});
    };
    issue(ks);
    commit(smem);
    __syncthreads();
    if (ph == 0) run(I0{}); else run(I1{});

    // raw slab[ks][pos][co][ci]; lanes run along ci (coalesced)
    float* sl = a.slab + (int64_t)ks * 16 * a.Cout * a.Cin;
    const int ci = ci0 + wn * 32 + l31;
    if (ci < a.Cin) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (co < a.Cout) sl[((int64_t)(ph * 8 + p) * a.Cout + co) * a.Cin + ci] = acc[p][r];
            }
        }
    }
}

// raw[pos][i] = sum_ks slab[ks][pos][i]   (i = co*Cin + ci); block = 64 i x 4 split-K groups
__global__ __launch_bounds__(256) void wino_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ raw,
                                                                int splitK, int64_t n) {
    __shared__ float red[256];
    const int p = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const int g = threadIdx.x >> 6;
    float s = 0.f;
    if (i < n)
        for (int k = g; k < splitK; k += 4) s += slab[((int64_t)k * 16 + p) * n + i];
    red[threadIdx.x] = s;
    __syncthreads();
    if (g == 0 && i < n)
        raw[(int64_t)p * n + i] = (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
}

// dw[co][ci][3][3] (+)= G^T M G,  G^T = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1]]
__global__ void wino_wgrad_transform_kernel(const float* __restrict__ raw, float* __restrict__ dw, int64_t n,
                                            int accumulate) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float m[4][4], r[3][4];
#pragma unroll
    for (int p = 0; p < 16; ++p) m[p >> 2][p & 3] = raw[(int64_t)p * n + i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[0][j] = m[0][j] + 0.5f * (m[1][j] + m[2][j]);
        r[1][j] = 0.5f * (m[1][j] - m[2][j]);
        r[2][j] = 0.5f * (m[1][j] + m[2][j]) + m[3][j];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float g0 = r[k][0] + 0.5f * (r[k][1] + r[k][2]);
        const float g1 = 0.5f * (r[k][1] - r[k][2]);
        const float g2 = 0.5f * (r[k][1] + r[k][2]) + r[k][3];
        float* o = dw + i * 9 + k * 3;
        o[0] = accumulate ? o[0] + g0 : g0;
        o[1] = accumulate ? o[1] + g1 : g1;
        o[2] = accumulate ? o[2] + g2 : g2;
    }
}

static void wino_wgrad_plan(int B, int Cin, int Cout, int H, int W, int& pw, int& splitK, int& sx, int& sy) {
    pw = (W > 16) ? 32 : 16;
    const int pr = 64 / pw;
    sx = cdiv(W, pw);
    sy = cdiv(H, pr);
    const int64_t nunits = (int64_t)B * sx * sy;
    const int64_t tiles = (int64_t)cdiv(Cout, 64) * cdiv(Cin, 64);
    static int target = -1;
    if (target < 0) { const char* e = getenv("ONET_W2W_BLOCKS"); target = (e && atoi(e) > 0) ? atoi(e) : 256; }
    int64_t s = (target + tiles - 1) / tiles;                    // one 8-wave block per CU, ONE round (256 vs 512: +2-6 %; 384 / 768: -20 %)
    const int64_t per = (int64_t)16 * Cout * Cin * 4;
    const int64_t cap = (160ll << 20) / (per > 0 ? per : 1);
    if (s > cap) s = cap;
    if (s > nunits) s = nunits;
    if (s < 1) s = 1;
    splitK = (int)s;
}

template <int PW>
static void launch_wino_wgrad(const WwArgs& a, int64_t blocks, hipStream_t st) {
    using C = WwCfg<PW>;
    auto kern = conv_wino_wgrad_kernel<PW>;
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  C::LDS_BYTES);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), C::LDS_BYTES, st, a);
}

extern "C" {

int onet_conv3x3_pack_weights_winograd(const float* w, float* wq_fwd, float* wq_dgrad, int Cout, int Cin,
                                       void* stream) {
    ONET_REQUIRE(w && Cout > 0 && Cin > 0, "pack3x3_winograd: bad args");
    const int64_t n = (int64_t)Cout * Cin;
    hipLaunchKernelGGL(pack3x3_wino_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n, 256), 4096)), dim3(256), 0,
                       as_stream(stream), w, wq_fwd, wq_dgrad, Cout, Cin);
    return check_launch("pack3x3_wino_kernel");
}

int onet_conv3x3_winograd_fwd(const float* x, int64_t x_bs, const float* wq, float* z, int64_t z_bs, int B, int Cin,
                              int Cout, int H, int W, void* stream) {
    ONET_REQUIRE(x && wq && z, "conv3x3_winograd_fwd: null pointer");
    ONET_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "conv3x3_winograd_fwd: bad shape");
    ONET_REQUIRE((Cout & 3) == 0, "conv3x3_winograd_fwd: Cout must be a multiple of 4 (use onet_conv_fwd)");
    ONET_REQUIRE(x_bs >= (int64_t)Cin * H * W && z_bs >= (int64_t)Cout * H * W, "conv3x3_winograd_fwd: batch stride too small");
    WinoArgs a{x, x_bs, wq, z, z_bs, B, Cin, Cout, H, W, 0, 0, 0};
    static int wm = -1;
    if (wm < 0) {
        const char* e = getenv("ONET_WINO_WM");      // tuning override: 2 = 8-wave blocks (default), 1 = 4-wave blocks x 2/CU
        wm = (e && e[0] == '1') ? 1 : 2;             // measured: 210 vs 207 effective TFLOP/s
    }
    static int dbg = -1;
    if (dbg < 0) {
        const char* e = getenv("ONET_WINO_DBG");     // timing-only ablations, W > 16 only
        dbg = e ? atoi(e) : 0;
    }
    if (dbg && W > 16) {
        switch (dbg) {
            case 1: return launch_wino<32, 8, 2, 1>(a, as_stream(stream));
            case 2: return launch_wino<32, 8, 2, 2>(a, as_stream(stream));
            case 4: return launch_wino<32, 8, 2, 4>(a, as_stream(stream));
            case 8: return launch_wino<32, 8, 2, 8>(a, as_stream(stream));
            case 14: return launch_wino<32, 8, 2, 14>(a, as_stream(stream));
            case 15: return launch_wino<32, 8, 2, 15>(a, as_stream(stream));
            default: break;
        }
    }
    static int pipe = -1;
    if (pipe < 0) {
        const char* e = getenv("ONET_WINO_PIPE");    // 0 = the pre-pipelining main loop (A/B timing)
        pipe = (e && e[0] == '0') ? 0 : 1;
    }
    if (wm == 2 && !pipe)
        return (W > 16) ? launch_wino<32, 8, 2, 0, false>(a, as_stream(stream)) : launch_wino<16, 8, 2, 0, false>(a, as_stream(stream));
    if (wm == 2) return (W > 16) ? launch_wino<32, 8, 2>(a, as_stream(stream)) : launch_wino<16, 8, 2>(a, as_stream(stream));
    return (W > 16) ? launch_wino<32, 8, 1>(a, as_stream(stream)) : launch_wino<16, 8, 1>(a, as_stream(stream));
}

int64_t onet_conv3x3_winograd_wgrad_ws_bytes(int B, int Cin, int Cout, int H, int W) {
    int pw, splitK, sx, sy;
    wino_wgrad_plan(B, Cin, Cout, H, W, pw, splitK, sx, sy);
    return ((int64_t)splitK + 1) * 16 * Cout * Cin * 4;
}

int onet_conv3x3_winograd_wgrad(const float* x, int64_t x_bs, const float* dz, int64_t dz_bs, float* dw, void* ws,
                                int64_t ws_bytes, int B, int Cin, int Cout, int H, int W, int accumulate,
                                void* stream) {
    ONET_REQUIRE(x && dz && dw && ws, "conv3x3_winograd_wgrad: null pointer");
    ONET_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "conv3x3_winograd_wgrad: bad shape");
    WwArgs a{x, x_bs, dz, dz_bs, (float*)ws, B, Cin, Cout, H, W, cdiv(Cin, 64), cdiv(Cout, 64), 1, 1, 1};
    int pw;
    wino_wgrad_plan(B, Cin, Cout, H, W, pw, a.splitK, a.stripsX, a.stripsY);
    const int64_t n = (int64_t)Cout * Cin;
    const int64_t need = ((int64_t)a.splitK + 1) * 16 * n * 4;
    ONET_REQUIRE(ws_bytes >= need, "conv3x3_winograd_wgrad: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)need);
    const int64_t blocks = (int64_t)a.splitK * a.ciTiles * a.coTiles;
    hipStream_t st = as_stream(stream);
    if (pw == 32) launch_wino_wgrad<32>(a, blocks, st); else launch_wino_wgrad<16>(a, blocks, st);
    int rc = check_launch("conv_wino_wgrad_kernel");
    if (rc) return rc;
    float* raw = (float*)ws + (int64_t)a.splitK * 16 * n;
    hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3((unsigned)cdiv(n, 64), 16u), dim3(256), 0, st, (const float*)ws, raw,
                       a.splitK, n);
    rc = check_launch("wino_wgrad_reduce_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(wino_wgrad_transform_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, (const float*)raw, dw,
                       n, accumulate);
    return check_launch("wino_wgrad_transform_kernel");
}

}  // extern "C"
