// K2/K3: BatchNorm2d (train: batch statistics; eval: running statistics) fused with ReLU.
// Replaces nn.BatchNorm2d + nn.ReLU(inplace=True) at OV:48-49,52-53 and their autograd.
// All kernels are HBM-bound streaming passes: float4 loads, wave-shuffle reductions,
// deterministic two-stage per-channel reductions (partials -> fp64 finalize), no atomics.
#include <type_traits>
#include "common.hpp"

using namespace onet;

#ifndef ONET_BN_BATCH
#define ONET_BN_BATCH 1
#endif

// Partial over one image plane chunk: part[p][c][3] = (n_k, mean_k, M2_k), M2_k = sum (z - mean_k)^2.
// ONE pass in fp64 over sums shifted by a pivot (the chunk's first element, within a few standard deviations of
// its mean): s1 = sum (z - pivot), s2 = sum (z - pivot)^2, mean = pivot + s1/n, M2 = s2 - s1^2/n.  With the shift
// the subtraction in M2 cancels at most a few bits of a 53-bit accumulator; the plain fp32 E[z^2]-mean^2 loses the
// variance when |mean| >> std or when B*H*W is tiny (1x1 bottleneck at 16^2), and the earlier two-pass form read
// every chunk twice (2.9 TB/s effective against 4.9 for the apply pass).
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ z, int64_t z_bs,
                                                               float* __restrict__ part, int C, int HW,
                                                               int chunks, int chunk_len) {
    __shared__ double red[16];
    const int c = blockIdx.x % C;
    const int p = blockIdx.x / C;
    const int b = p / chunks, ch = p % chunks;
    const float* src = z + (int64_t)b * z_bs + (int64_t)c * HW;
    const int beg = ch * chunk_len;
    const int end = min(beg + chunk_len, HW);
    const bool vec = ((HW & 3) == 0) && ((z_bs & 3) == 0) && ((chunk_len & 3) == 0);
    const double pivot = (double)src[beg];
    // fp64 accumulation like ATen's CPU batch norm (acc_type<float> = double); free in an HBM-bound pass
    double v[2] = {0.0, 0.0};
    if (vec) {
        int i = beg + threadIdx.x * 4;
#if ONET_BN_BATCH
        for (; i + 3072 < end; i += 4096) {       // four 16-byte loads in flight per thread; same summation order as below
            float4 q[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = *reinterpret_cast<const float4*>(src + i + 1024 * k);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double d0 = q[k].x - pivot, d1 = q[k].y - pivot, d2 = q[k].z - pivot, d3 = q[k].w - pivot;
                v[0] += (d0 + d1) + (d2 + d3);
                v[1] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
        }
#endif
        for (; i < end; i += 1024) {
            const float4 q = *reinterpret_cast<const float4*>(src + i);
            const double d0 = q.x - pivot, d1 = q.y - pivot, d2 = q.z - pivot, d3 = q.w - pivot;
            v[0] += (d0 + d1) + (d2 + d3);
            v[1] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    } else {
        for (int i = beg + threadIdx.x; i < end; i += 256) {
            const double d = src[i] - pivot;
            v[0] += d;
            v[1] += d * d;
        }
    }
    block_sum_256<double, 2>(v, red);
    if (threadIdx.x == 0) {
        const double n = (double)(end - beg);
        const double mean = pivot + v[0] / n;
        double m2 = v[1] - v[0] * v[0] / n;
        if (m2 < 0.0) m2 = 0.0;
        float* o = part + ((int64_t)p * C + c) * 3;
        o[0] = (float)n;
        o[1] = (float)mean;
        o[2] = (float)(m2 + n * (mean - (double)(float)mean) * (mean - (double)(float)mean));  // M2 about the ROUNDED mean
    }
}

// one wave per channel: Chan's parallel merge of the (n, mean, M2) partials in fp64, then the BN coefficients
__global__ __launch_bounds__(64) void bn_finalize_kernel(const float* __restrict__ part, int nparts,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* running_mean,
                                                         float* running_var, float momentum, float eps,
                                                         float* __restrict__ save, int C, unsigned* __restrict__ act_slots = nullptr) {
    // act_slots (pre-split storage): an upper bound of relu(bn(z)) over the channels goes into the activation's magnitude slots --
    // |gamma| sqrt(N - 1) + |beta|, since |z - mean| invstd <= sqrt(N - 1) for the N values the statistics were taken over --
    // from which the pass that writes the activation as fp16 parts takes its (guard) scale: 1 unless the bound reaches 2^15
    const int c = blockIdx.x;
    double n = 0.0, s = 0.0;
    for (int p = threadIdx.x; p < nparts; p += 64) {
        const float* o = part + ((int64_t)p * C + c) * 3;
        n += (double)o[0];
        s += (double)o[0] * (double)o[1];
    }
    n = wave_sum(n);
    s = wave_sum(s);
    const double mean = s / n;
    double m2 = 0.0;
    for (int p = threadIdx.x; p < nparts; p += 64) {
        const float* o = part + ((int64_t)p * C + c) * 3;
        const double d = (double)o[1] - mean;
        m2 += (double)o[2] + (double)o[0] * d * d;
    }
    m2 = wave_sum(m2);
    if (threadIdx.x == 0) {
        const double var = m2 / n;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
        const float scale = g * invstd;
        save[c] = (float)mean;
        save[C + c] = invstd;
        save[2 * C + c] = scale;
        save[3 * C + c] = bt;
        if (act_slots) {
            const float bound = fabsf(g) * sqrtf((float)(n > 1.0 ? n - 1.0 : 1.0)) * 1.000001f + fabsf(bt);
            if (bound == bound) atomicMax(act_slots + (c & 63) * AMAX_STRIDE, __builtin_bit_cast(unsigned, bound));
        }
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        if (running_var) {
            const double unb = n > 1.0 ? m2 / (n - 1.0) : var;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
        }
    }
}

// The same merge for partials laid out channel-major, `part[c * c_stride + p * 3]` (written by the F(4x4) convolution's
// epilogue, one per 16 x 32-pixel block: thousands per channel), one 256-thread block per channel.
__global__ __launch_bounds__(256) void bn_finalize_cm_kernel(const float* __restrict__ part, int nparts, int64_t c_stride,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* running_mean,
                                                            float* running_var, float momentum, float eps,
                                                            float* __restrict__ save, int C, unsigned* __restrict__ act_slots = nullptr,
                                                            int groups = 1) {
    // groups > 1: the statistics groups of a twin batch in one launch -- group g owns records g * nparts .. (g + 1) * nparts - 1 of
    // every channel and save[g][4][C]; the running statistics take the groups' updates in order, as one launch per group did
    __shared__ double red[16];
    __shared__ double bc[2];
    const int c = blockIdx.x;
    for (int g = 0; g < groups; ++g) {
        const float* src = part + (int64_t)c * c_stride + (int64_t)g * nparts * 3;
        float* sv = save + (int64_t)g * 4 * C;
        double v[2] = {0.0, 0.0};
        for (int p = threadIdx.x; p < nparts; p += 256) {
            const float* o = src + (int64_t)p * 3;
            v[0] += (double)o[0];
            v[1] += (double)o[0] * (double)o[1];
        }
        if (g) __syncthreads();                     // (the previous group's readers of red / bc are done)
        block_sum_256<double, 2>(v, red);
        if (threadIdx.x == 0) {
            bc[0] = v[0];
            bc[1] = v[1] / v[0];
        }
        __syncthreads();
        const double n = bc[0], mean = bc[1];
        double w[1] = {0.0};
        for (int p = threadIdx.x; p < nparts; p += 256) {
            const float* o = src + (int64_t)p * 3;
            const double d = (double)o[1] - mean;
            w[0] += (double)o[2] + (double)o[0] * d * d;
        }
        __syncthreads();
        block_sum_256<double, 1>(w, red);
        if (threadIdx.x == 0) {
            const double m2 = w[0];
            const double var = m2 / n;
            const float invstd = (float)(1.0 / sqrt(var + (double)eps));
            const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
            sv[c] = (float)mean;
            sv[C + c] = invstd;
            sv[2 * C + c] = gm * invstd;
            sv[3 * C + c] = bt;
            if (act_slots) {                            // (see bn_finalize_kernel)
                const float bound = fabsf(gm) * sqrtf((float)(n > 1.0 ? n - 1.0 : 1.0)) * 1.000001f + fabsf(bt);
                if (bound == bound) atomicMax(act_slots + (c & 63) * AMAX_STRIDE, __builtin_bit_cast(unsigned, bound));
            }
            if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            if (running_var) {
                const double unb = n > 1.0 ? m2 / (n - 1.0) : var;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
            }
        }
    }
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, float* save, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = 1.f / sqrtf(rv[c] + eps);
    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    const float scale = g * invstd;
    save[c] = rm[c];
    save[C + c] = invstd;
    save[2 * C + c] = scale;
    save[3 * C + c] = bt;
}

// a = max(0, (z-mean)*scale + beta)  (the centred form: z*scale+shift cancels badly when |mean| >> std); one block per 4096-element chunk of a (b, c) plane

// block maximum of v into one of the 64 magnitude slots.  The slots are 128 bytes apart (one cache line each: the updates of a launch's
// tens of thousands of blocks spread over 64 lines / L2 channels instead of queueing on one) and a block commits ONCE: wave maxima
// through LDS, then a plain read of the slot first -- once it holds a value >= the block's, after the first few blocks almost
// always, no atomic is issued.  atomicMax on the fp32 bit pattern of a non-negative value is an order-independent maximum.
// Every thread of a 256-thread block must call this (it contains a barrier).
__device__ __forceinline__ void amax_commit(float v, unsigned* slots) {
    if (!slots) return;
    __shared__ float wave_max[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float m = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
        if (m == m) {
            unsigned* s = slots + (blockIdx.x & 63) * AMAX_STRIDE;
            const unsigned bits = __builtin_bit_cast(unsigned, m);
            if (__builtin_nontemporal_load(s) < bits) atomicMax(s, bits);
        }
    }
}

// Round 5, BASELINE configs[2]: the convolution outputs z may be STORED as bf16 (conv == "bf16" with pre-split operands: z is rounded
// once, to nearest even, by the convolution's epilogue; the BatchNorm statistics still come from the fp32 accumulators) -- every pass
// that reads z then moves half the bytes.  The kernels below that read z are templates over its element type ZT.
template <typename ZT> __device__ __forceinline__ float bn_ldz(const ZT* p);
template <> __device__ __forceinline__ float bn_ldz<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float bn_ldz<__bf16>(const __bf16* p) { return (float)*p; }
template <typename ZT> __device__ __forceinline__ float4 bn_ldz4(const ZT* p);      // four consecutive elements (16 / 8 byte aligned)
template <> __device__ __forceinline__ float4 bn_ldz4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 bn_ldz4<__bf16>(const __bf16* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);      // (a 16-byte load of the aligned group with the half picked out: no faster, measured)
    return make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                       __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u));
}

template <typename ZT = float>
__global__ __launch_bounds__(256) void bn_relu_apply_kernel(const ZT* __restrict__ z, int64_t z_bs,
                                                            float* __restrict__ a, int64_t a_bs,
                                                            const float* __restrict__ save, int C, int HW,
                                                            int chunks, unsigned* __restrict__ amax = nullptr, int gimg = 0) {
    // amax: 64 magnitude slots of the activation (the range guard of the fp16-split convolution that consumes it)
    float vmax = 0.f;
    const int plane = blockIdx.x / chunks, ch = blockIdx.x % chunks;
    const int b = plane / C, c = plane % C;
    if (gimg) save += (int64_t)(b / gimg) * 4 * C;        // statistics groups = consecutive batch slices of gimg images, save [G][4][C]
    const float mean = save[c], sc = save[2 * C + c], sh = save[3 * C + c];
    const ZT* src = z + (int64_t)b * z_bs + (int64_t)c * HW;
    float* dst = a + (int64_t)b * a_bs + (int64_t)c * HW;
    const int beg = ch * 4096, end = min(beg + 4096, HW);
    if (((HW & 3) == 0) && ((z_bs & 3) == 0) && ((a_bs & 3) == 0)) {
#if ONET_BN_BATCH
        if (end - beg == 4096) {          // full chunk: all four 16-byte loads of the thread in flight before the first use
            const int i0 = beg + threadIdx.x * 4;
            float4 q[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = bn_ldz4<ZT>(src + i0 + 1024 * k);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                q[k].x = fmaxf(fmaf(q[k].x - mean, sc, sh), 0.f);
                q[k].y = fmaxf(fmaf(q[k].y - mean, sc, sh), 0.f);
                q[k].z = fmaxf(fmaf(q[k].z - mean, sc, sh), 0.f);
                q[k].w = fmaxf(fmaf(q[k].w - mean, sc, sh), 0.f);
                *reinterpret_cast<float4*>(dst + i0 + 1024 * k) = q[k];
                vmax = fmaxf(fmaxf(vmax, fmaxf(q[k].x, q[k].y)), fmaxf(q[k].z, q[k].w));
            }
            amax_commit(vmax, amax);
            return;
        }
#endif
        for (int i = beg + threadIdx.x * 4; i < end; i += 1024) {
            float4 q = bn_ldz4<ZT>(src + i);
            q.x = fmaxf(fmaf(q.x - mean, sc, sh), 0.f);
            q.y = fmaxf(fmaf(q.y - mean, sc, sh), 0.f);
            q.z = fmaxf(fmaf(q.z - mean, sc, sh), 0.f);
            q.w = fmaxf(fmaf(q.w - mean, sc, sh), 0.f);
            *reinterpret_cast<float4*>(dst + i) = q;
            vmax = fmaxf(fmaxf(vmax, fmaxf(q.x, q.y)), fmaxf(q.z, q.w));
        }
    } else {
        for (int i = beg + threadIdx.x; i < end; i += 256) {
            const float v = fmaxf(fmaf(bn_ldz<ZT>(src + i) - mean, sc, sh), 0.f);
            dst[i] = v;
            vmax = fmaxf(vmax, v);
        }
    }
    amax_commit(vmax, amax);
}

// The same pass for an encoder output that is max-pooled next (OV:49/53 -> nn.MaxPool2d(2), OV:67): a thread owns a 2 x 4 patch,
// writes the activation and the two pooled values of the patch -- the
// separate pooling pass re-read the whole activation.  Same arithmetic as bn_relu_apply_kernel + maxpool2_fwd_kernel: identical
// bits.  Requires H % 2 == 0, W % 4 == 0 and 16-byte aligned rows (host-checked).
__global__ __launch_bounds__(256) void bn_relu_apply_pool_kernel(const float* __restrict__ z, int64_t z_bs, float* __restrict__ a,
                                                                 int64_t a_bs, float* __restrict__ y, int64_t y_bs,
                                                                 const float* __restrict__ save, int C, int H, int W,
                                                                 int blocks_per_plane, unsigned* __restrict__ amax = nullptr) {
    const int plane = blockIdx.x / blocks_per_plane, blk = blockIdx.x % blocks_per_plane;
    const int b = plane / C, c = plane % C;
    const float mean = save[c], sc = save[2 * C + c], sh = save[3 * C + c];
    const int qw = W >> 2, npatch = (H >> 1) * qw;
    const int i = blk * 256 + threadIdx.x;
    if (i >= npatch) {
        amax_commit(0.f, amax);                     // (every thread of the block takes part in the reduction)
        return;
    }
    const int pr = i / qw, q = i % qw;
    const int64_t off = (int64_t)c * H * W + (int64_t)(2 * pr) * W + 4 * q;
    const float* src = z + (int64_t)b * z_bs + off;
    float4 r0 = *reinterpret_cast<const float4*>(src), r1 = *reinterpret_cast<const float4*>(src + W);
    r0.x = fmaxf(fmaf(r0.x - mean, sc, sh), 0.f); r0.y = fmaxf(fmaf(r0.y - mean, sc, sh), 0.f);
    r0.z = fmaxf(fmaf(r0.z - mean, sc, sh), 0.f); r0.w = fmaxf(fmaf(r0.w - mean, sc, sh), 0.f);
    r1.x = fmaxf(fmaf(r1.x - mean, sc, sh), 0.f); r1.y = fmaxf(fmaf(r1.y - mean, sc, sh), 0.f);
    r1.z = fmaxf(fmaf(r1.z - mean, sc, sh), 0.f); r1.w = fmaxf(fmaf(r1.w - mean, sc, sh), 0.f);
    float* d = a + (int64_t)b * a_bs + off;
    *reinterpret_cast<float4*>(d) = r0;
    *reinterpret_cast<float4*>(d + W) = r1;
    const float m0 = fmaxf(fmaxf(r0.x, r0.y), fmaxf(r1.x, r1.y)), m1 = fmaxf(fmaxf(r0.z, r0.w), fmaxf(r1.z, r1.w));
    const int64_t yo = (int64_t)c * (H >> 1) * (W >> 1) + (int64_t)pr * (W >> 1) + 2 * q;
    *reinterpret_cast<float2*>(y + (int64_t)b * y_bs + yo) = make_float2(m0, m1);
    amax_commit(fmaxf(m0, m1), amax);               // the pooled tensor has the same maximum: one set of slots serves both
}

// ------------------------------------------------------------------ pre-split producers (round 4)
// The consumers of these activations / gradients are the split-fp16 convolution kernels (conv_split.hip), which want every operand
// as fp16 (hi, mid) parts in the slot layout  xs [B][C/8][H][part 2][W][8]  (16 bytes = 8 channels of one pixel; 4 bytes per
// element, the fp32 tensor's footprint).  Written HERE, by the pass that produces the values, the MFMA kernels' staging becomes an
// LDS-DMA copy (no conversion VALU, no ds_write).  A thread owns 8 channels of one pixel: 8 coalesced 4-byte loads (one per channel
// plane), the arithmetic of the fp32 pass -- the values split are bit for bit the ones bn_relu_apply_kernel / bn_relu_bwd_apply_kernel
// write -- and two 16-byte stores.  C % 8 == 0.
typedef unsigned bn_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void bn_pixel_of(int p, int W, int& y, int& x) {          // p = y W + x without an integer division
    y = (int)((float)p * (1.0f / (float)W));
    if (y * W > p) --y;
    if ((y + 1) * W <= p) ++y;
    x = p - y * W;
}
// np = 2: fp16 (hi | mid) parts of s v (the split kernels); np = 1: ONE part, bf16(v) rounded to nearest even (BASELINE configs[2]'s
// plain-bf16 kernels: the rounding the bf16 convolution kernels apply to their operands)
__device__ __forceinline__ unsigned bn_pack_bf16(float a, float b) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const bf2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void bn_store_slots(unsigned* __restrict__ xs, int64_t slot, int W, const float (&v)[8], float s, int np = 2) {
    bn_u32x4 hi, mid;
    if (np == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) hi[k] = bn_pack_bf16(v[2 * k], v[2 * k + 1]);
        reinterpret_cast<bn_u32x4*>(xs)[slot] = hi;
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned h, m;
        split2h_s(v[2 * k], v[2 * k + 1], s, h, m);
        hi[k] = h;
        mid[k] = m;
    }
    bn_u32x4* d = reinterpret_cast<bn_u32x4*>(xs) + slot;
    d[0] = hi;
    d[W] = mid;
}

// The values of FOUR consecutive pixels per thread (float4 loads along x: the fp32 passes' instruction mix) leave as 16-byte slots of
// ONE pixel each -- written from the loading thread, a store instruction's 64 lanes hit 16 bytes out of every 64 (quarter-filled
// write transactions: measured 0.108 against 0.085 ms per launch for the forward pass).  So a 256-thread block trades its 1024
// pixels x 2 parts through LDS (one padding slot per four: the 64-byte lane stride of the writes becomes 80 bytes, conflict-free
// per 16 lanes) and stores lane-linear: every store instruction writes 1 KB of consecutive slots.
constexpr int BN_TR_SLOTS = 2 * (1024 + 256);       // 40 KB of LDS per block -- HALF of it with one part (plain bf16): the launches ask
                                                    // for what they use, so that twice the blocks fit a CU where the pass moves half the
                                                    // bytes per thread (a streaming pass runs at bytes in flight / latency)
__device__ __forceinline__ bn_u32x4* bn_tr_lds() {
    extern __shared__ __attribute__((aligned(16))) unsigned char bn_dyn_lds[];
    return reinterpret_cast<bn_u32x4*>(bn_dyn_lds);
}
static inline unsigned bn_tr_bytes(int nparts) { return (unsigned)(nparts == 1 ? BN_TR_SLOTS / 2 : BN_TR_SLOTS) * 16u; }
#ifndef BN_APPLY_SPLIT_TR
#define BN_APPLY_SPLIT_TR 1
#endif
// addr(px) -> global slot index of the hi part of the block's pixel px (0 .. 1023; < 0: outside), mid_off = slots from hi to mid
template <typename AddrFn>
__device__ __forceinline__ void bn_store_slots_block_fn(bn_u32x4* lds, unsigned* __restrict__ xs, int mid_off, const float (&v)[4][8], float s,
                                                        AddrFn addr, int np = 2) {
    const int t = threadIdx.x;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bn_u32x4 hi, mid;
        if (np == 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) hi[k] = bn_pack_bf16(v[q][2 * k], v[q][2 * k + 1]);
            lds[5 * t + q] = hi;
            continue;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned h, m;
            split2h_s(v[q][2 * k], v[q][2 * k + 1], s, h, m);
            hi[k] = h;
            mid[k] = m;
        }
        lds[5 * t + q] = hi;
        lds[1280 + 5 * t + q] = mid;
    }
    __syncthreads();
    bn_u32x4* dst = reinterpret_cast<bn_u32x4*>(xs);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int part = j >> 2, px = (j & 3) * 256 + t;
        if (part >= np) break;
        const int64_t a = addr(px);
        if (a >= 0) dst[a + (int64_t)part * mid_off] = lds[part * 1280 + px + (px >> 2)];
    }
}

__device__ __forceinline__ void bn_store_slots_block(bn_u32x4* lds, unsigned* __restrict__ xs, int c8, int H, int W, int p_blk, int HW,
                                                     const float (&v)[4][8], float s, int np = 2) {
    if (np == 1) {
        bn_store_slots_block_fn(lds, xs, 0, v, s, [&](int px) -> int64_t {
            const int p = p_blk + px;
            if (p >= HW) return -1;
            int y, x;
            bn_pixel_of(p, W, y, x);
            return (int64_t)(c8 * H + y) * W + x;
        }, 1);
        return;
    }
    const int t = threadIdx.x;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bn_u32x4 hi, mid;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned h, m;
            split2h_s(v[q][2 * k], v[q][2 * k + 1], s, h, m);
            hi[k] = h;
            mid[k] = m;
        }
        lds[5 * t + q] = hi;
        lds[1280 + 5 * t + q] = mid;
    }
    __syncthreads();
    bn_u32x4* dst = reinterpret_cast<bn_u32x4*>(xs);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int part = j >> 2, px = (j & 3) * 256 + t, p = p_blk + px;
        if (p < HW) {
            int y, x;
            bn_pixel_of(p, W, y, x);
            dst[((int64_t)(c8 * H + y) * 2 + part) * W + x] = lds[part * 1280 + px + (px >> 2)];
        }
    }
}

// a = relu(bn(z)) -> xs (pre-split) and, where `a` is given, the fp32 tensor too.  A thread owns 8 channels of four pixels 64 apart
// (a wave covers 256 consecutive pixels, four times 64): every load is a coalesced 256-byte row piece of one channel plane and
// every 16-byte slot store lands next to its neighbour lanes' (1 KB contiguous per store instruction).  Measured against the
// variant with four CONSECUTIVE pixels per thread (float4 loads, slot stores 64 bytes apart between lanes): 0.085 vs 0.108 ms per
// launch -- the strided 16-byte stores cost more than the narrower loads.
template <typename ZT = float>
__global__ __launch_bounds__(256) void bn_relu_apply_split_kernel(const ZT* __restrict__ z, int64_t z_bs, unsigned* __restrict__ xs,
                                                                  int64_t xs_bs, float* __restrict__ a, int64_t a_bs,
                                                                  const float* __restrict__ save, int C, int H, int W, int bpp, int np,
                                                                  const unsigned* __restrict__ slots, int gimg) {
    // slots: the activation's magnitude slots (the bound bn_finalize wrote): the fp16 parts are those of s a with the guard scale
    // s = 2^k they select -- 1 unless the bound reaches 2^15 -- and the consumers undo s; plain bf16 (np = 1) needs none
    float s_inv;
    const float s_act = np == 1 ? 1.f : amax_scale(amax_read(slots), false, s_inv);
    const int plane = blockIdx.x / bpp, blk = blockIdx.x % bpp;
    const int C8 = C >> 3, b = plane / C8, c8 = plane % C8;
    if (gimg) save += (int64_t)(b / gimg) * 4 * C;        // statistics groups = consecutive batch slices of gimg images, save [G][4][C]
#if BN_APPLY_SPLIT_TR
    {   // variant: four CONSECUTIVE pixels per thread (float4 loads) and the block's slots traded through LDS for coalesced stores
        bn_u32x4* const tr = bn_tr_lds();            // 40 KB (np = 2) / 20 KB (np = 1: one part) of dynamic LDS
        const int HW = H * W, p = (blk * 256 + threadIdx.x) * 4;
        const bool live = p < HW;
        const ZT* src = z + (int64_t)b * z_bs + (int64_t)c8 * 8 * HW + p;
        float v[4][8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = c8 * 8 + k;
            const float mean = save[c], sc = save[2 * C + c], sh = save[3 * C + c];
            const float4 q = live ? bn_ldz4<ZT>(src + (int64_t)k * HW) : make_float4(0.f, 0.f, 0.f, 0.f);
            v[0][k] = fmaxf(fmaf(q.x - mean, sc, sh), 0.f);
            v[1][k] = fmaxf(fmaf(q.y - mean, sc, sh), 0.f);
            v[2][k] = fmaxf(fmaf(q.z - mean, sc, sh), 0.f);
            v[3][k] = fmaxf(fmaf(q.w - mean, sc, sh), 0.f);
        }
        if (a && live) {
            float* d = a + (int64_t)b * a_bs + (int64_t)c8 * 8 * HW + p;
#pragma unroll
            for (int k = 0; k < 8; ++k) *reinterpret_cast<float4*>(d + (int64_t)k * HW) = make_float4(v[0][k], v[1][k], v[2][k], v[3][k]);
        }
        bn_store_slots_block(tr, xs + (int64_t)b * xs_bs, c8, H, W, blk * 1024, HW, v, s_act, np);
        return;
    }
#endif
    const int HW = H * W, lane = threadIdx.x & 63, p0 = blk * 1024 + (threadIdx.x >> 6) * 256 + lane;
    const ZT* src = z + (int64_t)b * z_bs + (int64_t)c8 * 8 * HW;
    float v[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = p0 + 64 * j;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[j][k] = p < HW ? bn_ldz<ZT>(src + (int64_t)k * HW + p) : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = c8 * 8 + k;
        const float mean = save[c], sc = save[2 * C + c], sh = save[3 * C + c];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j][k] = fmaxf(fmaf(v[j][k] - mean, sc, sh), 0.f);
    }
    unsigned* o = xs + (int64_t)b * xs_bs;
    float* d = a ? a + (int64_t)b * a_bs + (int64_t)c8 * 8 * HW : nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = p0 + 64 * j;
        if (p >= HW) continue;
        int y, x;
        bn_pixel_of(p, W, y, x);
        if (d) {
#pragma unroll
            for (int k = 0; k < 8; ++k) d[(int64_t)k * HW + p] = v[j][k];
        }
        bn_store_slots(o, ((int64_t)(c8 * H + y) * np) * W + x, W, v[j], s_act, np);
    }
}

// ... of an encoder output that is max-pooled next: a thread owns 8 channels of a 2 x 4 pixel patch (W % 4 == 0), writes the activation
// pre-split (xs: the skip groups of a concat buffer) and / or in fp32 (a), and the two pooled pixels pre-split (ys) or in fp32 (yf)
template <typename ZT = float>
__global__ __launch_bounds__(256) void bn_relu_apply_pool_split_kernel(const ZT* __restrict__ z, int64_t z_bs, unsigned* __restrict__ xs,
                                                                       int64_t xs_bs, float* __restrict__ a, int64_t a_bs,
                                                                       unsigned* __restrict__ ys, int64_t ys_bs, float* __restrict__ yf,
                                                                       int64_t yf_bs, const float* __restrict__ save, int C, int H, int W,
                                                                       int bpp, int np, const unsigned* __restrict__ slots, int gimg) {
    float s_inv;
    const float s_act = np == 1 ? 1.f : amax_scale(amax_read(slots), false, s_inv);      // (see bn_relu_apply_split_kernel)
    const int plane = blockIdx.x / bpp, blk = blockIdx.x % bpp;
    const int C8 = C >> 3, b = plane / C8, c8 = plane % C8;
    if (gimg) save += (int64_t)(b / gimg) * 4 * C;        // statistics groups = consecutive batch slices of gimg images, save [G][4][C]
    const int Hp = H >> 1, Wp = W >> 1, W4 = W >> 2, HW = H * W, i = blk * 256 + threadIdx.x;
    const bool live = i < Hp * W4;                  // (no early return: the block transposes its slots through LDS together)
    int yo, q;
    bn_pixel_of(live ? i : 0, W4, yo, q);
    const int64_t in_off = (int64_t)c8 * 8 * HW + (int64_t)(2 * yo) * W + 4 * q;
    const ZT* src = z + (int64_t)b * z_bs + in_off;
    float v[8][8], m[2][8];                     // [pixel: row 0 cols 0-3, row 1 cols 0-3][channel]; [pooled pixel][channel]
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = c8 * 8 + k;
        const float mean = save[c], sc = save[2 * C + c], sh = save[3 * C + c];
        const float4 r0 = bn_ldz4<ZT>(src + (int64_t)k * HW), r1 = bn_ldz4<ZT>(src + (int64_t)k * HW + W);
        v[0][k] = fmaxf(fmaf(r0.x - mean, sc, sh), 0.f);
        v[1][k] = fmaxf(fmaf(r0.y - mean, sc, sh), 0.f);
        v[2][k] = fmaxf(fmaf(r0.z - mean, sc, sh), 0.f);
        v[3][k] = fmaxf(fmaf(r0.w - mean, sc, sh), 0.f);
        v[4][k] = fmaxf(fmaf(r1.x - mean, sc, sh), 0.f);
        v[5][k] = fmaxf(fmaf(r1.y - mean, sc, sh), 0.f);
        v[6][k] = fmaxf(fmaf(r1.z - mean, sc, sh), 0.f);
        v[7][k] = fmaxf(fmaf(r1.w - mean, sc, sh), 0.f);
        m[0][k] = fmaxf(fmaxf(v[0][k], v[1][k]), fmaxf(v[4][k], v[5][k]));
        m[1][k] = fmaxf(fmaxf(v[2][k], v[3][k]), fmaxf(v[6][k], v[7][k]));
    }
    if (a && live) {
        float* d = a + (int64_t)b * a_bs + in_off;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            *reinterpret_cast<float4*>(d + (int64_t)k * HW) = make_float4(v[0][k], v[1][k], v[2][k], v[3][k]);
            *reinterpret_cast<float4*>(d + (int64_t)k * HW + W) = make_float4(v[4][k], v[5][k], v[6][k], v[7][k]);
        }
    }
    if (xs) {
        // the patch rows' slots leave through the block transpose (coalesced stores), one round per patch row: block pixel px =
        // (thread px / 4, column px % 4) of that row
        bn_u32x4* const tr = bn_tr_lds();            // 40 KB (np = 2) / 20 KB (np = 1: one part) of dynamic LDS
        unsigned* o = xs + (int64_t)b * xs_bs;
        const int npatch = Hp * W4, patch0 = blk * 256;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (r) __syncthreads();
            float vr[4][8];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int k = 0; k < 8; ++k) vr[e][k] = live ? v[4 * r + e][k] : 0.f;
            bn_store_slots_block_fn(tr, o, W, vr, s_act, [&](int px) -> int64_t {
                const int pi = patch0 + (px >> 2);
                if (pi >= npatch) return -1;
                int py, pq;
                bn_pixel_of(pi, W4, py, pq);
                return ((int64_t)(c8 * H + 2 * py + r) * np) * W + 4 * pq + (px & 3);
            }, np);
        }
    }
    if (!live) return;
    if (ys) {
        unsigned* o = ys + (int64_t)b * ys_bs;
        bn_store_slots(o, ((int64_t)(c8 * Hp + yo) * np) * Wp + 2 * q, Wp, m[0], s_act, np);
        bn_store_slots(o, ((int64_t)(c8 * Hp + yo) * np) * Wp + 2 * q + 1, Wp, m[1], s_act, np);
    }
    if (yf) {
        float* d = yf + (int64_t)b * yf_bs + (int64_t)c8 * 8 * Hp * Wp + (int64_t)yo * Wp + 2 * q;
#pragma unroll
        for (int k = 0; k < 8; ++k) *reinterpret_cast<float2*>(d + (int64_t)k * Hp * Wp) = make_float2(m[0][k], m[1][k]);
    }
}

// dz of the BatchNorm + ReLU backward (bn_relu_bwd_apply_kernel's arithmetic: fp64 per element, rounded once) pre-split: fp16 parts
// of 2^k dz with k = amax_scale(bound) from the magnitude slots (`slots`: an upper bound of |dz| written by bn_bwd_bound_kernel
// BEFORE this pass; the consumers read the same slots and undo 2^k on their accumulators).  8 channels x 4 pixels per thread.
#ifndef BN_BWD_OCC16
#define BN_BWD_OCC16 4
#endif
template <typename ZT = float>
__global__ __launch_bounds__(256, sizeof(ZT) == 2 ? BN_BWD_OCC16 : 3) void bn_relu_bwd_apply_split_kernel(const float* __restrict__ da, int64_t da_bs, const ZT* __restrict__ z,
                                                                      int64_t z_bs, const float* __restrict__ save, const float* __restrict__ coef,
                                                                      unsigned* __restrict__ dzs, int64_t dzs_bs, const unsigned* __restrict__ slots,
                                                                      int C, int H, int W, int bpp, int np, int gimg) {
    float inv;
    const float s = np == 1 ? 1.f : amax_scale(amax_read(slots), true, inv);     // (plain bf16: fp32's exponent range, no scale)
    const int plane = blockIdx.x / bpp, blk = blockIdx.x % bpp;
    const int C8 = C >> 3, b = plane / C8, c8 = plane % C8;
    if (gimg) {                                           // statistics groups: save, coef [G][4][C]
        save += (int64_t)(b / gimg) * 4 * C;
        if (coef) coef += (int64_t)(b / gimg) * 4 * C;
    }
    bn_u32x4* const tr = bn_tr_lds();            // 40 KB (np = 2) / 20 KB (np = 1: one part) of dynamic LDS
    const int HW = H * W, p = (blk * 256 + threadIdx.x) * 4;
    const bool live = p < HW;                     // (HW % 4 == 0: a thread's four pixels are all inside or all outside)
    const ZT* zs = z + (int64_t)b * z_bs + (int64_t)c8 * 8 * HW + p;
    const float* ds = da + (int64_t)b * da_bs + (int64_t)c8 * 8 * HW + p;
    // (a bf16 z stays PACKED in registers until it is used: 16 VGPRs instead of 32 -- the kernel fits 128 registers and a fourth
    // block per CU, which is what a pass with a third fewer bytes per thread in flight needs to reach the same rate)
    constexpr bool Z16 = sizeof(ZT) == 2;
    typename std::conditional<Z16, uint2, float4>::type zq[8];
    float4 gq[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {                 // all sixteen loads in flight before the first use
        if constexpr (Z16) zq[k] = live ? *reinterpret_cast<const uint2*>(zs + (int64_t)k * HW) : make_uint2(0u, 0u);
        else zq[k] = live ? *reinterpret_cast<const float4*>(zs + (int64_t)k * HW) : make_float4(0.f, 0.f, 0.f, 0.f);
        gq[k] = live ? *reinterpret_cast<const float4*>(ds + (int64_t)k * HW) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float v[4][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = c8 * 8 + k;
        const float mean = save[c], invstd = save[C + c], sc = save[2 * C + c], sh = save[3 * C + c];
        const double c1 = coef ? (double)coef[c] + (double)coef[C + c] : 0.0;
        const double c2 = coef ? (double)coef[2 * C + c] + (double)coef[3 * C + c] : 0.0;
        float zz[4];
        if constexpr (Z16) {
            zz[0] = __builtin_bit_cast(float, zq[k].x << 16); zz[1] = __builtin_bit_cast(float, zq[k].x & 0xffff0000u);
            zz[2] = __builtin_bit_cast(float, zq[k].y << 16); zz[3] = __builtin_bit_cast(float, zq[k].y & 0xffff0000u);
        } else {
            zz[0] = zq[k].x; zz[1] = zq[k].y; zz[2] = zq[k].z; zz[3] = zq[k].w;
        }
        const float gg[4] = {gq[k].x, gq[k].y, gq[k].z, gq[k].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double dy = fmaf(zz[e] - mean, sc, sh) > 0.f ? (double)gg[e] : 0.0;
            v[e][k] = (float)((double)sc * (dy - c1 - (((double)zz[e] - (double)mean) * (double)invstd) * c2));
        }
    }
    bn_store_slots_block(tr, dzs + (int64_t)b * dzs_bs, c8, H, W, blk * 1024, HW, v, s, np);
}

// |dz| <= |scale_c| (max |da| + |c1| + |c2| max |xhat|) with |xhat| <= sqrt(N - 1) for a channel of N values: the maximum over the
// channels goes into the dz magnitude slots (atomicMax; the launches of a twin batch's statistics groups share them).  da_slots:
// the exact max |da| recorded by the pass that reduced da (bn_relu_bwd_reduce_amax and the fused producers).
__global__ __launch_bounds__(64) void bn_bwd_bound_kernel(const float* __restrict__ save, const float* __restrict__ coef,
                                                          const unsigned* __restrict__ da_slots, float xhat_max, unsigned* __restrict__ dz_slots,
                                                          int C) {
    const float da_max = amax_read(da_slots);
    float bound = 0.f;
    for (int c = blockIdx.x * 64 + threadIdx.x; c < C; c += gridDim.x * 64) {
        const float c1 = coef ? fabsf(coef[c]) + fabsf(coef[C + c]) : 0.f, c2 = coef ? fabsf(coef[2 * C + c]) + fabsf(coef[3 * C + c]) : 0.f;
        bound = fmaxf(bound, fabsf(save[2 * C + c]) * (da_max + c1 + c2 * xhat_max));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bound = fmaxf(bound, __shfl_xor(bound, o, 64));
    if (threadIdx.x == 0 && bound == bound) atomicMax(dz_slots + (blockIdx.x & 63) * AMAX_STRIDE, __builtin_bit_cast(unsigned, bound));
}

// backward pass 1: part2[p][c] = (sum dy, sum dy*xhat), dy = da * [(z-mean)*scale+beta > 0].
// Sums are taken in fp64 (ATen's CPU kernel accumulates in double): both sums cancel heavily
// (BN outputs are zero-mean), so fp32 accumulation would cost orders of magnitude of accuracy.
template <typename ZT = float>
__global__ __launch_bounds__(256) void bn_relu_bwd_reduce_kernel(const float* __restrict__ da, int64_t da_bs,
                                                                 const ZT* __restrict__ z, int64_t z_bs,
                                                                 const float* __restrict__ save,
                                                                 float* __restrict__ part2, int C, int HW,
                                                                 int chunks, int chunk_len, unsigned* __restrict__ amax = nullptr,
                                                                 int gimg = 0) {
    __shared__ double red[8];
    float vmax = 0.f;                           // amax: magnitude slots of da (what bounds the dz this layer's apply pass writes)
    const int c = blockIdx.x % C;
    const int p = blockIdx.x / C;
    const int b = p / chunks, ch = p % chunks;
    if (gimg) save += (int64_t)(b / gimg) * 4 * C;        // statistics groups = consecutive batch slices of gimg images, save [G][4][C]
    const float mean = save[c], invstd = save[C + c], sc = save[2 * C + c], sh = save[3 * C + c];
    const double meand = mean, invd = invstd;
    const ZT* zs = z + (int64_t)b * z_bs + (int64_t)c * HW;
    const float* ds = da + (int64_t)b * da_bs + (int64_t)c * HW;
    const int beg = ch * chunk_len, end = min(beg + chunk_len, HW);
    double v[2] = {0.0, 0.0};
    if (((HW & 3) == 0) && ((z_bs & 3) == 0) && ((da_bs & 3) == 0) && ((chunk_len & 3) == 0)) {
        int i = beg + threadIdx.x * 4;
#if ONET_BN_BATCH
        for (; i + 3072 < end; i += 4096) {       // eight 16-byte loads in flight per thread; same summation order as below
            float4 q[4], g[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                q[k] = bn_ldz4<ZT>(zs + i + 1024 * k);
                g[k] = *reinterpret_cast<const float4*>(ds + i + 1024 * k);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float zz[4] = {q[k].x, q[k].y, q[k].z, q[k].w}, gg[4] = {g[k].x, g[k].y, g[k].z, g[k].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double dy = fmaf(zz[e] - mean, sc, sh) > 0.f ? (double)gg[e] : 0.0;
                    v[0] += dy;
                    v[1] += dy * (((double)zz[e] - meand) * invd);
                    vmax = fmaxf(vmax, fabsf(gg[e]));
                }
            }
        }
#endif
        for (; i < end; i += 1024) {
            const float4 q = bn_ldz4<ZT>(zs + i);
            const float4 g = *reinterpret_cast<const float4*>(ds + i);
            const float zz[4] = {q.x, q.y, q.z, q.w}, gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double dy = fmaf(zz[k] - mean, sc, sh) > 0.f ? (double)gg[k] : 0.0;
                v[0] += dy;
                v[1] += dy * (((double)zz[k] - meand) * invd);
                vmax = fmaxf(vmax, fabsf(gg[k]));
            }
        }
    } else {
        for (int i = beg + threadIdx.x; i < end; i += 256) {
            const float zi = bn_ldz<ZT>(zs + i);
            const double dy = fmaf(zi - mean, sc, sh) > 0.f ? (double)ds[i] : 0.0;
            v[0] += dy;
            v[1] += dy * (((double)zi - meand) * invd);
            vmax = fmaxf(vmax, fabsf(ds[i]));
        }
    }
    amax_commit(vmax, amax);
    block_sum_256<double, 2>(v, red);
    if (threadIdx.x == 0) {
        // two floats per sum (hi + lo) keep ~48 bits through the float partial buffer
        float* o = part2 + ((int64_t)p * C + c) * 4;
        o[0] = (float)v[0];
        o[1] = (float)(v[0] - (double)o[0]);
        o[2] = (float)v[1];
        o[3] = (float)(v[1] - (double)o[2]);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part2, int nparts,
                                                             double count, float* dgamma, float* dbeta,
                                                             float* coef, int accumulate, int C, const float* __restrict__ save = nullptr,
                                                             const unsigned* __restrict__ da_slots = nullptr,
                                                             unsigned* __restrict__ dz_slots = nullptr, float xhat_max = 0.f, int groups = 1) {
    // dz_slots (pre-split storage): this channel's bound of |dz| (see bn_bwd_bound_kernel) goes into the dz magnitude slots here,
    // saving that launch; da_slots must be complete (every reduce launch of the tensor precedes the first finalize).
    // groups > 1: the statistics groups of a twin batch in one launch: records [g][nparts][C][4], coef / save [g][4][C]; dgamma / dbeta
    // take the groups' sums one after the other in fp32, as one accumulating launch per group did
    // (256 threads per channel: a layer's 4096 .. 8192 records per group were a 64 .. 128-iteration chain of dependent fp64 adds per
    // thread with 64 -- 16 us per launch, 18 launches per step on the critical path; fixed summation order: waves, then the four
    // wave sums in index order)
    __shared__ double wsum[2][4];
    const float da_max = dz_slots ? amax_read(da_slots) : 0.f;
    const int c = blockIdx.x;
    float acc_b = 0.f, acc_g = 0.f;
    if (accumulate && threadIdx.x == 0) {
        acc_b = dbeta ? dbeta[c] : 0.f;
        acc_g = dgamma ? dgamma[c] : 0.f;
    }
    for (int g = 0; g < groups; ++g) {
        const float* rec = part2 + (int64_t)g * nparts * C * 4;
        double s = 0.0, sx = 0.0;
        for (int p = threadIdx.x; p < nparts; p += 256) {
            const float4 o = *reinterpret_cast<const float4*>(rec + ((int64_t)p * C + c) * 4);
            s += (double)o.x + (double)o.y;
            sx += (double)o.z + (double)o.w;
        }
        s = wave_sum(s);
        sx = wave_sum(sx);
        __syncthreads();                                   // (the previous group's sums have been read)
        if ((threadIdx.x & 63) == 0) {
            wsum[0][threadIdx.x >> 6] = s;
            wsum[1][threadIdx.x >> 6] = sx;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            s = ((wsum[0][0] + wsum[0][1]) + wsum[0][2]) + wsum[0][3];
            sx = ((wsum[1][0] + wsum[1][1]) + wsum[1][2]) + wsum[1][3];
            const bool first = g == 0 && !accumulate;
            acc_b = first ? (float)s : acc_b + (float)s;
            acc_g = first ? (float)sx : acc_g + (float)sx;
            float* cf = coef ? coef + (int64_t)g * 4 * C : nullptr;
            if (cf) {   // (hi, lo) float pairs of c1 = sum dy / N and c2 = sum dy*xhat / N
                const double c1 = s / count, c2 = sx / count;
                cf[c] = (float)c1;
                cf[C + c] = (float)(c1 - (double)(float)c1);
                cf[2 * C + c] = (float)c2;
                cf[3 * C + c] = (float)(c2 - (double)(float)c2);
            }
            if (dz_slots) {
                const double c1 = cf ? s / count : 0.0, c2 = cf ? sx / count : 0.0;
                const float bound = fabsf(save[(int64_t)g * 4 * C + 2 * C + c]) *
                                    (da_max + (float)fabs(c1) * 1.0000002f + (float)fabs(c2) * 1.0000002f * xhat_max);
                if (bound == bound) atomicMax(dz_slots + (c & 63) * AMAX_STRIDE, __builtin_bit_cast(unsigned, bound));
            }
        }
    }
    if (threadIdx.x == 0) {
        if (dbeta) dbeta[c] = acc_b;
        if (dgamma) dgamma[c] = acc_g;
    }
}

// the same finalize for the (sum dy, sum dy*xhat) records the F(4x4) dgrad epilogue writes per 16 x 32-pixel block,
// channel-major: part2[c * c_stride + 2 * p] (fp32 sums over 512 pixels each, merged here in fp64)
__global__ __launch_bounds__(256) void bn_bwd_finalize_cm_kernel(const float* __restrict__ part2, int nparts,
                                                                int64_t c_stride, double count, float* dgamma,
                                                                float* dbeta, float* coef, int accumulate, int C) {
    __shared__ double red[16];
    const int c = blockIdx.x;
    const float2* src = reinterpret_cast<const float2*>(part2 + (int64_t)c * c_stride);
    double v[2] = {0.0, 0.0};
    for (int p = threadIdx.x; p < nparts; p += 256) {
        const float2 o = src[p];
        v[0] += (double)o.x;
        v[1] += (double)o.y;
    }
    block_sum_256<double, 2>(v, red);
    if (threadIdx.x == 0) {
        const double s = v[0], sx = v[1];
        if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)s : (float)s;
        if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)sx : (float)sx;
        if (coef) {
            const double c1 = s / count, c2 = sx / count;
            coef[c] = (float)c1;
            coef[C + c] = (float)(c1 - (double)(float)c1);
            coef[2 * C + c] = (float)c2;
            coef[3 * C + c] = (float)(c2 - (double)(float)c2);
        }
    }
}

// backward pass 2: dz = scale * (dy - c1 - xhat*c2)  (train)  |  dz = scale*dy (eval, coef == NULL).
// Evaluated per element in fp64 and rounded once, as ATen's CPU kernel does (accscalar_t = double):
// the three terms cancel (strongly when B*H*W per channel is small), so fp32 arithmetic here
// is what limits end-to-end gradient parity, and fp64 VALU is free in this HBM-bound pass.
__global__ __launch_bounds__(256) void bn_relu_bwd_apply_kernel(const float* __restrict__ da, int64_t da_bs,
                                                                const float* __restrict__ z, int64_t z_bs,
                                                                const float* __restrict__ save,
                                                                const float* __restrict__ coef,
                                                                float* __restrict__ dz, int64_t dz_bs, int C,
                                                                int HW, int chunks, unsigned* __restrict__ amax = nullptr, int gimg = 0) {
    // amax: 64 magnitude slots of dz (fp32 bit patterns; atomicMax is an order-independent maximum): the fp16-split convolution
    // kernels that consume dz scale it by a power of two chosen from this (conv_split.hip, amax_scale)
    float vmax = 0.f;
    const int plane = blockIdx.x / chunks, ch = blockIdx.x % chunks;
    const int b = plane / C, c = plane % C;
    if (gimg) {                                           // statistics groups: save, coef [G][4][C]
        save += (int64_t)(b / gimg) * 4 * C;
        if (coef) coef += (int64_t)(b / gimg) * 4 * C;
    }
    const float mean = save[c], invstd = save[C + c], sc = save[2 * C + c], sh = save[3 * C + c];
    const double meand = mean, invd = invstd, scd = sc;
    const double c1 = coef ? (double)coef[c] + (double)coef[C + c] : 0.0;
    const double c2 = coef ? (double)coef[2 * C + c] + (double)coef[3 * C + c] : 0.0;
    const float* zs = z + (int64_t)b * z_bs + (int64_t)c * HW;
    const float* ds = da + (int64_t)b * da_bs + (int64_t)c * HW;
    float* out = dz + (int64_t)b * dz_bs + (int64_t)c * HW;
    const int beg = ch * 4096, end = min(beg + 4096, HW);
    if (((HW & 3) == 0) && ((z_bs & 3) == 0) && ((da_bs & 3) == 0) && ((dz_bs & 3) == 0)) {
#if ONET_BN_BATCH
        if (end - beg == 4096) {          // full chunk: the thread's eight 16-byte loads in flight before the first use
            const int i0 = beg + threadIdx.x * 4;
            float4 q[4], g[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                q[k] = *reinterpret_cast<const float4*>(zs + i0 + 1024 * k);
                g[k] = *reinterpret_cast<const float4*>(ds + i0 + 1024 * k);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float zz[4] = {q[k].x, q[k].y, q[k].z, q[k].w}, gg[4] = {g[k].x, g[k].y, g[k].z, g[k].w};
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double dy = fmaf(zz[e] - mean, sc, sh) > 0.f ? (double)gg[e] : 0.0;
                    o[e] = (float)(scd * (dy - c1 - (((double)zz[e] - meand) * invd) * c2));
                }
                *reinterpret_cast<float4*>(out + i0 + 1024 * k) = make_float4(o[0], o[1], o[2], o[3]);
                vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
            }
            amax_commit(vmax, amax);
            return;
        }
#endif
        for (int i = beg + threadIdx.x * 4; i < end; i += 1024) {
            const float4 q = *reinterpret_cast<const float4*>(zs + i);
            const float4 g = *reinterpret_cast<const float4*>(ds + i);
            const float zz[4] = {q.x, q.y, q.z, q.w}, gg[4] = {g.x, g.y, g.z, g.w};
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double dy = fmaf(zz[k] - mean, sc, sh) > 0.f ? (double)gg[k] : 0.0;
                o[k] = (float)(scd * (dy - c1 - (((double)zz[k] - meand) * invd) * c2));
            }
            *reinterpret_cast<float4*>(out + i) = make_float4(o[0], o[1], o[2], o[3]);
            vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
        }
    } else {
        for (int i = beg + threadIdx.x; i < end; i += 256) {
            const double dy = fmaf(zs[i] - mean, sc, sh) > 0.f ? (double)ds[i] : 0.0;
            const float v = (float)(scd * (dy - c1 - (((double)zs[i] - meand) * invd) * c2));
            out[i] = v;
            vmax = fmaxf(vmax, fabsf(v));
        }
    }
    amax_commit(vmax, amax);
}

// nparts must be B * chunks with chunks = ceil(HW / chunk_len); we derive chunk_len from nparts
static bool split_plan(int nparts, int B, int HW, int& chunks, int& chunk_len) {
    if (nparts <= 0 || nparts % B) return false;
    chunks = nparts / B;
    chunk_len = (HW + chunks - 1) / chunks;
    chunk_len = (chunk_len + 3) & ~3;
    return (int64_t)chunks * chunk_len >= HW;
}

extern "C" {

int onet_bn_stats_partial(const float* z, int64_t z_bs, float* part, int nparts, int B, int C, int HW,
                          void* stream) {
    ONET_REQUIRE(z && part && B > 0 && C > 0 && HW > 0, "bn_stats_partial: bad args");
    int chunks, chunk_len;
    ONET_REQUIRE(split_plan(nparts, B, HW, chunks, chunk_len), "bn_stats_partial: nparts=%d must be a multiple of B=%d", nparts, B);
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3((unsigned)((int64_t)nparts * C)), dim3(256), 0,
                       as_stream(stream), z, z_bs, part, C, HW, chunks, chunk_len);
    return check_launch("bn_stats_partial_kernel");
}

int onet_bn_finalize(const float* part, int nparts, int64_t count, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, float momentum, float eps, float* save, void* act_amax, int C, void* stream) {
    ONET_REQUIRE(part && save && nparts > 0 && count > 0 && C > 0, "bn_finalize: bad args");
    (void)count;   // the partials carry their own counts
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64), 0, as_stream(stream), part, nparts, gamma, beta, running_mean, running_var,
                       momentum, eps, save, C, (unsigned*)act_amax);
    return check_launch("bn_finalize_kernel");
}

int onet_bn_finalize_cm(const float* part, int nparts, int64_t c_stride, const float* gamma, const float* beta, float* running_mean,
                            float* running_var, float momentum, float eps, float* save, void* act_amax, int groups, int C, void* stream) {
    ONET_REQUIRE(part && save && nparts > 0 && C > 0 && groups >= 1 && c_stride >= (int64_t)nparts * groups * 3, "bn_finalize_cm: bad args");
    hipLaunchKernelGGL(bn_finalize_cm_kernel, dim3(C), dim3(256), 0, as_stream(stream), part, nparts, c_stride, gamma, beta, running_mean,
                       running_var, momentum, eps, save, C, (unsigned*)act_amax, groups);
    return check_launch("bn_finalize_cm_kernel");
}

int onet_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float eps, float* save, int C, void* stream) {
    ONET_REQUIRE(running_mean && running_var && save && C > 0, "bn_eval_coeffs: bad args");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(cdiv(C, 256)), dim3(256), 0, as_stream(stream), gamma, beta,
                       running_mean, running_var, eps, save, C);
    return check_launch("bn_eval_coeffs_kernel");
}

int onet_bn_relu_apply(const void* z, int z_bf16, int64_t z_bs, float* a, int64_t a_bs, const float* save, void* amax, int group_images,
                            int B, int C, int HW, void* stream) {
    ONET_REQUIRE(z && a && save && B > 0 && C > 0 && HW > 0 && group_images >= 0, "bn_relu_apply: bad args");
    const int chunks = cdiv(HW, 4096);
    const int64_t blocks = (int64_t)B * C * chunks;
    ONET_REQUIRE(blocks < (1ll << 31), "bn_relu_apply: grid too large");
    if (z_bf16) {
        ONET_REQUIRE((reinterpret_cast<uintptr_t>(z) & 7) == 0, "bn_relu_apply: 8-byte aligned bf16 rows required");
        hipLaunchKernelGGL(bn_relu_apply_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (const __bf16*)z, z_bs, a, a_bs,
                           save, C, HW, chunks, (unsigned*)amax, group_images);
    } else {
        hipLaunchKernelGGL(bn_relu_apply_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (const float*)z, z_bs, a, a_bs,
                           save, C, HW, chunks, (unsigned*)amax, group_images);
    }
    return check_launch("bn_relu_apply_kernel");
}

int onet_bn_relu_apply_pool(const float* z, int64_t z_bs, float* a, int64_t a_bs, float* y, int64_t y_bs, const float* save,
                                 void* amax, int B, int C, int H, int W, void* stream) {
    ONET_REQUIRE(z && a && y && save && B > 0 && C > 0 && H > 0 && W > 0, "bn_relu_apply_pool: bad args");
    auto al = [](const void* p, uintptr_t m) { return (reinterpret_cast<uintptr_t>(p) & m) == 0; };
    if ((H & 1) || (W & 3) || (z_bs & 3) || (a_bs & 3) || (y_bs & 1) || !al(z, 15) || !al(a, 15) || !al(y, 7)) return 1;
    const int npatch = (H / 2) * (W / 4), bpp = cdiv(npatch, 256);
    const int64_t blocks = (int64_t)B * C * bpp;
    ONET_REQUIRE(blocks < (1ll << 31), "bn_relu_apply_pool: grid too large");
    hipLaunchKernelGGL(bn_relu_apply_pool_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), z, z_bs, a, a_bs,
                       y, y_bs, save, C, H, W, bpp, (unsigned*)amax);
    return check_launch("bn_relu_apply_pool_kernel");
}

int onet_bn_relu_bwd_reduce(const float* da, int64_t da_bs, const void* z, int z_bf16, int64_t z_bs, const float* save, float* part2,
                                 int nparts, void* da_amax, int group_images, int B, int C, int HW, void* stream) {
    ONET_REQUIRE(da && z && save && part2 && B > 0 && C > 0 && HW > 0 && group_images >= 0, "bn_relu_bwd_reduce: bad args");
    int chunks, chunk_len;
    ONET_REQUIRE(split_plan(nparts, B, HW, chunks, chunk_len), "bn_relu_bwd_reduce: nparts=%d must be a multiple of B=%d", nparts, B);
    const dim3 grid((unsigned)((int64_t)nparts * C));
    if (z_bf16) {
        ONET_REQUIRE((reinterpret_cast<uintptr_t>(z) & 7) == 0, "bn_relu_bwd_reduce: 8-byte aligned bf16 rows required");
        hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel<__bf16>, grid, dim3(256), 0, as_stream(stream), da, da_bs, (const __bf16*)z, z_bs, save,
                           part2, C, HW, chunks, chunk_len, (unsigned*)da_amax, group_images);
    } else {
        hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel<float>, grid, dim3(256), 0, as_stream(stream), da, da_bs, (const float*)z, z_bs, save,
                           part2, C, HW, chunks, chunk_len, (unsigned*)da_amax, group_images);
    }
    return check_launch("bn_relu_bwd_reduce_kernel");
}

int onet_bn_bwd_bound(const float* save, const float* coef, const void* da_amax, int64_t count, void* dz_amax, int C, void* stream) {
    ONET_REQUIRE(save && da_amax && dz_amax && count > 0 && C > 0, "bn_bwd_bound: bad args");
    hipLaunchKernelGGL(bn_bwd_bound_kernel, dim3(cdiv(C, 64)), dim3(64), 0, as_stream(stream), save, coef, (const unsigned*)da_amax,
                       sqrtf((float)(count > 1 ? count - 1 : 1)), (unsigned*)dz_amax, C);
    return check_launch("bn_bwd_bound_kernel");
}

int onet_bn_relu_apply_split(const void* z, int z_bf16, int64_t z_bs, void* xs, int64_t xs_bs, float* a, int64_t a_bs, const float* save,
                             const void* act_amax, int nparts, int group_images, int B, int C, int H, int W, void* stream) {
    ONET_REQUIRE(nparts == 1 || nparts == 2, "bn_relu_apply_split: nparts must be 2 (fp16 hi | mid) or 1 (plain bf16)");
    ONET_REQUIRE(z && xs && save && B > 0 && C > 0 && (C % 8) == 0 && H > 0 && W > 0 && (W % 4) == 0, "bn_relu_apply_split: bad args (C %% 8 == 0, W %% 4 == 0)");
    ONET_REQUIRE((reinterpret_cast<uintptr_t>(xs) & 15) == 0 && (xs_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(z) & 15) == 0 && (z_bs & 3) == 0 &&
                 (reinterpret_cast<uintptr_t>(a) & 15) == 0 && (a_bs & 3) == 0, "bn_relu_apply_split: 16-byte aligned rows and slots required");
    const int bpp = cdiv((int64_t)H * W, 1024);
    const int64_t blocks = (int64_t)B * (C / 8) * bpp;
    ONET_REQUIRE(blocks < (1ll << 31) && (int64_t)H * W < (1 << 24), "bn_relu_apply_split: grid too large");
    if (z_bf16)
        hipLaunchKernelGGL(bn_relu_apply_split_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), bn_tr_bytes(nparts), as_stream(stream), (const __bf16*)z, z_bs,
                           (unsigned*)xs, xs_bs, a, a_bs, save, C, H, W, bpp, nparts, (const unsigned*)act_amax, group_images);
    else
        hipLaunchKernelGGL(bn_relu_apply_split_kernel<float>, dim3((unsigned)blocks), dim3(256), bn_tr_bytes(nparts), as_stream(stream), (const float*)z, z_bs,
                           (unsigned*)xs, xs_bs, a, a_bs, save, C, H, W, bpp, nparts, (const unsigned*)act_amax, group_images);
    return check_launch("bn_relu_apply_split_kernel");
}

int onet_bn_relu_apply_pool_split(const void* z, int z_bf16, int64_t z_bs, void* xs, int64_t xs_bs, float* a, int64_t a_bs, void* ys, int64_t ys_bs,
                                  float* y, int64_t y_bs, const float* save, const void* act_amax, int nparts, int group_images, int B, int C,
                                  int H, int W, void* stream) {
    ONET_REQUIRE(nparts == 1 || nparts == 2, "bn_relu_apply_pool_split: nparts must be 2 (fp16 hi | mid) or 1 (plain bf16)");
    ONET_REQUIRE(z && (xs || a) && (ys || y) && save && B > 0 && C > 0 && (C % 8) == 0 && H > 0 && W > 0, "bn_relu_apply_pool_split: bad args");
    auto al = [](const void* p, uintptr_t m) { return (reinterpret_cast<uintptr_t>(p) & m) == 0; };
    if ((H & 1) || (W & 3) || (z_bs & 3) || (a_bs & 3) || (xs_bs & 3) || (ys_bs & 3) || (y_bs & 1) || !al(z, 15) || !al(a, 15) || !al(xs, 15) ||
        !al(ys, 15) || !al(y, 7))
        return 1;                                  // not taken
    const int bpp = cdiv((int64_t)(H / 2) * (W / 4), 256);
    const int64_t blocks = (int64_t)B * (C / 8) * bpp;
    ONET_REQUIRE(blocks < (1ll << 31) && (int64_t)H * W < (1 << 24), "bn_relu_apply_pool_split: grid too large");
    if (z_bf16)
        hipLaunchKernelGGL(bn_relu_apply_pool_split_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), bn_tr_bytes(nparts), as_stream(stream), (const __bf16*)z, z_bs,
                           (unsigned*)xs, xs_bs, a, a_bs, (unsigned*)ys, ys_bs, y, y_bs, save, C, H, W, bpp, nparts, (const unsigned*)act_amax,
                           group_images);
    else
        hipLaunchKernelGGL(bn_relu_apply_pool_split_kernel<float>, dim3((unsigned)blocks), dim3(256), bn_tr_bytes(nparts), as_stream(stream), (const float*)z, z_bs,
                           (unsigned*)xs, xs_bs, a, a_bs, (unsigned*)ys, ys_bs, y, y_bs, save, C, H, W, bpp, nparts, (const unsigned*)act_amax,
                           group_images);
    return check_launch("bn_relu_apply_pool_split_kernel");
}

int onet_bn_relu_bwd_apply_split(const float* da, int64_t da_bs, const void* z, int z_bf16, int64_t z_bs, const float* save, const float* coef,
                                 void* dzs, int64_t dzs_bs, const void* dz_amax, int nparts, int group_images, int B, int C, int H, int W,
                                 void* stream) {
    ONET_REQUIRE(nparts == 1 || nparts == 2, "bn_relu_bwd_apply_split: nparts must be 2 (fp16 hi | mid) or 1 (plain bf16)");
    ONET_REQUIRE(da && z && save && dzs && (dz_amax || nparts == 1) && B > 0 && C > 0 && (C % 8) == 0 && H > 0 && W > 0 && (W % 4) == 0,
                 "bn_relu_bwd_apply_split: bad args");
    ONET_REQUIRE((reinterpret_cast<uintptr_t>(dzs) & 15) == 0 && (dzs_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(z) & 15) == 0 && (z_bs & 3) == 0 &&
                 (reinterpret_cast<uintptr_t>(da) & 15) == 0 && (da_bs & 3) == 0, "bn_relu_bwd_apply_split: 16-byte aligned rows and slots required");
    const int bpp = cdiv((int64_t)H * W, 1024);
    const int64_t blocks = (int64_t)B * (C / 8) * bpp;
    ONET_REQUIRE(blocks < (1ll << 31) && (int64_t)H * W < (1 << 24), "bn_relu_bwd_apply_split: grid too large");
    if (z_bf16)
        hipLaunchKernelGGL(bn_relu_bwd_apply_split_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), bn_tr_bytes(nparts), as_stream(stream), da, da_bs,
                           (const __bf16*)z, z_bs, save, coef, (unsigned*)dzs, dzs_bs, (const unsigned*)dz_amax, C, H, W, bpp, nparts, group_images);
    else
        hipLaunchKernelGGL(bn_relu_bwd_apply_split_kernel<float>, dim3((unsigned)blocks), dim3(256), bn_tr_bytes(nparts), as_stream(stream), da, da_bs,
                           (const float*)z, z_bs, save, coef, (unsigned*)dzs, dzs_bs, (const unsigned*)dz_amax, C, H, W, bpp, nparts, group_images);
    return check_launch("bn_relu_bwd_apply_split_kernel");
}

int onet_bn_bwd_finalize_cm(const float* part2, int nparts, int64_t c_stride, int64_t count, float* dgamma, float* dbeta,
                            float* coef, int accumulate, int C, void* stream) {
    ONET_REQUIRE(part2 && nparts > 0 && count > 0 && C > 0 && c_stride >= (int64_t)nparts * 2, "bn_bwd_finalize_cm: bad args");
    hipLaunchKernelGGL(bn_bwd_finalize_cm_kernel, dim3(C), dim3(256), 0, as_stream(stream), part2, nparts, c_stride,
                       (double)count, dgamma, dbeta, coef, accumulate, C);
    return check_launch("bn_bwd_finalize_cm_kernel");
}

int onet_bn_bwd_finalize(const float* part2, int nparts, int64_t count, float* dgamma, float* dbeta, float* coef, int accumulate,
                               int groups, int C, const float* save, const void* da_amax, void* dz_amax, void* stream) {
    ONET_REQUIRE(part2 && nparts > 0 && count > 0 && C > 0 && groups >= 1 && (!dz_amax || (save && da_amax)), "bn_bwd_finalize: bad args");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, as_stream(stream), part2, nparts, (double)count, dgamma, dbeta, coef,
                       accumulate, C, save, (const unsigned*)da_amax, (unsigned*)dz_amax, sqrtf((float)(count > 1 ? count - 1 : 1)), groups);
    return check_launch("bn_bwd_finalize_kernel");
}

int onet_bn_relu_bwd_apply(const float* da, int64_t da_bs, const float* z, int64_t z_bs, const float* save, const float* coef,
                                float* dz, int64_t dz_bs, void* amax, int group_images, int B, int C, int HW, void* stream) {
    ONET_REQUIRE(da && z && save && dz && B > 0 && C > 0 && HW > 0 && group_images >= 0, "bn_relu_bwd_apply: bad args");
    const int chunks = cdiv(HW, 4096);
    const int64_t blocks = (int64_t)B * C * chunks;
    ONET_REQUIRE(blocks < (1ll << 31), "bn_relu_bwd_apply: grid too large");
    hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), da, da_bs, z, z_bs, save, coef,
                       dz, dz_bs, C, HW, chunks, (unsigned*)amax, group_images);
    return check_launch("bn_relu_bwd_apply_kernel");
}

}  // extern "C"
