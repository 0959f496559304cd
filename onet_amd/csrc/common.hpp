// Shared helpers for the gfx950 kernels of libonet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <stdio.h>
#include "../../include/onet_hip.h"

namespace onet {

void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define ONET_REQUIRE(cond, ...)                    \
    do {                                           \
        if (!(cond)) {                             \
            onet::set_error(__VA_ARGS__);          \
            return ONET_EINVAL;                    \
        }                                          \
    } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// "first launch on THIS device" flag for per-function attributes (hipFuncSetAttribute is per device): one process may
// drive several GPUs, and two host threads may launch at once.
struct PerDeviceOnce {
    std::atomic<uint64_t> seen{0};
    bool first() {
        int d = 0;
        (void)hipGetDevice(&d);
        const uint64_t bit = 1ull << (d & 63);
        return !(seen.fetch_or(bit, std::memory_order_relaxed) & bit);
    }
};

// compute units of the current device (persistent-grid launches); cached per device
inline int device_cu_count() {
    static std::atomic<int> cached[64];
    int d = 0;
    (void)hipGetDevice(&d);
    int n = cached[d & 63].load(std::memory_order_relaxed);
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) n = 256;
        cached[d & 63].store(n, std::memory_order_relaxed);
    }
    return n;
}

// wave64 reductions via DPP/shuffles
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum of NV values for 256-thread blocks; result valid in thread 0
template <typename T, int NV>
__device__ __forceinline__ void block_sum_256(T (&v)[NV], T* smem /* >= 4*NV */) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) smem[wid * NV + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = smem[i] + smem[NV + i] + smem[2 * NV + i] + smem[3 * NV + i];
    }
}

// ---- fp16 operand splitting with per-tensor power-of-two scales (conv_split.hip consumers, bn.hip producers)
// Scaled fp16 split in FOUR VALU per value pair: hi = fp16(s v), mid = fp16(s v - hi) with v_fma_mixlo/mixhi_f16 -- the fma of
// fp32 operands (and, for mid, the fp16 hi part read straight from its register half) rounded once to fp16 into the low / high
// half of the destination.  s is a power of two, so s v is exact and hi, mid are bit for bit what split2h gives for s v (which the
// compiler builds from v_cvt_pk_f16_f32 + two v_cvt_f32_f16 + two v_sub + v_cvt_pk: six VALU per pair): the scale rides free.
__device__ __forceinline__ void split2h_s(float a, float b, float s, unsigned& hi, unsigned& mid) {
    unsigned h, m;
    asm("v_fma_mixlo_f16 %0, %1, %3, 0 op_sel_hi:[0,0,0]\n\tv_fma_mixhi_f16 %0, %2, %3, 0 op_sel_hi:[0,0,0]"
        : "=&v"(h) : "v"(a), "v"(b), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %3, -%4 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %0, %2, %3, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(m) : "v"(a), "v"(b), "v"(s), "v"(h));
    hi = h;
    mid = m;
}

// Per-tensor magnitude slots: 64 unsigned words holding max |v| as fp32 bits (non-negative floats order like their bit patterns,
// so atomicMax on the bits is an order-independent, i.e. deterministic, maximum); producers (bn.hip, the weight pack) spread their
// updates over the 64 words by block index, readers take the maximum of all 64.  -> wave-uniform fp32 amax (0: no information).
constexpr int AMAX_SLOTS = 64, AMAX_STRIDE = 32;      // 64 slots, one per 128-byte line (bn.hip: amax_commit)
__device__ __forceinline__ float amax_read(const unsigned* slots) {
    if (!slots) return 0.f;
    unsigned v = slots[(threadIdx.x & (AMAX_SLOTS - 1)) * AMAX_STRIDE];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o, 64));
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(v));
}
// power-of-two scale for an fp16 split of a tensor whose largest magnitude is amax: `always` -> amax lands in [2^13, 2^14) (the
// gradients: their magnitude is anyone's guess and fp16's 5-bit exponent is narrow); otherwise only when amax >= 2^15 would leave
// fp16's range (the forward activations: a range GUARD that leaves ordinary tensors -- and the bit-identity statements made about
// them -- untouched).  -> scale s = 2^k; the caller undoes it with 2^-k on the accumulators (exact).
__device__ __forceinline__ float amax_scale(float amax, bool always, float& inv) {
    const unsigned bits = __builtin_bit_cast(unsigned, amax);
    const int e = (int)((bits >> 23) & 255u) - 127;                    // floor(log2(amax))
    int k = 0;
    if (bits != 0u && e < 128) k = (always || e >= 15) ? 13 - e : 0;
    k = max(-100, min(100, k));
    inv = __builtin_bit_cast(float, (unsigned)(127 - k) << 23);
    return __builtin_bit_cast(float, (unsigned)(127 + k) << 23);
}

// conv_mfma.hip: dw (+)= sum_k slab[k][tap][co][ci] in the nn.Module layout (deterministic split-K reduction)
int launch_wgrad_reduce(const float* slab, float* dw, int splitK, int taps, int Cout, int Cin, int out_layout, int accumulate,
                        hipStream_t st);

// convt_gemm.hip: DMA-fed 128 x 128 GEMMs of ConvTranspose2d(k=2, s=2); return 1 when the shape is outside their fast path
int convt_gemm_fwd(const float* x, int64_t x_bs, const float* wq, const float* bias, float* y, int64_t y_bs, void* y16, int64_t y16_bs,
                   int B, int Cin, int Ct, int h, int w, int Ho, int Wo, int pt, int pl, int prec, hipStream_t st, int y16_split = 0,
                   const void* out_slots = nullptr);
int convt_out_bound(const float* w, const float* bias, int Cin, int Ct, const void* x_slots, void* out_slots, hipStream_t st);
int64_t convt_gemm_dbias_ws_bytes(int B, int Ct, int h, int w);
int convt_gemm_dgrad(const float* dy, int64_t dy_bs, const float* wd, float* dx, int64_t dx_bs, float* dbias, float* dbias_ws, int B,
                     int Cin, int Ct, int h, int w, int Ho, int Wo, int pt, int pl, int prec, hipStream_t st);
int64_t convt_gemm_wgrad_ws_bytes(int B, int Cin, int Ct, int h, int w);
int convt_gemm_wgrad(const float* x, int64_t x_bs, const float* dy, int64_t dy_bs, float* dw, void* ws, int64_t ws_bytes, int B,
                     int Cin, int Ct, int h, int w, int Ho, int Wo, int pt, int pl, int prec, hipStream_t st);

}  // namespace onet
