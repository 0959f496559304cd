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

// conv_mfma.hip: dw (+)= sum_k slab[k][tap][co][ci] in the nn.Module layout (deterministic split-K reduction)
int launch_wgrad_reduce(const float* slab, float* dw, int splitK, int taps, int Cout, int Cin, int out_layout, int accumulate,
                        hipStream_t st);

// convt_gemm.hip: DMA-fed 128 x 128 GEMMs of ConvTranspose2d(k=2, s=2); return 1 when the shape is outside their fast path
int convt_gemm_fwd(const float* x, int64_t x_bs, const float* wq, const float* bias, float* y, int64_t y_bs, void* y16, int64_t y16_bs,
                   int B, int Cin, int Ct, int h, int w, int Ho, int Wo, int pt, int pl, int prec, hipStream_t st);
int64_t convt_gemm_dbias_ws_bytes(int B, int Ct, int h, int w);
int convt_gemm_dgrad(const float* dy, int64_t dy_bs, const float* wd, float* dx, int64_t dx_bs, float* dbias, float* dbias_ws, int B,
                     int Cin, int Ct, int h, int w, int Ho, int Wo, int pt, int pl, int prec, hipStream_t st);
int64_t convt_gemm_wgrad_ws_bytes(int B, int Cin, int Ct, int h, int w);
int convt_gemm_wgrad(const float* x, int64_t x_bs, const float* dy, int64_t dy_bs, float* dw, void* ws, int64_t ws_bytes, int B,
                     int Cin, int Ct, int h, int w, int Ho, int Wo, int pt, int pl, int prec, hipStream_t st);

}  // namespace onet
