// Timing-only prototype of the INNER LOOP of a Winograd F(3x3,4x4) weight-gradient kernel: what
// fraction of the fp32 MFMA pipe does a K-step keep busy once both operands are transformed in registers from
// channel-major LDS strips?  No global staging, no barriers, no epilogue: an upper bound for the real kernel.
//   block = 8 waves = 4 position groups x 2 co halves, wave tile 32 co x 32 ci x 9 positions, K-step = 2 tiles.
//   hipcc --offload-arch=gfx950 -O3 -o wgrad4_inner tools/proto/wgrad4_inner.hip && ./wgrad4_inner
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int HALF>
static __device__ __forceinline__ void bt3(float d0, float d1, float d2, float d3, float d4, float& o0, float& o1, float& o2) {
    if constexpr (HALF == 0) {
        o0 = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
        const float a = fmaf(-4.f, d2, d4), b = fmaf(-4.f, d1, d3);
        o1 = a + b; o2 = a - b;
    } else {
        const float c = d3 - d1, e = d2 - d0;
        o0 = fmaf(2.f, e, c); o1 = fmaf(-2.f, e, c); o2 = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
    }
}
// half of G' (6x4, points 0, +-1, +-2, inf): HALF 0 -> points 0, 1, -1 ; HALF 1 -> 2, -2, inf
template <int HALF>
static __device__ __forceinline__ void g3(float d0, float d1, float d2, float d3, float& o0, float& o1, float& o2) {
    if constexpr (HALF == 0) {
        const float e = d0 + d2, o = d1 + d3;
        o0 = 0.25f * d0; o1 = (-1.f / 6.f) * (e + o); o2 = (-1.f / 6.f) * (e - o);
    } else {
        const float e = fmaf(4.f, d2, d0), o = fmaf(8.f, d3, 2.f * d1);
        o0 = (1.f / 24.f) * (e + o); o1 = (1.f / 24.f) * (e - o); o2 = d3;
    }
}

constexpr int SX = 220, SDZ = 132;      // channel strides (floats): 4 x odd -> conflict-free b128 over 16-lane groups

template <int RH, int CH>
static __device__ __forceinline__ void body(const float* lds, float* out, int iters) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wm = wid & 1;
    const int l31 = lane & 31, kh = lane >> 5;
    f32x16 acc[9];
    for (int p = 0; p < 9; ++p) for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    const float* xb = lds + l31 * SX + RH * 36 + kh * 4;                       // x patch rows RH .. RH+4 of tile kh
    const float* zb = lds + 32 * SX + (wm * 32 + l31) * SDZ + kh * 4;          // dy rows 0..3 of tile kh
    for (int it = 0; it < iters; ++it) {
        const int s = it & 3;                                                  // 4 K-steps per unit: tiles 2s, 2s+1
        float d[5][6];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(xb + r * 36 + s * 8);
            const f32x2 h = *reinterpret_cast<const f32x2*>(xb + r * 36 + s * 8 + 4);
            d[r][0] = q[0]; d[r][1] = q[1]; d[r][2] = q[2]; d[r][3] = q[3]; d[r][4] = h[0]; d[r][5] = h[1];
        }
        float y[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(zb + r * 32 + s * 8);
            y[r][0] = q[0]; y[r][1] = q[1]; y[r][2] = q[2]; y[r][3] = q[3];
        }
        float t[3][5], u[9], gt[3][4], g[9];
#pragma unroll
        for (int c = 0; c < 5; ++c) bt3<RH>(d[0][CH + c], d[1][CH + c], d[2][CH + c], d[3][CH + c], d[4][CH + c], t[0][c], t[1][c], t[2][c]);
#pragma unroll
        for (int i = 0; i < 3; ++i) bt3<CH>(t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], u[i * 3], u[i * 3 + 1], u[i * 3 + 2]);
#pragma unroll
        for (int c = 0; c < 4; ++c) g3<RH>(y[0][c], y[1][c], y[2][c], y[3][c], gt[0][c], gt[1][c], gt[2][c]);
#pragma unroll
        for (int i = 0; i < 3; ++i) g3<CH>(gt[i][0], gt[i][1], gt[i][2], gt[i][3], g[i * 3], g[i * 3 + 1], g[i * 3 + 2]);
#pragma unroll
        for (int p = 0; p < 9; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[p], u[p], acc[p], 0, 0, 0);
    }
    float s = 0.f;
    for (int p = 0; p < 9; ++p) for (int r = 0; r < 16; ++r) s += acc[p][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

__global__ __launch_bounds__(512, 2) void proto(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 32 * SX + 64 * SDZ; i += 512) lds[i] = 1e-3f * (float)((i * 37) % 101 - 50);
    __syncthreads();
    const int pg = (threadIdx.x >> 6) >> 1;
    switch (__builtin_amdgcn_readfirstlane(pg)) {
        case 0: body<0, 0>(lds, out, iters); break;
        case 1: body<0, 1>(lds, out, iters); break;
        case 2: body<1, 0>(lds, out, iters); break;
        default: body<1, 1>(lds, out, iters); break;
    }
}

int main() {
    const int blocks = 256, iters = 8192, lds_bytes = (32 * SX + 64 * SDZ) * 4;
    float* out;
    hipMalloc(&out, blocks * 512 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(proto), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(proto, dim3(blocks), dim3(512), lds_bytes, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        // per block and K-step: 8 waves x 9 MFMAs of 32x32x2 = 4096 FLOP each
        const double mfma_flops = (double)blocks * iters * 8 * 9 * 4096.0;
        const double hw_tf = mfma_flops / (ms * 1e-3) / 1e12;
        printf("rep %d: %.3f ms  MFMA rate %.1f TF = %.1f %% of 157.3  -> effective (x4) %.0f TF; F(2x2) wgrad today: 222 effective\n",
               rep, ms, hw_tf, 100.0 * hw_tf / 157.3, 4.0 * hw_tf);
    }
    return 0;
}
