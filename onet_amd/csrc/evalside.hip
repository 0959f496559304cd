// "Next" row f-3: the evaluation step either side of the path, kept on the GPU so eval
// batches are not copied to the host per image:
//   * per-frame min-max normalisation  tensor_normal_per_frame (UT:673-689, UT = utils_20231218.py)
//   * 2-class confusion counts per image (TP, FP, FN, TN) from which _acc / _miou / _target_iou /
//     _detection_rate / _false_alarm_rate (UT:100-192) and re_assign_label (UT:410-453) follow
//   * label flip (1 - pred) for the hard re-assignment of UT:429
// HBM-bound streaming reductions: one block per plane / image, wave-shuffle + LDS combine.
#include "common.hpp"

using namespace onet;

__global__ __launch_bounds__(256) void normalise_per_frame_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                  int HW, float spacing) {
    __shared__ float smin[4], smax[4];
    const float* src = x + (int64_t)blockIdx.x * HW;
    float* dst = y + (int64_t)blockIdx.x * HW;
    float lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < HW; i += 256) {
        const float v = src[i];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o, 64));
        hi = fmaxf(hi, __shfl_xor(hi, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        smin[threadIdx.x >> 6] = lo;
        smax[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    lo = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
    hi = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    const float den = (hi - lo) + spacing;      // (max - min + np.spacing(1)) evaluated in fp32 like torch does
    for (int i = threadIdx.x; i < HW; i += 256) dst[i] = (src[i] - lo) / den;
}

// counts[b] = (TP, FP, FN, TN) with positive class 1
__global__ __launch_bounds__(256) void confusion2_kernel(const int64_t* __restrict__ pred,
                                                         const int64_t* __restrict__ target,
                                                         int64_t* __restrict__ counts, int HW) {
    __shared__ int red[4][4];
    const int64_t* p = pred + (int64_t)blockIdx.x * HW;
    const int64_t* t = target + (int64_t)blockIdx.x * HW;
    int c[4] = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < HW; i += 256) {
        const int pp = p[i] != 0, tt = t[i] != 0;
        c[(pp && tt) ? 0 : (pp && !tt) ? 1 : (!pp && tt) ? 2 : 3] += 1;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int v = c[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4)
        counts[(int64_t)blockIdx.x * 4 + threadIdx.x] =
            (int64_t)red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ void flip_labels_kernel(const int64_t* __restrict__ p, int64_t* __restrict__ q, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        q[i] = 1 - p[i];
}

extern "C" {

int onet_normalise_per_frame(const float* x, float* y, int planes, int HW, void* stream) {
    ONET_REQUIRE(x && y && planes > 0 && HW > 0, "normalise_per_frame: bad args");
    hipLaunchKernelGGL(normalise_per_frame_kernel, dim3(planes), dim3(256), 0, as_stream(stream), x, y, HW,
                       2.220446049250313e-16f);
    return check_launch("normalise_per_frame_kernel");
}

int onet_confusion2(const int64_t* pred, const int64_t* target, int64_t* counts, int B, int HW, void* stream) {
    ONET_REQUIRE(pred && target && counts && B > 0 && HW > 0, "confusion2: bad args");
    hipLaunchKernelGGL(confusion2_kernel, dim3(B), dim3(256), 0, as_stream(stream), pred, target, counts, HW);
    return check_launch("confusion2_kernel");
}

int onet_flip_labels(const int64_t* pred, int64_t* out, int64_t n, void* stream) {
    ONET_REQUIRE(pred && out && n > 0, "flip_labels: bad args");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(flip_labels_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), pred, out, n);
    return check_launch("flip_labels_kernel");
}

}  // extern "C"
