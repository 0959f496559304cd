// Probe (GPU box): does `buffer_load_dwordx4 ... lds` accept a source address that is dword- but not 16-byte aligned?
//   hipcc --offload-arch=gfx950 -O2 tools/proto/dma_probe.hip -o /tmp/dma_probe && /tmp/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned lds_base, unsigned voff) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_base) : "memory");
}
__global__ void probe(const float* src, float* dst, int shift_floats, int nbytes) {
    __shared__ __attribute__((aligned(16))) float lds[256];
    const uint64_t p = reinterpret_cast<uint64_t>(src);
    i32x4 r; r.x = (int)(p & 0xffffffffu); r.y = (int)((p >> 32) & 0xffffu); r.z = nbytes; r.w = 0x00020000;
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
    dma16(r, base, (unsigned)((threadIdx.x * 4 + shift_floats) * 4));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) dst[i] = lds[i];
}
int main() {
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    float *s, *d;
    hipMalloc(&s, 4096); hipMalloc(&d, 1024);
    hipMemcpy(s, h.data(), 4096, hipMemcpyHostToDevice);
    for (int shift = 0; shift < 4; ++shift) {
        hipMemset(d, 0xff, 1024);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, s, d, shift, 4096);
        hipError_t e = hipDeviceSynchronize();
        std::vector<float> o(256);
        hipMemcpy(o.data(), d, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i) bad += (o[i] != (float)(i + shift));
        printf("shift %d floats: %s, mismatches %d (first values %g %g %g %g %g)\n", shift, hipGetErrorString(e), bad, o[0], o[1], o[2], o[3], o[4]);
    }
    // range check: last lane partially out of range
    hipMemset(d, 0xff, 1024);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, s, d, 1, 1024 - 8);   // buffer ends 8 bytes before lane 63's end
    hipDeviceSynchronize();
    std::vector<float> o(256);
    hipMemcpy(o.data(), d, 1024, hipMemcpyDeviceToHost);
    printf("partial OOB (last lane): %g %g %g %g | lane 62: %g %g %g %g\n", o[252], o[253], o[254], o[255], o[248], o[249], o[250], o[251]);
    return 0;
}
