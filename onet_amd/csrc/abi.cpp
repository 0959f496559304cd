// Error plumbing + device info for libonet_hip.so (host only).
#include <stdarg.h>
#include <string.h>
#include "common.hpp"

namespace onet {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return ONET_EHIP;
    }
    return ONET_OK;
}
}  // namespace onet

extern "C" {

const char* onet_last_error(void) { return onet::g_err; }

int onet_abi_version(void) { return 4; }

int onet_device_info(int* cu_count, int* lds_bytes, int* wave_size, char* arch, int arch_len) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        onet::set_error("hipGetDevice: %s", hipGetErrorString(e));
        return ONET_EHIP;
    }
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) {
        onet::set_error("hipGetDeviceProperties: %s", hipGetErrorString(e));
        return ONET_EHIP;
    }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)p.sharedMemPerBlock;
    if (wave_size) *wave_size = p.warpSize;
    if (arch && arch_len > 0) {
        strncpy(arch, p.gcnArchName, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return ONET_OK;
}
}
