// core.hip -- error reporting + device info for libws3d_hip.so
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>

#include "common.h"

namespace ws3d {

static thread_local char g_err[512] = "";
int g_tune[TUNE_COUNT] = {0};

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
        return WS3D_E_LAUNCH;
    }
    return WS3D_OK;
}

int raise_lds_cap(const void *fn, size_t bytes, const char *what) {
    if (bytes <= 64 * 1024) return WS3D_OK;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, size_t> caps;
    std::lock_guard<std::mutex> lock(mu);
    size_t &cur = caps[std::make_pair(fn, dev)];
    if (cur >= bytes) return WS3D_OK;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        set_error("%s: raising the dynamic LDS cap to %zu B on device %d failed: %s", what, bytes, dev, hipGetErrorString(e));
        (void)hipGetLastError();
        return WS3D_E_LAUNCH;
    }
    cur = bytes;
    return WS3D_OK;
}

}  // namespace ws3d

extern "C" int ws3d_abi_version(void) { return WS3D_ABI_VERSION; }

extern "C" int ws3d_tune(int key, int value) {
    if (key < 0 || key >= ws3d::TUNE_COUNT) { ws3d::set_error("ws3d_tune: unknown key %d", key); return WS3D_E_INVALID; }
    const int prev = ws3d::g_tune[key];
    if (value >= 0) ws3d::g_tune[key] = value;
    return prev;
}

extern "C" int ws3d_dist_mode(void) { return WS3D_DIST_MODE; }

extern "C" const char *ws3d_last_error(void) { return ws3d::g_err; }

extern "C" int ws3d_device_info(char *name, int name_len, int *cu_count, int *lds_bytes_per_block) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    hipDeviceProp_t p;
    if (e == hipSuccess) e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) {
        ws3d::set_error("ws3d_device_info: %s", hipGetErrorString(e));
        return WS3D_E_LAUNCH;
    }
    if (name && name_len > 0) {
        snprintf(name, (size_t)name_len, "%s (%s)", p.name, p.gcnArchName);
    }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (lds_bytes_per_block) *lds_bytes_per_block = (int)p.sharedMemPerBlock;
    return WS3D_OK;
}
