// gdkvm_api.hip -- ABI bookkeeping: version, thread-local error string, device check.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "gdkvm_common.hpp"

static thread_local char g_err[512] = "";

int gdkvm_fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int gdkvm_check_device(void)
{
    static thread_local int cached_dev = -1;
    static thread_local int cached_rc = 0;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return gdkvm_fail(GDKVM_ERR_ARCH, "hipGetDevice: %s", hipGetErrorString(e));
    if (dev == cached_dev) return cached_rc;
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return gdkvm_fail(GDKVM_ERR_ARCH, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    cached_dev = dev;
    cached_rc = strncmp(prop.gcnArchName, "gfx950", 6) == 0
                    ? GDKVM_OK
                    : gdkvm_fail(GDKVM_ERR_ARCH, "device %d is %s; this library is built for gfx950 only", dev, prop.gcnArchName);
    return cached_rc;
}

namespace {
__global__ __launch_bounds__(256) void zero_words_kernel(unsigned* p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
}  // namespace

int gdkvm_zero_async(void* p, size_t bytes, hipStream_t st)
{
    if (!bytes) return GDKVM_OK;
    if (!p || (reinterpret_cast<uintptr_t>(p) & 3u) || (bytes & 3u)) return gdkvm_fail(GDKVM_ERR_ARG, "zero_async: %p / %zu bytes is not a run of 4-byte words", p, bytes);
    const size_t n = bytes / 4, blocks = (n + 255) / 256;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, static_cast<unsigned*>(p), n);
    GDKVM_LAUNCH_CHECK("zero_words_kernel");
    return GDKVM_OK;
}

extern "C" int gdkvm_abi_version(void) { return GDKVM_ABI_VERSION; }
extern "C" const char* gdkvm_last_error(void) { return g_err; }
