// Error plumbing of the C ABI (no kernels here).
#include "common.h"

#include <map>
#include <mutex>
#include <utility>

static thread_local char g_err[512] = "";
static thread_local int g_launch_err = 0;          // a launch helper declined to launch (uem_allow_lds): uem_check_launch reports it

int uem_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
// Dynamic-LDS opt-in of a kernel above 48 KiB, once per (kernel, device) and size: the attribute is per device and the C ABI takes
// streams of any device, so the cache is keyed by the current device.  false = the error is recorded and the caller must NOT launch;
// the entry point's uem_check_launch then returns it (nothing relies on HIP's sticky last-error for this).
bool uem_allow_lds(const void* kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return true;
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> granted;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> lock(mu);
        size_t& have = granted[std::make_pair(kernel, dev)];
        if (have >= bytes) return true;
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e == hipSuccess) { have = bytes; return true; }
    }
    g_launch_err = uem_fail(UEM_ERR_LAUNCH, "dynamic LDS opt-in of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    return false;
}
int uem_check_launch(const char* what) {
    if (g_launch_err) {
        const int rc = g_launch_err;
        g_launch_err = 0;
        (void)hipGetLastError();
        return rc;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return uem_fail(UEM_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return UEM_OK;
}
extern "C" int uem_version(void) { return 100; }
extern "C" const char* uem_last_error(void) { return g_err; }
