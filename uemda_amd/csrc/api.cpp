// Error plumbing of the C ABI (no kernels here).
#include "common.h"

static thread_local char g_err[512] = "";

int uem_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
int uem_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return uem_fail(UEM_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return UEM_OK;
}
extern "C" int uem_version(void) { return 100; }
extern "C" const char* uem_last_error(void) { return g_err; }
