// C-ABI housekeeping of libavsiam_hip.so: error string + version.  Every entry point returns 0 on success,
// a negative code on failure (-1 launch/runtime error, -2 bad argument) and never throws, allocates or
// synchronises; avs_last_error() describes the most recent failure on the calling thread.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

extern "C" void avs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* avs_last_error(void) { return g_err; }

extern "C" int avs_abi_version(void) { return 1; }

// number of compute units of the current device (used by hosts to size split factors); <0 on error
extern "C" int avs_device_cu_count(void) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    return n;
}
