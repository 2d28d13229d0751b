// efts_common.hip -- error reporting, version and device check of libefts_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "efts_internal.h"

static thread_local char g_err[512] = "";

int efts_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

int efts_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return efts_fail(EFTS_ELAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
    return EFTS_OK;
}

extern "C" const char* efts_last_error(void) { return g_err; }
extern "C" int efts_version(void) { return EFTS_ABI_VERSION; }

extern "C" int efts_device_check(void) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
        return efts_fail(EFTS_EDEVICE, "efts_device_check: no HIP device");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return efts_fail(EFTS_EDEVICE, "efts_device_check: built for gfx950, device is %s", prop.gcnArchName);
    efts_gemm_init();
    return EFTS_OK;
}

int efts_num_cus(void) {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}
