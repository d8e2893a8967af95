// Error reporting and small runtime queries of the C-ABI library.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void mgen_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int mgen_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        mgen_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return MGEN_ELAUNCH;
    }
    return MGEN_OK;
}

extern "C" const char* mgen_last_error(void) { return g_err; }

extern "C" int mgen_version(void) { return 100; }

// identity of the sources this library was built from (mgen/_srcid.py, passed in by the Makefile)
#ifndef MGEN_SOURCE_ID
#define MGEN_SOURCE_ID "unknown"
#endif
extern "C" const char* mgen_source_id(void) { return MGEN_SOURCE_ID; }

// Fills name[] with the gcnArchName of `device` and returns the CU count (negative on error).
extern "C" int mgen_device_info(int device, char* name, int name_len) {
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, device);
    if (e != hipSuccess) {
        mgen_set_error("mgen_device_info: %s", hipGetErrorString(e));
        return MGEN_ELAUNCH;
    }
    if (name && name_len > 0) snprintf(name, name_len, "%s", p.gcnArchName);
    return p.multiProcessorCount;
}
