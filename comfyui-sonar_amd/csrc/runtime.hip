// Error plumbing + ABI version for libsonar_hip.so.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace sonar {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return SONAR_ERR_HIP;
    }
    return SONAR_OK;
}

}  // namespace sonar

extern "C" int sonar_abi_version(void) { return 1; }
extern "C" const char* sonar_last_error(void) { return sonar::g_err; }
