// Error plumbing + ABI version for libsonar_hip.so.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <set>
#include <utility>

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

void lds_attr(const void* kern, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    static const bool always = [] { const char* e = getenv("SONAR_LDS_ATTR_ALWAYS"); return e && e[0] == '1'; }();  // (A/B of the per-launch form)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) (void)hipGetLastError();
    if (!always) {
        std::lock_guard<std::mutex> lock(mu);
        if (!done.insert({kern, dev}).second) return;
    }
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) (void)hipGetLastError();
}

}  // namespace sonar

extern "C" int sonar_abi_version(void) { return 1; }
extern "C" const char* sonar_last_error(void) { return sonar::g_err; }
