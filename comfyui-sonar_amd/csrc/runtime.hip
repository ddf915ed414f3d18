// Error plumbing + ABI version for libsonar_hip.so.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <map>
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
    // The limit has to be raised BEFORE any thread may launch with more than 64 KB of dynamic LDS, so the lock is held across the
    // runtime call and the (kernel, device) key is recorded only once the call has succeeded: a second thread that meets the kernel for
    // the first time waits here instead of launching under the old limit, and a failed call is retried by the next launch (its error
    // text is kept for sonar_last_error(); the launch that follows reports the failure itself through check_launch).
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, int> done;  // (kernel, device) -> bytes granted
    static const bool always = [] { const char* e = getenv("SONAR_LDS_ATTR_ALWAYS"); return e && e[0] == '1'; }();  // (A/B of the per-launch form)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) (void)hipGetLastError();
    std::lock_guard<std::mutex> lock(mu);
    const std::pair<const void*, int> key{kern, dev};
    if (!always) {
        const auto it = done.find(key);
        if (it != done.end() && it->second >= bytes) return;
    }
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize, %d): %s", bytes, hipGetErrorString(e));
        return;
    }
    done[key] = bytes;
}

}  // namespace sonar

extern "C" int sonar_abi_version(void) { return 1; }
extern "C" int sonar_noise_stream_version(void) { return 6; }
extern "C" const char* sonar_last_error(void) { return sonar::g_err; }
