// SDXL bucket kernels, first half, and the dispatcher (see power_buckets.h).
#include "power_buckets.h"

namespace sonar {

int launch_power_bucket(SONAR_BUCKET_ARGS) {
#define SONAR_BUCKET_CASE(HH, WW, A, B, C, D) \
    if (H == HH && W == WW) return launch_power_any_t<A, B, C, D>(SONAR_BUCKET_PASS);
    SONAR_BUCKETS_A(SONAR_BUCKET_CASE)
#undef SONAR_BUCKET_CASE
    return launch_power_bucket_b(SONAR_BUCKET_PASS);
}

}  // namespace sonar

#ifdef SONAR_ANY_TRACE
extern "C" int sonar_debug_any_trace_a(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(sonar::g_any_trace), sizeof(sonar::g_any_trace));
}
#endif
