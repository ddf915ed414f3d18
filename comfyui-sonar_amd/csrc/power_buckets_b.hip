// SDXL bucket kernels, second half (see power_buckets.h).
#include "power_buckets.h"

namespace sonar {

int launch_power_bucket_b(SONAR_BUCKET_ARGS) {
#define SONAR_BUCKET_CASE(HH, WW, A, B, C, D) \
    if (H == HH && W == WW) return launch_power_any_t<A, B, C, D>(SONAR_BUCKET_PASS);
    SONAR_BUCKETS_B(SONAR_BUCKET_CASE)
#undef SONAR_BUCKET_CASE
    return kNotABucket;
}

}  // namespace sonar
