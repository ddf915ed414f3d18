// The general-size plane kernel (power_any_core.h) instantiated with compile-time factor pairs for the latent sizes of SDXL's resolution
// buckets -- most real SDXL runs are not 1024 x 1024: 832 x 1216 px is a 104 x 152 latent, 896 x 1152 a 112 x 144 one ... -- so that each
// kernel holds exactly its four codelets instead of a run-time switch over seventeen at four call sites (which spilled 32-74 vector
// registers in every instantiation, py/nodes/powernoise.py:356-377 is the path).  Two translation units (power_buckets_a.hip, _b.hip:
// the codelets of 13, 14, 17 and 19 points unroll to thousands of instructions per kernel) share this header; the factor pairs are
// checked against best_split at run time (launch_power_any_t), a bucket out of step falls back to the run-time-size kernel.
#pragma once
#include "power_any_core.h"

namespace sonar {

// X(H, W, hn1, hn2, mn1, mn2): latent H x W (pixels / 8), H = hn1 x hn2, W / 2 = mn1 x mn2 as best_split picks them
#define SONAR_BUCKETS_A(X) X(104, 152, 13, 8, 19, 4) X(152, 104, 19, 8, 13, 4) X(112, 144, 14, 8, 9, 8) X(144, 112, 12, 12, 8, 7)
#define SONAR_BUCKETS_B(X) X(96, 168, 12, 8, 12, 7) X(168, 96, 14, 12, 8, 6) X(80, 192, 10, 8, 12, 8) X(192, 80, 16, 12, 8, 5)

#define SONAR_BUCKET_ARGS                                                                                                                    \
    int what, const float *z, const float *filter, float *out, int64_t planes, int64_t H, int64_t W, uint64_t seed, uint64_t stream_id,       \
        int64_t plane_offset, int group, double *partials, NormArgs na, hipStream_t st, Ahead ah
#define SONAR_BUCKET_PASS what, z, filter, out, planes, H, W, seed, stream_id, plane_offset, group, partials, na, st, ah

int launch_power_bucket_b(SONAR_BUCKET_ARGS);  // power_buckets_b.hip

}  // namespace sonar
