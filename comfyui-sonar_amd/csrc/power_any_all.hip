// The general-size plane kernel with EVERY codelet length in its run-time switches (2 .. 16, 17, 19): for the plane sizes that need a
// factor of 13 .. 19 and are not one of the SDXL buckets (power_buckets_*.hip) -- 136 x 104, 160 x 160, 240 x 136 ...  This instantiation
// is the one kernel family of the library that spills (5-48 vector registers: the allocator provides for the largest of seventeen
// inlined codelets at four call sites); running those factors as direct sums instead costs these sizes 25-60 % more time
// (profiles/r05_sizes.txt).  A translation unit of its own: it compiles as long as everything else in power_fft.hip together.
#include "power_any_core.h"

namespace sonar {

int launch_power_any_all(int what, const float* z, const float* filter, float* out, int64_t planes, int64_t H, int64_t W, uint64_t seed,
                         uint64_t stream_id, int64_t plane_offset, int group, double* partials, NormArgs na, hipStream_t st, Ahead ah) {
    return launch_power_any_t<0, 0, 0, 0, kSetAll>(what, z, filter, out, planes, H, W, seed, stream_id, plane_offset, group, partials, na, st, ah);
}

}  // namespace sonar
