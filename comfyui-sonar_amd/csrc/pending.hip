// Entry points declared in include/sonar_hip.h whose kernels are not written yet: they fail loudly.
#include "common.h"
using namespace sonar;
extern "C" int sonar_rfft2_f32(const float*, const float*, float*, int64_t, int64_t, int64_t, void*) {
    set_error("sonar_rfft2_f32: not implemented yet");
    return SONAR_ERR_UNSUPPORTED;
}
