// Entry points declared in include/sonar_hip.h whose kernels are not written yet: they fail loudly.
#include "common.h"
using namespace sonar;
#define PENDING(name) do { set_error(name ": not implemented yet"); return SONAR_ERR_UNSUPPORTED; } while (0)
extern "C" int sonar_rfft2_f32(const float*, const float*, float*, int64_t, int64_t, int64_t, void*) { PENDING("sonar_rfft2_f32"); }
extern "C" int64_t sonar_dwt_out_len(int64_t, int64_t, int) { return -1; }
extern "C" int sonar_dwt2_fwd_f32(const float*, float*, float*, int64_t, int64_t, int64_t, const double*, const double*, int, int, void*) { PENDING("sonar_dwt2_fwd_f32"); }
extern "C" int sonar_dwt2_fwd_f64(const double*, double*, double*, int64_t, int64_t, int64_t, const double*, const double*, int, int, void*) { PENDING("sonar_dwt2_fwd_f64"); }
extern "C" int sonar_dwt2_inv_f32(const float*, const float*, float*, int64_t, int64_t, int64_t, int64_t, int64_t, const double*, const double*, int, int, void*) { PENDING("sonar_dwt2_inv_f32"); }
extern "C" int sonar_dwt2_inv_f64(const double*, const double*, double*, int64_t, int64_t, int64_t, int64_t, int64_t, const double*, const double*, int, int, void*) { PENDING("sonar_dwt2_inv_f64"); }
extern "C" int sonar_wcfg_band_f32(const float*, const float*, float*, int64_t, int64_t, const double*, const double*, const double*, const double*, int, double, void*) { PENDING("sonar_wcfg_band_f32"); }
extern "C" int sonar_wcfg_band_f64(const double*, const double*, double*, int64_t, int64_t, const double*, const double*, const double*, const double*, int, double, void*) { PENDING("sonar_wcfg_band_f64"); }
extern "C" int sonar_wcfg_output_f32(const float*, const void*, int, float*, int64_t, int64_t, int64_t, int64_t, int64_t, int, void*) { PENDING("sonar_wcfg_output_f32"); }
