// WaveletCFG with the coefficient bands resident in LDS (dwt_bands.h): the sonar_wcfg_bands_* entry points, and the same kernel as the
// "deeper levels" stage of sonar_wcfg_fused_* (dwt_tile.h), where its latent is the level-1 approximation in the workspace.
#include "dwt_bands.h"

using namespace sonar;

namespace sonar {

// levels 2 .. J of the band kernels' step on the level-1 approximation planes [planes][H1][W1] of type T, in place: ll <- kt * Phi_D(ll)
// (or, with `acc`, acc <- acc + Phi_D(ll)).  yh_scales: [levels][3] for the levels 2 .. J.  False: not taken (LDS, filter length).
template <typename T>
static bool bands_deep_impl(const T* ll, T* acc, T* out, int64_t planes, int H1, int W1, int levels, const double* dec_lo, const double* dec_hi,
                            const double* rec_lo, const double* rec_hi, int flen, int mode_fwd, int mode_inv, const double* yh_scales, double yl_scale,
                            hipStream_t st) {
    // acc: out = acc - (-Phi) = acc + Phi
    return wcfg_bands<T, T>(ll, (const T*)nullptr, acc, out, planes, H1, W1, levels, dec_lo, dec_hi, rec_lo, rec_hi, flen, mode_fwd, mode_inv, yh_scales,
                            yl_scale, 0.0, acc ? -1.0 : 1.0, acc ? 1 : 0, st, "sonar_wcfg_fused (deeper levels)") == SONAR_OK;
}

bool bands_deep(const float* ll, float* acc, float* out, int64_t planes, int H1, int W1, int levels, const double* dec_lo, const double* dec_hi,
                const double* rec_lo, const double* rec_hi, int flen, int mode_fwd, int mode_inv, const double* yh_scales, double yl_scale, hipStream_t st) {
    return bands_deep_impl<float>(ll, acc, out, planes, H1, W1, levels, dec_lo, dec_hi, rec_lo, rec_hi, flen, mode_fwd, mode_inv, yh_scales, yl_scale, st);
}
bool bands_deep(const double* ll, double* acc, double* out, int64_t planes, int H1, int W1, int levels, const double* dec_lo, const double* dec_hi,
                const double* rec_lo, const double* rec_hi, int flen, int mode_fwd, int mode_inv, const double* yh_scales, double yl_scale, hipStream_t st) {
    return bands_deep_impl<double>(ll, acc, out, planes, H1, W1, levels, dec_lo, dec_hi, rec_lo, rec_hi, flen, mode_fwd, mode_inv, yh_scales, yl_scale, st);
}

}  // namespace sonar

extern "C" int64_t sonar_wcfg_bands_lds_bytes(int64_t H, int64_t W, int levels, int flen, int mode_fwd, int mode_inv, int elem_size, int per_orientation) {
    size_t lds = 0;
    bool ok;
    if (elem_size == 8) {
        BandsArgs<double> a{};
        ok = bands_plan(a, lds, H, W, levels, flen, mode_fwd, mode_inv, per_orientation != 0);
    } else {
        BandsArgs<float> a{};
        ok = bands_plan(a, lds, H, W, levels, flen, mode_fwd, mode_inv, per_orientation != 0);
    }
    return ok ? (int64_t)lds : -1;
}

extern "C" int sonar_wcfg_bands_f32(const float* a, const float* b, const float* x, float* out, int64_t planes, int64_t H, int64_t W, int levels,
                                    const double* dec_lo, const double* dec_hi, const double* rec_lo, const double* rec_hi, int flen, int mode_fwd,
                                    int mode_inv, const double* yh_scales, double yl_scale, double ku, double kt, int subtract_from_x, void* stream) {
    return wcfg_bands<float, float>(a, b, x, out, planes, H, W, levels, dec_lo, dec_hi, rec_lo, rec_hi, flen, mode_fwd, mode_inv, yh_scales, yl_scale,
                                    ku, kt, subtract_from_x, (hipStream_t)stream, "sonar_wcfg_bands_f32");
}

extern "C" int sonar_wcfg_bands_f64(const float* a, const float* b, const float* x, float* out, int64_t planes, int64_t H, int64_t W, int levels,
                                    const double* dec_lo, const double* dec_hi, const double* rec_lo, const double* rec_hi, int flen, int mode_fwd,
                                    int mode_inv, const double* yh_scales, double yl_scale, double ku, double kt, int subtract_from_x, void* stream) {
    return wcfg_bands<double, float>(a, b, x, out, planes, H, W, levels, dec_lo, dec_hi, rec_lo, rec_hi, flen, mode_fwd, mode_inv, yh_scales, yl_scale,
                                     ku, kt, subtract_from_x, (hipStream_t)stream, "sonar_wcfg_bands_f64");
}

#ifdef SONAR_BANDS_TRACE
extern "C" int sonar_debug_bands_trace(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(sonar::g_bands_trace), sizeof(sonar::g_bands_trace));
}
#endif
