// Dual-tree complex wavelet transform (DTCWT) building blocks: the reference reaches it through pytorch_wavelets'
// DTCWTForward / DTCWTInverse (py/wavelet_functions.py:56-73; Kingsbury's dtwavexfm2 / dtwaveifm2).  Every stage of that
// algorithm -- odd-length column / row filters with symmetric extension (level 1), the decimating dual-tree filters (coldfilt,
// levels >= 2) and the interpolating ones (colifilt) -- is a sparse linear map along ONE axis with a fixed number of taps per output
// row.  The host builds (source index, coefficient) tables for a given length once (comfyui-sonar_amd/py/dtcwt.py) and this file
// applies them: sonar_axis_taps_{f32,f64}.  The quad <-> complex-pair shuffles between levels (q2c / c2q) are the other two kernels.
// HBM-bound gathers; nothing here is shaped for the matrix cores on purpose.
#include <hip/hip_runtime.h>

#include "common.h"

namespace sonar {

// out[o][j][i] (+)= sum_k coef[j][k] * x[o][idx[j][k]][i];  lanes run along i (inner > 1: rows of a plane) or along j (inner == 1)
template <typename T>
__global__ void __launch_bounds__(kBlock) axis_taps_kernel(const T* __restrict__ x, T* out, int64_t outer, int n_in, int n_out, int inner,
                                                            const int* __restrict__ idx, const T* __restrict__ coef, int taps, int accumulate) {
    kernarg_touch_for(x, out, outer, n_in, n_out, inner, idx, coef, taps, accumulate);
    const int64_t per = (int64_t)n_out * inner, total = outer * per;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
        const int64_t o = e / per;
        const int r = (int)(e - o * per);
        const int j = r / inner, i = r - j * inner;
        const T* src = x + o * (int64_t)n_in * inner + i;
        const int* ix = idx + (int64_t)j * taps;
        const T* cf = coef + (int64_t)j * taps;
        T acc = T(0);
        for (int k = 0; k < taps; ++k) acc = fma_t(cf[k], src[(int64_t)ix[k] * inner], acc);
        out[e] = accumulate ? out[e] + acc : acc;
    }
}

// q2c: three real planes lh / hh / hl [P][2h][2w] -> bands [P][6][h][w][2] (15, 45, 75, 105, 135, 165 degrees; re, im):
// quad (a b / c d) -> ((a - d) + i (b + c)) / sqrt 2 and ((a + d) + i (b - c)) / sqrt 2
template <typename T>
__global__ void __launch_bounds__(kBlock) dtcwt_q2c_kernel(const T* __restrict__ lh, const T* __restrict__ hh, const T* __restrict__ hl, T* out,
                                                            int64_t planes, int h, int w) {
    kernarg_touch_for(lh, hh, hl, out, planes, h, w);
    const int64_t hw = (int64_t)h * w, total = planes * 3 * hw;
    const T s = (T)0.70710678118654752440;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
        const int64_t p = e / (3 * hw);
        const int64_t r = e - p * 3 * hw;
        const int pair = (int)(r / hw);
        const int at = (int)(r - (int64_t)pair * hw);
        const int y = at / w, x0 = at - y * w;
        const T* src = (pair == 0 ? lh : pair == 1 ? hh : hl) + p * 4 * hw + (int64_t)(2 * y) * (2 * w) + 2 * x0;
        const T a = src[0], b = src[1], c = src[2 * w], d = src[2 * w + 1];
        // pair 0 -> orientations (0, 5), pair 1 -> (1, 4), pair 2 -> (2, 3)
        T* o1 = out + ((p * 6 + pair) * hw + at) * 2;
        T* o2 = out + ((p * 6 + (5 - pair)) * hw + at) * 2;
        o1[0] = (a - d) * s;
        o1[1] = (b + c) * s;
        o2[0] = (a + d) * s;
        o2[1] = (b - c) * s;
    }
}

// c2q: the inverse shuffle
template <typename T>
__global__ void __launch_bounds__(kBlock) dtcwt_c2q_kernel(const T* __restrict__ bands, T* lh, T* hh, T* hl, int64_t planes, int h, int w) {
    kernarg_touch_for(bands, lh, hh, hl, planes, h, w);
    const int64_t hw = (int64_t)h * w, total = planes * 3 * hw;
    const T s = (T)0.70710678118654752440;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
        const int64_t p = e / (3 * hw);
        const int64_t r = e - p * 3 * hw;
        const int pair = (int)(r / hw);
        const int at = (int)(r - (int64_t)pair * hw);
        const int y = at / w, x0 = at - y * w;
        const T* z1 = bands + ((p * 6 + pair) * hw + at) * 2;
        const T* z2 = bands + ((p * 6 + (5 - pair)) * hw + at) * 2;
        const T pr = (z1[0] + z2[0]) * s, pi = (z1[1] + z2[1]) * s;  // a + i b
        const T qr = (z1[0] - z2[0]) * s, qi = (z1[1] - z2[1]) * s;  // -d + i c
        T* dst = (pair == 0 ? lh : pair == 1 ? hh : hl) + p * 4 * hw + (int64_t)(2 * y) * (2 * w) + 2 * x0;
        dst[0] = pr;
        dst[1] = pi;
        dst[2 * w] = qi;
        dst[2 * w + 1] = -qr;
    }
}

template <typename T>
static int axis_taps(const T* x, T* out, int64_t outer, int64_t n_in, int64_t n_out, int64_t inner, const int* idx, const T* coef, int taps,
                     int accumulate, hipStream_t st, const char* what) {
    SONAR_REQUIRE(x && out && idx && coef && x != out && outer >= 0 && n_in > 0 && n_out > 0 && inner > 0 && taps > 0 && taps <= 64, SONAR_ERR_ARG,
                  "%s: bad argument", what);
    SONAR_REQUIRE(n_in < (1 << 24) && n_out < (1 << 24) && inner < (1 << 24), SONAR_ERR_UNSUPPORTED, "%s: axis too long", what);
    if (outer == 0) return SONAR_OK;
    hipLaunchKernelGGL((axis_taps_kernel<T>), dim3(grid_for(outer * n_out * inner, kBlock * 2)), dim3(kBlock), 0, st, x, out, outer, (int)n_in,
                       (int)n_out, (int)inner, idx, coef, taps, accumulate);
    return check_launch(what);
}

}  // namespace sonar

using namespace sonar;

extern "C" int sonar_axis_taps_f32(const float* x, float* out, int64_t outer, int64_t n_in, int64_t n_out, int64_t inner, const int* idx,
                                   const float* coef, int taps, int accumulate, void* stream) {
    return axis_taps<float>(x, out, outer, n_in, n_out, inner, idx, coef, taps, accumulate, (hipStream_t)stream, "sonar_axis_taps_f32");
}
extern "C" int sonar_axis_taps_f64(const double* x, double* out, int64_t outer, int64_t n_in, int64_t n_out, int64_t inner, const int* idx,
                                   const double* coef, int taps, int accumulate, void* stream) {
    return axis_taps<double>(x, out, outer, n_in, n_out, inner, idx, coef, taps, accumulate, (hipStream_t)stream, "sonar_axis_taps_f64");
}

#define SONAR_Q2C(NAME, T)                                                                                                                      \
    extern "C" int sonar_dtcwt_q2c_##NAME(const T* lh, const T* hh, const T* hl, T* bands, int64_t planes, int64_t h, int64_t w, void* stream) { \
        SONAR_REQUIRE(lh && hh && hl && bands && planes >= 0 && h > 0 && w > 0 && h < (1 << 15) && w < (1 << 15), SONAR_ERR_ARG,                 \
                      "sonar_dtcwt_q2c_" #NAME ": bad argument");                                                                                \
        if (planes == 0) return SONAR_OK;                                                                                                        \
        hipLaunchKernelGGL((dtcwt_q2c_kernel<T>), dim3(grid_for(planes * 3 * h * w, kBlock * 2)), dim3(kBlock), 0, (hipStream_t)stream, lh, hh,  \
                           hl, bands, planes, (int)h, (int)w);                                                                                   \
        return check_launch("sonar_dtcwt_q2c_" #NAME);                                                                                           \
    }                                                                                                                                            \
    extern "C" int sonar_dtcwt_c2q_##NAME(const T* bands, T* lh, T* hh, T* hl, int64_t planes, int64_t h, int64_t w, void* stream) {             \
        SONAR_REQUIRE(lh && hh && hl && bands && planes >= 0 && h > 0 && w > 0 && h < (1 << 15) && w < (1 << 15), SONAR_ERR_ARG,                 \
                      "sonar_dtcwt_c2q_" #NAME ": bad argument");                                                                                \
        if (planes == 0) return SONAR_OK;                                                                                                        \
        hipLaunchKernelGGL((dtcwt_c2q_kernel<T>), dim3(grid_for(planes * 3 * h * w, kBlock * 2)), dim3(kBlock), 0, (hipStream_t)stream, bands,   \
                           lh, hh, hl, planes, (int)h, (int)w);                                                                                  \
        return check_launch("sonar_dtcwt_c2q_" #NAME);                                                                                           \
    }
SONAR_Q2C(f32, float)
SONAR_Q2C(f64, double)
#undef SONAR_Q2C
