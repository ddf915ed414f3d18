// 2-D DWT / IDWT (one level per call) and the WaveletCFG band arithmetic, fp32 and fp64.
// Semantics: PyWavelets dwt/idwt (== pytorch_wavelets DWTForward/DWTInverse) applied separably:
//   analysis   out[i] = sum_j f[j] * ext(x)[2i + 1 - j]                      (non-periodization)
//              out[i] = sum_j f[j] * xe[(2i - j + F/2) mod Ne]               (periodization, xe = x padded to even)
//   synthesis  x[o]   = sum_{2i + j = o + F - 2} a[i] lo[j] + d[i] hi[j]      (length 2n - F + 2)
//              x[o]   = sum_{2i + j == o + F/2 - 1 (mod 2n)} ...              (periodization, length 2n)
// Row pass and column pass are separate launches through a caller-provided intermediate; both are
// coalesced along W.  Taps travel as kernel arguments (wave-uniform index -> scalar loads).
#include "dwt_common.h"
#include "dwt_tile.h"
#include "dwt_lowpass.h"

namespace sonar {

// ---- analysis along W: x[rows][W] -> tmp[rows][2][w]
template <typename T>
__global__ void __launch_bounds__(kBlock) dwt_rows_kernel(const T* __restrict__ x, T* __restrict__ tmp, int64_t rows, int W,
                                                           int w, Taps<T> tp, int mode) {
    kernarg_touch_for(x, tmp, rows, W, w, tp, mode);
    const int64_t total = rows * w;
    const int F = tp.len;
    const int We = (mode == kPeriodization && (W & 1)) ? W + 1 : W;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int xo = (int)(i % w);
        const int64_t r = i / w;
        const T* row = x + r * W;
        T a = T(0), d = T(0);
        for (int j = 0; j < F; ++j) {
            int src;
            if (mode == kPeriodization) {
                int p = (2 * xo - j + F / 2) % We;
                if (p < 0) p += We;
                src = p < W ? p : W - 1;  // the padded sample repeats the last one
            } else {
                src = ext_index(2 * xo + 1 - j, W, mode);
            }
            if (src >= 0) {
                const T v = row[src];
                a += tp.lo[j] * v;
                d += tp.hi[j] * v;
            }
        }
        tmp[(r * 2 + 0) * w + xo] = a;
        tmp[(r * 2 + 1) * w + xo] = d;
    }
}

// ---- analysis along H: tmp[planes][H][2][w] -> ll[planes][h][w], hi[planes][3][h][w]
template <typename T>
__global__ void __launch_bounds__(kBlock) dwt_cols_kernel(const T* __restrict__ tmp, T* __restrict__ ll, T* __restrict__ hi,
                                                           int64_t planes, int H, int h, int w, Taps<T> tp, int mode) {
    kernarg_touch_for(tmp, ll, hi, planes, H, h, w, tp, mode);
    const int64_t total = planes * h * w;
    const int F = tp.len;
    const int He = (mode == kPeriodization && (H & 1)) ? H + 1 : H;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int xo = (int)(i % w);
        const int yo = (int)((i / w) % h);
        const int64_t p = i / ((int64_t)w * h);
        const T* base = tmp + p * (int64_t)H * 2 * w + xo;
        T a_lo = T(0), d_lo = T(0), a_hi = T(0), d_hi = T(0);  // {a,d} along H of the {lo,hi}-along-W signals
        for (int j = 0; j < F; ++j) {
            int src;
            if (mode == kPeriodization) {
                int q = (2 * yo - j + F / 2) % He;
                if (q < 0) q += He;
                src = q < H ? q : H - 1;
            } else {
                src = ext_index(2 * yo + 1 - j, H, mode);
            }
            if (src >= 0) {
                const T vl = base[(int64_t)src * 2 * w];
                const T vh = base[(int64_t)src * 2 * w + w];
                a_lo += tp.lo[j] * vl;
                d_lo += tp.hi[j] * vl;
                a_hi += tp.lo[j] * vh;
                d_hi += tp.hi[j] * vh;
            }
        }
        const int64_t o = ((int64_t)yo) * w + xo;
        const int64_t hw = (int64_t)h * w;
        ll[p * hw + o] = a_lo;
        hi[(p * 3 + 0) * hw + o] = d_lo;  // cH: high along H, low along W
        hi[(p * 3 + 1) * hw + o] = a_hi;  // cV: low along H, high along W
        hi[(p * 3 + 2) * hw + o] = d_hi;  // cD
    }
}

// ---- synthesis along H: (ll, cH) -> lo_w ; (cV, cD) -> hi_w ; tmp[planes][2][Hr][w]
template <typename T>
__global__ void __launch_bounds__(kBlock) idwt_cols_kernel(const T* __restrict__ ll, int ll_h, int ll_w,
                                                            const T* __restrict__ hi, T* __restrict__ tmp, int64_t planes, int h,
                                                            int w, int Hr, Taps<T> tp, int mode) {
    kernarg_touch_for(ll, ll_h, ll_w, hi, tmp, planes, h, w, Hr, tp, mode);
    const int64_t total = planes * Hr * w;
    const int64_t hw = (int64_t)h * w;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int xo = (int)(i % w);
        const int yo = (int)((i / w) % Hr);
        const int64_t p = i / ((int64_t)w * Hr);
        const T* pll = ll + p * (int64_t)ll_h * ll_w + xo;
        const T* pch = hi + (p * 3 + 0) * hw + xo;
        const T* pcv = hi + (p * 3 + 1) * hw + xo;
        const T* pcd = hi + (p * 3 + 2) * hw + xo;
        tmp[((p * 2 + 0) * Hr + yo) * w + xo] = synth<T>(pll, ll_w, pch, w, h, yo, tp, mode);
        tmp[((p * 2 + 1) * Hr + yo) * w + xo] = synth<T>(pcv, w, pcd, w, h, yo, tp, mode);
    }
}

// ---- synthesis along W: tmp[planes][2][Hr][w] -> out[planes][Ho][Wo] (cropped)
template <typename T>
__global__ void __launch_bounds__(kBlock) idwt_rows_kernel(const T* __restrict__ tmp, T* __restrict__ out, int64_t planes, int w,
                                                            int Hr, int Ho, int Wo, Taps<T> tp, int mode) {
    kernarg_touch_for(tmp, out, planes, w, Hr, Ho, Wo, tp, mode);
    const int64_t total = planes * Ho * Wo;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int xo = (int)(i % Wo);
        const int yo = (int)((i / Wo) % Ho);
        const int64_t p = i / ((int64_t)Wo * Ho);
        const T* lo_w = tmp + ((p * 2 + 0) * Hr + yo) * w;
        const T* hi_w = tmp + ((p * 2 + 1) * Hr + yo) * w;
        out[i] = synth<T>(lo_w, 1, hi_w, 1, w, xo, tp, mode);
    }
}

// HEAD: the reference's band scaling on 1-D bands [B, C, l] (py/wavelet_functions.py:212-215 indexes axis 2, the coefficient axis
// there): group 0 = the first coefficient of every row of group_size elements, group 1 (unit scales) = the rest.
template <typename T, bool HEAD>
__global__ void __launch_bounds__(kBlock) wcfg_band_kernel(const T* cond, const T* uncond, T* out /* may alias cond / uncond */, int64_t n, int64_t group_size, int groups,
                                                            BandScales<T> sc, int blend_mode, T strength) {
    kernarg_touch_for(cond, uncond, out, n, group_size, groups, sc, blend_mode, strength);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int g = HEAD ? (int)(i % group_size != 0) : groups > 1 ? (int)((i / group_size) % groups) : 0;
        out[i] = band_combine<T>(cond[i], uncond[i], sc, g, blend_mode, strength);
    }
}

template <typename T>
__global__ void __launch_bounds__(kBlock) wcfg_output_kernel(const float* __restrict__ x, const T* __restrict__ res,
                                                              float* __restrict__ out, int64_t planes, int H, int W, int Hr,
                                                              int Wr, int subtract) {
    kernarg_touch_for(x, res, out, planes, H, W, Hr, Wr, subtract);
    const int64_t total = planes * H * W;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int xo = (int)(i % W);
        const int yo = (int)((i / W) % H);
        const int64_t p = i / ((int64_t)W * H);
        const float r = (float)res[(p * Hr + yo) * (int64_t)Wr + xo];
        out[i] = subtract ? x[i] - r : r;
    }
}

// ---- 1-D transform of flattened latents (py/wavelet_functions.py:56-57 use_1d_dwt; py/wavelet_cfg.py:713-715 flattens to
// [B, C, H*W]): analysis x[rows][L] -> lo[rows][n], hi[rows][n]; a thread owns one coefficient pair.  Away from the borders
// (the common case: L is thousands of samples) the taps are read straight, without the extension arithmetic.
template <typename T>
__global__ void __launch_bounds__(kBlock) dwt1_fwd_kernel(const T* __restrict__ x, T* __restrict__ lo, T* __restrict__ hi,
                                                           int64_t rows, int L, int n, Taps<T> tp, int mode) {
    kernarg_touch_for(x, lo, hi, rows, L, n, tp, mode);
    const int64_t total = rows * n;
    const int F = tp.len;
    const int Le = (mode == kPeriodization && (L & 1)) ? L + 1 : L;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int xo = (int)(i % n);
        const T* row = x + (i / n) * L;
        const int top = mode == kPeriodization ? 2 * xo + F / 2 : 2 * xo + 1;  // index read by tap 0; tap j reads top - j
        T a = T(0), d = T(0);
        if (top < L && top - (F - 1) >= 0) {
            for (int j = 0; j < F; ++j) {
                const T v = row[top - j];
                a += tp.lo[j] * v;
                d += tp.hi[j] * v;
            }
        } else {
            for (int j = 0; j < F; ++j) {
                int src;
                if (mode == kPeriodization) {
                    int p = (top - j) % Le;
                    if (p < 0) p += Le;
                    src = p < L ? p : L - 1;  // the padded sample repeats the last one
                } else {
                    src = ext_index(top - j, L, mode);
                }
                if (src >= 0) {
                    const T v = row[src];
                    a += tp.lo[j] * v;
                    d += tp.hi[j] * v;
                }
            }
        }
        lo[i] = a;
        hi[i] = d;
    }
}

// synthesis lo[rows][lo_len] (leading n used), hi[rows][n] -> out[rows][Lo] (cropped); taps walked by parity, so a
// periodization output costs F / 2 terms like the others (synth() above scans all n coefficients in that mode)
template <typename T>
__global__ void __launch_bounds__(kBlock) dwt1_inv_kernel(const T* __restrict__ lo, int lo_len, const T* __restrict__ hi,
                                                           T* __restrict__ out, int64_t rows, int n, int Lo, Taps<T> tp, int mode) {
    kernarg_touch_for(lo, lo_len, hi, out, rows, n, Lo, tp, mode);
    const int64_t total = rows * Lo;
    const int F = tp.len;
    for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kBlock) {
        const int o = (int)(idx % Lo);
        const int64_t r = idx / Lo;
        const T* a = lo + r * lo_len;
        const T* d = hi + r * n;
        const int t = mode == kPeriodization ? o + F / 2 - 1 : o + F - 2;  // 2 i + j == t (mod 2n when periodized)
        T acc = T(0);
        for (int j = t & 1; j < F; j += 2) {
            int i = (t - j) >> 1;  // t - j is even; arithmetic shift = floor
            if (mode == kPeriodization) {
                i %= n;
                if (i < 0) i += n;
            } else if (i < 0 || i >= n) {
                continue;
            }
            acc += a[i] * tp.lo[j] + d[i] * tp.hi[j];
        }
        out[idx] = acc;
    }
}

template <typename T>
static int dwt2_fwd(const T* x, T* ll, T* hi, int64_t planes, int64_t H, int64_t W, const double* dec_lo, const double* dec_hi,
                    int flen, int mode, void* ws, hipStream_t st, const char* what) {
    SONAR_REQUIRE(x && ll && hi && ws && planes >= 0 && mode >= 0 && mode <= 5, SONAR_ERR_ARG, "%s: bad argument", what);
    SONAR_REQUIRE(dims_ok(H, W), SONAR_ERR_UNSUPPORTED, "%s: bad plane size", what);
    Taps<T> tp;
    SONAR_REQUIRE(make_taps(tp, dec_lo, dec_hi, flen), SONAR_ERR_ARG, "%s: 1..%d filter taps required", what, kMaxTaps);
    if (planes == 0) return SONAR_OK;
    const int h = (int)dwt_len(H, flen, mode), w = (int)dwt_len(W, flen, mode);
    if (dwt2_fwd_tiled<T>(x, ll, hi, planes, (int)H, (int)W, h, w, tp, mode, st)) return check_launch(what);  // LDS-staged, one launch
    T* tmp = (T*)ws;
    hipLaunchKernelGGL((dwt_rows_kernel<T>), dim3(grid_for(planes * H * w, kBlock)), dim3(kBlock), 0, st, x, tmp, planes * H, (int)W, w,
                       tp, mode);
    hipLaunchKernelGGL((dwt_cols_kernel<T>), dim3(grid_for(planes * h * w, kBlock)), dim3(kBlock), 0, st, tmp, ll, hi, planes, (int)H, h,
                       w, tp, mode);
    return check_launch(what);
}

template <typename T>
static int dwt2_inv(const T* ll, int64_t ll_h, int64_t ll_w, const T* hi, T* out, int64_t planes, int64_t h, int64_t w, int64_t Ho,
                    int64_t Wo, const double* rec_lo, const double* rec_hi, int flen, int mode, void* ws, hipStream_t st,
                    const char* what) {
    SONAR_REQUIRE(ll && hi && out && ws && planes >= 0 && mode >= 0 && mode <= 5, SONAR_ERR_ARG, "%s: bad argument", what);
    SONAR_REQUIRE(dims_ok(h, w) && ll_h >= h && ll_w >= w, SONAR_ERR_ARG, "%s: bad coefficient size", what);
    Taps<T> tp;
    SONAR_REQUIRE(make_taps(tp, rec_lo, rec_hi, flen), SONAR_ERR_ARG, "%s: 1..%d filter taps required", what, kMaxTaps);
    const int64_t Hr = mode == kPeriodization ? 2 * h : 2 * h - flen + 2;
    const int64_t Wr = mode == kPeriodization ? 2 * w : 2 * w - flen + 2;
    SONAR_REQUIRE(Hr > 0 && Wr > 0 && Ho > 0 && Wo > 0 && Ho <= Hr && Wo <= Wr, SONAR_ERR_ARG,
                  "%s: requested output %lldx%lld exceeds the reconstruction %lldx%lld", what, (long long)Ho, (long long)Wo,
                  (long long)Hr, (long long)Wr);
    if (planes == 0) return SONAR_OK;
    if (dwt2_inv_tiled<T>(ll, (int)ll_h, (int)ll_w, hi, out, planes, (int)h, (int)w, (int)Ho, (int)Wo, tp, mode, st)) return check_launch(what);
    T* tmp = (T*)ws;
    hipLaunchKernelGGL((idwt_cols_kernel<T>), dim3(grid_for(planes * Hr * w, kBlock)), dim3(kBlock), 0, st, ll, (int)ll_h, (int)ll_w, hi,
                       tmp, planes, (int)h, (int)w, (int)Hr, tp, mode);
    hipLaunchKernelGGL((idwt_rows_kernel<T>), dim3(grid_for(planes * Ho * Wo, kBlock)), dim3(kBlock), 0, st, tmp, out, planes, (int)w,
                       (int)Hr, (int)Ho, (int)Wo, tp, mode);
    return check_launch(what);
}

template <typename T>
static int dwt1_fwd(const T* x, T* lo, T* hi, int64_t rows, int64_t L, const double* dec_lo, const double* dec_hi, int flen, int mode,
                    hipStream_t st, const char* what) {
    SONAR_REQUIRE(x && lo && hi && rows >= 0 && mode >= 0 && mode <= 5, SONAR_ERR_ARG, "%s: bad argument", what);
    SONAR_REQUIRE(L > 0 && L < (1 << 30), SONAR_ERR_UNSUPPORTED, "%s: bad signal length", what);
    Taps<T> tp;
    SONAR_REQUIRE(make_taps(tp, dec_lo, dec_hi, flen), SONAR_ERR_ARG, "%s: 1..%d filter taps required", what, kMaxTaps);
    if (rows == 0) return SONAR_OK;
    const int n = (int)dwt_len(L, flen, mode);
    hipLaunchKernelGGL((dwt1_fwd_kernel<T>), dim3(grid_for(rows * n, kBlock)), dim3(kBlock), 0, st, x, lo, hi, rows, (int)L, n, tp, mode);
    return check_launch(what);
}

template <typename T>
static int dwt1_inv(const T* lo, int64_t lo_len, const T* hi, T* out, int64_t rows, int64_t n, int64_t Lo, const double* rec_lo,
                    const double* rec_hi, int flen, int mode, hipStream_t st, const char* what) {
    SONAR_REQUIRE(lo && hi && out && rows >= 0 && mode >= 0 && mode <= 5, SONAR_ERR_ARG, "%s: bad argument", what);
    SONAR_REQUIRE(n > 0 && n < (1 << 29) && lo_len >= n, SONAR_ERR_ARG, "%s: bad coefficient length", what);
    Taps<T> tp;
    SONAR_REQUIRE(make_taps(tp, rec_lo, rec_hi, flen), SONAR_ERR_ARG, "%s: 1..%d filter taps required", what, kMaxTaps);
    const int64_t Lr = mode == kPeriodization ? 2 * n : 2 * n - flen + 2;
    SONAR_REQUIRE(Lr > 0 && Lo > 0 && Lo <= Lr, SONAR_ERR_ARG, "%s: requested output %lld exceeds the reconstruction %lld", what,
                  (long long)Lo, (long long)Lr);
    if (rows == 0) return SONAR_OK;
    hipLaunchKernelGGL((dwt1_inv_kernel<T>), dim3(grid_for(rows * Lo, kBlock)), dim3(kBlock), 0, st, lo, (int)lo_len, hi, out, rows, (int)n,
                       (int)Lo, tp, mode);
    return check_launch(what);
}

template <typename T, bool HEAD = false>
static int wcfg_band(const T* cond, const T* uncond, T* out, int64_t n, int64_t group_size, int64_t groups, const double* s_cond,
                     const double* s_uncond, const double* s_diff, const double* s_final, int blend_mode, double strength,
                     hipStream_t st, const char* what) {
    SONAR_REQUIRE(cond && uncond && out && n >= 0 && groups >= 1 && groups <= kMaxBandGroups && group_size > 0 && blend_mode >= 0 && blend_mode <= 2,
                  SONAR_ERR_ARG, "%s: bad argument", what);
    BandScales<T> sc;
    for (int g = 0; g < kMaxBandGroups; ++g) {
        const bool in = g < (HEAD ? 1 : groups);
        sc.cond[g] = in && s_cond ? (T)s_cond[g] : T(1);
        sc.uncond[g] = in && s_uncond ? (T)s_uncond[g] : T(1);
        sc.diff[g] = in && s_diff ? (T)s_diff[g] : T(1);
        sc.fin[g] = in && s_final ? (T)s_final[g] : T(1);
    }
    if (n == 0) return SONAR_OK;
    hipLaunchKernelGGL((wcfg_band_kernel<T, HEAD>), dim3(grid_for(n, kBlock * 2)), dim3(kBlock), 0, st, cond, uncond, out, n, group_size,
                       (int)groups, sc, blend_mode, (T)strength);
    return check_launch(what);
}

}  // namespace sonar

using namespace sonar;

extern "C" int64_t sonar_dwt_out_len(int64_t n, int64_t flen, int mode) {
    if (n <= 0 || flen <= 0 || mode < 0 || mode > 5) return -1;
    return dwt_len(n, flen, mode);
}

extern "C" int64_t sonar_dwt2_ws_bytes(int64_t planes, int64_t H, int64_t W, int flen, int mode, int elem_size, int inverse) {
    if (planes < 0 || H <= 0 || W <= 0 || flen <= 0 || mode < 0 || mode > 5 || (elem_size != 4 && elem_size != 8)) return -1;
    if (!inverse) return planes * H * 2 * dwt_len(W, flen, mode) * elem_size;  // H, W = input plane
    const int64_t Hr = mode == kPeriodization ? 2 * H : 2 * H - flen + 2;     // H, W = coefficient plane (h, w)
    return planes * 2 * (Hr > 0 ? Hr : 0) * W * elem_size;
}

extern "C" int sonar_dwt2_fwd_f32(const float* x, float* ll, float* hi, int64_t planes, int64_t H, int64_t W, const double* dec_lo,
                                  const double* dec_hi, int flen, int mode, void* ws, void* stream) {
    return dwt2_fwd<float>(x, ll, hi, planes, H, W, dec_lo, dec_hi, flen, mode, ws, (hipStream_t)stream, "sonar_dwt2_fwd_f32");
}
extern "C" int sonar_dwt2_fwd_f64(const double* x, double* ll, double* hi, int64_t planes, int64_t H, int64_t W, const double* dec_lo,
                                  const double* dec_hi, int flen, int mode, void* ws, void* stream) {
    return dwt2_fwd<double>(x, ll, hi, planes, H, W, dec_lo, dec_hi, flen, mode, ws, (hipStream_t)stream, "sonar_dwt2_fwd_f64");
}
extern "C" int sonar_dwt2_inv_f32(const float* ll, int64_t ll_h, int64_t ll_w, const float* hi, float* out, int64_t planes, int64_t h,
                                  int64_t w, int64_t Ho, int64_t Wo, const double* rec_lo, const double* rec_hi, int flen, int mode,
                                  void* ws, void* stream) {
    return dwt2_inv<float>(ll, ll_h, ll_w, hi, out, planes, h, w, Ho, Wo, rec_lo, rec_hi, flen, mode, ws, (hipStream_t)stream,
                           "sonar_dwt2_inv_f32");
}
extern "C" int sonar_dwt2_inv_f64(const double* ll, int64_t ll_h, int64_t ll_w, const double* hi, double* out, int64_t planes,
                                  int64_t h, int64_t w, int64_t Ho, int64_t Wo, const double* rec_lo, const double* rec_hi, int flen,
                                  int mode, void* ws, void* stream) {
    return dwt2_inv<double>(ll, ll_h, ll_w, hi, out, planes, h, w, Ho, Wo, rec_lo, rec_hi, flen, mode, ws, (hipStream_t)stream,
                            "sonar_dwt2_inv_f64");
}
extern "C" int sonar_dwt1_fwd_f32(const float* x, float* lo, float* hi, int64_t rows, int64_t L, const double* dec_lo,
                                  const double* dec_hi, int flen, int mode, void* stream) {
    return dwt1_fwd<float>(x, lo, hi, rows, L, dec_lo, dec_hi, flen, mode, (hipStream_t)stream, "sonar_dwt1_fwd_f32");
}
extern "C" int sonar_dwt1_fwd_f64(const double* x, double* lo, double* hi, int64_t rows, int64_t L, const double* dec_lo,
                                  const double* dec_hi, int flen, int mode, void* stream) {
    return dwt1_fwd<double>(x, lo, hi, rows, L, dec_lo, dec_hi, flen, mode, (hipStream_t)stream, "sonar_dwt1_fwd_f64");
}
extern "C" int sonar_dwt1_inv_f32(const float* lo, int64_t lo_len, const float* hi, float* out, int64_t rows, int64_t n, int64_t Lo,
                                  const double* rec_lo, const double* rec_hi, int flen, int mode, void* stream) {
    return dwt1_inv<float>(lo, lo_len, hi, out, rows, n, Lo, rec_lo, rec_hi, flen, mode, (hipStream_t)stream, "sonar_dwt1_inv_f32");
}
extern "C" int sonar_dwt1_inv_f64(const double* lo, int64_t lo_len, const double* hi, double* out, int64_t rows, int64_t n, int64_t Lo,
                                  const double* rec_lo, const double* rec_hi, int flen, int mode, void* stream) {
    return dwt1_inv<double>(lo, lo_len, hi, out, rows, n, Lo, rec_lo, rec_hi, flen, mode, (hipStream_t)stream, "sonar_dwt1_inv_f64");
}
extern "C" int sonar_wcfg_band_f32(const float* cond, const float* uncond, float* out, int64_t n, int64_t group_size, int64_t groups,
                                   const double* s_cond, const double* s_uncond, const double* s_diff, const double* s_final,
                                   int blend_mode, double strength, void* stream) {
    return wcfg_band<float>(cond, uncond, out, n, group_size, groups, s_cond, s_uncond, s_diff, s_final, blend_mode, strength,
                            (hipStream_t)stream, "sonar_wcfg_band_f32");
}
extern "C" int sonar_wcfg_band_f64(const double* cond, const double* uncond, double* out, int64_t n, int64_t group_size,
                                   int64_t groups, const double* s_cond, const double* s_uncond, const double* s_diff,
                                   const double* s_final, int blend_mode, double strength, void* stream) {
    return wcfg_band<double>(cond, uncond, out, n, group_size, groups, s_cond, s_uncond, s_diff, s_final, blend_mode, strength,
                             (hipStream_t)stream, "sonar_wcfg_band_f64");
}
extern "C" int sonar_wcfg_band_head_f32(const float* cond, const float* uncond, float* out, int64_t n, int64_t row_len, const double* s_cond,
                                        const double* s_uncond, const double* s_diff, const double* s_final, int blend_mode,
                                        double strength, void* stream) {
    return wcfg_band<float, true>(cond, uncond, out, n, row_len, 2, s_cond, s_uncond, s_diff, s_final, blend_mode, strength,
                                  (hipStream_t)stream, "sonar_wcfg_band_head_f32");
}
extern "C" int sonar_wcfg_band_head_f64(const double* cond, const double* uncond, double* out, int64_t n, int64_t row_len,
                                        const double* s_cond, const double* s_uncond, const double* s_diff, const double* s_final,
                                        int blend_mode, double strength, void* stream) {
    return wcfg_band<double, true>(cond, uncond, out, n, row_len, 2, s_cond, s_uncond, s_diff, s_final, blend_mode, strength,
                                   (hipStream_t)stream, "sonar_wcfg_band_head_f64");
}
extern "C" int64_t sonar_wcfg_fused_ws_bytes(int64_t planes, int64_t H, int64_t W, int levels, int dec_len, int mode_fwd, int rec_len,
                                             int mode_inv, int elem_size) {
    WcfgPlan pl;
    if (planes < 0 || !dims_ok(H, W) || dec_len < 1 || rec_len < 1 || mode_fwd < 0 || mode_fwd > 5 || mode_inv < 0 || mode_inv > 5 ||
        (elem_size != 4 && elem_size != 8) || !wcfg_plan(pl, planes, H, W, levels, dec_len, mode_fwd, rec_len, mode_inv))
        return -1;
    return pl.total * elem_size;
}
extern "C" int sonar_wcfg_fused_f32(const float* cond, const float* uncond, const float* x, float* out, int64_t planes, int64_t H,
                                    int64_t W, int levels, const double* dec_lo, const double* dec_hi, int dec_len, int mode_fwd,
                                    const double* rec_lo, const double* rec_hi, int rec_len, int mode_inv, const double* yl_scales,
                                    const double* yh_scales, int blend_mode, double strength, int subtract_from_x,
                                    int perfect_reconstruction, void* ws, int64_t ws_bytes, void* stream) {
    return wcfg_fused<float>(cond, uncond, x, out, planes, H, W, levels, dec_lo, dec_hi, dec_len, mode_fwd, rec_lo, rec_hi, rec_len,
                             mode_inv, yl_scales, yh_scales, blend_mode, strength, subtract_from_x, ws, ws_bytes, (hipStream_t)stream,
                             "sonar_wcfg_fused_f32", perfect_reconstruction != 0);
}
extern "C" int sonar_wcfg_hi_storage(int fp32) {
    const int before = wcfg_hi_fp32_switch();
    if (fp32 >= 0) wcfg_hi_fp32_switch() = fp32 != 0;
    return before;
}

extern "C" int sonar_wcfg_fused_f64(const float* cond, const float* uncond, const float* x, float* out, int64_t planes, int64_t H,
                                    int64_t W, int levels, const double* dec_lo, const double* dec_hi, int dec_len, int mode_fwd,
                                    const double* rec_lo, const double* rec_hi, int rec_len, int mode_inv, const double* yl_scales,
                                    const double* yh_scales, int blend_mode, double strength, int subtract_from_x,
                                    int perfect_reconstruction, void* ws, int64_t ws_bytes, void* stream) {
    return wcfg_fused<double>(cond, uncond, x, out, planes, H, W, levels, dec_lo, dec_hi, dec_len, mode_fwd, rec_lo, rec_hi, rec_len,
                              mode_inv, yl_scales, yh_scales, blend_mode, strength, subtract_from_x, ws, ws_bytes, (hipStream_t)stream,
                              "sonar_wcfg_fused_f64", perfect_reconstruction != 0);
}
extern "C" int64_t sonar_wcfg_lowpass_lds_bytes(int64_t H, int64_t W, int levels, int flen, int mode_fwd, int mode_inv, int elem_size) {
    size_t lds = 0;
    if (mode_fwd < 0 || mode_fwd > 5 || mode_inv < 0 || mode_inv > 5) return -1;
    if (elem_size == 8) {
        LowArgs<double> a{};
        return lowpass_plan(a, lds, H, W, levels, flen, mode_fwd, mode_inv) ? (int64_t)lds : -1;
    }
    if (elem_size == 4) {
        LowArgs<float> a{};
        return lowpass_plan(a, lds, H, W, levels, flen, mode_fwd, mode_inv) ? (int64_t)lds : -1;
    }
    return -1;
}
extern "C" int sonar_wcfg_lowpass_f32(const float* cond, const float* uncond, const float* x, float* out, int64_t planes, int64_t H, int64_t W,
                                      int levels, const double* dec_lo, const double* rec_lo, int flen, int mode_fwd, int mode_inv,
                                      const double* g, double ku, double kt, int subtract_from_x, void* stream) {
    return wcfg_lowpass<float>(cond, uncond, x, out, planes, H, W, levels, dec_lo, rec_lo, flen, mode_fwd, mode_inv, g, ku, kt, subtract_from_x,
                               (hipStream_t)stream, "sonar_wcfg_lowpass_f32");
}
extern "C" int sonar_wcfg_lowpass_f64(const float* cond, const float* uncond, const float* x, float* out, int64_t planes, int64_t H, int64_t W,
                                      int levels, const double* dec_lo, const double* rec_lo, int flen, int mode_fwd, int mode_inv,
                                      const double* g, double ku, double kt, int subtract_from_x, void* stream) {
    return wcfg_lowpass<double>(cond, uncond, x, out, planes, H, W, levels, dec_lo, rec_lo, flen, mode_fwd, mode_inv, g, ku, kt, subtract_from_x,
                                (hipStream_t)stream, "sonar_wcfg_lowpass_f64");
}
extern "C" int sonar_wcfg_output_f32(const float* x, const void* result, int result_is_f64, float* out, int64_t planes, int64_t H,
                                     int64_t W, int64_t Hr, int64_t Wr, int subtract_from_x, void* stream) {
    SONAR_REQUIRE(result && out && (x || !subtract_from_x) && planes >= 0 && dims_ok(H, W) && Hr >= H && Wr >= W, SONAR_ERR_ARG,
                  "sonar_wcfg_output_f32: bad argument");
    if (planes == 0) return SONAR_OK;
    const int g = grid_for(planes * H * W, kBlock * 2);
    if (result_is_f64)
        hipLaunchKernelGGL((wcfg_output_kernel<double>), dim3(g), dim3(kBlock), 0, (hipStream_t)stream, x, (const double*)result, out,
                           planes, (int)H, (int)W, (int)Hr, (int)Wr, subtract_from_x);
    else
        hipLaunchKernelGGL((wcfg_output_kernel<float>), dim3(g), dim3(kBlock), 0, (hipStream_t)stream, x, (const float*)result, out,
                           planes, (int)H, (int)W, (int)Hr, (int)Wr, subtract_from_x);
    return check_launch("sonar_wcfg_output_f32");
}
