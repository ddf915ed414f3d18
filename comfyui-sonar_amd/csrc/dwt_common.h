// Shared by dwt.hip (one pass per launch) and dwt_tile.hip (LDS-staged fused passes): PyWavelets index rules, taps,
// the synthesis sum and the WaveletCFG band arithmetic.
#pragma once
#include "common.h"

namespace sonar {

constexpr int kMaxTaps = 64;
constexpr int kMaxLevels = 12;
enum DwtMode { kZero = 0, kSymmetric = 1, kReflect = 2, kPeriodization = 3, kPeriodic = 4, kConstant = 5 };

template <typename T>
struct Taps {
    T lo[kMaxTaps];
    T hi[kMaxTaps];
    int len;
};

__host__ __device__ inline int64_t dwt_len(int64_t n, int64_t flen, int mode) {
    return mode == kPeriodization ? (n + 1) / 2 : (n + flen - 1) / 2;
}

// extended-signal index -> source index (or -1 for an implicit zero)
__device__ __forceinline__ int ext_index(int idx, int n, int mode) {
    if (idx >= 0 && idx < n) return idx;
    switch (mode) {
        case kZero: return -1;
        case kConstant: return idx < 0 ? 0 : n - 1;
        case kPeriodic: {
            int r = idx % n;
            return r < 0 ? r + n : r;
        }
        case kSymmetric: {
            const int period = 2 * n;
            int p = idx % period;
            if (p < 0) p += period;
            return p < n ? p : period - 1 - p;
        }
        case kReflect: {
            if (n == 1) return 0;
            const int period = 2 * n - 2;
            int p = idx % period;
            if (p < 0) p += period;
            return p < n ? p : period - p;
        }
        default: return -1;
    }
}

// one synthesis output from two coefficient sequences a (stride sa) and d (stride sd)
template <typename T>
__device__ __forceinline__ T synth(const T* __restrict__ a, int64_t sa, const T* __restrict__ d, int64_t sd, int n, int o,
                                   const Taps<T>& tp, int mode) {
    const int F = tp.len;
    T acc = T(0);
    if (mode == kPeriodization) {
        const int N = 2 * n;
        for (int i = 0; i < n; ++i) {
            int j = (o + F / 2 - 1 - 2 * i) % N;
            if (j < 0) j += N;
            for (; j < F; j += N) acc += a[(int64_t)i * sa] * tp.lo[j] + d[(int64_t)i * sd] * tp.hi[j];
        }
    } else {
        // 2i + j = o + F - 2 with 0 <= j < F  ->  i in [ceil((o - 1) / 2), floor((o + F - 2) / 2)]
        const int t = o + F - 2;
        int i0 = o > 0 ? (o >> 1) : 0;   // ceil((o - 1) / 2) for o >= 0
        int i1 = t >> 1;
        if (i1 > n - 1) i1 = n - 1;
        for (int i = i0; i <= i1; ++i) {
            const int j = t - 2 * i;
            acc += a[(int64_t)i * sa] * tp.lo[j] + d[(int64_t)i * sd] * tp.hi[j];
        }
    }
    return acc;
}

// ---- WaveletCFG band arithmetic
constexpr int kMaxBandGroups = 8;  // orientations per band: 3 (DWT), 6 (dual-tree complex transform)
template <typename T>
struct BandScales {
    T cond[kMaxBandGroups], uncond[kMaxBandGroups], diff[kMaxBandGroups], fin[kMaxBandGroups];
};

// blend(u * s_u, (c * s_c - u * s_u) * s_d, strength) * s_f  (py/wavelet_cfg.py:765-787), scales of 1 are skipped like the reference's `!= 1.0` tests
template <typename T>
__device__ __forceinline__ T band_combine(T c, T u, const BandScales<T>& sc, int g, int blend_mode, T strength) {
    if (sc.cond[g] != T(1)) c = c * sc.cond[g];
    if (sc.uncond[g] != T(1)) u = u * sc.uncond[g];
    T d = c - u;
    if (sc.diff[g] != T(1)) d = d * sc.diff[g];
    T r = blend<T>(blend_mode, u, d, strength);
    if (sc.fin[g] != T(1)) r = r * sc.fin[g];
    return r;
}

template <typename T>
static bool make_taps(Taps<T>& tp, const double* lo, const double* hi, int flen) {
    if (!lo || !hi || flen < 1 || flen > kMaxTaps) return false;
    tp.len = flen;
    for (int j = 0; j < kMaxTaps; ++j) {
        tp.lo[j] = j < flen ? (T)lo[j] : T(0);
        tp.hi[j] = j < flen ? (T)hi[j] : T(0);
    }
    return true;
}

static bool dims_ok(int64_t a, int64_t b) { return a > 0 && b > 0 && a < (1 << 24) && b < (1 << 24); }

}  // namespace sonar
