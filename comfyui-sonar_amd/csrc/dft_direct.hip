// Direct (O(N^2) per line) real 2-D DFT building blocks for planes the LDS-resident FFT kernels do not take: odd heights or widths
// (1080-line video: 135-row latents), planes too large for LDS.  torch.fft.rfft2 / irfft2 semantics (py/nodes/powernoise.py:338-408,
// py/noise_generation.py:680-759, py/nodes/freeu_extreme.py:10-29), three passes through a caller-owned complex workspace:
//   rows r2c (real [rows][W] -> complex [rows][W/2+1]), columns (complex DFT along H, optionally x a real filter on the way in),
//   rows c2r (complex [rows][W/2+1] -> real [rows][W], the imaginary parts of the DC / Nyquist columns ignored as irfft does).
// Twiddles come from a per-workgroup LDS table e^{2 pi i j / N} built in fp64; the index j = (k n) mod N is kept incrementally.
// A fallback: ~N / log N times the arithmetic of the FFT kernels, bandwidth from L2.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "common.h"

namespace sonar {

constexpr int kDirectMax = 2048;  // longest line: table + staged row stay under 32 KB of LDS

__device__ __forceinline__ void build_table(float2* tw, int N, int sign) {
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
        double s, c;
        sincospi(2.0 * (double)j / (double)N, &s, &c);
        tw[j] = make_float2((float)c, (float)(sign * s));
    }
}

// y[row][k] = sum_x x[row][x] e^{-2 pi i k x / W}, k = 0 .. W/2
__global__ void __launch_bounds__(kBlock) dft_rows_r2c_kernel(const float* __restrict__ x, float2* __restrict__ y, int64_t rows, int W) {
    kernarg_touch_for(x, y, rows, W);
    extern __shared__ __align__(16) unsigned char dft_lds[];
    float2* tw = reinterpret_cast<float2*>(dft_lds);
    float* line = reinterpret_cast<float*>(tw + W);
    const int Wh = W / 2 + 1;
    build_table(tw, W, -1);
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        __syncthreads();
        for (int i = threadIdx.x; i < W; i += kBlock) line[i] = x[r * W + i];
        __syncthreads();
        for (int k = threadIdx.x; k < Wh; k += kBlock) {
            float re = 0.0f, im = 0.0f;
            int idx = 0;
            for (int i = 0; i < W; ++i) {
                const float2 t = tw[idx];
                re = __builtin_fmaf(line[i], t.x, re);
                im = __builtin_fmaf(line[i], t.y, im);
                idx += k;
                idx -= idx >= W ? W : 0;
            }
            y[r * Wh + k] = make_float2(re, im);
        }
    }
}

// out[p][n][k] = sum_m in[p][m][k] (* filter[m][k]) e^{-+ 2 pi i m n / H}; lanes = consecutive k (coalesced), one wave per n
__global__ void __launch_bounds__(kBlock) dft_cols_kernel(const float2* __restrict__ in, const float* __restrict__ filter, float2* __restrict__ out,
                                                          int64_t planes, int H, int K, int inverse) {
    kernarg_touch_for(in, filter, out, planes, H, K, inverse);
    extern __shared__ __align__(16) unsigned char dft_lds[];
    float2* tw = reinterpret_cast<float2*>(dft_lds);
    build_table(tw, H, inverse ? 1 : -1);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int kWaves = kBlock / 64;
    const int ktiles = (K + 63) / 64, nchunks = (H + kWaves - 1) / kWaves;
    const int64_t units = planes * ktiles * nchunks;  // a workgroup: 64 columns x kWaves output rows of one plane (one row per wave)
    for (int64_t u = blockIdx.x; u < units; u += gridDim.x) {
        const int64_t p = u / ((int64_t)ktiles * nchunks);
        const int rem = (int)(u % ((int64_t)ktiles * nchunks));
        const int k = (rem / nchunks) * 64 + lane;
        const int n = (rem % nchunks) * kWaves + wave;
        if (k >= K || n >= H) continue;
        const float2* src = in + p * (int64_t)H * K + k;
        {
            float re = 0.0f, im = 0.0f;
            int idx = 0;
            for (int m = 0; m < H; ++m) {
                float2 v = src[(int64_t)m * K];
                if (filter) {
                    const float f = filter[(int64_t)m * K + k];
                    v.x *= f;
                    v.y *= f;
                }
                const float2 t = tw[idx];
                re = __builtin_fmaf(v.x, t.x, __builtin_fmaf(-v.y, t.y, re));
                im = __builtin_fmaf(v.x, t.y, __builtin_fmaf(v.y, t.x, im));
                idx += n;
                idx -= idx >= H ? H : 0;
            }
            out[(p * H + n) * (int64_t)K + k] = make_float2(re, im);
        }
    }
}

// out[row][x] = scale * (Re y0 + sum_{k >= 1} w_k Re(y_k e^{+2 pi i k x / W})), w_k = 2 (1 for the Nyquist column of an even W)
template <bool STATS>
__global__ void __launch_bounds__(kBlock) dft_rows_c2r_kernel(const float2* __restrict__ y, float* __restrict__ out, int64_t rows, int W, float scale,
                                                              double* partials) {
    kernarg_touch_for(y, out, rows, W, scale, partials);
    extern __shared__ __align__(16) unsigned char dft_lds[];
    __shared__ double red[2 * kBlock / 64];
    float2* tw = reinterpret_cast<float2*>(dft_lds);
    float2* line = tw + W;
    const int Wh = W / 2 + 1;
    build_table(tw, W, 1);
    double s = 0.0, q = 0.0;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        __syncthreads();
        for (int k = threadIdx.x; k < Wh; k += kBlock) {
            float2 v = y[r * Wh + k];
            const float wgt = (k == 0 || (2 * k == W)) ? 1.0f : 2.0f;
            line[k] = make_float2(v.x * wgt, v.y * wgt);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < W; i += kBlock) {
            float acc = line[0].x;
            int idx = i;  // (k i) mod W for k = 1
            for (int k = 1; k < Wh; ++k) {
                const float2 t = tw[idx];
                acc = __builtin_fmaf(line[k].x, t.x, __builtin_fmaf(-line[k].y, t.y, acc));
                idx += i;
                idx -= idx >= W ? W : 0;
            }
            const float v = acc * scale;
            out[r * W + i] = v;
            if constexpr (STATS) {
                s += (double)v;
                q += (double)v * (double)v;
            }
        }
    }
    if constexpr (STATS) write_partial<kBlock>(s, q, partials, red);
}

}  // namespace sonar

using namespace sonar;

extern "C" int sonar_dft_rows_r2c_f32(const float* x, float* y, int64_t rows, int64_t W, void* stream) {
    SONAR_REQUIRE(x && y && rows >= 0 && W >= 1 && W <= kDirectMax, SONAR_ERR_ARG, "sonar_dft_rows_r2c_f32: bad argument (1 <= W <= %d)", kDirectMax);
    if (rows == 0) return SONAR_OK;
    if ((reinterpret_cast<uintptr_t>(y) & 7u) == 0 && sonar_lines_rows_r2c(x, y, rows, W, (hipStream_t)stream)) return check_launch("sonar_dft_rows_r2c_f32");
    hipLaunchKernelGGL(dft_rows_r2c_kernel, dim3((int)std::min<int64_t>(rows, 4096)), dim3(kBlock), (size_t)W * (sizeof(float2) + sizeof(float)),
                       (hipStream_t)stream, x, reinterpret_cast<float2*>(y), rows, (int)W);
    return check_launch("sonar_dft_rows_r2c_f32");
}

extern "C" int sonar_dft_cols_f32(const float* in, const float* filter, float* out, int64_t planes, int64_t H, int64_t K, int inverse,
                                  void* stream) {
    SONAR_REQUIRE(in && out && in != out && planes >= 0 && H >= 1 && H <= kDirectMax && K >= 1 && K <= kDirectMax, SONAR_ERR_ARG,
                  "sonar_dft_cols_f32: bad argument (out of place, lines of at most %d)", kDirectMax);
    SONAR_REQUIRE(inverse >= 0 && inverse <= 2 && (inverse != 2 || filter), SONAR_ERR_ARG, "sonar_dft_cols_f32: inverse is 0, 1 or 2 (2 needs the filter)");
    if (planes == 0) return SONAR_OK;
    if (((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 7u) == 0 &&
        sonar_lines_cols(in, filter, out, planes, H, K, inverse, (hipStream_t)stream))
        return check_launch("sonar_dft_cols_f32");
    // the round trip (forward, x filter, inverse) only exists with the columns in LDS: the caller runs the two passes instead
    SONAR_REQUIRE(inverse != 2, SONAR_ERR_UNSUPPORTED, "sonar_dft_cols_f32: the fused round trip does not take these buffers");
    const int64_t units = planes * ((K + 63) / 64) * ((H + kBlock / 64 - 1) / (kBlock / 64));
    hipLaunchKernelGGL(dft_cols_kernel, dim3((int)std::min<int64_t>(units, 1 << 16)), dim3(kBlock), (size_t)H * sizeof(float2), (hipStream_t)stream,
                       reinterpret_cast<const float2*>(in), filter, reinterpret_cast<float2*>(out), planes, (int)H, (int)K, inverse);
    return check_launch("sonar_dft_cols_f32");
}

extern "C" int sonar_dft_rows_c2r_f32(const float* y, float* out, int64_t rows, int64_t W, float scale, double* partials, void* stream) {
    SONAR_REQUIRE(y && out && rows >= 0 && W >= 1 && W <= kDirectMax, SONAR_ERR_ARG, "sonar_dft_rows_c2r_f32: bad argument (1 <= W <= %d)", kDirectMax);
    if (rows > 0 && (reinterpret_cast<uintptr_t>(y) & 7u) == 0 && sonar_lines_rows_c2r(y, out, rows, W, scale, partials, (hipStream_t)stream))
        return check_launch("sonar_dft_rows_c2r_f32");
    const size_t lds = (size_t)W * sizeof(float2) + (size_t)(W / 2 + 1) * sizeof(float2);
    const int g = (int)std::max<int64_t>(1, std::min<int64_t>(rows, kNPart));
    if (partials)
        hipLaunchKernelGGL((dft_rows_c2r_kernel<true>), dim3(g), dim3(kBlock), lds, (hipStream_t)stream, reinterpret_cast<const float2*>(y), out, rows,
                           (int)W, scale, partials);
    else if (rows > 0)
        hipLaunchKernelGGL((dft_rows_c2r_kernel<false>), dim3(g), dim3(kBlock), lds, (hipStream_t)stream, reinterpret_cast<const float2*>(y), out,
                           rows, (int)W, scale, partials);
    return check_launch("sonar_dft_rows_c2r_f32");
}
