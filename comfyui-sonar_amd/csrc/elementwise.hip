// Streaming elementwise + reduction kernels: whole-tensor normalisation (scale_noise), blends,
// chain accumulation, mask mix, and the fused Sonar momentum steps.  All HBM-bound: 16 B/lane
// coalesced accesses, grid-stride over <= kMaxGrid blocks of 256 threads, fp64 block reductions.
#include <math.h>
#include <string.h>

#include "common.h"

namespace sonar {

// ------------------------------------------------------------------------------------------------
// V-wide pack of floats (V = 4 -> one dwordx4 access per lane; V = 1 for unaligned buffers).
template <int V>
struct Pack {
    float v[V];
};
template <int V>
__device__ __forceinline__ Pack<V> load(const float* p, int64_t i) {
    Pack<V> r;
    if constexpr (V == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p + i);
        r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
    } else {
        r.v[0] = p[i];
    }
    return r;
}
#ifndef SONAR_EW_NT
#define SONAR_EW_NT 0  // profiling builds: every elementwise kernel's 16-byte stores with the non-temporal hint
#endif
template <int V, bool NT = (SONAR_EW_NT != 0)>
__device__ __forceinline__ void store(float* p, int64_t i, const Pack<V>& r) {
    if constexpr (V == 4) {
        store4<NT>(p + i, r.v[0], r.v[1], r.v[2], r.v[3]);
    } else {
        p[i] = r.v[0];
    }
}

// Generic driver: op.template run<V>(elem_index) handles V consecutive elements.
template <int V, typename Op>
__global__ void __launch_bounds__(kBlock) ew_kernel(Op op, int64_t n) {
    kernarg_touch_for(op, n);
    const int64_t nv = n / V;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nv; i += stride) op.template run<V>(i * V);
    if constexpr (V > 1) {
        if (blockIdx.x == 0)
            for (int64_t i = nv * V + threadIdx.x; i < n; i += kBlock) op.template run<1>(i);
    }
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Pure streaming kernels: one 16-byte item per thread (measured on MI355X, 3 reads + 2 writes of 134 MB each:
// 32768 blocks x 1 item 5.8 TB/s, 2048 persistent blocks x 16 items 5.3 TB/s -- scratch/stream5.cpp)
static inline int grid_stream(int64_t items) { return (int)std::max<int64_t>(1, std::min<int64_t>((items + kBlock - 1) / kBlock, 1 << 20)); }

template <typename Op>
static int launch_ew(Op op, int64_t n, bool vec_ok, hipStream_t st, const char* what) {
    if (n == 0) return SONAR_OK;
    if (vec_ok) {
        hipLaunchKernelGGL((ew_kernel<4, Op>), dim3(grid_stream(n / 4 + 1)), dim3(kBlock), 0, st, op, n);
    } else {
        hipLaunchKernelGGL((ew_kernel<1, Op>), dim3(grid_stream(n)), dim3(kBlock), 0, st, op, n);
    }
    return check_launch(what);
}

// ------------------------------------------------------------------------------------------------
// stats: (sum, sumsq) partials in fp64
template <int V>
__global__ void __launch_bounds__(kBlock) stats_kernel(const float* __restrict__ x, int64_t n, double* partials) {
    kernarg_touch_for(x, n, partials);
    __shared__ double red[2 * kBlock / 64];
    double s = 0.0, q = 0.0;
    const int64_t nv = n / V;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    // two independent loads in flight per iteration
    for (; i + stride < nv; i += 2 * stride) {
        const Pack<V> a = load<V>(x, i * V);
        const Pack<V> b = load<V>(x, (i + stride) * V);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const double da = a.v[k], db = b.v[k];
            s += da; q += da * da;
            s += db; q += db * db;
        }
    }
    for (; i < nv; i += stride) {
        const Pack<V> a = load<V>(x, i * V);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const double da = a.v[k];
            s += da; q += da * da;
        }
    }
    if (V > 1 && blockIdx.x == 0)
        for (int64_t j = nv * V + threadIdx.x; j < n; j += kBlock) {
            const double da = x[j];
            s += da; q += da * da;
        }
    write_partial<kBlock>(s, q, partials, red);
}

__global__ void __launch_bounds__(kBlock) stats_finalize_kernel(const double* __restrict__ partials, int64_t npart,
                                                                 int64_t n, double* out3) {
    kernarg_touch_for(partials, npart, n, out3);
    __shared__ double red[2 * kBlock / 64];
    double s = 0.0, q = 0.0;
    for (int64_t i = threadIdx.x; i < npart; i += kBlock) {
        s += partials[2 * i];
        q += partials[2 * i + 1];
    }
    block_sum2<kBlock>(s, q, red);
    if (threadIdx.x == 0) {
        out3[0] = s;
        out3[1] = q;
        out3[2] = (double)n;
    }
}

template <int V, bool NT = false /* common.h store4: launch-bound sizes */>
__global__ void __launch_bounds__(kBlock) scale_noise_kernel(float* x, int64_t n, float factor, int normalized,
                                                             float thr_sd, const double* __restrict__ partials,
                                                             int64_t npart, int64_t n_total, double* out_partials) {
    kernarg_touch_for(x, n, factor, normalized, thr_sd, partials, npart, n_total, out_partials);
    __shared__ double red[2 * kBlock / 64];
    __shared__ NormDecision sh;
    NormDecision d{0.f, 1.f, 0, 0};
    if (normalized) d = decide_norm<kBlock>(partials, npart, n_total, thr_sd, red, &sh);
    if (normalized == 2) {  // x *= factor / std (py/noise_generation.py:702): one multiplier, no mean shift
        factor = factor / d.stdv;
        d.do_sub = 0;
        d.do_div = 0;
    }
    const bool do_mul = factor != 1.0f;
    if (out_partials && blockIdx.x == 0) {
        // statistics of the RESULT, derived from the input's: y = ((x - m) / s) * f  ->  sum y = f (S - n m) / s,
        // sum y^2 = f^2 (Q - 2 m S + n m^2) / s^2.  A wrapper that normalises this tensor again reads them instead of the tensor.
        double S = 0.0, Q = 0.0;
        for (int64_t i = threadIdx.x; i < npart; i += kBlock) {
            S += partials[2 * i];
            Q += partials[2 * i + 1];
        }
        __syncthreads();
        block_sum2<kBlock>(S, Q, red);
        if (threadIdx.x == 0) {
            const double nt = (double)n_total, m = d.do_sub ? (double)d.mean : 0.0, sd = d.do_div ? (double)d.stdv : 1.0;
            const double f = do_mul ? (double)factor : 1.0;
            out_partials[0] = f * (S - nt * m) / sd;
            out_partials[1] = f * f * (Q - 2.0 * m * S + nt * m * m) / (sd * sd);
        }
        for (int j = 1 + threadIdx.x; j < kNPart; j += kBlock) {
            out_partials[2 * j] = 0.0;
            out_partials[2 * j + 1] = 0.0;
        }
    }
    if (!d.do_sub && !d.do_div && !do_mul) return;  // nothing to change (every block takes the same decision)
    const int64_t nv = n / V;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    auto f = [&](float v) {
        if (d.do_sub) v = v - d.mean;
        if (d.do_div) v = v / d.stdv;
        if (do_mul) v = v * factor;
        return v;
    };
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    for (; i + stride < nv; i += 2 * stride) {
        Pack<V> a = load<V>(x, i * V);
        Pack<V> b = load<V>(x, (i + stride) * V);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            a.v[k] = f(a.v[k]);
            b.v[k] = f(b.v[k]);
        }
        store<V, NT>(x, i * V, a);
        store<V, NT>(x, (i + stride) * V, b);
    }
    for (; i < nv; i += stride) {
        Pack<V> a = load<V>(x, i * V);
#pragma unroll
        for (int k = 0; k < V; ++k) a.v[k] = f(a.v[k]);
        store<V, NT>(x, i * V, a);
    }
    if (V > 1 && blockIdx.x == 0)
        for (int64_t j = nv * V + threadIdx.x; j < n; j += kBlock) x[j] = f(x[j]);
}

// Kernels with ONE workgroup per row (a row is usually a whole latent or plane, and there are few of them) run at the latency of that
// workgroup's loads, not at any throughput: kRowThreads threads, and row_visit keeps four independent 16-byte loads per thread in
// flight (amax over four rows of 65536 values: 63 -> 6 us).  f sees every element once, in no particular order.
// Short rows (there are then usually many) keep 256-thread workgroups.
constexpr int kRowThreads = 1024;
constexpr int64_t kLongRow = 8192;
template <int THREADS, typename F>
__device__ __forceinline__ void row_visit(const float* __restrict__ row, int64_t n, F&& f) {
    if ((reinterpret_cast<uintptr_t>(row) & 15u) == 0 && (n & 3) == 0) {
        const float4* row4 = reinterpret_cast<const float4*>(row);
        const int64_t n4 = n >> 2;
        int64_t i = threadIdx.x;
        for (; i + 3 * THREADS < n4; i += 4 * THREADS) {
            const float4 a = row4[i], b = row4[i + THREADS], c = row4[i + 2 * THREADS], d = row4[i + 3 * THREADS];
            f(a.x); f(a.y); f(a.z); f(a.w);
            f(b.x); f(b.y); f(b.z); f(b.w);
            f(c.x); f(c.y); f(c.z); f(c.w);
            f(d.x); f(d.y); f(d.z); f(d.w);
        }
        for (; i < n4; i += THREADS) {
            const float4 a = row4[i];
            f(a.x); f(a.y); f(a.z); f(a.w);
        }
    } else {
        for (int64_t i = threadIdx.x; i < n; i += THREADS) f(row[i]);
    }
}

// one workgroup per row: 1024 threads for long rows, 256 for short ones
#define SONAR_ROW_LAUNCH(kernel, row_len, nrows, lds, st, ...) \
    do { \
        if ((row_len) >= kLongRow) hipLaunchKernelGGL(kernel<kRowThreads>, dim3(grid_for(nrows, 1)), dim3(kRowThreads), lds, st, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel<kBlock>, dim3(grid_for(nrows, 1)), dim3(kBlock), lds, st, __VA_ARGS__); \
    } while (0)

// normalize_dims variant (py/utils.py:96-99): one block per row of `inner` contiguous elements
template <int THREADS>
__global__ void __launch_bounds__(THREADS) scale_rows_kernel(float* x, int64_t rows, int64_t inner, float factor) {
    kernarg_touch_for(x, rows, inner, factor);
    __shared__ double red[2 * THREADS / 64];
    __shared__ float sh_val;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        float* row = x + r * inner;
        double s = 0.0, q = 0.0;
        row_visit<THREADS>(row, inner, [&](float f) {
            const double v = f;
            s += v; q += v * v;
        });
        block_sum2<THREADS>(s, q, red);
        if (threadIdx.x == 0) {
            const double nt = (double)inner;
            const double var = (q - s * (s / nt)) / (nt - 1.0);
            sh_val = (float)sqrt(var > 0.0 || !(var == var) ? var : 0.0);
        }
        __syncthreads();
        const float sd = sh_val;
        double s2 = 0.0, q2 = 0.0;
        row_visit<THREADS>(row, inner, [&](float f) { s2 += (double)(f / sd); });
        __syncthreads();
        block_sum2<THREADS>(s2, q2, red);
        if (threadIdx.x == 0) sh_val = (float)(s2 / (double)inner);
        __syncthreads();
        const float mean = sh_val;
        if ((reinterpret_cast<uintptr_t>(row) & 15u) == 0 && (inner & 3) == 0) {
            float4* row4 = reinterpret_cast<float4*>(row);
            for (int64_t i = threadIdx.x; i < (inner >> 2); i += THREADS) {
                float4 v = row4[i];
                v.x = (v.x / sd - mean) * factor; v.y = (v.y / sd - mean) * factor;
                v.z = (v.z / sd - mean) * factor; v.w = (v.w / sd - mean) * factor;
                row4[i] = v;
            }
        } else {
            for (int64_t i = threadIdx.x; i < inner; i += THREADS) row[i] = (row[i] / sd - mean) * factor;
        }
        __syncthreads();
    }
}

template <int THREADS>
__global__ void __launch_bounds__(THREADS) minmax_rows_kernel(const float* __restrict__ x, int64_t rows, int64_t inner,
                                                              float* out_min, float* out_max) {
    kernarg_touch_for(x, rows, inner, out_min, out_max);
    __shared__ float smin[THREADS / 64], smax[THREADS / 64];
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const float* row = x + r * inner;
        float lo = INFINITY, hi = -INFINITY;
        row_visit<THREADS>(row, inner, [&](float v) {
            lo = fminf(lo, v);
            hi = fmaxf(hi, v);
        });
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo = fminf(lo, __shfl_down(lo, off, 64));
            hi = fmaxf(hi, __shfl_down(hi, off, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            smin[threadIdx.x >> 6] = lo;
            smax[threadIdx.x >> 6] = hi;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < THREADS / 64; ++w) {
                lo = fminf(lo, smin[w]);
                hi = fmaxf(hi, smax[w]);
            }
            out_min[r] = lo;
            out_max[r] = hi;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
struct BlendOp {
    int mode;
    const float *a, *b;
    float t;
    float* out;
    template <int V>
    __device__ __forceinline__ void run(int64_t i) const {
        const Pack<V> pa = load<V>(a, i), pb = load<V>(b, i);
        Pack<V> r;
#pragma unroll
        for (int k = 0; k < V; ++k) r.v[k] = blend<float>(mode, pa.v[k], pb.v[k], t);
        store<V>(out, i, r);
    }
};

struct BlendTensorOp {
    int mode;
    const float *a, *b, *t;
    int64_t tn;
    float* out;
    template <int V>
    __device__ __forceinline__ void run(int64_t i) const {
        const Pack<V> pa = load<V>(a, i), pb = load<V>(b, i);
        Pack<V> r;
#pragma unroll
        for (int k = 0; k < V; ++k) r.v[k] = blend<float>(mode, pa.v[k], pb.v[k], t[(i + k) % tn]);
        store<V>(out, i, r);
    }
};

struct AxpbyOp {
    float* y;
    float ymul;
    const float* x;
    float xmul;
    template <int V>
    __device__ __forceinline__ void run(int64_t i) const {
        Pack<V> py = load<V>(y, i);
        const Pack<V> px = load<V>(x, i);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            // y*ymul and x*xmul are skipped when the multiplier is exactly 1 (the reference's
            // scale_noise/add_ sequence does not multiply then)
            const float yy = ymul != 1.0f ? py.v[k] * ymul : py.v[k];
            const float xx = xmul != 1.0f ? px.v[k] * xmul : px.v[k];
            py.v[k] = yy + xx;
        }
        store<V>(y, i, py);
    }
};

struct AffineOp {
    float* x;
    float sub, mul, add;
    template <int V>
    __device__ __forceinline__ void run(int64_t i) const {
        Pack<V> p = load<V>(x, i);
#pragma unroll
        for (int k = 0; k < V; ++k) p.v[k] = (p.v[k] - sub) * mul + add;
        store<V>(x, i, p);
    }
};

struct ScalarOp {
    int op;
    const float *a, *b;
    float s;
    float* out;
    template <int V>
    __device__ __forceinline__ void run(int64_t i) const {
        const Pack<V> pa = load<V>(a, i);
        Pack<V> pb, r;
        if (op == 2) pb = load<V>(b, i);
#pragma unroll
        for (int k = 0; k < V; ++k) r.v[k] = op == 0 ? pa.v[k] * s : op == 1 ? pa.v[k] / s : (pa.v[k] - pb.v[k]) / s;
        store<V>(out, i, r);
    }
};

template <int THREADS>
__global__ void __launch_bounds__(THREADS) rowstats_kernel(const float* __restrict__ x, int64_t rows, int64_t inner,
                                                           float* mean, float* stdv) {
    kernarg_touch_for(x, rows, inner, mean, stdv);
    __shared__ double red[2 * THREADS / 64];
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const float* row = x + r * inner;
        double s = 0.0, q = 0.0;
        row_visit<THREADS>(row, inner, [&](float f) {
            const double v = f;
            s += v; q += v * v;
        });
        block_sum2<THREADS>(s, q, red);
        if (threadIdx.x == 0) {
            const double nt = (double)inner, m = s / nt;
            const double var = (q - s * m) / (nt - 1.0);
            mean[r] = (float)m;
            stdv[r] = (float)sqrt(var > 0.0 || !(var == var) ? var : 0.0);
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(kBlock) row_affine_kernel(int op, const float* __restrict__ x, int64_t rows,
                                                             int64_t inner, const float* __restrict__ a,
                                                             const float* __restrict__ b, float* out) {
    kernarg_touch_for(op, x, rows, inner, a, b, out);
    const int64_t total = rows * inner;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / inner;
        out[i] = op == 0 ? (x[i] - a[r]) / b[r] : x[i] * b[r] + a[r];
    }
}

// normalize_to_scale's tail (py/utils.py:462-469): ((x - lo) / ((hi - lo) + eps)) * (tmax - tmin) + tmin, clamped; each step rounded
// on its own as the reference's in-place tensor ops are.  The targets are Python floats there: their difference is formed in double
// and rounded to fp32 ONCE (`span`; 0.3 - 0.1 is 0.2, not 0.20000002)
__global__ void __launch_bounds__(kBlock) minmax_rescale_kernel(const float* __restrict__ x, int64_t rows, int64_t inner,
                                                                 const float* __restrict__ lo, const float* __restrict__ hi, float eps,
                                                                 float tmin, float tmax, float span, float* out) {
    kernarg_touch_for(x, rows, inner, lo, hi, eps, tmin, tmax, span, out);
    const int64_t total = rows * inner;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / inner;
        const float denom = __fadd_rn(__fsub_rn(hi[r], lo[r]), eps);
        float v = __fsub_rn(x[i], lo[r]) / denom;
        v = __fadd_rn(__fmul_rn(v, span), tmin);
        out[i] = v != v ? v : fminf(fmaxf(v, tmin), tmax);  // clamp_ keeps NaN
    }
}

// normalize_to_scale_adv (py/utils.py:473-510): the negative and the positive values of a row are rescaled separately, each between its
// own extremes.  Pass 1: per row (min, max) over the negatives and over the positives (+-inf where a sign has no value).
__global__ void __launch_bounds__(kBlock) signed_minmax_rows_kernel(const float* __restrict__ x, int64_t rows, int64_t inner, float4* __restrict__ out) {
    kernarg_touch_for(x, rows, inner, out);
    __shared__ float red[4][kBlock / 64];
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        float nlo = INFINITY, nhi = -INFINITY, plo = INFINITY, phi = -INFINITY;
        const float* row = x + r * inner;
        for (int64_t i = threadIdx.x; i < inner; i += kBlock) {
            const float v = row[i];
            if (v < 0.0f) {
                nlo = fminf(nlo, v);
                nhi = fmaxf(nhi, v);
            } else if (v > 0.0f) {
                plo = fminf(plo, v);
                phi = fmaxf(phi, v);
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            nlo = fminf(nlo, __shfl_xor(nlo, off));
            nhi = fmaxf(nhi, __shfl_xor(nhi, off));
            plo = fminf(plo, __shfl_xor(plo, off));
            phi = fmaxf(phi, __shfl_xor(phi, off));
        }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
            const int w = threadIdx.x >> 6;
            red[0][w] = nlo; red[1][w] = nhi; red[2][w] = plo; red[3][w] = phi;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < kBlock / 64; ++w) {
                nlo = fminf(nlo, red[0][w]);
                nhi = fmaxf(nhi, red[1][w]);
                plo = fminf(plo, red[2][w]);
                phi = fmaxf(phi, red[3][w]);
            }
            out[r] = make_float4(nlo, nhi, plo, phi);
        }
    }
}

// Pass 2: x < 0 -> normalize_to_scale over the negatives to [min_neg, max_neg] (max_neg >= 0: the row's own largest negative), x > 0 ->
// over the positives to [min_pos, max_pos] (min_pos < 0: the row's own smallest positive), 0 stays 0; a skipped sign is copied.  Each
// step rounded on its own, as the reference's in-place tensor ops are (the same sequence as minmax_rescale_kernel).
__global__ void __launch_bounds__(kBlock) signed_rescale_kernel(const float* __restrict__ x, int64_t rows, int64_t inner,
                                                                 const float4* __restrict__ stats, double min_neg, double max_neg, double min_pos,
                                                                 double max_pos, int skip_neg, int skip_pos, float eps, float* __restrict__ out) {
    kernarg_touch_for(x, rows, inner, stats, min_neg, max_neg, min_pos, max_pos, skip_neg, skip_pos, eps, out);
    const int64_t total = rows * inner;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const float4 st = stats[i / inner];
        const float v = x[i];
        float res = 0.0f;
        if (v < 0.0f || v > 0.0f) {
            const bool neg = v < 0.0f;
            if (neg ? skip_neg : skip_pos) {
                res = v;
            } else {
                const float lo = neg ? st.x : st.z, hi = neg ? st.y : st.w;
                // the targets are Python floats in the reference (the data-derived ones `.item()` of an fp32 value): their difference is
                // formed in double and meets the fp32 tensor as one rounded scalar; add_ / clamp_ round each bound on its own
                const double tmin_d = neg ? min_neg : (min_pos < 0.0 ? (double)st.z : min_pos);
                const double tmax_d = neg ? (max_neg >= 0.0 ? (double)st.y : max_neg) : max_pos;
                const float span = (float)(tmax_d - tmin_d), tmin = (float)tmin_d, tmax = (float)tmax_d;
                const float denom = __fadd_rn(__fsub_rn(hi, lo), eps);
                float q = __fsub_rn(v, lo) / denom;
                q = __fadd_rn(__fmul_rn(q, span), tmin);
                res = q != q ? q : fminf(fmaxf(q, tmin), tmax);
            }
        } else if (v != v) {
            res = 0.0f;  // NaN is neither < 0 nor > 0: the reference's masks leave the zero of zeros_like
        }
        out[i] = res;
    }
}

__global__ void __launch_bounds__(kBlock) amax_mid_kernel(const float* __restrict__ x, int64_t outer, int64_t mid,
                                                           int64_t inner, int use_abs, float* peak) {
    kernarg_touch_for(x, outer, mid, inner, use_abs, peak);
    const int64_t total = outer * inner;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
        const int64_t o = t / inner, i = t - o * inner;
        const float* p = x + o * mid * inner + i;
        float hi = -INFINITY;
        bool nan = false;
        for (int64_t m = 0; m < mid; ++m) {
            float v = p[m * inner];
            if (use_abs) v = fabsf(v);
            nan |= v != v;
            hi = fmaxf(hi, v);
        }
        peak[t] = nan ? NAN : hi;  // torch.amax propagates NaN
    }
}

// inner == 1: one workgroup per row of `mid` contiguous elements
template <int THREADS>
__global__ void __launch_bounds__(THREADS) amax_row_kernel(const float* __restrict__ x, int64_t rows, int64_t len, int use_abs,
                                                                float* peak) {
    kernarg_touch_for(x, rows, len, use_abs, peak);
    __shared__ float smax[THREADS / 64];
    __shared__ int snan[THREADS / 64];
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        float hi = -INFINITY;
        int nan = 0;
        row_visit<THREADS>(x + r * len, len, [&](float v) {
            if (use_abs) v = fabsf(v);
            nan |= v != v;
            hi = fmaxf(hi, v);
        });
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            hi = fmaxf(hi, __shfl_down(hi, off, 64));
            nan |= __shfl_down(nan, off, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            smax[threadIdx.x >> 6] = hi;
            snan[threadIdx.x >> 6] = nan;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < THREADS / 64; ++w) {
                hi = fmaxf(hi, smax[w]);
                nan |= snan[w];
            }
            peak[r] = nan ? NAN : hi;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(kBlock) div_mid_kernel(float* x, int64_t outer, int64_t mid, int64_t inner,
                                                          const float* __restrict__ d) {
    kernarg_touch_for(x, outer, mid, inner, d);
    const int64_t total = outer * mid * inner, plane = mid * inner;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
        const int64_t o = t / plane;
        const int64_t i = (t - o * plane) % inner;
        x[t] = x[t] / d[o * inner + i];
    }
}

// ---- ModulatedNoise (py/noise.py:784-866) ----------------------------------------------------------------------------------
// unbiased std over the middle axis of x[outer][mid][inner] (torch.std(dim=-3, keepdim=True)); mid == 1 gives NaN like torch
__global__ void __launch_bounds__(kBlock) std_mid_kernel(const float* __restrict__ x, int64_t outer, int64_t mid, int64_t inner,
                                                          float* stdv) {
    kernarg_touch_for(x, outer, mid, inner, stdv);
    const int64_t total = outer * inner;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
        const int64_t o = t / inner, i = t - o * inner;
        const float* p = x + o * mid * inner + i;
        double s = 0.0;
        for (int64_t m = 0; m < mid; ++m) s += (double)p[m * inner];
        const double mean = s / (double)mid;
        double q = 0.0;
        for (int64_t m = 0; m < mid; ++m) {
            const double d = (double)p[m * inner] - mean;
            q += d * d;
        }
        stdv[t] = (float)sqrt(q / (double)(mid - 1));
    }
}

// v = x * k * (1 / (std * |strength| + 1) + 1), std broadcast through (so, sm, si) strides over x's [outer][mid][inner] view
// (py/noise.py:799-803: noise * scaling + noise).  Optional store; optional partials: slot pair (sum x^2, sum v^2) per block.
__global__ void __launch_bounds__(kBlock) bcast_gain_kernel(const float* __restrict__ x, const float* __restrict__ stdv, int64_t outer,
                                                             int64_t mid, int64_t inner, int64_t so, int64_t sm, int64_t si,
                                                             float abs_strength, float k, float* out, double* partials) {
    kernarg_touch_for(x, stdv, outer, mid, inner, so, sm, si, abs_strength, k, out, partials);
    __shared__ double red[2 * kBlock / 64];
    const int64_t total = outer * mid * inner, plane = mid * inner;
    double sx = 0.0, sv = 0.0;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
        const int64_t o = t / plane, r = t - o * plane, m = r / inner, i = r - m * inner;
        const float sd = stdv[o * so + m * sm + i * si];
        const float gain = 1.0f / (sd * abs_strength + 1.0f);
        const float xv = x[t];
        const float plain = xv * k;
        const float v = plain * gain + plain;
        if (out) out[t] = v;
        sx += (double)xv * (double)xv;
        sv += (double)v * (double)v;
    }
    if (partials) write_partial<kBlock>(sx, sv, partials, red);
}

// out = a * (a_mul * rho) + x * x_mul with rho = sqrt(num_mul * sum(num[2i]) / sum(den[2i+1])) -- the "scale to normal noise
// strength" ratio of two L2 norms (py/noise.py:805-810), taken from partial slots so the host never reads it back
__global__ void __launch_bounds__(kBlock) ratio_mix_kernel(const float* __restrict__ a, float a_mul, const float* __restrict__ x,
                                                            float x_mul, const double* __restrict__ num, double num_mul,
                                                            const double* __restrict__ den, float* out, int64_t n) {
    kernarg_touch_for(a, a_mul, x, x_mul, num, num_mul, den, out, n);
    __shared__ double red[2 * kBlock / 64];
    __shared__ float srho;
    double sn = 0.0, sd = 0.0;
    for (int j = threadIdx.x; j < kNPart; j += kBlock) {
        sn += num[2 * j];
        sd += den[2 * j + 1];
    }
    block_sum2<kBlock>(sn, sd, red);
    if (threadIdx.x == 0) srho = (float)sqrt(num_mul * sn / sd);
    __syncthreads();
    const float am = a_mul * srho;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < n; t += (int64_t)gridDim.x * kBlock)
        out[t] = a[t] * am + x[t] * x_mul;
}

// LaplacianNoiseGenerator (py/noise_generation.py:789-802): x = x / div_fac + Laplace(loc, scale) with the Laplace variate built
// from a uniform u in (eps - 1, 1) exactly as torch.distributions.Laplace.rsample does: loc - scale * sign(u) * log1p(-|u|)
struct LaplaceAddOp {
    float* x;
    const float* u;
    float div_fac, loc, scale;
    template <int V>
    __device__ __forceinline__ void run(int64_t i) const {
        Pack<V> p = load<V>(x, i);
        const Pack<V> pu = load<V>(u, i);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float uu = pu.v[k];
            const float a = fmaxf(fabsf(uu), 1.17549435e-38f);  // clamp(min=finfo.tiny)
            const float sgn = uu > 0.0f ? 1.0f : uu < 0.0f ? -1.0f : 0.0f;
            p.v[k] = p.v[k] / div_fac + (loc - scale * sgn * log1pf(-a));
        }
        store<V>(x, i, p);
    }
};

// ---- StudentTNoiseGenerator (py/noise_generation.py:652-677) --------------------------------------------------------------------
// x (a standard normal draw) -> loc + scale * x * rsqrt(max(g / 0.5, tiny) / df), g the torch._standard_gamma(df / 2) draw
// (torch.distributions.StudentT.rsample / Chi2 / Gamma.rsample)
struct StudentTOp {
    float* x;
    const float* g;
    float loc, scale, df;
    template <int V>
    __device__ __forceinline__ void run(int64_t i) const {
        Pack<V> p = load<V>(x, i);
        const Pack<V> pg = load<V>(g, i);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float z = fmaxf(pg.v[k] / 0.5f, 1.17549435e-38f);
            p.v[k] = loc + scale * (p.v[k] * (1.0f / sqrtf(z / df)));
        }
        store<V>(x, i, p);
    }
};

// q-quantile (torch.quantile, linear interpolation) of |x| over each row of `inner` contiguous values: radix select on the bit
// patterns (non-negative floats order like unsigned integers), four 8-bit histogram passes in LDS, then one pass for the next
// order statistic.  One workgroup per row.  rank = lo + frac is computed by the host in fp32 like torch does.
// The first pass sees the exponent byte -- two or three bins take every value -- so each bin is kept in kQuantReplicas copies (lane
// mod 32 picks one, copies of a bin in distinct banks): same-address LDS atomics serialise, 226 -> 20 us for four rows of 65536.
constexpr int kQuantReplicas = 32;
template <int THREADS>
__global__ void __launch_bounds__(THREADS) abs_quantile_rows_kernel(const float* __restrict__ x, int64_t rows, int64_t inner, int64_t lo,
                                                                        float frac, float* out) {
    kernarg_touch_for(x, rows, inner, lo, frac, out);
    __shared__ unsigned hist[256 * kQuantReplicas];
    __shared__ unsigned total[256];
    __shared__ unsigned sh_prefix, sh_k, sh_cnt, sh_min;
    const unsigned rep = threadIdx.x & (kQuantReplicas - 1);
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        const float* row = x + r * inner;
        const bool vec = (reinterpret_cast<uintptr_t>(row) & 15u) == 0 && (inner & 3) == 0;
        const float4* row4 = reinterpret_cast<const float4*>(row);
        unsigned prefix = 0, mask = 0, k = (unsigned)lo;
        for (int shift = 24; shift >= 0; shift -= 8) {
            for (int b = threadIdx.x; b < 256 * kQuantReplicas; b += THREADS) hist[b] = 0;
            __syncthreads();
            auto tally = [&](float v) {
                const unsigned bits = __float_as_uint(v) & 0x7FFFFFFFu;
                if ((bits & mask) == prefix) atomicAdd(&hist[((bits >> shift) & 255u) * kQuantReplicas + rep], 1u);
            };
            if (vec) {
                for (int64_t i = threadIdx.x; i < (inner >> 2); i += THREADS) {
                    const float4 v = row4[i];
                    tally(v.x); tally(v.y); tally(v.z); tally(v.w);
                }
            } else {
                for (int64_t i = threadIdx.x; i < inner; i += THREADS) tally(row[i]);
            }
            __syncthreads();
            if (threadIdx.x < 256) {
                unsigned sum = 0;
                for (int j = 0; j < kQuantReplicas; ++j) sum += hist[threadIdx.x * kQuantReplicas + ((j + threadIdx.x) & (kQuantReplicas - 1))];
                total[threadIdx.x] = sum;
            }
            __syncthreads();
            if (threadIdx.x < 64) {
                // the first bin b with (bins 0..b) > k, 255 if there is none: lane l owns bins 4l..4l+3
                const unsigned a0 = total[4 * threadIdx.x], a1 = total[4 * threadIdx.x + 1], a2 = total[4 * threadIdx.x + 2], a3 = total[4 * threadIdx.x + 3];
                const unsigned own = a0 + a1 + a2 + a3;
                unsigned incl = own;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const unsigned up = __shfl_up(incl, off, 64);
                    if ((int)threadIdx.x >= off) incl += up;
                }
                const unsigned excl = incl - own;
                if ((excl <= k && k < incl) || (threadIdx.x == 63 && k >= incl)) {
                    unsigned cum = excl, b = 4 * threadIdx.x;
                    if (cum + a0 <= k) {
                        cum += a0; ++b;
                        if (cum + a1 <= k) {
                            cum += a1; ++b;
                            if (cum + a2 <= k) { cum += a2; ++b; }
                        }
                    }
                    sh_prefix = prefix | (b << shift);
                    sh_k = k - cum;
                }
            }
            __syncthreads();
            prefix = sh_prefix;
            k = sh_k;
            mask |= 255u << shift;
            __syncthreads();
        }
        // prefix = bits of the lo-th smallest |x|; the next order statistic is the same value if it repeats, else the smallest larger one
        if (threadIdx.x == 0) {
            sh_cnt = 0;
            sh_min = 0x7FFFFFFFu;
        }
        __syncthreads();
        unsigned cnt = 0, mn = 0x7FFFFFFFu;
        auto look = [&](float v) {
            const unsigned bits = __float_as_uint(v) & 0x7FFFFFFFu;
            cnt += bits <= prefix;
            if (bits > prefix) mn = min(mn, bits);
        };
        if (vec) {
            for (int64_t i = threadIdx.x; i < (inner >> 2); i += THREADS) {
                const float4 v = row4[i];
                look(v.x); look(v.y); look(v.z); look(v.w);
            }
        } else {
            for (int64_t i = threadIdx.x; i < inner; i += THREADS) look(row[i]);
        }
        atomicAdd(&sh_cnt, cnt);
        atomicMin(&sh_min, mn);
        __syncthreads();
        if (threadIdx.x == 0) {
            const float vlo = __uint_as_float(prefix);
            const float vhi = (lo + 1 < (int64_t)sh_cnt || lo + 1 >= inner) ? vlo : __uint_as_float(sh_min);
            out[r] = blend<float>(SONAR_BLEND_LERP, vlo, vhi, frac);
        }
        __syncthreads();
    }
}

// x = copysign(|clamp(x, -lim, lim)|^p, x) per row, lim = limit[row] * mul (StudentT: clamp to the quantile, compress the tails)
__global__ void __launch_bounds__(kBlock) clamp_signpow_rows_kernel(float* x, int64_t rows, int64_t inner, const float* __restrict__ limit,
                                                                     float mul, float p) {
    kernarg_touch_for(x, rows, inner, limit, mul, p);
    const int64_t total = rows * inner;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const float lim = limit[i / inner] * mul;
        const float v = fminf(fmaxf(x[i], -lim), lim);
        const float a = fabsf(v);
        const float m = p == 0.5f ? sqrtf(a) : p == 1.0f ? a : p == 2.0f ? a * a : powf(a, p);
        x[i] = copysignf(m, v);
    }
}

// acc = (first ? 0 : acc) + mul * z^2: chi-square of an integer df built from squared normals (on-device StudentT draws)
struct SqAccOp {
    float* acc;
    const float* z;
    float mul;
    int first;
    template <int V>
    __device__ __forceinline__ void run(int64_t i) const {
        Pack<V> a = load<V>(acc, i);
        const Pack<V> pz = load<V>(z, i);
#pragma unroll
        for (int k = 0; k < V; ++k) a.v[k] = (first ? 0.0f : a.v[k]) + mul * (pz.v[k] * pz.v[k]);
        store<V>(acc, i, a);
    }
};

// y = y*ymul + x*xmul with the (sum, sumsq) partials of the result: the last accumulation of a noise chain feeds the chain's
// normalisation without a separate statistics sweep
__global__ void __launch_bounds__(kBlock) axpby_stats_kernel(float* y, float ymul, const float* __restrict__ x, float xmul, int64_t n,
                                                              double* partials) {
    kernarg_touch_for(y, ymul, x, xmul, n, partials);
    __shared__ double red[2 * kBlock / 64];
    double s = 0.0, q = 0.0;
    const int64_t nv = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nv; i += (int64_t)gridDim.x * kBlock) {
        Pack<4> py = load<4>(y, i * 4);
        const Pack<4> px = load<4>(x, i * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float yy = ymul != 1.0f ? py.v[k] * ymul : py.v[k];
            const float xx = xmul != 1.0f ? px.v[k] * xmul : px.v[k];
            const float v = yy + xx;
            py.v[k] = v;
            s += (double)v;
            q += (double)v * (double)v;
        }
        store<4>(y, i * 4, py);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = nv * 4; i < n; ++i) {
            const float v = (ymul != 1.0f ? y[i] * ymul : y[i]) + (xmul != 1.0f ? x[i] * xmul : x[i]);
            y[i] = v;
            s += (double)v;
            q += (double)v * (double)v;
        }
    write_partial<kBlock>(s, q, partials, red);
}

// RippleFilteredNoise (py/noise.py:1134-1202): x *= s[(i / inner) % len] with a small periodic table s broadcast over the
// tensor; with `follow_sign` the result takes the sign of 1 - s (torch.copysign(result, 1 - s))
__global__ void __launch_bounds__(kBlock) mul_table_kernel(float* x, const float* __restrict__ s, int64_t n, int64_t inner, int64_t len,
                                                            int follow_sign) {
    kernarg_touch_for(x, s, n, inner, len, follow_sign);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const float sv = s[(i / inner) % len];
        float v = x[i] * sv;
        if (follow_sign) v = copysignf(v, 1.0f - sv);
        x[i] = v;
    }
}

struct PowerLawOp {
    float* x;
    float alpha;
    int use_sign;
    template <int V>
    __device__ __forceinline__ void run(int64_t i) const {
        Pack<V> p = load<V>(x, i);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float v = p.v[k];
            const float av = fabsf(v);  // ATen special-cases these exponents (pow_tensor_scalar), keep them exact
            const float mod = alpha == 0.0f ? 1.0f : alpha == 1.0f ? av : alpha == 0.5f ? sqrtf(av) : alpha == 2.0f ? av * av : powf(av, alpha);
            const float base = use_sign ? (v > 0.0f ? 1.0f : v < 0.0f ? -1.0f : v) : v;  // torch.sign keeps 0 and NaN
            p.v[k] = base * mod;
        }
        store<V>(x, i, p);
    }
};

struct MaskMixOp {
    const float *dst, *src, *mask;
    int64_t mask_n;
    float* out;
    template <int V>
    __device__ __forceinline__ void run(int64_t i) const {
        const Pack<V> pd = load<V>(dst, i), ps = load<V>(src, i);
        Pack<V> r;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float m = mask[(i + k) % mask_n];
            // py/noise.py:528-530: noise_dst *= (1 - mask); noise_src *= mask; dst + src
            r.v[k] = pd.v[k] * (1.0f - m) + ps.v[k] * m;
        }
        store<V>(out, i, r);
    }
};

// ------------------------------------------------------------------------------------------------
// Momentum building blocks (py/sonar.py:227-307), evaluated per element in registers.
struct HistState {
    float h;
    bool present;
};

// update_hist: h <- v                       when no history yet
//              h <- hblend(v*ms, h*hs, hr)  otherwise          (py/sonar.py:231-236)
__device__ __forceinline__ void hist_update(const sonar_momentum_cfg& c, HistState& hs, float v) {
    if (!c.update_hist) return;
    if (!hs.present) {
        hs.h = v;
        hs.present = true;
    } else {
        hs.h = blend<float>(c.history_blend, v * c.md_scale, hs.h * c.hist_scale, c.hist_ratio);
    }
}

// init_hist_d for SAMPLE / SAMPLE_NORM (py/sonar.py:184-191)
__device__ __forceinline__ void hist_init(const sonar_momentum_cfg& c, HistState& hs, float x, float den, float sigma) {
    if (hs.present || c.init_kind == SONAR_INIT_NONE) return;
    const float src = c.mode == SONAR_MODE_DENOISED ? den : x;
    hs.h = c.init_kind == SONAR_INIT_SAMPLE_NORM ? src / sigma : src;
    hs.present = true;
}

// get_momentum_denoised (py/sonar.py:262-283): returns den_m, updates history with den/sigma
__device__ __forceinline__ float momentum_denoised(const sonar_momentum_cfg& c, HistState& hs, bool h_usable, float x,
                                                   float den, float sigma) {
    float den_m = den;
    if (c.mode == SONAR_MODE_DENOISED && hs.present && h_usable && c.momentum != 1.0f)
        den_m = blend<float>(c.momentum_blend, hs.h * sigma, den, c.momentum);
    hist_init(c, hs, x, den, sigma);
    hist_update(c, hs, den / sigma);
    return c.use_momentum ? den_m : den;
}

// get_momentum_d (py/sonar.py:285-307) for a given d; `early_out` = momentum==1 || DENOISED mode
__device__ __forceinline__ float momentum_d(const sonar_momentum_cfg& c, HistState& hs, bool early_out, float x,
                                            float den, float sigma, float d) {
    if (early_out) return d;
    const float md = hs.present ? blend<float>(c.momentum_blend, hs.h, d, c.momentum) : d;
    hist_init(c, hs, x, den, sigma);
    hist_update(c, hs, c.mode == SONAR_MODE_NEW ? d : md);
    return c.use_momentum ? md : d;
}

// A noise tensor whose global normalisation (py/utils.py:100-105) is still pending: the decision -- taken on the device from the
// tensor's (sum, sumsq) partials by sonar_norm_decision_f32 -- rides along and the step kernel that consumes the noise applies it on
// the fly, with scale_noise_kernel's own operation sequence (subtract, divide, multiply: same bits), instead of a separate read +
// write of the tensor.
// The quotient v / std without the ten-instruction IEEE division sequence per element: std is the same for every element, so its
// correctly rounded reciprocal is taken once (sonar_norm_decision_f32) and one residual step q' = q + (v - std q) / std on q = v / std
// gives the correctly rounded quotient (Markstein: exact remainder by fma, correctly rounded reciprocal) for every v in the normal
// range -- the same bits as scale_noise_kernel's `v / std`, at three instructions.
struct PendingNorm {  // one thread's copy of the decision (read once per item, not once per element)
    float mean, stdv, inv_std, factor;
    bool sub, div, mul;
    __device__ __forceinline__ explicit PendingNorm(const sonar_noise_norm* __restrict__ nn) {
        if (nn) {
            mean = nn->mean; stdv = nn->stdv; inv_std = nn->inv_std; factor = nn->factor;
            sub = nn->do_sub != 0; div = nn->do_div != 0; mul = factor != 1.0f;
        } else {
            mean = 0.0f; stdv = inv_std = factor = 1.0f;
            sub = div = mul = false;
        }
    }
    __device__ __forceinline__ float operator()(float v) const {
        if (sub) v = v - mean;
        if (div) {
            const float q = v * inv_std;
            v = __builtin_fmaf(__builtin_fmaf(-stdv, q, v), inv_std, q);
        }
        if (mul) v = v * factor;
        return v;
    }
};

__global__ void __launch_bounds__(kBlock) norm_decision_kernel(const double* __restrict__ partials, int64_t npart, int64_t n_total, float factor,
                                                                float thr_sd, sonar_noise_norm* out) {
    kernarg_touch_for(partials, npart, n_total, factor, thr_sd, out);
    __shared__ double red[2 * kBlock / 64];
    __shared__ NormDecision sh;
    const NormDecision d = decide_norm<kBlock>(partials, npart, n_total, thr_sd, red, &sh);
    if (threadIdx.x == 0) {
        out->mean = d.mean;
        out->stdv = d.stdv;
        out->inv_std = 1.0f / d.stdv;  // correctly rounded (IEEE division), once
        out->factor = factor;
        out->do_sub = d.do_sub;
        out->do_div = d.do_div;
    }
}

struct ApplyNormOp {
    float* x;
    const sonar_noise_norm* nn;
    template <int V>
    __device__ __forceinline__ void run(int64_t i) const {
        Pack<V> p = load<V>(x, i);
        const PendingNorm pending(nn);
#pragma unroll
        for (int k = 0; k < V; ++k) p.v[k] = pending(p.v[k]);
        store<V>(x, i, p);
    }
};

// NT: both outputs stored with the non-temporal hint -- for a streaming kernel with three inputs and two outputs the lines are better off
// not sitting in the L2 (512 SDXL latents: 120 -> 115 us; at 64 latents and below the write-back cache wins by a little)
template <bool NT>
struct EulerOpT {
    const float *x, *den, *h_in;
    float *x_out, *h_out;
    const float* noise;
    const sonar_noise_norm* noise_norm;
    float noise_scale, sigma, dt;
    sonar_momentum_cfg c;
    template <int V>
    __device__ __forceinline__ void run(int64_t i) const {
        const Pack<V> px = load<V>(x, i), pd = load<V>(den, i);
        Pack<V> ph, pn, rx, rh;
        if (h_in) ph = load<V>(h_in, i);
        if (noise) pn = load<V>(noise, i);
        const bool early = c.momentum == 1.0f || c.mode == SONAR_MODE_DENOISED;
        const PendingNorm pending(noise_norm);
        bool present = false;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            HistState hs{h_in ? ph.v[k] : 0.0f, h_in != nullptr};
            const float den_m = momentum_denoised(c, hs, !c.h_in_fresh, px.v[k], pd.v[k], sigma);
            const float d = (px.v[k] - den_m) / sigma;  // to_d
            const float md = momentum_d(c, hs, early, px.v[k], den_m, sigma, d);
            float xn = md * dt + px.v[k];
            if (noise) xn = xn + pending(pn.v[k]) * noise_scale;
            rx.v[k] = xn;
            rh.v[k] = hs.h;
            present = hs.present;
        }
        store<V, NT>(x_out, i, rx);
        if (present && h_out) store<V, NT>(h_out, i, rh);
    }
};
using EulerOp = EulerOpT<false>;

struct Dpmpp1Op {
    const float *x, *den, *h_in;
    float *x2_out, *md1_out, *h_out;
    const float* noise;
    const sonar_noise_norm* noise_norm;
    float noise_scale, sigma, expm1_a, ratio_a;
    int adj_is_one;
    sonar_momentum_cfg c;
    template <int V>
    __device__ __forceinline__ void run(int64_t i) const {
        const Pack<V> px = load<V>(x, i), pd = load<V>(den, i);
        Pack<V> ph, pn, rx, rm, rh;
        if (h_in) ph = load<V>(h_in, i);
        if (noise) pn = load<V>(noise, i);
        const bool early = adj_is_one || c.mode == SONAR_MODE_DENOISED;
        const PendingNorm pending(noise_norm);
        bool present = false;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            HistState hs{h_in ? ph.v[k] : 0.0f, h_in != nullptr};
            const float md1 = momentum_denoised(c, hs, !c.h_in_fresh, px.v[k], pd.v[k], sigma);
            const float diff2 = expm1_a * md1;
            const float m_d = momentum_d(c, hs, early, px.v[k], md1, sigma, diff2);
            float x2 = ratio_a * px.v[k] - m_d;
            if (noise) x2 = x2 + pending(pn.v[k]) * noise_scale;
            rx.v[k] = x2;
            rm.v[k] = md1;
            rh.v[k] = hs.h;
            present = hs.present;
        }
        store<V>(x2_out, i, rx);
        store<V>(md1_out, i, rm);
        if (present && h_out) store<V>(h_out, i, rh);
    }
};

struct Dpmpp2Op {
    const float *x, *den2, *md1, *h_in;
    float *x_out, *dd_out, *h_out;
    const float* noise;
    const sonar_noise_norm* noise_norm;
    float noise_scale, sigma_s, expm1_b, ratio_b, fac;
    int adj_is_one;
    sonar_momentum_cfg c;
    template <int V>
    __device__ __forceinline__ void run(int64_t i) const {
        const Pack<V> px = load<V>(x, i), pd = load<V>(den2, i), pm = load<V>(md1, i);
        Pack<V> ph, pn, rx, rd, rh;
        if (h_in) ph = load<V>(h_in, i);
        if (noise) pn = load<V>(noise, i);
        const bool early = adj_is_one || c.mode == SONAR_MODE_DENOISED;
        const float one_minus_fac = 1.0f - fac;
        const PendingNorm pending(noise_norm);
        bool present = false;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            HistState hs{h_in ? ph.v[k] : 0.0f, h_in != nullptr};
            const float md2 = momentum_denoised(c, hs, true, px.v[k], pd.v[k], sigma_s);
            const float dd = one_minus_fac * pm.v[k] + fac * md2;
            const float diff1 = expm1_b * dd;
            const float m_d = momentum_d(c, hs, early, px.v[k], md2, sigma_s, diff1);
            float xn = ratio_b * px.v[k] - m_d;
            if (noise) xn = xn + pending(pn.v[k]) * noise_scale;
            rx.v[k] = xn;
            rd.v[k] = dd;
            rh.v[k] = hs.h;
            present = hs.present;
        }
        store<V>(x_out, i, rx);
        if (dd_out) store<V>(dd_out, i, rd);
        if (present && h_out) store<V>(h_out, i, rh);
    }
};

// Whether a history exists after the step (host-side mirror of the device logic).
static int hist_present_after(const sonar_momentum_cfg& c, bool h_in, bool second_update_possible) {
    if (h_in) return 1;
    if (c.init_kind != SONAR_INIT_NONE) return 1;
    if (c.update_hist) return 1;
    (void)second_update_possible;
    return 0;
}

struct CastOp {
    const float* in;
    double* out;
    template <int V>
    __device__ __forceinline__ void run(int64_t i) const {
        const Pack<V> p = load<V>(in, i);
        if constexpr (V == 4) {
            *reinterpret_cast<double2*>(out + i) = make_double2((double)p.v[0], (double)p.v[1]);
            *reinterpret_cast<double2*>(out + i + 2) = make_double2((double)p.v[2], (double)p.v[3]);
        } else {
            out[i] = (double)p.v[0];
        }
    }
};

// ---- ModulatedNoise spectral_signum (py/noise.py:938-1015) --------------------------------------------------------------------
// DFT along the middle axis of complex z[outer][C][inner] (the channel axis of fftn over dims (-3) or (-3, -2, -1)); out of place.
// inverse: e^{+...}, no 1/C (the caller folds it into the mask gain).  real_in: the input is a real float array (imaginary part 0);
// real_out: only the real part is written, to a float array.  C <= 64: every output sums C terms read through L1 / L2.
__global__ void __launch_bounds__(kBlock) cdft_mid_kernel(const float* __restrict__ zin, float* __restrict__ zout, int64_t outer, int C,
                                                          int64_t inner, int inverse, int real_in, int real_out) {
    kernarg_touch_for(zin, zout, outer, C, inner, inverse, real_in, real_out);
    const int64_t total = outer * C * inner;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
        const int64_t i = t % inner;
        const int k = (int)((t / inner) % C);
        const int64_t o = t / (inner * C);
        float re = 0.0f, im = 0.0f;
        for (int c = 0; c < C; ++c) {
            const int64_t at = (o * C + c) * inner + i;
            float xr, xi;
            if (real_in) {
                xr = zin[at];
                xi = 0.0f;
            } else {
                const float2 v = reinterpret_cast<const float2*>(zin)[at];
                xr = v.x;
                xi = v.y;
            }
            float sn, cs;
            sincospif(2.0f * (float)((c * k) % C) / (float)C, &sn, &cs);
            if (!inverse) sn = -sn;
            re += xr * cs - xi * sn;
            im += xr * sn + xi * cs;
        }
        if (real_out) zout[t] = re;
        else reinterpret_cast<float2*>(zout)[t] = make_float2(re, im);
    }
}

// la = log(sqrt(re^2 + im^2)) of a spectrum z[planes][H][Wz] (Wz = W/2 + 1 columns of an rfft2 half-spectrum, or Wz = W for a full
// one) and the multiset of |la| over the FULL spectrum: full[planes][H][W]; the columns an rfft2 drops come from their Hermitian
// partners ((C - c) % C, (H - ky) % H, W - kx) -- C > 1 when a channel DFT was applied on top (fftn over (-3, -2, -1)).
__global__ void __launch_bounds__(kBlock) spectral_logamp_kernel(const float2* __restrict__ z, float* __restrict__ la, float* __restrict__ full,
                                                                 int64_t planes, int C, int H, int W, int Wz) {
    kernarg_touch_for(z, la, full, planes, C, H, W, Wz);
    const int64_t nz = planes * H * Wz;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < nz; t += (int64_t)gridDim.x * kBlock) {
        const float2 v = z[t];
        la[t] = logf(sqrtf(v.x * v.x + v.y * v.y));
    }
    (void)full;
}
__global__ void __launch_bounds__(kBlock) spectral_full_kernel(const float* __restrict__ la, float* __restrict__ full, int64_t planes, int C,
                                                               int H, int W, int Wz) {
    kernarg_touch_for(la, full, planes, C, H, W, Wz);
    const int64_t nf = planes * H * W;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < nf; t += (int64_t)gridDim.x * kBlock) {
        const int kx = (int)(t % W), ky = (int)((t / W) % H);
        const int64_t p = t / ((int64_t)W * H);
        int64_t src;
        if (kx < Wz) {
            src = (p * H + ky) * Wz + kx;
        } else {
            const int64_t b = p / C;
            const int c = (int)(p % C);
            src = ((b * C + (C - c) % C) * H + (H - ky) % H) * Wz + (W - kx);
        }
        full[t] = fabsf(la[src]);
    }
}

// z *= gain * (mult_low * mult_high) ^ intensity with, per bin (:975-1003),
//   mult_high = la > q_high ? 1 - min((la - q_high) / (q_max - q_high), 0.5) : 1,  mult_low = la < q_low ? 1 + min(1 - la / q_low, 0.5) : 1
// q[nq][3] = (low, high, max) quantiles of |la| per SAMPLE; the reference expands that [B] vector against [B, C, H, W] as [B, 1, 1],
// so the row used for a bin of plane p is 0 when nq == 1 and the bin's CHANNEL index p % C when nq == C.
// channel_sym (a channel DFT was applied and nq == C): the Hermitian partner of a bin sits at channel frequency (C - c) % C and meets
// ANOTHER quantile row, so the reference's mask is not Hermitian and it keeps the real part of a complex inverse; that real part equals
// the inverse of the spectrum times the symmetrised mask (m(k) + m(-k)) / 2 (|z| is symmetric), which a half-spectrum inverse can take.
__device__ __forceinline__ float signum_mult(float a, const float* __restrict__ qq, float intensity) {
    const float ql = qq[0], qh = qq[1], qm = qq[2];
    const float hi = a > qh ? 1.0f - fminf((a - qh) / (qm - qh), 0.5f) : 1.0f;
    const float lo = a < ql ? 1.0f + fminf(1.0f - a / ql, 0.5f) : 1.0f;
    return powf(lo * hi, intensity);
}
__global__ void __launch_bounds__(kBlock) spectral_signum_mask_kernel(float2* z, const float* __restrict__ la, const float* __restrict__ q, int nq,
                                                                      int64_t planes, int C, int64_t plane_elems, float intensity, float gain,
                                                                      int channel_sym) {
    kernarg_touch_for(z, la, q, nq, planes, C, plane_elems, intensity, gain, channel_sym);
    const int64_t n = planes * plane_elems;
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < n; t += (int64_t)gridDim.x * kBlock) {
        const int64_t p = t / plane_elems;
        const int c = (int)(p % C);
        const float a = la[t];
        float m = signum_mult(a, q + 3 * (nq == 1 ? 0 : c), intensity);
        if (channel_sym && nq != 1) m = 0.5f * (m + signum_mult(a, q + 3 * ((C - c) % C), intensity));
        m *= gain;
        float2 v = z[t];
        v.x *= m;
        v.y *= m;
        z[t] = v;
    }
}

}  // namespace sonar

using namespace sonar;

// ================================================================================================
extern "C" int sonar_stats_f32(const float* x, int64_t n, double* partials, void* stream) {
    SONAR_REQUIRE(x && partials && n >= 0, SONAR_ERR_ARG, "sonar_stats_f32: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (aligned16(x)) {
        const int g = (int)std::min<int64_t>(kNPart, grid_for(n / 4 + 1, kBlock * 2));
        hipLaunchKernelGGL((stats_kernel<4>), dim3(g), dim3(kBlock), 0, st, x, n, partials);
    } else {
        const int g = (int)std::min<int64_t>(kNPart, grid_for(n, kBlock * 4));
        hipLaunchKernelGGL((stats_kernel<1>), dim3(g), dim3(kBlock), 0, st, x, n, partials);
    }
    return check_launch("sonar_stats_f32");
}

extern "C" int sonar_stats_finalize(const double* partials, int64_t npart, int64_t n, double* out3, void* stream) {
    SONAR_REQUIRE(partials && out3 && npart > 0, SONAR_ERR_ARG, "sonar_stats_finalize: bad argument");
    hipLaunchKernelGGL(stats_finalize_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, partials, npart, n, out3);
    return check_launch("sonar_stats_finalize");
}

static int scale_noise_launch(float* x, int64_t n, float factor, int normalized, float threshold_std_devs, const double* partials,
                              int64_t npart, int64_t n_total, double* out_partials, void* stream) {
    SONAR_REQUIRE(x && n >= 0, SONAR_ERR_ARG, "sonar_scale_noise_f32: bad argument");
    SONAR_REQUIRE(!normalized || (partials && npart > 0 && n_total > 0), SONAR_ERR_ARG,
                  "sonar_scale_noise_f32: normalized=1 needs partials");
    SONAR_REQUIRE(!out_partials || (normalized && out_partials != partials), SONAR_ERR_ARG,
                  "sonar_scale_noise_stats_f32: result statistics need normalized=1 and a separate buffer");
    if (n == 0 || (!normalized && factor == 1.0f)) return SONAR_OK;
    hipStream_t st = (hipStream_t)stream;
    if (aligned16(x) && nt_stores_host(n)) {
        hipLaunchKernelGGL((scale_noise_kernel<4, true>), dim3(grid_for(n / 4 + 1, kBlock * 2)), dim3(kBlock), 0, st, x, n,
                           factor, normalized, threshold_std_devs, partials, npart, n_total, out_partials);
    } else if (aligned16(x)) {
        hipLaunchKernelGGL((scale_noise_kernel<4>), dim3(grid_for(n / 4 + 1, kBlock * 2)), dim3(kBlock), 0, st, x, n,
                           factor, normalized, threshold_std_devs, partials, npart, n_total, out_partials);
    } else {
        hipLaunchKernelGGL((scale_noise_kernel<1>), dim3(grid_for(n, kBlock * 4)), dim3(kBlock), 0, st, x, n, factor,
                           normalized, threshold_std_devs, partials, npart, n_total, out_partials);
    }
    return check_launch("sonar_scale_noise_f32");
}

extern "C" int sonar_scale_noise_f32(float* x, int64_t n, float factor, int normalized, float threshold_std_devs,
                                     const double* partials, int64_t npart, int64_t n_total, void* stream) {
    return scale_noise_launch(x, n, factor, normalized, threshold_std_devs, partials, npart, n_total, nullptr, stream);
}

extern "C" int sonar_scale_noise_stats_f32(float* x, int64_t n, float factor, float threshold_std_devs, const double* partials,
                                           int64_t npart, int64_t n_total, double* out_partials, void* stream) {
    SONAR_REQUIRE(out_partials, SONAR_ERR_ARG, "sonar_scale_noise_stats_f32: bad argument");
    return scale_noise_launch(x, n, factor, 1, threshold_std_devs, partials, npart, n_total, out_partials, stream);
}

extern "C" int sonar_std_scale_f32(float* x, int64_t n, float mul, const double* partials, int64_t npart, int64_t n_total,
                                   void* stream) {
    SONAR_REQUIRE(x && partials && n >= 0 && npart > 0 && n_total >= 1, SONAR_ERR_ARG, "sonar_std_scale_f32: bad argument");  // one element: its std is NaN, as in torch
    if (n == 0) return SONAR_OK;
    hipStream_t st = (hipStream_t)stream;
    if (aligned16(x))
        hipLaunchKernelGGL((scale_noise_kernel<4>), dim3(grid_for(n / 4 + 1, kBlock * 2)), dim3(kBlock), 0, st, x, n, mul, 2, 0.0f,
                           partials, npart, n_total, (double*)nullptr);
    else
        hipLaunchKernelGGL((scale_noise_kernel<1>), dim3(grid_for(n, kBlock * 4)), dim3(kBlock), 0, st, x, n, mul, 2, 0.0f, partials,
                           npart, n_total, (double*)nullptr);
    return check_launch("sonar_std_scale_f32");
}

extern "C" int sonar_scale_noise_rows_f32(float* x, int64_t rows, int64_t inner, float factor, void* stream) {
    SONAR_REQUIRE(x && rows >= 0 && inner > 0, SONAR_ERR_ARG, "sonar_scale_noise_rows_f32: bad argument");
    if (rows == 0) return SONAR_OK;
    SONAR_ROW_LAUNCH(scale_rows_kernel, inner, rows, 0, (hipStream_t)stream, x, rows, inner,
                       factor);
    return check_launch("sonar_scale_noise_rows_f32");
}

extern "C" int sonar_minmax_rows_f32(const float* x, int64_t rows, int64_t inner, float* out_min, float* out_max,
                                     void* stream) {
    SONAR_REQUIRE(x && out_min && out_max && rows >= 0 && inner > 0, SONAR_ERR_ARG, "sonar_minmax_rows_f32: bad argument");
    if (rows == 0) return SONAR_OK;
    SONAR_ROW_LAUNCH(minmax_rows_kernel, inner, rows, 0, (hipStream_t)stream, x, rows, inner,
                       out_min, out_max);
    return check_launch("sonar_minmax_rows_f32");
}

extern "C" int sonar_blend_f32(int mode, const float* a, const float* b, float t, float* out, int64_t n, void* stream) {
    SONAR_REQUIRE(a && b && out && n >= 0 && mode >= 0 && mode <= 2, SONAR_ERR_ARG, "sonar_blend_f32: bad argument");
    return launch_ew(BlendOp{mode, a, b, t, out}, n, aligned16(a) && aligned16(b) && aligned16(out),
                     (hipStream_t)stream, "sonar_blend_f32");
}

extern "C" int sonar_blend_tensor_f32(int mode, const float* a, const float* b, const float* t, int64_t tn, float* out,
                                      int64_t n, void* stream) {
    SONAR_REQUIRE(a && b && t && out && n >= 0 && tn > 0 && mode >= 0 && mode <= 2, SONAR_ERR_ARG,
                  "sonar_blend_tensor_f32: bad argument");
    return launch_ew(BlendTensorOp{mode, a, b, t, tn, out}, n, aligned16(a) && aligned16(b) && aligned16(out),
                     (hipStream_t)stream, "sonar_blend_tensor_f32");
}

extern "C" int sonar_axpby_stats_f32(float* y, float ymul, const float* x, float xmul, int64_t n, double* partials, void* stream) {
    SONAR_REQUIRE(x && y && partials && n >= 0 && aligned16(x) && aligned16(y), SONAR_ERR_ARG,
                  "sonar_axpby_stats_f32: bad argument (16-byte aligned buffers required)");
    const int g = (int)std::min<int64_t>(kNPart, std::max<int64_t>(1, grid_for(n / 4 + 1, kBlock)));
    hipLaunchKernelGGL(axpby_stats_kernel, dim3(g), dim3(kBlock), 0, (hipStream_t)stream, y, ymul, x, xmul, n, partials);
    return check_launch("sonar_axpby_stats_f32");
}

extern "C" int sonar_axpby_f32(float* y, float ymul, const float* x, float xmul, int64_t n, void* stream) {
    SONAR_REQUIRE(x && y && n >= 0, SONAR_ERR_ARG, "sonar_axpby_f32: bad argument");
    return launch_ew(AxpbyOp{y, ymul, x, xmul}, n, aligned16(x) && aligned16(y), (hipStream_t)stream, "sonar_axpby_f32");
}

extern "C" int sonar_affine_f32(float* x, float sub, float mul, float add, int64_t n, void* stream) {
    SONAR_REQUIRE(x && n >= 0, SONAR_ERR_ARG, "sonar_affine_f32: bad argument");
    return launch_ew(AffineOp{x, sub, mul, add}, n, aligned16(x), (hipStream_t)stream, "sonar_affine_f32");
}

extern "C" int sonar_scalar_op_f32(int op, const float* a, const float* b, float s, float* out, int64_t n, void* stream) {
    SONAR_REQUIRE(a && out && n >= 0 && op >= 0 && op <= 2 && (op != 2 || b), SONAR_ERR_ARG, "sonar_scalar_op_f32: bad argument");
    return launch_ew(ScalarOp{op, a, b, s, out}, n, aligned16(a) && aligned16(out) && (!b || aligned16(b)), (hipStream_t)stream,
                     "sonar_scalar_op_f32");
}

extern "C" int sonar_rowstats_f32(const float* x, int64_t rows, int64_t inner, float* mean, float* stdv, void* stream) {
    SONAR_REQUIRE(x && mean && stdv && rows >= 0 && inner > 0, SONAR_ERR_ARG, "sonar_rowstats_f32: bad argument");
    if (rows == 0) return SONAR_OK;
    SONAR_ROW_LAUNCH(rowstats_kernel, inner, rows, 0, (hipStream_t)stream, x, rows, inner, mean, stdv);
    return check_launch("sonar_rowstats_f32");
}

extern "C" int sonar_row_affine_f32(int op, const float* x, int64_t rows, int64_t inner, const float* a, const float* b,
                                    float* out, void* stream) {
    SONAR_REQUIRE(x && a && b && out && rows >= 0 && inner > 0 && (op == 0 || op == 1), SONAR_ERR_ARG,
                  "sonar_row_affine_f32: bad argument");
    if (rows == 0) return SONAR_OK;
    hipLaunchKernelGGL(row_affine_kernel, dim3(grid_for(rows * inner, kBlock * 2)), dim3(kBlock), 0, (hipStream_t)stream, op, x,
                       rows, inner, a, b, out);
    return check_launch("sonar_row_affine_f32");
}

extern "C" int sonar_minmax_rescale_f32(const float* x, int64_t rows, int64_t inner, const float* lo, const float* hi, float eps,
                                        double target_min, double target_max, float* out, void* stream) {
    SONAR_REQUIRE(x && lo && hi && out && rows >= 0 && inner > 0, SONAR_ERR_ARG, "sonar_minmax_rescale_f32: bad argument");
    if (rows == 0) return SONAR_OK;
    hipLaunchKernelGGL(minmax_rescale_kernel, dim3(grid_for(rows * inner, kBlock * 2)), dim3(kBlock), 0, (hipStream_t)stream, x, rows,
                       inner, lo, hi, eps, (float)target_min, (float)target_max, (float)(target_max - target_min), out);
    return check_launch("sonar_minmax_rescale_f32");
}

extern "C" int sonar_signed_rescale_f32(const float* x, int64_t rows, int64_t inner, double min_neg, double max_neg, double min_pos, double max_pos,
                                        float eps, float* stats_ws, float* out, void* stream) {
    SONAR_REQUIRE(x && out && stats_ws && rows >= 0 && inner > 0 && (reinterpret_cast<uintptr_t>(stats_ws) & 15u) == 0, SONAR_ERR_ARG,
                  "sonar_signed_rescale_f32: bad argument (a 16-byte aligned workspace of 4 floats per row is required)");
    if (rows == 0) return SONAR_OK;
    // py/utils.py:482-483
    const int skip_pos = max_pos <= 0.0 || min_pos >= max_pos, skip_neg = min_neg >= 0.0 || min_neg >= max_neg;
    float4* st = reinterpret_cast<float4*>(stats_ws);
    hipLaunchKernelGGL(signed_minmax_rows_kernel, dim3((int)std::min<int64_t>(rows, 4096)), dim3(kBlock), 0, (hipStream_t)stream, x, rows, inner, st);
    hipLaunchKernelGGL(signed_rescale_kernel, dim3(grid_for(rows * inner, kBlock * 2)), dim3(kBlock), 0, (hipStream_t)stream, x, rows, inner, st,
                       min_neg, max_neg, min_pos, max_pos, skip_neg, skip_pos, eps, out);
    return check_launch("sonar_signed_rescale_f32");
}

extern "C" int sonar_powerlaw_f32(float* x, float alpha, int use_sign, int64_t n, void* stream) {
    SONAR_REQUIRE(x && n >= 0, SONAR_ERR_ARG, "sonar_powerlaw_f32: bad argument");
    return launch_ew(PowerLawOp{x, alpha, use_sign}, n, aligned16(x), (hipStream_t)stream, "sonar_powerlaw_f32");
}

extern "C" int sonar_amax_mid_f32(const float* x, int64_t outer, int64_t mid, int64_t inner, int use_abs, float* peak,
                                  void* stream) {
    SONAR_REQUIRE(x && peak && outer >= 0 && mid > 0 && inner > 0, SONAR_ERR_ARG, "sonar_amax_mid_f32: bad argument");
    if (outer == 0) return SONAR_OK;
    if (inner == 1)
        SONAR_ROW_LAUNCH(amax_row_kernel, mid, outer, 0, (hipStream_t)stream, x, outer, mid, use_abs, peak);
    else
        hipLaunchKernelGGL(amax_mid_kernel, dim3(grid_for(outer * inner, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, x, outer,
                           mid, inner, use_abs, peak);
    return check_launch("sonar_amax_mid_f32");
}

extern "C" int sonar_div_mid_f32(float* x, int64_t outer, int64_t mid, int64_t inner, const float* d, void* stream) {
    SONAR_REQUIRE(x && d && outer >= 0 && mid > 0 && inner > 0, SONAR_ERR_ARG, "sonar_div_mid_f32: bad argument");
    if (outer == 0) return SONAR_OK;
    hipLaunchKernelGGL(div_mid_kernel, dim3(grid_for(outer * mid * inner, kBlock * 2)), dim3(kBlock), 0, (hipStream_t)stream, x,
                       outer, mid, inner, d);
    return check_launch("sonar_div_mid_f32");
}

extern "C" int sonar_laplace_add_f32(float* x, const float* u, float div_fac, float loc, float scale, int64_t n, void* stream) {
    SONAR_REQUIRE(x && u && n >= 0 && div_fac != 0.0f, SONAR_ERR_ARG, "sonar_laplace_add_f32: bad argument");
    return launch_ew(LaplaceAddOp{x, u, div_fac, loc, scale}, n, aligned16(x) && aligned16(u), (hipStream_t)stream, "sonar_laplace_add_f32");
}

extern "C" int sonar_studentt_f32(float* x, const float* gamma, float loc, float scale, float df, int64_t n, void* stream) {
    SONAR_REQUIRE(x && gamma && n >= 0 && df > 0.0f, SONAR_ERR_ARG, "sonar_studentt_f32: bad argument");
    return launch_ew(StudentTOp{x, gamma, loc, scale, df}, n, aligned16(x) && aligned16(gamma), (hipStream_t)stream, "sonar_studentt_f32");
}

extern "C" int sonar_abs_quantile_rows_f32(const float* x, int64_t rows, int64_t inner, int64_t rank_lo, float rank_frac, float* out,
                                           void* stream) {
    SONAR_REQUIRE(x && out && rows >= 0 && inner > 0 && rank_lo >= 0 && rank_lo < inner && rank_frac >= 0.0f && rank_frac <= 1.0f,
                  SONAR_ERR_ARG, "sonar_abs_quantile_rows_f32: bad argument");
    if (rows == 0) return SONAR_OK;
    SONAR_ROW_LAUNCH(abs_quantile_rows_kernel, inner, rows, 0, (hipStream_t)stream, x, rows, inner, rank_lo,
                       rank_frac, out);
    return check_launch("sonar_abs_quantile_rows_f32");
}

extern "C" int sonar_clamp_signpow_rows_f32(float* x, int64_t rows, int64_t inner, const float* limit, float mul, float p, void* stream) {
    SONAR_REQUIRE(x && limit && rows >= 0 && inner > 0, SONAR_ERR_ARG, "sonar_clamp_signpow_rows_f32: bad argument");
    if (rows == 0) return SONAR_OK;
    hipLaunchKernelGGL(clamp_signpow_rows_kernel, dim3(grid_for(rows * inner, kBlock * 4)), dim3(kBlock), 0, (hipStream_t)stream, x, rows,
                       inner, limit, mul, p);
    return check_launch("sonar_clamp_signpow_rows_f32");
}

extern "C" int sonar_sq_acc_f32(float* acc, const float* z, float mul, int first, int64_t n, void* stream) {
    SONAR_REQUIRE(acc && z && n >= 0, SONAR_ERR_ARG, "sonar_sq_acc_f32: bad argument");
    return launch_ew(SqAccOp{acc, z, mul, first}, n, aligned16(acc) && aligned16(z), (hipStream_t)stream, "sonar_sq_acc_f32");
}

extern "C" int sonar_mul_table_f32(float* x, const float* table, int64_t n, int64_t inner, int64_t len, int follow_sign, void* stream) {
    SONAR_REQUIRE(x && table && n >= 0 && inner > 0 && len > 0, SONAR_ERR_ARG, "sonar_mul_table_f32: bad argument");
    if (n == 0) return SONAR_OK;
    hipLaunchKernelGGL(mul_table_kernel, dim3(grid_for(n, kBlock * 4)), dim3(kBlock), 0, (hipStream_t)stream, x, table, n, inner, len,
                       follow_sign);
    return check_launch("sonar_mul_table_f32");
}

extern "C" int sonar_std_mid_f32(const float* x, int64_t outer, int64_t mid, int64_t inner, float* stdv, void* stream) {
    SONAR_REQUIRE(x && stdv && outer >= 0 && mid > 0 && inner > 0, SONAR_ERR_ARG, "sonar_std_mid_f32: bad argument");
    if (outer == 0) return SONAR_OK;
    hipLaunchKernelGGL(std_mid_kernel, dim3(grid_for(outer * inner, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, x, outer, mid, inner,
                       stdv);
    return check_launch("sonar_std_mid_f32");
}

extern "C" int sonar_bcast_gain_f32(const float* x, const float* stdv, int64_t outer, int64_t mid, int64_t inner, int bcast,
                                    float abs_strength, float k, float* out, double* partials, void* stream) {
    SONAR_REQUIRE(x && stdv && (out || partials) && outer >= 0 && mid > 0 && inner > 0 && bcast >= 0 && bcast <= 2, SONAR_ERR_ARG,
                  "sonar_bcast_gain_f32: bad argument");
    // bcast 0: std[outer] (dims -3,-2,-1)   1: std[outer][mid] (dims -2,-1)   2: std[outer][inner] (dim -3)
    const int64_t so = bcast == 0 ? 1 : bcast == 1 ? mid : inner, sm = bcast == 1 ? 1 : 0, si = bcast == 2 ? 1 : 0;
    const int64_t n = outer * mid * inner;
    const int grid = (int)std::min<int64_t>(std::max<int64_t>(grid_for(n, kBlock * 4), 1), kNPart);
    hipLaunchKernelGGL(bcast_gain_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, x, stdv, outer, mid, inner, so, sm, si,
                       abs_strength, k, out, partials);
    return check_launch("sonar_bcast_gain_f32");
}

extern "C" int sonar_ratio_mix_f32(const float* a, float a_mul, const float* x, float x_mul, const double* num_partials,
                                   double num_mul, const double* den_partials, float* out, int64_t n, void* stream) {
    SONAR_REQUIRE(a && x && num_partials && den_partials && out && n >= 0, SONAR_ERR_ARG, "sonar_ratio_mix_f32: bad argument");
    if (n == 0) return SONAR_OK;
    hipLaunchKernelGGL(ratio_mix_kernel, dim3(grid_for(n, kBlock * 4)), dim3(kBlock), 0, (hipStream_t)stream, a, a_mul, x, x_mul,
                       num_partials, num_mul, den_partials, out, n);
    return check_launch("sonar_ratio_mix_f32");
}

extern "C" int sonar_cdft_mid_f32(const float* z_in, float* z_out, int64_t outer, int64_t C, int64_t inner, int inverse, int real_in,
                                  int real_out, void* stream) {
    SONAR_REQUIRE(z_in && z_out && z_in != z_out && outer >= 0 && C >= 1 && C <= 64 && inner > 0, SONAR_ERR_ARG,
                  "sonar_cdft_mid_f32: bad argument (1..64 channels, out of place)");
    if (outer == 0) return SONAR_OK;
    hipLaunchKernelGGL(cdft_mid_kernel, dim3(grid_for(outer * C * inner, kBlock * 2)), dim3(kBlock), 0, (hipStream_t)stream, z_in, z_out, outer,
                       (int)C, inner, inverse, real_in, real_out);
    return check_launch("sonar_cdft_mid_f32");
}

extern "C" int sonar_spectral_logamp_f32(const float* z, float* la, float* full, int64_t planes, int64_t C, int64_t H, int64_t W,
                                         int64_t Wz, void* stream) {
    SONAR_REQUIRE(z && la && full && planes >= 0 && C >= 1 && planes % C == 0 && H > 0 && W > 0 && (Wz == W || Wz == W / 2 + 1), SONAR_ERR_ARG,
                  "sonar_spectral_logamp_f32: bad argument");
    if (planes == 0) return SONAR_OK;
    hipLaunchKernelGGL(spectral_logamp_kernel, dim3(grid_for(planes * H * Wz, kBlock * 2)), dim3(kBlock), 0, (hipStream_t)stream,
                       reinterpret_cast<const float2*>(z), la, full, planes, (int)C, (int)H, (int)W, (int)Wz);
    hipLaunchKernelGGL(spectral_full_kernel, dim3(grid_for(planes * H * W, kBlock * 2)), dim3(kBlock), 0, (hipStream_t)stream, la, full, planes,
                       (int)C, (int)H, (int)W, (int)Wz);
    return check_launch("sonar_spectral_logamp_f32");
}

extern "C" int sonar_spectral_signum_mask_f32(float* z, const float* la, const float* q, int64_t nq, int64_t planes, int64_t C,
                                              int64_t plane_elems, float intensity, float gain, int channel_sym, void* stream) {
    SONAR_REQUIRE(z && la && q && planes >= 0 && C >= 1 && plane_elems > 0 && (nq == 1 || nq == C), SONAR_ERR_ARG,
                  "sonar_spectral_signum_mask_f32: bad argument (one quantile row, or one per channel)");
    if (planes == 0) return SONAR_OK;
    hipLaunchKernelGGL(spectral_signum_mask_kernel, dim3(grid_for(planes * plane_elems, kBlock * 2)), dim3(kBlock), 0, (hipStream_t)stream,
                       reinterpret_cast<float2*>(z), la, q, (int)nq, planes, (int)C, plane_elems, intensity, gain, channel_sym);
    return check_launch("sonar_spectral_signum_mask_f32");
}

// torch.max semantics (a NaN anywhere gives NaN) over a small vector, written straight into host-visible pinned memory: one launch +
// one wait replaces torch's reduce kernel, device-to-host copy and wait for WaveletCFG's `sigma.max().item()`
// (py/wavelet_cfg.py:795-796).  The result and the ticket of the request leave as ONE 8-byte store, so a host that sees the ticket
// sees the value.
__global__ void __launch_bounds__(kBlock) max_to_host_kernel(const float* __restrict__ x, int64_t n, unsigned long long* __restrict__ host_out,
                                                             unsigned ticket) {
    kernarg_touch_for(x, n, host_out, ticket);
    __shared__ float part[kBlock / 64];
    __shared__ int nan_part[kBlock / 64];
    float m = -INFINITY;
    int bad = 0;
    for (int64_t i = threadIdx.x; i < n; i += kBlock) {
        const float v = x[i];
        bad |= v != v;
        m = fmaxf(m, v);
    }
    for (int off = 32; off > 0; off >>= 1) {
        m = fmaxf(m, __shfl_xor(m, off));
        bad |= __shfl_xor(bad, off);
    }
    if ((threadIdx.x & 63) == 0) {
        part[threadIdx.x >> 6] = m;
        nan_part[threadIdx.x >> 6] = bad;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) {
            m = fmaxf(m, part[w]);
            bad |= nan_part[w];
        }
        const unsigned long long word = ((unsigned long long)ticket << 32) | __float_as_uint(bad ? NAN : m);
        __hip_atomic_store(host_out, word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

namespace {
// one request in flight per host thread and device: the pinned word it lands in and the ticket it carries
struct MaxSlot {
    unsigned long long* word = nullptr;
    hipEvent_t done = nullptr;  // recorded right behind the reduction: the wait must not include what the caller queues after `begin`
    unsigned ticket = 0;
    bool pending = false;
};
thread_local MaxSlot g_max_slot[64];
}  // namespace

extern "C" int sonar_max_to_host_begin_f32(const float* x, int64_t n, void* stream) {
    SONAR_REQUIRE(x && n >= 1, SONAR_ERR_ARG, "sonar_max_to_host_begin_f32: bad argument (a non-empty device vector is required)");
    int dev = 0;
    SONAR_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64, SONAR_ERR_HIP, "sonar_max_to_host_begin_f32: no current device");
    MaxSlot& s = g_max_slot[dev];
    SONAR_REQUIRE(!s.pending, SONAR_ERR_ARG, "sonar_max_to_host_begin_f32: the previous request of this thread was not collected");
    if (!s.word) {
        SONAR_REQUIRE(hipHostMalloc((void**)&s.word, 64, hipHostMallocMapped) == hipSuccess, SONAR_ERR_HIP,
                      "sonar_max_to_host_begin_f32: pinned allocation failed");
        *s.word = 0;
    }
    if (!s.done)
        SONAR_REQUIRE(hipEventCreateWithFlags(&s.done, hipEventDisableTiming) == hipSuccess, SONAR_ERR_HIP,
                      "sonar_max_to_host_begin_f32: event creation failed");
    s.ticket = s.ticket + 1 ? s.ticket + 1 : 1;  // never 0: the word starts as 0
    hipLaunchKernelGGL(max_to_host_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, x, n, s.word, s.ticket);
    int rc = check_launch("sonar_max_to_host_begin_f32");
    if (rc == SONAR_OK && hipEventRecord(s.done, (hipStream_t)stream) != hipSuccess) {
        set_error("sonar_max_to_host_begin_f32: event record failed");
        rc = SONAR_ERR_HIP;
    }
    s.pending = rc == SONAR_OK;
    return rc;
}

extern "C" int sonar_max_to_host_end_f32(float* result, void* stream) {
    SONAR_REQUIRE(result, SONAR_ERR_ARG, "sonar_max_to_host_end_f32: bad argument");
    int dev = 0;
    SONAR_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64, SONAR_ERR_HIP, "sonar_max_to_host_end_f32: no current device");
    MaxSlot& s = g_max_slot[dev];
    SONAR_REQUIRE(s.pending, SONAR_ERR_ARG, "sonar_max_to_host_end_f32: no request in flight on this thread");
    s.pending = false;
    // the word usually lands within microseconds: look at it for a short while before handing the wait to the runtime
    unsigned long long word = 0;
    bool seen = false;
    for (int spin = 0; spin < 16384 && !seen; ++spin) {
        word = __atomic_load_n(s.word, __ATOMIC_ACQUIRE);
        seen = (unsigned)(word >> 32) == s.ticket;
    }
    if (!seen) {
        // not the stream: WaveletCFG queues its kernels behind the reduction before it asks for the value
        SONAR_REQUIRE(hipEventSynchronize(s.done) == hipSuccess, SONAR_ERR_HIP, "sonar_max_to_host_end_f32: wait failed");
        word = __atomic_load_n(s.word, __ATOMIC_ACQUIRE);
        SONAR_REQUIRE((unsigned)(word >> 32) == s.ticket, SONAR_ERR_HIP, "sonar_max_to_host_end_f32: the result never arrived");
    }
    const unsigned bits = (unsigned)word;
    memcpy(result, &bits, sizeof(float));
    return SONAR_OK;
}

extern "C" int sonar_max_to_host_f32(const float* x, int64_t n, float* result, void* stream) {
    SONAR_REQUIRE(result, SONAR_ERR_ARG, "sonar_max_to_host_f32: bad argument (a non-empty device vector is required)");
    const int rc = sonar_max_to_host_begin_f32(x, n, stream);
    return rc != SONAR_OK ? rc : sonar_max_to_host_end_f32(result, stream);
}

extern "C" int sonar_mask_mix_f32(const float* dst, const float* src, const float* mask, int64_t mask_n, float* out,
                                  int64_t n, void* stream) {
    SONAR_REQUIRE(dst && src && mask && out && n >= 0 && mask_n > 0, SONAR_ERR_ARG, "sonar_mask_mix_f32: bad argument");
    return launch_ew(MaskMixOp{dst, src, mask, mask_n, out}, n, aligned16(dst) && aligned16(src) && aligned16(out),
                     (hipStream_t)stream, "sonar_mask_mix_f32");
}

static bool cfg_ok(const sonar_momentum_cfg* c) {
    return c && c->mode >= 0 && c->mode <= 2 && c->momentum_blend >= 0 && c->momentum_blend <= 2 &&
           c->history_blend >= 0 && c->history_blend <= 2 && c->init_kind >= 0 && c->init_kind <= 2;
}

extern "C" int sonar_momentum_euler_f32(const float* x, const float* denoised, const float* h_in, float* x_out,
                                        float* h_out, const float* noise, float noise_scale, float sigma, float dt,
                                        const sonar_momentum_cfg* cfg, int64_t n, int* h_out_present,
                                        const sonar_noise_norm* noise_norm, void* stream) {
    SONAR_REQUIRE(x && denoised && x_out && n >= 0 && cfg_ok(cfg), SONAR_ERR_ARG, "sonar_momentum_euler_f32: bad argument");
    const int present = hist_present_after(*cfg, h_in != nullptr, true);
    SONAR_REQUIRE(!present || h_out, SONAR_ERR_ARG, "sonar_momentum_euler_f32: h_out required (history is produced)");
    if (h_out_present) *h_out_present = present;
    const bool v = aligned16(x) && aligned16(denoised) && aligned16(x_out) && (!h_in || aligned16(h_in)) &&
                   (!h_out || aligned16(h_out)) && (!noise || aligned16(noise));
    if (v && !nt_stores_host(n))  // beyond 32 MiB: streaming stores
        return launch_ew(EulerOpT<true>{x, denoised, h_in, x_out, h_out, noise, noise ? noise_norm : nullptr, noise_scale, sigma, dt, *cfg}, n, v,
                         (hipStream_t)stream, "sonar_momentum_euler_f32");
    return launch_ew(EulerOp{x, denoised, h_in, x_out, h_out, noise, noise ? noise_norm : nullptr, noise_scale, sigma, dt, *cfg}, n, v,
                     (hipStream_t)stream, "sonar_momentum_euler_f32");
}

extern "C" int sonar_dpmpp_stage1_f32(const float* x, const float* denoised, const float* h_in, float* x2_out,
                                      float* md1_out, float* h_out, const float* noise, float noise_scale, float sigma,
                                      float expm1_a, float ratio_a, int adj_is_one, const sonar_momentum_cfg* cfg,
                                      int64_t n, int* h_out_present, const sonar_noise_norm* noise_norm, void* stream) {
    SONAR_REQUIRE(x && denoised && x2_out && md1_out && n >= 0 && cfg_ok(cfg), SONAR_ERR_ARG,
                  "sonar_dpmpp_stage1_f32: bad argument");
    const int present = hist_present_after(*cfg, h_in != nullptr, true);
    SONAR_REQUIRE(!present || h_out, SONAR_ERR_ARG, "sonar_dpmpp_stage1_f32: h_out required");
    if (h_out_present) *h_out_present = present;
    const bool v = aligned16(x) && aligned16(denoised) && aligned16(x2_out) && aligned16(md1_out) &&
                   (!h_in || aligned16(h_in)) && (!h_out || aligned16(h_out)) && (!noise || aligned16(noise));
    return launch_ew(Dpmpp1Op{x, denoised, h_in, x2_out, md1_out, h_out, noise, noise ? noise_norm : nullptr, noise_scale, sigma, expm1_a, ratio_a,
                              adj_is_one, *cfg},
                     n, v, (hipStream_t)stream, "sonar_dpmpp_stage1_f32");
}

extern "C" int sonar_dpmpp_stage2_f32(const float* x, const float* denoised2, const float* md1, const float* h_in,
                                      float* x_out, float* dd_out, float* h_out, const float* noise, float noise_scale,
                                      float sigma_s, float expm1_b, float ratio_b, float fac, int adj_is_one,
                                      const sonar_momentum_cfg* cfg, int64_t n, int* h_out_present,
                                      const sonar_noise_norm* noise_norm, void* stream) {
    SONAR_REQUIRE(x && denoised2 && md1 && x_out && n >= 0 && cfg_ok(cfg), SONAR_ERR_ARG,
                  "sonar_dpmpp_stage2_f32: bad argument");
    const int present = hist_present_after(*cfg, h_in != nullptr, true);
    SONAR_REQUIRE(!present || h_out, SONAR_ERR_ARG, "sonar_dpmpp_stage2_f32: h_out required");
    if (h_out_present) *h_out_present = present;
    const bool v = aligned16(x) && aligned16(denoised2) && aligned16(md1) && aligned16(x_out) &&
                   (!dd_out || aligned16(dd_out)) && (!h_in || aligned16(h_in)) && (!h_out || aligned16(h_out)) &&
                   (!noise || aligned16(noise));
    return launch_ew(Dpmpp2Op{x, denoised2, md1, h_in, x_out, dd_out, h_out, noise, noise ? noise_norm : nullptr, noise_scale, sigma_s, expm1_b,
                              ratio_b, fac, adj_is_one, *cfg},
                     n, v, (hipStream_t)stream, "sonar_dpmpp_stage2_f32");
}

extern "C" int sonar_norm_decision_f32(const double* partials, int64_t npart, int64_t n_total, float factor, float threshold_std_devs,
                                       sonar_noise_norm* out, void* stream) {
    SONAR_REQUIRE(partials && out && npart > 0 && n_total > 1, SONAR_ERR_ARG, "sonar_norm_decision_f32: bad argument");
    hipLaunchKernelGGL(norm_decision_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, partials, npart, n_total, factor, threshold_std_devs,
                       out);
    return check_launch("sonar_norm_decision_f32");
}

extern "C" int sonar_apply_norm_f32(float* x, int64_t n, const sonar_noise_norm* norm, void* stream) {
    SONAR_REQUIRE(x && norm && n >= 0, SONAR_ERR_ARG, "sonar_apply_norm_f32: bad argument");
    return launch_ew(ApplyNormOp{x, norm}, n, aligned16(x), (hipStream_t)stream, "sonar_apply_norm_f32");
}

extern "C" int sonar_cast_f32_f64(const float* in, double* out, int64_t n, void* stream) {
    SONAR_REQUIRE(in && out && n >= 0, SONAR_ERR_ARG, "sonar_cast_f32_f64: bad argument");
    return launch_ew(CastOp{in, out}, n, aligned16(in) && aligned16(out), (hipStream_t)stream, "sonar_cast_f32_f64");
}
