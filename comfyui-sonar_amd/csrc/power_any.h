// General-size planes for the power-noise path, the single-translation-unit part: the statistics / spectrum kernels, the launcher and
// the pass-per-launch line kernels for planes beyond LDS.  The codelets, the passes and the plane kernel are in power_any_core.h.
// Included by power_fft.hip.
#pragma once
#include "power_any_core.h"

namespace sonar {

__global__ void __launch_bounds__(kFftThreads) power_stats_any_kernel(const float* __restrict__ filter, int64_t planes, AnyPlan pl,
                                                                      uint64_t seed, uint64_t stream_id, int64_t plane_offset, int group,
                                                                      int split, double* partials) {
    kernarg_touch_for(filter, planes, pl, seed, stream_id, plane_offset, group, split, partials);
    extern __shared__ __align__(16) unsigned char any_lds[];
    __shared__ double red[2 * kFftThreads / 64];
    power_stats_any_body(filter, planes, pl, seed, stream_id, plane_offset, group, split, partials, blockIdx.x, gridDim.x,
                         reinterpret_cast<c32*>(any_lds), red);
}

__global__ void __launch_bounds__(kFftThreads) power_spectrum_any_kernel(float* zout, int64_t planes, AnyPlan pl, uint64_t seed,
                                                                         uint64_t stream_id, int64_t plane_offset, int group, int split) {
    kernarg_touch_for(zout, planes, pl, seed, stream_id, plane_offset, group, split);
    const int H = pl.H, M = pl.M, S = pl.S, NC = H * S;
    const int tid = threadIdx.x;
    for (int64_t unit = blockIdx.x; unit < (split ? planes : planes / group); unit += gridDim.x) {
        const GroupWalk gw(unit, group, split);
        SpectrumRng rng = spectrum_rng_dyn<true>(seed, stream_id, plane_offset / group + gw.grp, tid, H);
        for (int i = 0; i < gw.first; ++i)
            draw_plane_dyn<true>(rng, tid, H, M, [](uint32_t, uint32_t, uint32_t) {}, [](int, int, uint32_t, uint32_t, uint32_t) {});
        for (int gp = gw.first; gp < gw.first + gw.count; ++gp) {
            c32* zp = reinterpret_cast<c32*>(zout) + (gw.grp * group + gp) * NC;
            draw_plane_dyn<true>(
                rng, tid, H, M,
                [&](uint32_t r0, uint32_t rm, uint32_t t) {
                    zp[tid * S] = unit_complex_normal(r0, angle_lo(t));
                    zp[tid * S + M] = unit_complex_normal(rm, angle_hi(t));
                },
                [&](int ky, int kx, uint32_t ra, uint32_t rb, uint32_t t) {
                    if (kx < M) {
                        zp[ky * S + kx] = unit_complex_normal(ra, angle_lo(t));
                        zp[(ky + H / 2) * S + kx] = unit_complex_normal(rb, angle_hi(t));
                    }
                });
        }
    }
}

// what: as launch_power.  The SDXL buckets' plane sizes have kernels with compile-time factor pairs (power_buckets_*.hip); every other
// size runs the run-time-size kernel.
int launch_power_bucket(int what, const float* z, const float* filter, float* out, int64_t planes, int64_t H, int64_t W, uint64_t seed,
                        uint64_t stream_id, int64_t plane_offset, int group, double* partials, NormArgs na, hipStream_t st, Ahead ah);
int launch_power_any_all(int what, const float* z, const float* filter, float* out, int64_t planes, int64_t H, int64_t W, uint64_t seed,
                         uint64_t stream_id, int64_t plane_offset, int group, double* partials, NormArgs na, hipStream_t st, Ahead ah);  // power_any_all.hip
static int launch_power_any(int what, const float* z, const float* filter, float* out, int64_t planes, int64_t H, int64_t W, uint64_t seed,
                            uint64_t stream_id, int64_t plane_offset, int group, double* partials, NormArgs na, hipStream_t st,
                            Ahead ah = Ahead()) {
    if (what != 2) {
        const int rc = launch_power_bucket(what, z, filter, out, planes, H, W, seed, stream_id, plane_offset, group, partials, na, st, ah);
        if (rc != kNotABucket) return rc;
        if (any_needs_all(H, W))
            return launch_power_any_all(what, z, filter, out, planes, H, W, seed, stream_id, plane_offset, group, partials, na, st, ah);
    }
    return launch_power_any_t<0, 0, 0, 0>(what, z, filter, out, planes, H, W, seed, stream_id, plane_offset, group, partials, na, st, ah);
}

// ---- planes beyond LDS: the same line transforms, a pass per launch through a complex workspace ---------------------------------
// (sonar_dft_rows_r2c_f32 / sonar_dft_cols_f32 / sonar_dft_rows_c2r_f32 for even widths: dft_direct.hip keeps the odd ones.)  A
// workgroup stages a batch of lines in LDS -- rows as W/2 complex values with the half-length trick, columns as a block of adjacent
// columns of one plane -- runs line_dft over them and writes the result; lengths up to kLinesMax, any factorisation (a factor the
// codelets do not cover costs that pass its direct sums: 512 = 16 x 32 is sixteen terms per value, not 512).
constexpr int kLinesThreads = 512;
#ifndef SONAR_LINES_AHEAD
#define SONAR_LINES_AHEAD 2  // lines_c2r_kernel<.., CR1 > 0>: chunks of 8 values per thread of the next batch requested a batch ahead (0: none)
#endif
constexpr int kLinesMax = 2048;
constexpr size_t kLinesLds = 64 * 1024;  // two workgroups per CU

// (row, column) of the flat item index j = thread + k x kLinesThreads over rows of `cols` items, walked instead of divided out
struct LinesWalk {
    int j, r, c, dr, dc;
    __device__ __forceinline__ LinesWalk(int tid, int cols) : j(tid), r(tid / cols), c(tid - (tid / cols) * cols), dr(kLinesThreads / cols), dc(kLinesThreads - (kLinesThreads / cols) * cols) {}
    __device__ __forceinline__ void next(int cols) {
        j += kLinesThreads;
        r += dr;
        c += dc;
        if (c >= cols) {
            c -= cols;
            ++r;
        }
    }
};

// A staging loop over `total` items in rows of `cols`, kLinesDepth global loads in flight per thread before the first use: one load per
// iteration waited for a memory latency per item (a workgroup stages 15-30 items per thread and batch).
constexpr int kLinesDepth = 8;
template <typename T, typename Load, typename Use>
__device__ __forceinline__ void lines_stage(int tid, int cols, int total, Load&& load, Use&& use) {
    LinesWalk lw(tid, cols);
    while (lw.j < total) {
        T v[kLinesDepth];
        int rr[kLinesDepth], cc[kLinesDepth];
#pragma unroll
        for (int u = 0; u < kLinesDepth; ++u) {
            rr[u] = lw.j < total ? lw.r : -1;
            cc[u] = lw.c;
            if (rr[u] >= 0) v[u] = load(rr[u], cc[u]);
            lw.next(cols);
        }
#pragma unroll
        for (int u = 0; u < kLinesDepth; ++u)
            if (rr[u] >= 0) use(rr[u], cc[u], v[u]);
    }
}

__device__ __forceinline__ void lines_table(c32* tw, int N, int tid) {  // e^{2 pi i j / N}
    for (int j = tid; j < N; j += kLinesThreads) {
        double sn, cs;
        sincospi(2.0 * (double)j / (double)N, &sn, &cs);
        tw[j] = make_float2((float)cs, (float)sn);
    }
}

// y[row][k] = sum_x x[row][x] e^{-2 pi i k x / W}, k = 0 .. W/2; `per` rows per batch
__global__ void __launch_bounds__(kLinesThreads, 4) lines_r2c_kernel(const float* __restrict__ x, c32* __restrict__ y, int64_t rows, int W, int n1,
                                                                     int n2, int per) {
    kernarg_touch_for(x, y, rows, W, n1, n2, per);
    extern __shared__ __align__(16) unsigned char any_lds[];
    const int M = W / 2, S = M + 1, tid = threadIdx.x;
    c32* const A = reinterpret_cast<c32*>(any_lds);
    c32* const tw = A + per * S;
    lines_table(tw, W, tid);
    for (int64_t r0 = (int64_t)blockIdx.x * per; r0 < rows; r0 += (int64_t)gridDim.x * per) {
        const int nr = (int)min<int64_t>(per, rows - r0);
        __syncthreads();
        lines_stage<float2>(tid, M, nr * M, [&](int r, int m) { return *reinterpret_cast<const float2*>(x + (r0 + r) * W + 2 * m); },
                            [&](int r, int m, float2 v) { A[r * S + m] = v; });
        __syncthreads();
        line_dft<kLinesThreads, true, 0, 0, kSetLines>(A, tw, W, 2, n1, n2, nr, 1, S, tid);
        // X[k] = E + w^k O, X[M-k] = conj(E - w^k O), E = (C[k] + conj C[M-k]) / 2, O = (C[k] - conj C[M-k]) / 2i, w = e^{-2 pi i / W}
        for (LinesWalk lw(tid, (M / 2 + 1)); lw.j < nr * (M / 2 + 1); lw.next((M / 2 + 1))) {
            const int r = lw.r, k = lw.c;
            c32* row = A + r * S;
            if (k == 0) {
                const c32 c0 = row[0];
                row[0] = make_float2(c0.x + c0.y, 0.0f);
                row[M] = make_float2(c0.x - c0.y, 0.0f);
            } else {
                const int kk = M - k;
                const c32 a = row[k], b = row[kk];
                const c32 e = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
                const c32 o = make_float2(0.5f * (a.y + b.y), -0.5f * (a.x - b.x));
                const c32 w = tw[k];
                const c32 t = make_float2(o.x * w.x + o.y * w.y, o.y * w.x - o.x * w.y);
                row[k] = make_float2(e.x + t.x, e.y + t.y);
                if (kk != k) row[kk] = make_float2(e.x - t.x, -(e.y - t.y));
            }
        }
        __syncthreads();
        for (LinesWalk lw(tid, S); lw.j < nr * S; lw.next(S)) {
            const int r = lw.r, k = lw.c;
            y[(r0 + r) * S + k] = A[r * S + k];
        }
    }
}

// MODE 0 / 1: out[p][n][k] = sum_m in[p][m][k] (* filter[m][k]) e^{-+ 2 pi i m n / H}; MODE 2: the spectral filter's middle -- forward
// columns, x filter[ky][k], inverse columns -- with the columns staying in LDS (one pass over the workspace instead of two).  A
// workgroup owns `cols` adjacent columns of one plane.
template <int MODE>
__global__ void __launch_bounds__(kLinesThreads, 4) lines_cols_kernel(const c32* __restrict__ in, const float* __restrict__ filter,
                                                                      c32* __restrict__ out, int64_t planes, int H, int K, int n1, int n2,
                                                                      int cols) {
    kernarg_touch_for(in, filter, out, planes, H, K, n1, n2, cols);
    extern __shared__ __align__(16) unsigned char any_lds[];
    const int S = cols | 1, tid = threadIdx.x;  // odd row stride
    c32* const A = reinterpret_cast<c32*>(any_lds);
    c32* const tw = A + (size_t)H * S;
    lines_table(tw, H, tid);
    const int blocks = (K + cols - 1) / cols;
    for (int64_t u = blockIdx.x; u < planes * blocks; u += gridDim.x) {
        const int64_t p = u / blocks;
        const int k0 = (int)(u - p * blocks) * cols, nc = min(cols, K - k0);
        const c32* src = in + p * (int64_t)H * K + k0;
        __syncthreads();
        lines_stage<c32>(tid, nc, H * nc, [&](int m, int c) { return src[(int64_t)m * K + c]; },
                         [&](int m, int c, c32 v) {
                             if (MODE == 1 && filter) {
                                 const float f = filter[(int64_t)m * K + k0 + c];
                                 v.x *= f;
                                 v.y *= f;
                             }
                             A[m * S + c] = v;
                         });
        __syncthreads();
        line_dft<kLinesThreads, MODE != 1, 0, 0, kSetLines>(A, tw, H, 1, n1, n2, nc, S, 1, tid);
        if constexpr (MODE == 2) {
            for (LinesWalk lw(tid, nc); lw.j < H * nc; lw.next(nc)) {
                const int m = lw.r, c = lw.c;
                const float f = filter[(int64_t)m * K + k0 + c];
                c32 v = A[m * S + c];
                v.x *= f;
                v.y *= f;
                A[m * S + c] = v;
            }
            __syncthreads();
            line_dft<kLinesThreads, false, 0, 0, kSetLines>(A, tw, H, 1, n1, n2, nc, S, 1, tid);
        }
        c32* dst = out + p * (int64_t)H * K + k0;
        for (LinesWalk lw(tid, nc); lw.j < H * nc; lw.next(nc)) {
            const int m = lw.r, c = lw.c;
            dst[(int64_t)m * K + c] = A[m * S + c];
        }
    }
}

// out[row][x] = scale * (Re y0 + sum_{k >= 1} w_k Re(y_k e^{+2 pi i k x / W})), w_k = 2 (1 for the Nyquist column)
// NORM: out = (v * scale - mean) / std * factor by the statistics in na.partials (computed before the rows exist: power_block.h)
#ifdef SONAR_LINES_TRACE  // profiling builds (scratch/lines_trace.py): thread 0's cycle stamps of a workgroup's first batches
static __device__ unsigned long long g_lines_trace[512 * 4 * 8];
#define SONAR_LINES_STAMP(slot) do { if (threadIdx.x == 0 && blockIdx.x < 512 && lt_b < 4) g_lines_trace[(blockIdx.x * 4 + lt_b) * 8 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define SONAR_LINES_STAMP(slot) do { } while (0)
#endif
// CR1, CR2 > 0: the row length's factor pair at compile time (256-wide rows: 128 = 16 x 8 -- the planes beyond LDS of 2048 px latents)
template <bool STATS, bool NORM = false, int CR1 = 0, int CR2 = 0>
__global__ void __launch_bounds__(kLinesThreads, 4) lines_c2r_kernel(const c32* __restrict__ y, float* __restrict__ out, int64_t rows, int W, int n1,
                                                                     int n2, int per, float scale, double* partials, NormArgs na) {
    kernarg_touch_for(y, out, rows, W, n1, n2, per, scale, partials, na);
    extern __shared__ __align__(16) unsigned char any_lds[];
    __shared__ double red[2 * kLinesThreads / 64];
    __shared__ NormDecision shd;
    const int M = W / 2, S = M + 1, tid = threadIdx.x;
    c32* const A = reinterpret_cast<c32*>(any_lds);
    c32* const tw = A + per * S;
    lines_table(tw, W, tid);
    [[maybe_unused]] float nc = 0.0f;
    if constexpr (NORM) {
        const NormDecision dec = decide_norm<kLinesThreads>(na.partials, kNPart, na.n_total, na.thr_sd, red, &shd);
        const float g = (dec.do_div ? 1.0f / dec.stdv : 1.0f) * na.factor;
        scale *= g;
        nc = dec.do_sub ? dec.mean * g : 0.0f;
    }
    double s = 0.0, q = 0.0;
    [[maybe_unused]] int lt_b = 0;  // (trace builds)
    // A batch is one contiguous run of the workspace and of LDS (same row stride).  The compile-time-pair instantiation (256-wide rows)
    // requests the NEXT batch's values -- all of them: 2 x 8 per thread -- before the current one is transformed and stored: a workgroup sat a
    // third of a batch's time behind its own loads (`scratch/lines_trace.py`: 8.5 k of 26 k ticks).  The run-time instantiation has no
    // registers to spare for it (it spilled 4-12 and lost what the request gained).  The loop is split at the request, not closed behind the
    // stores (see spectral_filter128_kernel).
    constexpr int U = 8, PF = CR1 > 0 ? SONAR_LINES_AHEAD : 0;
    [[maybe_unused]] c32 ahead[PF > 0 ? PF : 1][U];
    auto request = [&](int64_t r0) {
        const c32* __restrict__ src = y + r0 * S;
        const int total = (int)min<int64_t>(per, rows - r0) * S;
#pragma unroll
        for (int c = 0; c < PF; ++c) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = tid + (c * U + u) * kLinesThreads;
                ahead[c][u] = src[i < total ? i : 0];
            }
        }
    };
    auto transform_and_store = [&](int64_t r0, int nr) {
        // G[k] = (X[k] + conj X[M-k]) + i (X[k] - conj X[M-k]) w^k (X[0], X[M] contribute their real parts), formed by the first pass as it
        // loads when that pass is a codelet (c2r_pass0: one LDS round trip and one barrier fewer per batch), then the length-M inverse DFT
        c2r_rows<kLinesThreads, CR1, CR2, kSetLines>(A, tw, W, n1, n2, nr, M, S, 1, tid);
        SONAR_LINES_STAMP(2);
        for (LinesWalk lw(tid, M); lw.j < nr * M; lw.next(M)) {
            const int r = lw.r, m = lw.c;
            const c32 g = A[r * S + m];
            const float a = NORM ? __builtin_fmaf(g.x, scale, -nc) : g.x * scale, b = NORM ? __builtin_fmaf(g.y, scale, -nc) : g.y * scale;
            *reinterpret_cast<float2*>(out + (r0 + r) * W + 2 * m) = make_float2(a, b);
            if constexpr (STATS) {  // fp64 per value, like the direct pass (its callers compare the sums with the tensor's)
                const double da = a, db = b;
                s += da + db;
                q += da * da + db * db;
            }
        }
        SONAR_LINES_STAMP(3);
        ++lt_b;
    };
    int64_t r0 = (int64_t)blockIdx.x * per;
    if (r0 < rows) {
        if constexpr (PF > 0) request(r0);
        for (;;) {
            const int nr = (int)min<int64_t>(per, rows - r0);
            const int total = nr * S;
            __syncthreads();
            SONAR_LINES_STAMP(0);
            if constexpr (PF > 0) {
#pragma unroll
                for (int c = 0; c < PF; ++c) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int i = tid + (c * U + u) * kLinesThreads;
                        if (i < total) A[i] = ahead[c][u];
                    }
                }
            }
            {   // what the request did not cover: eight loads in flight per thread
                const c32* __restrict__ src = y + r0 * S;
                for (int i0 = tid + PF * U * kLinesThreads; i0 < total; i0 += U * kLinesThreads) {
                    c32 v[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int i = i0 + u * kLinesThreads;
                        v[u] = src[i < total ? i : i0];
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int i = i0 + u * kLinesThreads;
                        if (i < total) A[i] = v[u];
                    }
                }
            }
            __syncthreads();
            SONAR_LINES_STAMP(1);
            const int64_t next = r0 + (int64_t)gridDim.x * per;
            if (next >= rows) {  // uniform
                transform_and_store(r0, nr);
                break;
            }
            if constexpr (PF > 0) request(next);
            transform_and_store(r0, nr);
            r0 = next;
        }
    }
    if constexpr (STATS) write_partial<kLinesThreads>(s, q, partials, red);
}
#ifdef SONAR_LINES_TRACE
extern "C" int sonar_debug_lines_trace(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(sonar::g_lines_trace), sizeof(sonar::g_lines_trace));
}
#endif

// Odd widths have no half-length trick: a row is transformed as W complex values (imaginary parts zero on the way in, discarded on the
// way out) -- twice the arithmetic of an even neighbour, not the O(W) per value of the direct sums (135 = 15 x 9 runs as two codelets).
__global__ void __launch_bounds__(kLinesThreads, 4) lines_r2c_odd_kernel(const float* __restrict__ x, c32* __restrict__ y, int64_t rows, int W, int n1,
                                                                         int n2, int per) {
    kernarg_touch_for(x, y, rows, W, n1, n2, per);
    extern __shared__ __align__(16) unsigned char any_lds[];
    const int K = W / 2 + 1, S = W | 1, tid = threadIdx.x;
    c32* const A = reinterpret_cast<c32*>(any_lds);
    c32* const tw = A + per * S;
    lines_table(tw, W, tid);
    for (int64_t r0 = (int64_t)blockIdx.x * per; r0 < rows; r0 += (int64_t)gridDim.x * per) {
        const int nr = (int)min<int64_t>(per, rows - r0);
        __syncthreads();
        for (LinesWalk lw(tid, W); lw.j < nr * W; lw.next(W)) {
            const int r = lw.r, i = lw.c;
            A[r * S + i] = make_float2(x[(r0 + r) * W + i], 0.0f);
        }
        __syncthreads();
        line_dft<kLinesThreads, true, 0, 0, kSetLines>(A, tw, W, 1, n1, n2, nr, 1, S, tid);
        for (LinesWalk lw(tid, K); lw.j < nr * K; lw.next(K)) {
            const int r = lw.r, k = lw.c;
            y[(r0 + r) * K + k] = A[r * S + k];
        }
    }
}

template <bool STATS>
__global__ void __launch_bounds__(kLinesThreads, 4) lines_c2r_odd_kernel(const c32* __restrict__ y, float* __restrict__ out, int64_t rows, int W, int n1,
                                                                         int n2, int per, float scale, double* partials) {
    kernarg_touch_for(y, out, rows, W, n1, n2, per, scale, partials);
    extern __shared__ __align__(16) unsigned char any_lds[];
    __shared__ double red[2 * kLinesThreads / 64];
    const int K = W / 2 + 1, S = W | 1, tid = threadIdx.x;
    c32* const A = reinterpret_cast<c32*>(any_lds);
    c32* const tw = A + per * S;
    lines_table(tw, W, tid);
    double s = 0.0, q = 0.0;
    for (int64_t r0 = (int64_t)blockIdx.x * per; r0 < rows; r0 += (int64_t)gridDim.x * per) {
        const int nr = (int)min<int64_t>(per, rows - r0);
        __syncthreads();
        // the full Hermitian line: X[k] = y[k], X[W - k] = conj y[k]; the DC value's imaginary part is ignored, as irfft does
        for (LinesWalk lw(tid, K); lw.j < nr * K; lw.next(K)) {
            const int r = lw.r, k = lw.c;
            const c32 v = y[(r0 + r) * K + k];
            if (k == 0) {
                A[r * S] = make_float2(v.x, 0.0f);
            } else {
                A[r * S + k] = v;
                A[r * S + W - k] = make_float2(v.x, -v.y);
            }
        }
        __syncthreads();
        line_dft<kLinesThreads, false, 0, 0, kSetLines>(A, tw, W, 1, n1, n2, nr, 1, S, tid);
        for (LinesWalk lw(tid, W); lw.j < nr * W; lw.next(W)) {
            const int r = lw.r, i = lw.c;
            const float a = A[r * S + i].x * scale;
            out[(r0 + r) * W + i] = a;
            if constexpr (STATS) {
                const double da = a;
                s += da;
                q += da * da;
            }
        }
    }
    if constexpr (STATS) write_partial<kLinesThreads>(s, q, partials, red);
}

// rows per batch / columns per block that fit kLinesLds beside the twiddle table
static inline int lines_per(size_t line_bytes, size_t table_bytes, int want) {
    const size_t room = kLinesLds > table_bytes ? kLinesLds - table_bytes : 0;
    return (int)std::max<size_t>(1, std::min<size_t>(want, room / line_bytes));
}

template <typename K>
static void lines_lds_attr(K kern) {
    lds_attr(reinterpret_cast<const void*>(kern), (int)kAnyLdsLimit);
}

}  // namespace sonar

// even widths / any height up to kLinesMax; false: not taken (the direct passes of dft_direct.hip run)
bool sonar_lines_rows_r2c(const float* x, float* y, int64_t rows, int64_t W, hipStream_t st) {
    using namespace sonar;
    if (W < 3 || W > kLinesMax || (!(W & 1) && W < 4)) return false;
    if (W & 1) {
        int n1, n2;
        best_split((int)W, n1, n2, kSetLines);
        if (n1 == 1) return false;  // a prime above the codelets: the direct sums' own kernel is the faster one (135 x 241: 2.4 against 2.65 ms)
        const size_t line = (size_t)(W | 1) * sizeof(c32), table = (size_t)W * sizeof(c32);
        const int per = lines_per(line, table, 512);
        lines_lds_attr(lines_r2c_odd_kernel);
        const int g = (int)std::max<int64_t>(1, std::min<int64_t>((rows + per - 1) / per, 1024));
        hipLaunchKernelGGL(lines_r2c_odd_kernel, dim3(g), dim3(kLinesThreads), per * line + table, st, x, reinterpret_cast<c32*>(y), rows, (int)W, n1, n2, per);
        return true;
    }
    if (reinterpret_cast<uintptr_t>(x) & 7u) return false;
    const int M = (int)W / 2;
    int n1, n2;
    best_split(M, n1, n2, kSetLines);
    const size_t line = (size_t)(M + 1) * sizeof(c32), table = (size_t)W * sizeof(c32);
    const int per = lines_per(line, table, 512);
    lines_lds_attr(lines_r2c_kernel);
    const int g = (int)std::max<int64_t>(1, std::min<int64_t>((rows + per - 1) / per, 1024));
    hipLaunchKernelGGL(lines_r2c_kernel, dim3(g), dim3(kLinesThreads), per * line + table, st, x, reinterpret_cast<c32*>(y), rows, (int)W, n1, n2, per);
    return true;
}

bool sonar_lines_cols(const float* in, const float* filter, float* out, int64_t planes, int64_t H, int64_t K, int inverse, hipStream_t st) {
    using namespace sonar;
    if (H < 2 || H > kLinesMax) return false;
    int n1, n2;
    best_split((int)H, n1, n2, kSetLines);
    const size_t table = (size_t)H * sizeof(c32);
    // a block of columns is a run of cols x 8 bytes in every row of the workspace: whole 128-byte lines when the budget allows (two
    // workgroups per CU: 74 KB each with the odd LDS row stride)
    const size_t room = 74 * 1024 - table;
    int cols = (int)std::max<size_t>(1, std::min<size_t>(64, room / ((size_t)H * sizeof(c32)) - 1));
    for (int unit = 16; unit >= 2; unit /= 2)
        if (cols >= unit) {
            cols -= cols % unit;
            break;
        }
    cols = (int)std::min<int64_t>(cols, K);
    const size_t lds = (size_t)H * (cols | 1) * sizeof(c32) + table;
    if (lds > kAnyLdsLimit) return false;
    const int64_t units = planes * ((K + cols - 1) / cols);
    const int g = (int)std::max<int64_t>(1, std::min<int64_t>(units, 4096));
#define SONAR_LINES_COLS(MODE)                                                                                                            \
    do {                                                                                                                                   \
        lines_lds_attr(lines_cols_kernel<MODE>);                                                                                           \
        hipLaunchKernelGGL(lines_cols_kernel<MODE>, dim3(g), dim3(kLinesThreads), lds, st, reinterpret_cast<const c32*>(in), filter,       \
                           reinterpret_cast<c32*>(out), planes, (int)H, (int)K, n1, n2, cols);                                             \
    } while (0)
    if (inverse == 2) SONAR_LINES_COLS(2); else if (inverse) SONAR_LINES_COLS(1); else SONAR_LINES_COLS(0);
#undef SONAR_LINES_COLS
    return true;
}

bool sonar_lines_rows_c2r(const float* y, float* out, int64_t rows, int64_t W, float scale, double* partials, hipStream_t st) {
    return sonar_lines_rows_c2r_norm(y, out, rows, W, scale, partials, nullptr, st);
}

// `norm`: write the rows normalised by the statistics it names (even widths only; no statistics of the output then)
bool sonar_lines_rows_c2r_norm(const float* y, float* out, int64_t rows, int64_t W, float scale, double* partials, const sonar::NormArgs* norm,
                               hipStream_t st) {
    using namespace sonar;
    if (W < 3 || W > kLinesMax || (!(W & 1) && W < 4)) return false;
    if (norm && ((W & 1) || partials)) return false;
    const NormArgs na = norm ? *norm : NormArgs{nullptr, 0, 1.0f, 0.0f};
    if (W & 1) {
        int n1, n2;
        best_split((int)W, n1, n2, kSetLines);
        if (n1 == 1) return false;
        const size_t line = (size_t)(W | 1) * sizeof(c32), table = (size_t)W * sizeof(c32);
        const int per = lines_per(line, table, 512);
        const int g = (int)std::max<int64_t>(1, std::min<int64_t>((rows + per - 1) / per, kNPart));
        if (partials) {
            lines_lds_attr(lines_c2r_odd_kernel<true>);
            hipLaunchKernelGGL(lines_c2r_odd_kernel<true>, dim3(g), dim3(kLinesThreads), per * line + table, st, reinterpret_cast<const c32*>(y), out, rows,
                               (int)W, n1, n2, per, scale, partials);
        } else {
            lines_lds_attr(lines_c2r_odd_kernel<false>);
            hipLaunchKernelGGL(lines_c2r_odd_kernel<false>, dim3(g), dim3(kLinesThreads), per * line + table, st, reinterpret_cast<const c32*>(y), out, rows,
                               (int)W, n1, n2, per, scale, partials);
        }
        return true;
    }
    if (reinterpret_cast<uintptr_t>(out) & 7u) return false;
    const int M = (int)W / 2;
    int n1, n2;
    best_split(M, n1, n2, kSetLines);
    const size_t line = (size_t)(M + 1) * sizeof(c32), table = (size_t)W * sizeof(c32);
    const int per = lines_per(line, table, 512);
    const int g = (int)std::max<int64_t>(1, std::min<int64_t>((rows + per - 1) / per, kNPart));
    if (norm && n1 == 16 && n2 == 8) {
        lines_lds_attr(lines_c2r_kernel<false, true, 16, 8>);
        hipLaunchKernelGGL((lines_c2r_kernel<false, true, 16, 8>), dim3(g), dim3(kLinesThreads), per * line + table, st, reinterpret_cast<const c32*>(y), out,
                           rows, (int)W, n1, n2, per, scale, partials, na);
    } else if (norm) {
        lines_lds_attr(lines_c2r_kernel<false, true>);
        hipLaunchKernelGGL((lines_c2r_kernel<false, true>), dim3(g), dim3(kLinesThreads), per * line + table, st, reinterpret_cast<const c32*>(y), out, rows,
                           (int)W, n1, n2, per, scale, partials, na);
    } else if (partials) {
        lines_lds_attr(lines_c2r_kernel<true>);
        hipLaunchKernelGGL(lines_c2r_kernel<true>, dim3(g), dim3(kLinesThreads), per * line + table, st, reinterpret_cast<const c32*>(y), out, rows, (int)W,
                           n1, n2, per, scale, partials, na);
    } else {
        lines_lds_attr(lines_c2r_kernel<false>);
        hipLaunchKernelGGL(lines_c2r_kernel<false>, dim3(g), dim3(kLinesThreads), per * line + table, st, reinterpret_cast<const c32*>(y), out, rows, (int)W,
                           n1, n2, per, scale, partials, na);
    }
    return true;
}
