// Base generators (Philox normal / uniform), Perlin lattice noise and pyramid (multi-resolution)
// noise.  Output-write-bound: every kernel streams 16 B/lane stores of contiguous NCHW latents
// and, when asked, folds the whole-tensor (sum, sumsq) partials of the normaliser into the same pass.
#include <math.h>
#include <stdlib.h>

#include "common.h"

namespace sonar {

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#ifdef SONAR_NG_TRACE  // profiling builds: cycle stamps of thread 0 of the first 256 workgroups at the phase boundaries (scratch/ng_trace.py)
__device__ unsigned long long g_ng_trace[256 * 16];
#define SONAR_NG_STAMP(slot) do { if (threadIdx.x == 0 && blockIdx.x < 256) g_ng_trace[blockIdx.x * 16 + (slot)] = __builtin_readcyclecounter(); } while (0)
#define SONAR_NG_STAMP_T(thread, slot) do { if ((int)threadIdx.x == (thread) && blockIdx.x < 256) g_ng_trace[blockIdx.x * 16 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define SONAR_NG_STAMP(slot) do { } while (0)
#define SONAR_NG_STAMP_T(thread, slot) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// Draw loop shared by every generator: one wave per tile (common.h: kTileElems elements), each lane
// runs its own Philox-seeded multiply-with-carry burst (common.h, Mwc) and hands 4 consecutive elements per step to `f`.
// `f(e, v)` receives the LOCAL index e (may be < 0 or >= n at the two ends of a shard) of v[0].
enum class Dist { Normal, Uniform };

template <Dist D, typename F>
__device__ __forceinline__ void for_each_group(int64_t n, uint64_t seed, uint64_t stream_id, int64_t elem_offset, F&& f) {
    const uint32_t lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * kBlock) >> 6;
    const int64_t first = elem_offset / kTileElems, last = (elem_offset + n - 1) / kTileElems;
    for (int64_t tile = first + wave; tile <= last; tile += nwaves) {
        TileRng rng = rng_stream(seed, stream_id, (uint64_t)tile, lane);
        const int64_t base = tile * kTileElems + (int64_t)lane * 4 - elem_offset;
#pragma unroll 4
        for (int it = 0; it < kTileIters; ++it) {
            float v[4];
            if constexpr (D == Dist::Normal) rng.normal4(v); else rng.uniform4(v);
            f(base + it * 256, v);
        }
    }
}

// store 4 values at local index e with range / alignment handling; returns how many were in range
template <bool VEC>
__device__ __forceinline__ void store_group(float* out, int64_t n, int64_t e, const float (&v)[4], double& s, double& q, bool stats) {
    if (VEC && e >= 0 && e + 4 <= n) {
        *reinterpret_cast<float4*>(out + e) = make_float4(v[0], v[1], v[2], v[3]);
        if (stats) {
            const float ps = (v[0] + v[1]) + (v[2] + v[3]);
            const float pq = __builtin_fmaf(v[0], v[0], __builtin_fmaf(v[1], v[1], __builtin_fmaf(v[2], v[2], v[3] * v[3])));
            s += (double)ps;
            q += (double)pq;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (e + k >= 0 && e + k < n) {
                out[e + k] = v[k];
                if (stats) {
                    s += (double)v[k];
                    q += (double)v[k] * (double)v[k];
                }
            }
        }
    }
}

// Accumulating epilogue of the generators: instead of writing a fresh tensor that a chain then folds into its running sum with
// sonar_axpby_f32 (read 8N, write 4N), the generator reads the sum and writes it back: v <- y * ya + v * f with the products
// rounded on their own and skipped when the multiplier is exactly 1, as AxpbyOp does (elementwise.hip), so the chain's value does
// not change by a bit.  y == nullptr: plain store.  y may be the output itself (in place).
struct Accum {
    const float* y;
    float ya, f;
    __device__ __forceinline__ float operator()(float yv, float v) const {
        const float yy = ya != 1.0f ? yv * ya : yv;
        const float xx = f != 1.0f ? v * f : v;
        return yy + xx;
    }
};
template <bool VEC>
__device__ __forceinline__ void accumulate_group(const Accum& acc, int64_t n, int64_t e, float (&v)[4]) {
    if (!acc.y) return;
    if (VEC && e >= 0 && e + 4 <= n) {
        const float4 y = *reinterpret_cast<const float4*>(acc.y + e);
        v[0] = acc(y.x, v[0]);
        v[1] = acc(y.y, v[1]);
        v[2] = acc(y.z, v[2]);
        v[3] = acc(y.w, v[3]);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (e + k >= 0 && e + k < n) v[k] = acc(acc.y[e + k], v[k]);
    }
}
static constexpr Accum kNoAccum{nullptr, 1.0f, 1.0f};

// A tile-keyed generator's fold y <- y * ya + x * f riding in ANOTHER generator's pass over y (sonar_fold_prefix): the chain's previous
// item (Gaussian draw or Perlin) is evaluated on the fly and applied BEFORE the hosting kernel's own fold, so its read + write of the
// running sum disappears.  Same stream keys, same operations in the same order as its own kernel -> the same bits.  fresh: that item
// is the chain's first, the sum holds nothing yet: y1 = x, no read at all.
struct Prefix {
    float ya, f, div_fac;
    uint64_t seed, stream_id;
    const float* terms;
    int chw;
    int fresh;
};
// kind: 0 none, 1 Gaussian draw, 2 Perlin (summed lattice).  One group of four values: x drawn by the caller's copy of the prefix's
// generator (`rng`, advanced here), `term` the lattice vector of these four elements (Perlin).
template <int PRE, typename DIV>
__device__ __forceinline__ void prefix_draw(TileRng& rng, const DIV& div, const float4& term, float (&x)[4]) {
    if constexpr (PRE == 1) {
        rng.normal4(x);
    } else {
        uint32_t r[4];
        rng.words4_high(r);
        x[0] = div.from_word(r[0], term.x); x[1] = div.from_word(r[1], term.y);
        x[2] = div.from_word(r[2], term.z); x[3] = div.from_word(r[3], term.w);
    }
}
// v <- fold(y1, v) with y1 = x (fresh) or y * ya + x * f, y read from the running sum
__device__ __forceinline__ void prefix_fold(const Prefix& pre, const Accum& pfold, const Accum& fold, int64_t e, const float (&x)[4], float (&v)[4]) {
    if (pre.fresh) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = fold(x[k], v[k]);
    } else {
        const float4 y = *reinterpret_cast<const float4*>(fold.y + e);
        v[0] = fold(pfold(y.x, x[0]), v[0]);
        v[1] = fold(pfold(y.y, x[1]), v[1]);
        v[2] = fold(pfold(y.z, x[2]), v[2]);
        v[3] = fold(pfold(y.w, x[3]), v[3]);
    }
}
static int make_prefix(const sonar_fold_prefix* pre, Prefix& px, const char* what) {
    SONAR_REQUIRE(pre->kind == SONAR_PREFIX_NORMAL || pre->kind == SONAR_PREFIX_PERLIN, SONAR_ERR_ARG, "%s: unknown prefix kind", what);
    const bool perlin = pre->kind == SONAR_PREFIX_PERLIN;
    SONAR_REQUIRE(!perlin || (pre->terms && pre->chw > 0 && pre->chw < (1LL << 31) && (reinterpret_cast<uintptr_t>(pre->terms) & 15u) == 0),
                  SONAR_ERR_ARG, "%s: a Perlin prefix needs its 16-byte aligned lattice", what);
    SONAR_REQUIRE(!pre->fresh || pre->x_mul == 1.0f, SONAR_ERR_ARG, "%s: a fresh prefix is the first item's raw values (x_mul 1)", what);
    px = Prefix{pre->y_mul, pre->x_mul, perlin ? pre->div_fac : 1.0f, pre->seed, pre->stream_id, pre->terms, perlin ? (int)pre->chw : 1,
                pre->fresh ? 1 : 0};
    return SONAR_OK;
}

struct Affine {
    float sub, mul, add;
    int active;
    __device__ __forceinline__ float operator()(float u) const { return active ? (u - sub) * mul + add : u; }
};

// the affine map and the normalisation with their run-time switches decided ONCE per launch (template flags): inside the tile loop every
// switch was a v_cndmask per value, and the range checks of a shard's ragged ends a branch per group -- 220 instructions per step of
// eight values where the arithmetic needs 90 (the launch was instruction-bound: 33 us against the 24 us of its stores)
template <bool ACTIVE>
struct AffineT {
    float sub, mul, add;
    __device__ __forceinline__ float operator()(float u) const {
        if constexpr (ACTIVE) return (u - sub) * mul + add;
        else return u;
    }
};
template <bool SUB, bool SCALE>
struct NormT {
    float mean, inv_std, factor;
    __device__ __forceinline__ float operator()(float v) const {
        if constexpr (SUB) v = v - mean;
        if constexpr (SCALE) v = v * inv_std * factor;
        return v;
    }
};
template <typename F>
__device__ __forceinline__ void with_affine_norm(const Affine& aff, const NormFast& nf, F&& f) {
    auto with_norm = [&](auto a) {
        if (nf.do_sub) {
            if (nf.do_scale) f(a, NormT<true, true>{nf.mean, nf.inv_std, nf.factor});
            else f(a, NormT<true, false>{nf.mean, nf.inv_std, nf.factor});
        } else {
            if (nf.do_scale) f(a, NormT<false, true>{nf.mean, nf.inv_std, nf.factor});
            else f(a, NormT<false, false>{nf.mean, nf.inv_std, nf.factor});
        }
    };
    if (aff.active) with_norm(AffineT<true>{aff.sub, aff.mul, aff.add});
    else with_norm(AffineT<false>{aff.sub, aff.mul, aff.add});
}

// scale_noise_kernel's own sequence (subtract, IEEE division; factor 1): what the one-pass N(0,1) route applies when a threshold fails
template <bool SUB, bool DIV>
struct NormExactT {
    float mean, stdv;
    __device__ __forceinline__ float operator()(float v) const {
        if constexpr (SUB) v = v - mean;
        if constexpr (DIV) v = v / stdv;
        return v;
    }
};
template <typename F>
__device__ __forceinline__ void with_exact_norm(const NormDecision& d, F&& f) {
    if (d.do_sub) {
        if (d.do_div) f(NormExactT<true, true>{d.mean, d.stdv});
        else f(NormExactT<true, false>{d.mean, d.stdv});
    } else {
        if (d.do_div) f(NormExactT<false, true>{d.mean, d.stdv});
        else f(NormExactT<false, false>{d.mean, d.stdv});
    }
}
template <typename F>
__device__ __forceinline__ void with_norm_flags(const NormFast& nf, F&& f) {
    if (nf.do_sub) {
        if (nf.do_scale) f(NormT<true, true>{nf.mean, nf.inv_std, nf.factor});
        else f(NormT<true, false>{nf.mean, nf.inv_std, nf.factor});
    } else {
        if (nf.do_scale) f(NormT<false, true>{nf.mean, nf.inv_std, nf.factor});
        else f(NormT<false, false>{nf.mean, nf.inv_std, nf.factor});
    }
}

// VEC: out 16-B aligned and elem_offset % 4 == 0 -> dwordx4 stores
// WHOLE (round 6): whole tiles only, a plain store (no running sum to fold into): no range checks, the affine map's switch decided once per
// launch (store_group's checks and the switch were a branch and a select per group: the fill of 134 MB was ~10 % instruction-bound).  The
// same values, the same per-thread sums in the same order.
template <Dist D, bool VEC, bool STATS, bool WHOLE = false>
__global__ void __launch_bounds__(kBlock) stream_fill_kernel(float* out, int64_t n, uint64_t seed, uint64_t stream_id,
                                                             int64_t elem_offset, Affine aff, double* partials, Accum acc) {
    kernarg_touch_for(out, n, seed, stream_id, elem_offset, aff, partials, acc);
    __shared__ double red[2 * kBlock / 64];
    double s = 0.0, q = 0.0;
    if constexpr (WHOLE) {
        const uint32_t lane = threadIdx.x & 63;
        const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
        const int64_t nwaves = ((int64_t)gridDim.x * kBlock) >> 6;
        const int64_t first = elem_offset / kTileElems, last = (elem_offset + n - 1) / kTileElems;
        auto tiles = [&](auto af) {
            for (int64_t tile = first + wave; tile <= last; tile += nwaves) {
                TileRng rng = rng_stream(seed, stream_id, (uint64_t)tile, lane);
                float* const o = out + (tile * kTileElems + (int64_t)lane * 4 - elem_offset);
#pragma unroll 4
                for (int it = 0; it < kTileIters; ++it) {
                    float v[4];
                    if constexpr (D == Dist::Normal) rng.normal4(v); else rng.uniform4(v);
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = af(v[k]);
                    *reinterpret_cast<float4*>(o + it * 256) = make_float4(v[0], v[1], v[2], v[3]);
                    if constexpr (STATS) {
                        const float ps = (v[0] + v[1]) + (v[2] + v[3]);
                        const float pq = __builtin_fmaf(v[0], v[0], __builtin_fmaf(v[1], v[1], __builtin_fmaf(v[2], v[2], v[3] * v[3])));
                        s += (double)ps;
                        q += (double)pq;
                    }
                }
            }
        };
        if (aff.active) tiles(AffineT<true>{aff.sub, aff.mul, aff.add});
        else tiles(AffineT<false>{aff.sub, aff.mul, aff.add});
    } else {
        for_each_group<D>(n, seed, stream_id, elem_offset, [&](int64_t e, float (&v)[4]) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = aff(v[k]);
            accumulate_group<VEC>(acc, n, e, v);
            store_group<VEC>(out, n, e, v, s, q, STATS);
        });
    }
    if constexpr (STATS) write_partial<kBlock>(s, q, partials, red);
}

static inline int tile_grid(int64_t n, int64_t elem_offset) {
    const int64_t tiles = (elem_offset + n - 1) / kTileElems - elem_offset / kTileElems + 1;
    return (int)std::max<int64_t>(1, std::min<int64_t>(kNPart, (tiles + 3) / 4));  // 4 waves (tiles) per block
}

template <Dist D>
static int launch_fill(float* out, int64_t n, uint64_t seed, uint64_t stream_id, int64_t elem_offset, Affine aff,
                       double* partials, hipStream_t st, const char* what, Accum acc = kNoAccum) {
    if (n == 0) return SONAR_OK;
    const bool vec = aligned16(out) && aligned16(acc.y) && (elem_offset & 3) == 0;
    const int g = tile_grid(n, elem_offset);
    if (vec && !acc.y && n % kTileElems == 0 && elem_offset % kTileElems == 0) {
        if (partials) hipLaunchKernelGGL((stream_fill_kernel<D, true, true, true>), dim3(g), dim3(kBlock), 0, st, out, n, seed, stream_id, elem_offset, aff, partials, acc);
        else hipLaunchKernelGGL((stream_fill_kernel<D, true, false, true>), dim3(g), dim3(kBlock), 0, st, out, n, seed, stream_id, elem_offset, aff, partials, acc);
        return check_launch(what);
    }
#define SONAR_FILL(A, S) \
    hipLaunchKernelGGL((stream_fill_kernel<D, A, S>), dim3(g), dim3(kBlock), 0, st, out, n, seed, stream_id, elem_offset, aff, partials, acc)
    if (vec) {
        if (partials) SONAR_FILL(true, true); else SONAR_FILL(true, false);
    } else {
        if (partials) SONAR_FILL(false, true); else SONAR_FILL(false, false);
    }
#undef SONAR_FILL
    return check_launch(what);
}

// Two tile-keyed generators in one pass over a chain's running sum: the HOST item (1 Gaussian draw, 2 Perlin) folds its values as its
// own accumulating kernel would, after the previous item (PRE, same kinds; `pre.fresh`: the chain's first) has been applied on the
// fly.  Whole groups of four only (n, elem_offset multiples of 4, 16-byte aligned tensors, Perlin latents of a multiple of 4).
template <int HOST, int PRE, bool STATS>
__global__ void __launch_bounds__(kBlock) pair_fold_kernel(Accum fold, int64_t n, int64_t elem_offset, Prefix host, Prefix pre, double* partials) {
    kernarg_touch_for(fold, n, elem_offset, host, pre, partials);
    __shared__ double red[2 * kBlock / 64];
    double s = 0.0, q = 0.0;
    const uint32_t lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * kBlock) >> 6;
    const int64_t first = elem_offset / kTileElems, last = (elem_offset + n - 1) / kTileElems;
    const Accum pfold{fold.y, pre.ya, pre.f};
    const Divider hdiv(HOST == 2 ? host.div_fac : 1.0f), pdiv(PRE == 2 ? pre.div_fac : 1.0f);
    for (int64_t tile = first + wave; tile <= last; tile += nwaves) {
        TileRng hrng = rng_stream(host.seed, host.stream_id, (uint64_t)tile, lane);
        TileRng prng = rng_stream(pre.seed, pre.stream_id, (uint64_t)tile, lane);
        const int64_t base = tile * kTileElems + (int64_t)lane * 4 - elem_offset;
        // positions inside the latent (the lattices repeat per latent; shards start on a latent boundary), kept incrementally
        int rh = HOST == 2 ? (int)(((base % host.chw) + host.chw) % host.chw) : 0;
        int rp = PRE == 2 ? (int)(((base % pre.chw) + pre.chw) % pre.chw) : 0;
#pragma unroll 4
        for (int it = 0; it < kTileIters; ++it) {
            const int64_t e = base + it * 256;
            const bool ours = e >= 0 && e < n;
            float4 th = make_float4(0.0f, 0.0f, 0.0f, 0.0f), tp = th;
            if constexpr (HOST == 2) {
                if (ours) th = *reinterpret_cast<const float4*>(host.terms + rh);
                rh += 256;
                while (rh >= host.chw) rh -= host.chw;
            }
            if constexpr (PRE == 2) {
                if (ours) tp = *reinterpret_cast<const float4*>(pre.terms + rp);
                rp += 256;
                while (rp >= pre.chw) rp -= pre.chw;
            }
            float v[4], x[4];
            prefix_draw<HOST>(hrng, hdiv, th, v);
            prefix_draw<PRE>(prng, pdiv, tp, x);
            if (!ours) continue;
            prefix_fold(pre, pfold, fold, e, x, v);
            store_group<true>(const_cast<float*>(fold.y), n, e, v, s, q, STATS);
        }
    }
    if constexpr (STATS) write_partial<kBlock>(s, q, partials, red);
}

static int launch_pair_fold(const sonar_accumulate* acc, const sonar_fold_prefix* pre, int host_kind, const Prefix& host, int64_t n,
                            int64_t elem_offset, hipStream_t st, const char* what) {
    SONAR_REQUIRE(acc && acc->y && pre && n >= 0 && elem_offset >= 0, SONAR_ERR_ARG, "%s: bad argument", what);
    Prefix px;
    const int rc = make_prefix(pre, px, what);
    if (rc != SONAR_OK) return rc;
    SONAR_REQUIRE(n % 4 == 0 && elem_offset % 4 == 0 && aligned16(acc->y) && (host_kind != SONAR_PREFIX_PERLIN || host.chw % 4 == 0) &&
                      (pre->kind != SONAR_PREFIX_PERLIN || pre->chw % 4 == 0),
                  SONAR_ERR_UNSUPPORTED, "%s: hosting a fold prefix needs whole 4-element groups and 16-byte aligned tensors", what);
    if (n == 0) return SONAR_OK;
    const Accum fold{acc->y, acc->y_mul, acc->x_mul};
    const int g = tile_grid(n, elem_offset);
#define SONAR_PF(H, P) \
    do { \
        if (acc->partials) hipLaunchKernelGGL((pair_fold_kernel<H, P, true>), dim3(g), dim3(kBlock), 0, st, fold, n, elem_offset, host, px, acc->partials); \
        else hipLaunchKernelGGL((pair_fold_kernel<H, P, false>), dim3(g), dim3(kBlock), 0, st, fold, n, elem_offset, host, px, acc->partials); \
    } while (0)
    const bool hp = host_kind == SONAR_PREFIX_PERLIN, pp = pre->kind == SONAR_PREFIX_PERLIN;
    if (hp && pp) SONAR_PF(2, 2); else if (hp) SONAR_PF(2, 1); else if (pp) SONAR_PF(1, 2); else SONAR_PF(1, 1);
#undef SONAR_PF
    return check_launch(what);
}

// Normalised fill without a second sweep: MODE 1 re-draws the values and only reduces their statistics (no stores), MODE 2
// re-draws them again, normalises with the decision derived from those statistics and stores -- the tensor is written once
// (GaussianNoiseGenerator / UniformNoiseGenerator followed by scale_noise(normalized=True), py/noise_generation.py:252-260,496-514)
template <Dist D, bool VEC, int MODE>
__global__ void __launch_bounds__(kBlock) stream_fill_norm_kernel(float* out, int64_t n, uint64_t seed, uint64_t stream_id,
                                                                  int64_t elem_offset, Affine aff, double* partials, NormArgs na) {
    kernarg_touch_for(out, n, seed, stream_id, elem_offset, aff, partials, na);
    __shared__ double red[2 * kBlock / 64];
    __shared__ NormDecision sh;
    NormDecision dec{0.f, 1.f, 0, 0};
    if constexpr (MODE == 2) dec = decide_norm<kBlock>(na.partials, kNPart, na.n_total, na.thr_sd, red, &sh);
    const NormFast norm(dec, na.factor);
    double s = 0.0, q = 0.0;
    for_each_group<D>(n, seed, stream_id, elem_offset, [&](int64_t e, float (&v)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = aff(v[k]);
        if constexpr (MODE == 1) {
            if (e >= 0 && e + 4 <= n) {
                const float ps = (v[0] + v[1]) + (v[2] + v[3]);
                const float pq = __builtin_fmaf(v[0], v[0], __builtin_fmaf(v[1], v[1], __builtin_fmaf(v[2], v[2], v[3] * v[3])));
                s += (double)ps;
                q += (double)pq;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (e + k >= 0 && e + k < n) {
                        s += (double)v[k];
                        q += (double)v[k] * (double)v[k];
                    }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = norm(v[k]);
            store_group<VEC>(out, n, e, v, s, q, false);
        }
    });
    if constexpr (MODE == 1) write_partial<kBlock>(s, q, partials, red);
}

template <Dist D>
static int launch_fill_norm(float* out, int64_t n, uint64_t seed, uint64_t stream_id, int64_t elem_offset, Affine aff, float factor,
                            float thr, double* partials, hipStream_t st, const char* what) {
    if (n == 0) return SONAR_OK;
    const bool vec = aligned16(out) && (elem_offset & 3) == 0;
    const int g = tile_grid(n, elem_offset);
    const NormArgs na{partials, n, factor, thr};
    hipLaunchKernelGGL((stream_fill_norm_kernel<D, true, 1>), dim3(g), dim3(kBlock), 0, st, out, n, seed, stream_id, elem_offset, aff, partials,
                       NormArgs{nullptr, 0, 1.0f, 0.0f});
    if (vec)
        hipLaunchKernelGGL((stream_fill_norm_kernel<D, true, 2>), dim3(g), dim3(kBlock), 0, st, out, n, seed, stream_id, elem_offset, aff, partials, na);
    else
        hipLaunchKernelGGL((stream_fill_norm_kernel<D, false, 2>), dim3(g), dim3(kBlock), 0, st, out, n, seed, stream_id, elem_offset, aff, partials, na);
    return check_launch(what);
}

// A sampler's steady state inside a prepared plan (the stream id of the call that follows is known): the final pass of THIS call and
// the statistics pass of the NEXT call as ONE launch (round 6; sonar_philox_noise_ahead_f32).  The final pass is store-bound (a 134 MB
// tensor: 24 us, 21 of them the write itself) with the vector ALUs idle most of the time; the statistics pass is nothing but vector-ALU
// work (15 us as a launch of its own).  A wave draws the next call's tile right behind this call's -- its statistics arithmetic issues
// while the stores of the step before are in flight.  Same grid, same tile -> wave -> step order, same per-thread sums and block
// reduction as stream_fill_norm_kernel<D, VEC, 1>: the partials left for the next call are the bits its own statistics pass would write.
// ALIGNED: whole tiles only (n and elem_offset multiples of kTileElems, 16-byte aligned output): no range checks at all
// EXACT: N(0,1) with factor 1, whose ordinary route stores the raw draws and leaves the (rare) correction to scale_noise_kernel
template <Dist D, bool VEC, bool ALIGNED, bool EXACT = false>
__global__ void __launch_bounds__(kBlock) stream_fill_ahead_kernel(float* out, int64_t n, uint64_t seed, uint64_t stream_id, uint64_t next_stream,
                                                                   int64_t elem_offset, Affine aff, NormArgs na, double* partials_next) {
    kernarg_touch_for(out, n, seed, stream_id, next_stream, elem_offset, aff, na, partials_next);
    __shared__ double red[2 * kBlock / 64];
    __shared__ NormDecision sh;
    const NormDecision dec = decide_norm<kBlock>(na.partials, kNPart, na.n_total, na.thr_sd, red, &sh);
    const NormFast norm(dec, na.factor);
    double s = 0.0, q = 0.0;
    const uint32_t lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * kBlock) >> 6;
    const int64_t first = elem_offset / kTileElems, last = (elem_offset + n - 1) / kTileElems;
    if constexpr (ALIGNED) {
        auto tiles = [&](auto af, auto nm) {
            for (int64_t tile = first + wave; tile <= last; tile += nwaves) {
                TileRng now = rng_stream(seed, stream_id, (uint64_t)tile, lane);
                TileRng nxt = rng_stream(seed, next_stream, (uint64_t)tile, lane);
                float* const o = out + (tile * kTileElems + (int64_t)lane * 4 - elem_offset);
#pragma unroll 4
                for (int it = 0; it < kTileIters; ++it) {
                    float v[4], u[4];
                    if constexpr (D == Dist::Normal) now.normal4(v); else now.uniform4(v);
                    *reinterpret_cast<float4*>(o + it * 256) = make_float4(nm(af(v[0])), nm(af(v[1])), nm(af(v[2])), nm(af(v[3])));
                    if constexpr (D == Dist::Normal) nxt.normal4(u); else nxt.uniform4(u);
#pragma unroll
                    for (int k = 0; k < 4; ++k) u[k] = af(u[k]);
                    const float ps = (u[0] + u[1]) + (u[2] + u[3]);
                    const float pq = __builtin_fmaf(u[0], u[0], __builtin_fmaf(u[1], u[1], __builtin_fmaf(u[2], u[2], u[3] * u[3])));
                    s += (double)ps;
                    q += (double)pq;
                }
            }
        };
        if constexpr (EXACT) with_exact_norm(dec, [&](auto nm) { tiles(AffineT<false>{0.0f, 1.0f, 0.0f}, nm); });
        else with_affine_norm(aff, norm, tiles);
    } else {
        double unused_s = 0.0, unused_q = 0.0;
        for (int64_t tile = first + wave; tile <= last; tile += nwaves) {
            TileRng now = rng_stream(seed, stream_id, (uint64_t)tile, lane);
            TileRng nxt = rng_stream(seed, next_stream, (uint64_t)tile, lane);
            const int64_t base = tile * kTileElems + (int64_t)lane * 4 - elem_offset;
#pragma unroll 4
            for (int it = 0; it < kTileIters; ++it) {
                const int64_t e = base + it * 256;
                float v[4], u[4];
                if constexpr (D == Dist::Normal) now.normal4(v); else now.uniform4(v);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = EXACT ? apply_norm(v[k], dec, 1.0f, false) : norm(aff(v[k]));
                store_group<VEC>(out, n, e, v, unused_s, unused_q, false);
                if constexpr (D == Dist::Normal) nxt.normal4(u); else nxt.uniform4(u);
#pragma unroll
                for (int k = 0; k < 4; ++k) u[k] = aff(u[k]);
                // (EXACT: the statistics the next call's own launch would leave are store_group<VEC>'s -- per element when the shape has no
                // vector path; the other shapes' statistics pass always takes whole groups where it can)
                if ((VEC || !EXACT) && e >= 0 && e + 4 <= n) {
                    const float ps = (u[0] + u[1]) + (u[2] + u[3]);
                    const float pq = __builtin_fmaf(u[0], u[0], __builtin_fmaf(u[1], u[1], __builtin_fmaf(u[2], u[2], u[3] * u[3])));
                    s += (double)ps;
                    q += (double)pq;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (e + k >= 0 && e + k < n) {
                            s += (double)u[k];
                            q += (double)u[k] * (double)u[k];
                        }
                }
            }
        }
    }
    __syncthreads();  // (red: the decision's reduction above, the partial's below)
    write_partial<kBlock>(s, q, partials_next, red);
}

template <Dist D>
static int launch_fill_ahead(float* out, int64_t n, uint64_t seed, uint64_t stream_id, int64_t elem_offset, Affine aff, float factor, float thr,
                             double* partials, int have_stats, uint64_t next_stream, double* partials_next, hipStream_t st, const char* what) {
    if (n == 0) return SONAR_OK;
    const bool vec = aligned16(out) && (elem_offset & 3) == 0;
    const int g = tile_grid(n, elem_offset);
    const NormArgs na{partials, n, factor, thr};
    const bool unit_normal = D == Dist::Normal && factor == 1.0f;
    if (!have_stats) {  // nobody left this call's statistics (the first call, a reseed, another generator drew in between): its own pass
        if (unit_normal) {
            // the ordinary route's first launch (raw draws stored with their statistics; the stores are repeated below): its partials
            const int rc = launch_fill<D>(out, n, seed, stream_id, elem_offset, aff, partials, st, what);
            if (rc != SONAR_OK) return rc;
        } else {
            hipLaunchKernelGGL((stream_fill_norm_kernel<D, true, 1>), dim3(g), dim3(kBlock), 0, st, out, n, seed, stream_id, elem_offset, aff, partials,
                               NormArgs{nullptr, 0, 1.0f, 0.0f});
        }
    }
    const bool whole = vec && n % kTileElems == 0 && elem_offset % kTileElems == 0;
#define SONAR_FA(V, AL) \
    hipLaunchKernelGGL((stream_fill_ahead_kernel<D, V, AL>), dim3(g), dim3(kBlock), 0, st, out, n, seed, stream_id, next_stream, elem_offset, aff, na, partials_next)
    if (unit_normal) {
#define SONAR_FAE(V, AL) \
    hipLaunchKernelGGL((stream_fill_ahead_kernel<Dist::Normal, V, AL, true>), dim3(g), dim3(kBlock), 0, st, out, n, seed, stream_id, next_stream, elem_offset, aff, na, partials_next)
        if (whole) SONAR_FAE(true, true);
        else if (vec) SONAR_FAE(true, false);
        else SONAR_FAE(false, false);
#undef SONAR_FAE
    } else if (whole) SONAR_FA(true, true);
    else if (vec) SONAR_FA(true, false);
    else SONAR_FA(false, false);
#undef SONAR_FA
    return check_launch(what);
}

// ------------------------------------------------------------------------------------------------
// Perlin: lattice term at cell centre (py/noise_generation.py:388-405 with positions == (0.5, 0.5)).
__global__ void __launch_bounds__(kBlock) perlin_terms_kernel(const float* __restrict__ angles, float* terms,
                                                               int64_t planes /*iters*C*/, int H, int W, int blend_mode) {
    kernarg_touch_for(angles, terms, planes, H, W, blend_mode);
    const int64_t total = planes * H * W;
    const int gw = W + 1;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const int64_t p = i / ((int64_t)W * H);
        const float* a = angles + p * (int64_t)(H + 1) * gw + (int64_t)y * gw + x;
        const float a00 = a[0], a01 = a[1], a10 = a[gw], a11 = a[gw + 1];
        // gradient = (cos, sin); corner dot products with (pos - corner), pos = (0.5, 0.5)
        const float d0 = cosf(a00) * 0.5f + sinf(a00) * 0.5f;      // TL: ( 0.5,  0.5)
        const float d1 = cosf(a01) * -0.5f + sinf(a01) * 0.5f;     // TR: (-0.5,  0.5)
        const float d2 = cosf(a10) * 0.5f + sinf(a10) * -0.5f;     // BL: ( 0.5, -0.5)
        const float d3 = cosf(a11) * -0.5f + sinf(a11) * -0.5f;    // BR: (-0.5, -0.5)
        // smooth_step(0.5) = 0.5*0.5*(3 - 2*0.5) = 0.5 exactly
        const float row0 = blend<float>(blend_mode, d0, d1, 0.5f);
        const float row1 = blend<float>(blend_mode, d2, d3, 0.5f);
        terms[i] = blend<float>(blend_mode, row0, row1, 0.5f);
    }
}

// Generate mode: the lattice angles are drawn where they are used.  angle(iteration, c, gy, gx) = 2 pi u with u a
// counter-based uniform (Philox4x32-10, counter = (lattice point, iteration group), key = seed; word `it % 4`), so the four
// corners of a cell are random-access and a neighbouring cell recomputes the same values.  Writes the SUM over iterations
// of the cell-centre terms: one launch instead of angle fill + terms + pre-add.
// block `bid` of `nb` blocks of the lattice's work (the kernel below: the whole grid; perlin_ahead_kernel: its leading blocks)
template <int BLOCK = kBlock>
__device__ __forceinline__ void perlin_lattice_cells(float* terms_sum, int iters, int64_t C, int H, int W, int blend_mode, uint64_t seed,
                                                     uint64_t stream_id, int64_t bid, int64_t nb) {
    const int64_t total = C * H * W;
    const int gw = W + 1;
    for (int64_t i = bid * BLOCK + threadIdx.x; i < total; i += nb * BLOCK) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const int64_t c = i / ((int64_t)W * H);
        const int64_t p00 = (c * (H + 1) + y) * gw + x;  // lattice point index of the top-left corner
        float acc = 0.0f;
        for (int g = 0; g < iters; g += 4) {
            auto words = [&](int64_t pt) { return philox4x32_10((uint32_t)pt, (uint32_t)(pt >> 32), (uint32_t)(g >> 2), (uint32_t)stream_id,
                                                                (uint32_t)seed, (uint32_t)(seed >> 32)); };
            const Philox4 w00 = words(p00), w01 = words(p00 + 1), w10 = words(p00 + gw), w11 = words(p00 + gw + 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // fully unrolled: the word index must be a compile-time constant (registers, not LDS)
                if (g + k >= iters) break;
                // v_sin / v_cos take revolutions: the angle 2 pi u is never formed
                const float u00 = u01(w00.v[k]), u01_ = u01(w01.v[k]), u10 = u01(w10.v[k]), u11 = u01(w11.v[k]);
                const float d0 = __builtin_amdgcn_cosf(u00) * 0.5f + __builtin_amdgcn_sinf(u00) * 0.5f;
                const float d1 = __builtin_amdgcn_cosf(u01_) * -0.5f + __builtin_amdgcn_sinf(u01_) * 0.5f;
                const float d2 = __builtin_amdgcn_cosf(u10) * 0.5f + __builtin_amdgcn_sinf(u10) * -0.5f;
                const float d3 = __builtin_amdgcn_cosf(u11) * -0.5f + __builtin_amdgcn_sinf(u11) * -0.5f;
                const float row0 = blend<float>(blend_mode, d0, d1, 0.5f);
                const float row1 = blend<float>(blend_mode, d2, d3, 0.5f);
                acc += blend<float>(blend_mode, row0, row1, 0.5f);
            }
        }
        terms_sum[i] = acc;
    }
}

__global__ void __launch_bounds__(kBlock) perlin_lattice_kernel(float* terms_sum, int iters, int64_t C, int H, int W, int blend_mode,
                                                                uint64_t seed, uint64_t stream_id) {
    kernarg_touch_for(terms_sum, iters, C, H, W, blend_mode, seed, stream_id);
    perlin_lattice_cells(terms_sum, iters, C, H, W, blend_mode, seed, stream_id, blockIdx.x, gridDim.x);
}

// out[b][i] = base[b][i]/div + terms[0][i] + terms[1][i] + ...   (terms broadcast over batch)
// replay: base read from memory.  One float4 per lane; chw % 4 == 0 on the vector path.
template <bool STATS, int V>
__global__ void __launch_bounds__(kBlock) perlin_apply_kernel(const float* __restrict__ base,
                                                               const float* __restrict__ terms, float* out, int64_t B,
                                                               int64_t chw, int iters, float div_fac, double* partials) {
    kernarg_touch_for(base, terms, out, B, chw, iters, div_fac, partials);
    __shared__ double red[2 * kBlock / 64];
    double s = 0.0, q = 0.0;
    const int64_t n = B * chw;
    const int64_t nv = n / V;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nv; i += stride) {
        const int64_t e = i * V;
        const int64_t r = e % chw;
        float v[V];
        if constexpr (V == 4) {
            const float4 t = *reinterpret_cast<const float4*>(base + e);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
            v[0] = base[e];
        }
#pragma unroll
        for (int k = 0; k < V; ++k) v[k] = v[k] / div_fac;
        for (int it = 0; it < iters; ++it) {
            const float* t = terms + (int64_t)it * chw + r;
            if constexpr (V == 4) {
                const float4 tt = *reinterpret_cast<const float4*>(t);
                v[0] += tt.x; v[1] += tt.y; v[2] += tt.z; v[3] += tt.w;
            } else {
                v[0] += t[0];
            }
        }
        if constexpr (V == 4) {
            *reinterpret_cast<float4*>(out + e) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            out[e] = v[0];
        }
        if constexpr (STATS) {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const double d = v[k];
                s += d; q += d * d;
            }
        }
    }
    if constexpr (STATS) write_partial<kBlock>(s, q, partials, red);
}

// The fast path's tile loop (device-drawn calls: ONE summed term table, vector-aligned latents): wave `wave` of `nwaves` takes the tiles
// first + wave, first + wave + nwaves, ...  MODE as in perlin_generate_kernel.  Shared by that kernel and perlin_ahead_kernel.
template <int MODE, bool STATS, bool ALIGNED, typename DIV, typename NORM>
__device__ __forceinline__ void perlin_fast_tiles(const float* __restrict__ terms, float* out, int64_t n, int64_t chw, uint64_t seed,
                                                  uint64_t stream_id, int64_t elem_offset, const NORM& norm, const DIV& divide,
                                                  const Accum& acc, int64_t wave, int64_t nwaves, double& s, double& q) {
    const uint32_t lane = threadIdx.x & 63;
    const int64_t first = elem_offset / kTileElems, last = (elem_offset + n - 1) / kTileElems;
    const int ichw = (int)chw;
    for (int64_t tile = first + wave; tile <= last; tile += nwaves) {
        TileRng rng = rng_stream(seed, stream_id, (uint64_t)tile, lane);
        const int64_t base = tile * kTileElems + (int64_t)lane * 4 - elem_offset;
        // position of the tile's first vector inside the latent; the prefetch cursor walks on in steps of 256 elements and wraps
        // at the latent's end (a tile may straddle two latents when chw is not a multiple of the tile)
        int rp = (int)(((base % chw) + chw) % chw);
        const float4* const trow = reinterpret_cast<const float4*>(terms + rp);  // ALIGNED: the tile's vectors sit at trow[it * 64]
        int fetched = 0;
        auto next_terms = [&]() {
            if constexpr (ALIGNED) return trow[64 * fetched++];  // unrolled callers: constant offsets in the load instructions
            const float4 t = *reinterpret_cast<const float4*>(terms + rp);
            rp += 256;
            rp -= rp >= ichw ? ichw : 0;  // chw >= 256 (launcher)
            return t;
        };
        float4 cur[4], nxt[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] = next_terms();
        float ts = 0.0f, tq = 0.0f;  // the tile's 64 values per lane in fp32, folded into the fp64 sums once per tile
#pragma unroll
        for (int g = 0; g < kTileIters / 4; ++g) {
            if (g + 1 < kTileIters / 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) nxt[j] = next_terms();
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[4];
                uint32_t r[4];
                rng.words4_high(r);
                const int64_t e = base + (4 * g + j) * 256;
                if constexpr (!ALIGNED) {
                    if (e < 0 || e >= n) continue;  // the two ends of a shard that does not start / end on a tile (whole groups of 4)
                }
                v[0] = divide.from_word(r[0], cur[j].x); v[1] = divide.from_word(r[1], cur[j].y);
                v[2] = divide.from_word(r[2], cur[j].z); v[3] = divide.from_word(r[3], cur[j].w);
                if constexpr (MODE == 2) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = norm(v[k]);
                }
                if constexpr (MODE != 1) {
                    if constexpr (MODE == 0) accumulate_group<true>(acc, n, e, v);
                    if constexpr (ALIGNED) {  // whole tiles inside the shard: no range checks (store_group's were a branch per group)
                        *reinterpret_cast<float4*>(out + e) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        double unused_s = 0.0, unused_q = 0.0;
                        store_group<true>(out, n, e, v, unused_s, unused_q, false);
                    }
                }
                if constexpr (STATS || MODE == 1) {
                    ts += (v[0] + v[1]) + (v[2] + v[3]);
                    tq = __builtin_fmaf(v[0], v[0], __builtin_fmaf(v[1], v[1], __builtin_fmaf(v[2], v[2], __builtin_fmaf(v[3], v[3], tq))));
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
        }
        if constexpr (STATS || MODE == 1) {
            s += (double)ts;
            q += (double)tq;
        }
    }
}

// The final pass of one call (MODE 2: `terms`, `stream_id`, normalised, stored) and the statistics pass of the next (MODE 1: `terms_next`,
// `next_stream`, summed) for the same tiles in ONE wave, step by step (round 6): the statistics arithmetic issues while the stores of the
// steps before are in flight.  Per tile and lane the same values in the same order as the two separate loops above -- the stored bits and
// the (sum, sumsq) this thread adds up are theirs.  Whole tiles per latent only (ALIGNED).
template <typename DIV, typename NORM>
__device__ __forceinline__ void perlin_fast_tiles_pair(const float* __restrict__ terms, const float* __restrict__ terms_next, float* out, int64_t n,
                                                       int64_t chw, uint64_t seed, uint64_t stream_id, uint64_t next_stream, int64_t elem_offset,
                                                       const NORM& norm, const DIV& divide, int64_t wave, int64_t nwaves, double& s, double& q) {
    const uint32_t lane = threadIdx.x & 63;
    const int64_t first = elem_offset / kTileElems, last = (elem_offset + n - 1) / kTileElems;
    for (int64_t tile = first + wave; tile <= last; tile += nwaves) {
        TileRng rng = rng_stream(seed, stream_id, (uint64_t)tile, lane);
        TileRng rnx = rng_stream(seed, next_stream, (uint64_t)tile, lane);
        const int64_t base = tile * kTileElems + (int64_t)lane * 4 - elem_offset;
        const int rp = (int)(((base % chw) + chw) % chw);
        const float4* const trow = reinterpret_cast<const float4*>(terms + rp);
        const float4* const tnxt = reinterpret_cast<const float4*>(terms_next + rp);
        float4 cur[4], nxt[4], curn[4], nxtn[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            cur[j] = trow[64 * j];
            curn[j] = tnxt[64 * j];
        }
        float ts = 0.0f, tq = 0.0f;
#pragma unroll
        for (int g = 0; g < kTileIters / 4; ++g) {
            if (g + 1 < kTileIters / 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    nxt[j] = trow[64 * (4 * (g + 1) + j)];
                    nxtn[j] = tnxt[64 * (4 * (g + 1) + j)];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t r[4];
                rng.words4_high(r);
                *reinterpret_cast<float4*>(out + base + (4 * g + j) * 256) =
                    make_float4(norm(divide.from_word(r[0], cur[j].x)), norm(divide.from_word(r[1], cur[j].y)),
                                norm(divide.from_word(r[2], cur[j].z)), norm(divide.from_word(r[3], cur[j].w)));
                uint32_t w[4];
                rnx.words4_high(w);
                const float v0 = divide.from_word(w[0], curn[j].x), v1 = divide.from_word(w[1], curn[j].y);
                const float v2 = divide.from_word(w[2], curn[j].z), v3 = divide.from_word(w[3], curn[j].w);
                ts += (v0 + v1) + (v2 + v3);
                tq = __builtin_fmaf(v0, v0, __builtin_fmaf(v1, v1, __builtin_fmaf(v2, v2, __builtin_fmaf(v3, v3, tq))));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                cur[j] = nxt[j];
                curn[j] = nxtn[j];
            }
        }
        s += (double)ts;
        q += (double)tq;
    }
}

// Normalisation folded into the generating pass (SURVEY.md §8d "stats-before-write"): MODE 0 writes the
// raw values (+ optional statistics), MODE 1 only reduces the statistics (no stores), MODE 2 re-draws the
// same values, normalises them with the decision derived from `norm_partials` and writes the final tensor.
// generate: base u ~ U[0,1) drawn on device.  VEC: chw % 4 == 0, aligned pointers, elem_offset % 4 == 0.
// FAST (what every device-drawn call has: ONE summed term table, vector-aligned latents): the tile's 16 term vectors are requested
// four iterations ahead of their use by a cursor of their own (the general loop below waits for each load right where it issues it:
// at 4 waves per SIMD that wait, not the arithmetic, set the kernel's time), and there is no per-element tail path.
template <int MODE, bool VEC, bool STATS, bool FAST = false, bool ALIGNED = false>
__global__ void __launch_bounds__(kBlock) perlin_generate_kernel(const float* __restrict__ terms, float* out, int64_t B,
                                                                  int64_t chw, int iters, float div_fac, uint64_t seed,
                                                                  uint64_t stream_id, int64_t elem_offset, double* partials,
                                                                  NormArgs na, Accum acc) {
    kernarg_touch_for(terms, out, B, chw, iters, div_fac, seed, stream_id, elem_offset, partials, na, acc);
    __shared__ double red[2 * kBlock / 64];
    __shared__ NormDecision sh;
    NormDecision dec{0.f, 1.f, 0, 0};
    if constexpr (MODE == 2) dec = decide_norm<kBlock>(na.partials, kNPart, na.n_total, na.thr_sd, red, &sh);
    const NormFast norm(dec, na.factor);
    const Divider divide(div_fac);
    double s = 0.0, q = 0.0;
    const int64_t n = B * chw;
    const uint32_t lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * kBlock) >> 6;
    const int64_t first = elem_offset / kTileElems, last = (elem_offset + n - 1) / kTileElems;
    const int ichw = (int)chw;  // launcher guarantees chw < 2^31
    if constexpr (FAST) {
        static_assert(VEC, "the fast path is a vector path");
        with_divider(divide, [&](auto dv) {
            if constexpr (MODE == 2) {  // the normalisation's switches decided once per launch, not per value
                with_norm_flags(norm, [&](auto nm) {
                    perlin_fast_tiles<MODE, STATS, ALIGNED>(terms, out, n, chw, seed, stream_id, elem_offset, nm, dv, acc, wave, nwaves, s, q);
                });
            } else {
                perlin_fast_tiles<MODE, STATS, ALIGNED>(terms, out, n, chw, seed, stream_id, elem_offset, norm, dv, acc, wave, nwaves, s, q);
            }
        });
        if constexpr (STATS || MODE == 1) write_partial<kBlock>(s, q, partials, red);
        return;
    }
    for (int64_t tile = first + wave; tile <= last; tile += nwaves) {
        TileRng rng = rng_stream(seed, stream_id, (uint64_t)tile, lane);
        const int64_t base = tile * kTileElems + (int64_t)lane * 4 - elem_offset;
        // position inside the latent (terms repeat per latent; shards start on a latent boundary), kept incrementally
        int r = (int)(((base % chw) + chw) % chw);
#pragma unroll 4
        for (int it = 0; it < kTileIters; ++it) {
            float v[4];
            rng.uniform4(v);
            const int64_t e = base + it * 256;
            const int rr = r;
            r += 256;
            while (r >= ichw) r -= ichw;
            if (e + 4 <= 0 || e >= n) continue;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = divide(v[k]);
            for (int t_i = 0; t_i < iters; ++t_i) {
                const float* t = terms + (int64_t)t_i * chw;
                if (VEC) {
                    const float4 tt = *reinterpret_cast<const float4*>(t + rr);
                    v[0] += tt.x; v[1] += tt.y; v[2] += tt.z; v[3] += tt.w;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] += t[(rr + k) % ichw];
                }
            }
            if constexpr (MODE == 2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = norm(v[k]);
            }
            if constexpr (MODE == 1) {
                if (VEC || (e >= 0 && e + 4 <= n)) {
                    const float ps = (v[0] + v[1]) + (v[2] + v[3]);
                    const float pq = __builtin_fmaf(v[0], v[0], __builtin_fmaf(v[1], v[1], __builtin_fmaf(v[2], v[2], v[3] * v[3])));
                    s += (double)ps;
                    q += (double)pq;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (e + k >= 0 && e + k < n) {
                            s += (double)v[k];
                            q += (double)v[k] * (double)v[k];
                        }
                }
            } else {
                if constexpr (MODE == 0) accumulate_group<VEC>(acc, n, e, v);
                store_group<VEC>(out, n, e, v, s, q, STATS);
            }
        }
    }
    if constexpr (STATS || MODE == 1) write_partial<kBlock>(s, q, partials, red);
}

static int launch_perlin_apply(const float* base, const float* terms, float* out, int64_t B, int64_t chw, int64_t iters,
                               float div_fac, double* partials, hipStream_t st, const char* what) {
    const int64_t n = B * chw;
    if (n == 0) return SONAR_OK;
    const bool vec = (chw % 4 == 0) && aligned16(out) && aligned16(terms) && aligned16(base);
    const int g = (int)std::min<int64_t>(kNPart, grid_for(vec ? n / 4 : n, kBlock));
#define SONAR_PA(S, V) \
    hipLaunchKernelGGL((perlin_apply_kernel<S, V>), dim3(g), dim3(kBlock), 0, st, base, terms, out, B, chw, (int)iters, div_fac, partials)
    if (vec) {
        if (partials) SONAR_PA(true, 4); else SONAR_PA(false, 4);
    } else {
        if (partials) SONAR_PA(true, 1); else SONAR_PA(false, 1);
    }
#undef SONAR_PA
    return check_launch(what);
}

template <int MODE>
static int launch_perlin_generate(const float* terms, float* out, int64_t B, int64_t chw, int64_t iters, float div_fac,
                                  uint64_t seed, uint64_t stream_id, int64_t elem_offset, double* partials, NormArgs na,
                                  hipStream_t st, const char* what, Accum acc = kNoAccum) {
    const int64_t n = B * chw;
    if (n == 0) return SONAR_OK;
    const bool vec = (chw % 4 == 0) && (MODE == 1 || aligned16(out)) && (iters == 0 || aligned16(terms)) && (elem_offset % 4 == 0) &&
                     aligned16(acc.y);
    const int g = tile_grid(n, elem_offset);
    const bool fast = vec && iters == 1 && chw >= 256;
    const bool aligned = fast && chw % kTileElems == 0 && elem_offset % kTileElems == 0;  // every tile inside one latent and inside the shard
#define SONAR_PG(V, S) \
    hipLaunchKernelGGL((perlin_generate_kernel<MODE, V, S>), dim3(g), dim3(kBlock), 0, st, terms, out, B, chw, (int)iters, div_fac, seed, stream_id, elem_offset, partials, na, acc)
#define SONAR_PGF(S, AL) \
    hipLaunchKernelGGL((perlin_generate_kernel<MODE, true, S, true, AL>), dim3(g), dim3(kBlock), 0, st, terms, out, B, chw, (int)iters, div_fac, seed, stream_id, elem_offset, partials, na, acc)
    if (aligned) {
        if (partials && MODE == 0) SONAR_PGF(true, true); else SONAR_PGF(false, true);
    } else if (fast) {
        if (partials && MODE == 0) SONAR_PGF(true, false); else SONAR_PGF(false, false);
    } else if (vec) {
        if (partials && MODE == 0) SONAR_PG(true, true); else SONAR_PG(true, false);
    } else {
        if (partials && MODE == 0) SONAR_PG(false, true); else SONAR_PG(false, false);
    }
#undef SONAR_PG
#undef SONAR_PGF
    return check_launch(what);
}

// A sampler's steady state at launch-bound batch sizes (prepared plans: the stream ids of the calls that follow are known): the three
// dependent launches of a normalised Perlin call -- lattice, statistics pass, final pass -- become ONE, with nothing inside the launch
// depending on anything else inside it.  Blocks [lat_blocks, ...) run the final pass of THIS call (lattice `terms`, statistics `partials`:
// both left by earlier launches) and then the statistics pass of the NEXT call (its lattice `terms_next` left by the previous launch) into
// `partials_next`; blocks [0, lat_blocks) compute a lattice for a LATER call into `lattice_out`.  Same tile -> wave -> slot mapping, same
// arithmetic as perlin_generate_kernel<1 / 2, FAST, ALIGNED> and perlin_lattice_kernel: the same bits.
struct PerlinAhead {
    const float* terms;
    float* out;
    int64_t n, chw;
    float div_fac;
    uint64_t seed, stream_id;
    int64_t elem_offset;
    NormArgs na;                  // this call's statistics (complete before the launch) and its normalisation
    uint64_t next_stream_id;      // statistics of the next call: tiles drawn with this stream id against terms_next
    const float* terms_next;      // nullable: no statistics ahead
    double* partials_next;
    float* lattice_out;           // nullable: no lattice ahead
    int lat_blocks, tile_blocks, lat_iters, blend_mode;
    int fused;                    // one wave runs both passes for its tiles (the bandwidth-bound sizes); 0: separate, interleaved blocks
    int64_t C;
    int H, W;
    uint64_t lattice_stream_id;
};

__global__ void __launch_bounds__(kBlock) perlin_ahead_kernel(PerlinAhead a) {
    kernarg_touch_for(a);
    __shared__ double red[2 * kBlock / 64];
    __shared__ NormDecision sh;
    if ((int)blockIdx.x < a.lat_blocks) {
        perlin_lattice_cells(a.lattice_out, a.lat_iters, a.C, a.H, a.W, a.blend_mode, a.seed, a.lattice_stream_id, blockIdx.x, a.lat_blocks);
        return;
    }
    // the final pass of this call and the statistics pass of the next one are independent: separate blocks, so that at the launch-bound
    // sizes a wave runs ONE tile's chain, not two back to back -- and INTERLEAVED block by block (round 5), so that at the bandwidth-bound
    // sizes the two kinds are resident side by side: the final pass waits for its stores (a 134 MB tensor: 29 us, 21 of them the
    // write itself) with the vector ALUs idle, the statistics pass is nothing but vector ALU work (13 us as a launch of its own)
    const int nb = a.tile_blocks;
    const int blk = (int)blockIdx.x - a.lat_blocks;
    const Divider divide(a.div_fac);
    double s = 0.0, q = 0.0;
    if (a.fused) {
        // bandwidth-bound sizes (round 6): ONE wave runs this call's final pass and the next call's statistics pass for its tiles, step by
        // step -- the interleaved blocks below left the statistics waves done after 13 us and the storing waves on their own for the rest
        const int64_t wave = ((int64_t)blk * kBlock + threadIdx.x) >> 6, nwaves = ((int64_t)nb * kBlock) >> 6;
        const NormDecision dec = decide_norm<kBlock>(a.na.partials, kNPart, a.na.n_total, a.na.thr_sd, red, &sh);
        const NormFast norm(dec, a.na.factor);
        with_divider(divide, [&](auto dv) {
            with_norm_flags(norm, [&](auto nm) {
                perlin_fast_tiles_pair(a.terms, a.terms_next, a.out, a.n, a.chw, a.seed, a.stream_id, a.next_stream_id, a.elem_offset, nm, dv, wave, nwaves,
                                       s, q);
            });
        });
        __syncthreads();  // (red: the decision's reduction above, the partial's below)
        write_partial_at<kBlock>(s, q, a.partials_next, red, blk, nb);
        return;
    }
    const bool ahead = a.terms_next != nullptr && (blk & 1);
    const int bid = a.terms_next != nullptr ? blk >> 1 : blk;
    const int64_t wave = ((int64_t)bid * kBlock + threadIdx.x) >> 6, nwaves = ((int64_t)nb * kBlock) >> 6;
    if (!ahead) {
        const NormDecision dec = decide_norm<kBlock>(a.na.partials, kNPart, a.na.n_total, a.na.thr_sd, red, &sh);
        const NormFast norm(dec, a.na.factor);
        with_divider(divide, [&](auto dv) {
            with_norm_flags(norm, [&](auto nm) {
                perlin_fast_tiles<2, false, true>(a.terms, a.out, a.n, a.chw, a.seed, a.stream_id, a.elem_offset, nm, dv, kNoAccum, wave, nwaves, s, q);
            });
        });
    } else {
        const NormFast norm(NormDecision{0.f, 1.f, 0, 0}, 1.0f);
        with_divider(divide, [&](auto dv) {
            perlin_fast_tiles<1, false, true>(a.terms_next, nullptr, a.n, a.chw, a.seed, a.next_stream_id, a.elem_offset, norm, dv, kNoAccum, wave,
                                              nwaves, s, q);
        });
        write_partial_at<kBlock>(s, q, a.partials_next, red, bid, nb);
    }
}

// ------------------------------------------------------------------------------------------------
// Resampling (F.interpolate semantics, ATen UpSampleKernel index/weight rules).
struct alignas(16) Lin {
    int i0, i1;
    float w0, w1;
};
__device__ __forceinline__ Lin lin_coord(int dst, float scale, int in_size) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;   // align_corners=False
    src = src < 0.0f ? 0.0f : src;
    int i0 = (int)floorf(src);
    i0 = i0 < in_size - 1 ? i0 : in_size - 1;
    float l1 = src - (float)i0;
    l1 = fminf(fmaxf(l1, 0.0f), 1.0f);
    Lin r;
    r.i0 = i0;
    r.i1 = i0 + 1 < in_size - 1 ? i0 + 1 : in_size - 1;
    r.w0 = 1.0f - l1;
    r.w1 = l1;
    return r;
}
__device__ __forceinline__ float bilerp(const float* __restrict__ plane, int w, const Lin& ly, const Lin& lx) {
    const float* r0 = plane + (int64_t)ly.i0 * w;
    const float* r1 = plane + (int64_t)ly.i1 * w;
    const float t0 = r0[lx.i0] * lx.w0 + r0[lx.i1] * lx.w1;
    const float t1 = r1[lx.i0] * lx.w0 + r1[lx.i1] * lx.w1;
    return t0 * ly.w0 + t1 * ly.w1;
}
__device__ __forceinline__ int nearest_exact_idx(int dst, float scale, int in_size) {
    const int i = (int)floorf(((float)dst + 0.5f) * scale);
    return i < in_size - 1 ? i : in_size - 1;
}
__device__ __forceinline__ int nearest_idx(int dst, float scale, int in_size) {  // legacy "nearest": floor(dst * scale)
    const int i = (int)floorf((float)dst * scale);
    return i < in_size - 1 ? i : in_size - 1;
}
// align_corners=True bilinear: src = dst * (in - 1) / (out - 1)
__device__ __forceinline__ Lin lin_coord_aligned(int dst, float scale, int in_size) {
    const float src = scale * (float)dst;
    int i0 = (int)src;
    i0 = i0 < in_size - 1 ? i0 : in_size - 1;
    const float l1 = src - (float)i0;
    Lin r;
    r.i0 = i0;
    r.i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    r.w0 = 1.0f - l1;
    r.w1 = l1;
    return r;
}
// bicubic (A = -0.75, ATen UpSampleBicubic2d): taps at floor(src) - 1 .. + 2 with clamped indices; weights from t = src - floor(src)
struct Cubic {
    int i[4];
    float c[4];
};
__device__ __forceinline__ Cubic cubic_coord(int dst, float scale, int in_size, bool aligned) {
    const float A = -0.75f;
    const float src = aligned ? scale * (float)dst : scale * ((float)dst + 0.5f) - 0.5f;
    const float fl = floorf(src);
    const int i0 = (int)fl;
    const float t = src - fl, u = 1.0f - t;
    Cubic r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.i[k] = min(max(i0 - 1 + k, 0), in_size - 1);
    const float t1 = t + 1.0f, u1 = u + 1.0f;
    r.c[0] = ((A * t1 - 5.0f * A) * t1 + 8.0f * A) * t1 - 4.0f * A;
    r.c[1] = ((A + 2.0f) * t - (A + 3.0f)) * t * t + 1.0f;
    r.c[2] = ((A + 2.0f) * u - (A + 3.0f)) * u * u + 1.0f;
    r.c[3] = ((A * u1 - 5.0f * A) * u1 + 8.0f * A) * u1 - 4.0f * A;
    return r;
}
__device__ __forceinline__ float bicubic(const float* __restrict__ plane, int w, const Cubic& cy, const Cubic& cx) {
    float rows[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float* r = plane + (int64_t)cy.i[k] * w;
        rows[k] = r[cx.i[0]] * cx.c[0] + r[cx.i[1]] * cx.c[1] + r[cx.i[2]] * cx.c[2] + r[cx.i[3]] * cx.c[3];
    }
    return rows[0] * cy.c[0] + rows[1] * cy.c[1] + rows[2] * cy.c[2] + rows[3] * cy.c[3];
}
__device__ __forceinline__ float area_sample(const float* __restrict__ plane, int h, int w, int H, int W, int y, int x) {
    // adaptive average pooling window: [floor(i*in/out), ceil((i+1)*in/out)); 32-bit quotients whenever the products fit (a 64-bit
    // division is ~80 instructions, four of them per value were this mode's cost)
    int y0, y1, x0, x1;
    if ((int64_t)(H + 1) * h < (1ll << 31) && (int64_t)(W + 1) * w < (1ll << 31)) {
        y0 = (int)(((unsigned)y * (unsigned)h) / (unsigned)H), y1 = (int)((((unsigned)y + 1u) * (unsigned)h + (unsigned)H - 1u) / (unsigned)H);
        x0 = (int)(((unsigned)x * (unsigned)w) / (unsigned)W), x1 = (int)((((unsigned)x + 1u) * (unsigned)w + (unsigned)W - 1u) / (unsigned)W);
    } else {
        y0 = (int)(((int64_t)y * h) / H), y1 = (int)((((int64_t)y + 1) * h + H - 1) / H);
        x0 = (int)(((int64_t)x * w) / W), x1 = (int)((((int64_t)x + 1) * w + W - 1) / W);
    }
    float acc = 0.0f;
    for (int yy = y0; yy < y1; ++yy)
        for (int xx = x0; xx < x1; ++xx) acc += plane[(int64_t)yy * w + xx];
    return acc / (float)((y1 - y0) * (x1 - x0));
}

template <bool STATS>
__global__ void __launch_bounds__(kBlock) resample_acc_kernel(float* dst, const float* __restrict__ src, int64_t planes,
                                                               int H, int W, int h, int w, float scale, int mode,
                                                               int accumulate, double* partials) {
    kernarg_touch_for(dst, src, planes, H, W, h, w, scale, mode, accumulate, partials);
    __shared__ double red[2 * kBlock / 64];
    double s = 0.0, q = 0.0;
    const int64_t total = planes * H * W;
    const bool aligned = mode == 5 || mode == 6;  // align_corners=True: (in - 1) / (out - 1), 0 for a single output
    const float sy = aligned ? (H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.0f) : (float)h / (float)H;
    const float sx = aligned ? (W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.0f) : (float)w / (float)W;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const int64_t p = i / ((int64_t)W * H);
        const float* plane = src + p * (int64_t)h * w;
        float v;
        if (mode == 0) {
            v = bilerp(plane, w, lin_coord(y, sy, h), lin_coord(x, sx, w));
        } else if (mode == 1) {
            v = plane[(int64_t)nearest_exact_idx(y, sy, h) * w + nearest_exact_idx(x, sx, w)];
        } else if (mode == 2) {
            v = area_sample(plane, h, w, H, W, y, x);
        } else if (mode == 3) {
            v = plane[(int64_t)nearest_idx(y, sy, h) * w + nearest_idx(x, sx, w)];
        } else if (mode == 6) {
            v = bilerp(plane, w, lin_coord_aligned(y, sy, h), lin_coord_aligned(x, sx, w));
        } else if (H == h && W == w) {
            v = plane[(int64_t)y * w + x];  // same size: ATen copies
        } else {
            v = bicubic(plane, w, cubic_coord(y, sy, h, aligned), cubic_coord(x, sx, w, aligned));
        }
        if (scale != 1.0f) v = v * scale;
        if (accumulate) v = dst[i] + v;
        dst[i] = v;
        if constexpr (STATS) {
            const double d = v;
            s += d; q += d * d;
        }
    }
    if constexpr (STATS) write_partial<kBlock>(s, q, partials, red);
}

// Pyramid, generate mode: out = N(0,1)[stream] * base_scale + sum_l bilerp(level_l) * w_l.  A level with the latent's own
// size (the reference's i = 0 level: ratio r^0 = 1) is one more independent N(0,1) per element times w0; the sum of the two
// independent normals is drawn as ONE normal scaled by sqrt(1 + w0^2) (same distribution, half the generator work).
constexpr int kMaxLevels = 12;
struct PyramidLevels {
    const float* ptr[kMaxLevels];
    int h[kMaxLevels], w[kMaxLevels];
    float weight[kMaxLevels];
    int count;       // small-grid levels
    int supplied;    // ... of them with a grid passed in (ptr != nullptr: replay mode)
    unsigned long long draw_stream[kMaxLevels];  // ptr == nullptr: the level grid is drawn by the plane kernel from this stream id
    int fullres;     // number of leading full-resolution levels folded into the base draw (0 or 1)
    float fullres_weight;
    float base_scale;  // sqrt(1 + fullres_weight^2) when fullres else 1
};

// MODE as in perlin_generate_kernel (0 raw (+stats), 1 statistics only, 2 normalised final).
template <int MODE, bool STATS>
__global__ void __launch_bounds__(kBlock) pyramid_generate_kernel(float* out, int64_t planes, int H, int W,
                                                                  PyramidLevels lv, int mode, uint64_t seed,
                                                                  uint64_t stream_id, int64_t elem_offset,
                                                                  double* partials, NormArgs na) {
    kernarg_touch_for(out, planes, H, W, lv, mode, seed, stream_id, elem_offset, partials, na);
    __shared__ double red[2 * kBlock / 64];
    __shared__ NormDecision sh;
    NormDecision dec{0.f, 1.f, 0, 0};
    if constexpr (MODE == 2) dec = decide_norm<kBlock>(na.partials, kNPart, na.n_total, na.thr_sd, red, &sh);
    const NormFast norm(dec, na.factor);
    double s = 0.0, q = 0.0;
    const int64_t n = planes * (int64_t)H * W;
    // W % 4 == 0 and elem_offset % 4 == 0 (launcher): a 4-group never straddles a row or the shard ends
    const uint32_t lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * kBlock) >> 6;
    const int64_t first = elem_offset / kTileElems, last = (elem_offset + n - 1) / kTileElems;
    for (int64_t tile = first + wave; tile <= last; tile += nwaves) {
        TileRng rng = rng_stream(seed, stream_id, (uint64_t)tile, lane);
        const int64_t base = tile * kTileElems + (int64_t)lane * 4 - elem_offset;
        for (int it = 0; it < kTileIters; ++it) {
            float v[4];
            rng.normal4(v);
            if (lv.fullres) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] *= lv.base_scale;
            }
            const int64_t e = base + it * 256;
            if (e < 0 || e >= n) continue;
            const int x4 = (int)(e % W);
            const int y = (int)((e / W) % H);
            const int64_t p = e / ((int64_t)W * H);
            for (int l = 0; l < lv.count; ++l) {
                const int h = lv.h[l], w = lv.w[l];
                const float* plane = lv.ptr[l] + p * (int64_t)h * w;
                const float sy = (float)h / (float)H, sx = (float)w / (float)W;
                const float wt = lv.weight[l];
                if (mode == 0) {
                    const Lin ly = lin_coord(y, sy, h);
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] += bilerp(plane, w, ly, lin_coord(x4 + k, sx, w)) * wt;
                } else if (mode == 1) {
                    const float* row = plane + (int64_t)nearest_exact_idx(y, sy, h) * w;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] += row[nearest_exact_idx(x4 + k, sx, w)] * wt;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] += area_sample(plane, h, w, H, W, y, x4 + k) * wt;
                }
            }
            if constexpr (MODE == 2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = norm(v[k]);
            }
            if constexpr (MODE == 1) {
                const float ps = (v[0] + v[1]) + (v[2] + v[3]);
                const float pq = __builtin_fmaf(v[0], v[0], __builtin_fmaf(v[1], v[1], __builtin_fmaf(v[2], v[2], v[3] * v[3])));
                s += (double)ps;
                q += (double)pq;
            } else {
                store_group<true>(out, n, e, v, s, q, STATS);
            }
        }
    }
    if constexpr (STATS || MODE == 1) write_partial<kBlock>(s, q, partials, red);
}

// One workgroup per plane, the plane's level grids staged in LDS (they are re-read ~4 H W / (h w) times each by the
// bilinear gathers); the four waves stride over the plane's RNG tiles.  Same values as the flat kernel above (to rounding).
// Requires W % 4 == 0 and elem_offset % (H * W) == 0 (whole planes; a plane may straddle RNG tiles).
// XROWS (bilinear, when it fits): every level grid is first stretched along x to the full width W in LDS (h_l rows of W
// floats), so an output needs, per level, two conflict-free 16-byte reads and two FMAs per value instead of four gathers, a
// coordinate-table read and eleven arithmetic instructions; the level weight is folded into the y weights.
constexpr size_t kPyramidLdsBudget = 64 * 1024;
// The grids and the stretched rows of a plane take up to 64 KB of LDS, so two workgroups fit a CU: with 256 threads that is two
// waves per SIMD (one at batch 64, where there is a plane per CU) and the tile loop runs at the latency of its dependent chains.
// kPyrParts 256-thread parts share one plane's LDS instead.
#ifndef SONAR_PYR_PARTS
#define SONAR_PYR_PARTS 2
#endif
constexpr int kPyrParts = SONAR_PYR_PARTS;
#ifndef SONAR_PYR_STRETCH
#define SONAR_PYR_STRETCH 1
#endif
// items a thread stretches along x per round (XROWS).  Re-swept once the column-fixed path had made an item cheap (same box, batch 512 / 64,
// normalised call): 8 items 92.1 / 22.3 us, 4 items (the setting while an item was a table read and two dependent gathers) 89.7 / 21.9, 3 items
// 88.6 / 22.0, 2 items 88.5 / 21.7, **1 item 87.3 / 21.6**
constexpr int kPyrStretch = SONAR_PYR_STRETCH;
#ifndef SONAR_PYR_GROUP
#define SONAR_PYR_GROUP 1
#endif
#ifndef SONAR_PYR_GROUP_PRE
#define SONAR_PYR_GROUP_PRE 1
#endif
constexpr int kPyrGroup = SONAR_PYR_GROUP, kPyrGroupPre = SONAR_PYR_GROUP_PRE;  // burst steps whose level reads are in flight together (XROWS)
constexpr int kPyrBlock = kPyrParts * kBlock;
static_assert((kPyrParts & (kPyrParts - 1)) == 0 && kTileIters % kPyrParts == 0, "parts split a tile's burst evenly");
#ifndef SONAR_PYR_UNROLL_N
#define SONAR_PYR_UNROLL_N 1
#endif
#define SONAR_PYR_PRAGMA(x) _Pragma(#x)
#define SONAR_PYR_UNROLL_(n) SONAR_PYR_PRAGMA(unroll n)
#define SONAR_PYR_UNROLL_X(n) SONAR_PYR_UNROLL_(n)
#define SONAR_PYR_UNROLL SONAR_PYR_UNROLL_X(SONAR_PYR_UNROLL_N)

// fold (nullable y): the values are folded into a chain's running sum, out == fold.y (sonar_pyramid_generate_acc_f32); PRE != 0: the
// chain's previous item rides along (Prefix above) -- its generator shares this kernel's tile keying, so its state simply walks the
// same (tile, iteration) sequence.
// A Perlin lattice computed in `blocks` extra leading workgroups of a plane-kernel launch: the lattice of a LATER call's hosted Perlin item
// (sonar_pyramid_generate_acc_ahead_f32) -- independent of everything else in the launch.  blocks == 0: none.
// N words of a stream passed over as straight-line code: a counted loop around one multiply-add spent 96 cycles per word on its own
// bookkeeping (vector-exec loop, trace build of round 5: 3.1 k cycles for the 32 words of half a burst)
template <int N>
__device__ __forceinline__ void skip_words(TileRng& r) {
#pragma unroll
    for (int k = 0; k < N; ++k) r.next();
}

struct LatticeJob {
    float* out;
    int blocks, iters, blend_mode;
    int64_t C;
    uint64_t seed, stream_id;
};
static constexpr LatticeJob kNoLattice{nullptr, 0, 0, 0, 0, 0, 0};

// AHEAD (round 6, sonar_pyramid_noise_ahead_f32): a normalised call inside a prepared plan as ONE launch.  Workgroups [0, ah.main_blocks)
// run THIS call's planes and store them normalised -- the statistics were left by the previous call's launch -- with scale_noise_kernel's own
// operation sequence (ExactNorm: the same bits as the in-place pass); the workgroups behind them run the NEXT call's planes (its level
// table `ah.lv`, its stream ids) without storing anything and leave that call's (sum, sumsq) partials, workgroup by workgroup what its own
// generating launch would leave.  At the launch-bound sizes the two kinds sit side by side on the chip (one plane workgroup per CU before).
struct PyrAhead {
    NormArgs na;
    int npart;                      // partial pairs of THIS call's statistics (the workgroups of the launch that left them)
    PyramidLevels lv;               // the next call's level table, stream id and grid size
    unsigned long long stream_id;
    int grid_floats;
    int main_blocks;
    double* partials;               // the next call's statistics
};
// the decision as scale_noise_kernel's 256-thread blocks take it (same strides, same reduction order: same bits), inside a larger block
template <int SUB, int BLOCK>
__device__ __forceinline__ NormDecision decide_norm_as(const double* __restrict__ partials, int64_t npart, int64_t n_total, float thr_sd, double* red,
                                                       NormDecision* sh) {
    static_assert(BLOCK >= SUB && SUB % 64 == 0, "the first SUB threads of the block");
    constexpr int NW = SUB / 64;
    double s = 0.0, q = 0.0;
    if ((int)threadIdx.x < SUB)
        for (int64_t i = threadIdx.x; i < npart; i += SUB) {
            s += partials[2 * i];
            q += partials[2 * i + 1];
        }
    s = wave_sum(s);
    q = wave_sum(q);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0 && wid < NW) {
        red[wid] = s;
        red[NW + wid] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ss = 0.0, qq = 0.0;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            ss += red[i];
            qq += red[NW + i];
        }
        *sh = decision_from_totals(ss, qq, n_total, thr_sd);
    }
    __syncthreads();
    return *sh;
}
// scale_noise_kernel's value sequence -- subtract, IEEE quotient, multiply -- with the quotient by the correctly rounded reciprocal and one
// residual step (elementwise.hip, PendingNorm: the same bits as `v / std` for every v in the normal range)
struct ExactNorm {
    float mean, stdv, inv_std, factor;
    bool sub, div, mul;
    __device__ __forceinline__ ExactNorm(const NormDecision& d, float f)
        : mean(d.mean), stdv(d.stdv), inv_std(1.0f / d.stdv), factor(f), sub(d.do_sub != 0), div(d.do_div != 0), mul(f != 1.0f) {}
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

// ROLE: 0 an ordinary launch; the look-ahead launch's workgroups: 1 this call's planes (normalised stores, no statistics), 2 the next call's
// planes (statistics into ah.partials, no stores).  The body is instantiated per role and handed ITS level table by reference -- the kernel
// arguments stay in scalar registers (selecting between the two tables at run time put a copy in scratch memory: 2 x slower).
template <bool STATS, bool XROWS, int PRE, bool NT, int ROLE>
__device__ __forceinline__ void pyramid_plane_body(float* out, int64_t planes, int H, int W, const PyramidLevels& lv, int mode, uint64_t seed,
                                                   uint64_t stream_id, int64_t elem_offset, double* partials, int grid_floats, const Accum& fold,
                                                   const Prefix& pre, const PyrAhead& ah, const int bid, const int nblocks) {
    static_assert(ROLE == 0 || (XROWS && !STATS && PRE == 0), "the look-ahead form is the stretched-rows kernel of a single generator");
    extern __shared__ __align__(16) float pyr_lds[];
    __shared__ double red[2 * kPyrBlock / 64];
    __shared__ NormDecision sh_dec;
    // the levels' parameters where a thread can index them by a run-time level (kernel arguments live in scalar registers: per-level
    // loops over them are sixteen short dependent loops; flattened over (level, item) a thread has four independent items in flight)
    __shared__ int lvl_h[kMaxLevels], lvl_w[kMaxLevels], lvl_off[kMaxLevels + 1], lvl_item0[kMaxLevels + 1], lvl_row0[kMaxLevels];
    __shared__ float lvl_weight[kMaxLevels];
    __shared__ unsigned long long lvl_stream[kMaxLevels];
    SONAR_NG_STAMP(0);
    double s = 0.0, q = 0.0;
    NormDecision dec{0.f, 1.f, 0, 0};
    if constexpr (ROLE == 1) dec = decide_norm_as<kBlock, kPyrBlock>(ah.na.partials, ah.npart, ah.na.n_total, ah.na.thr_sd, red, &sh_dec);
    const ExactNorm enorm(dec, ROLE == 1 ? ah.na.factor : 1.0f);
    const int HW = H * W;
    const uint32_t lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int part = wave / (kBlock / 64);  // 256-thread parts of the workgroup (wave-uniform: the step-ahead below is straight-line code)
    const int dy = 256 / W, dx = 256 - dy * W;  // one burst step advances 256 elements
    // bilinear source coordinates depend on (level, x) and (level, y) only: tabulated once per workgroup
    Lin* const xtab = reinterpret_cast<Lin*>(pyr_lds + grid_floats);  // [level][W]
    Lin* const ytab = xtab + lv.count * W;                            // [level][H]
    float* const xrows = reinterpret_cast<float*>(ytab + lv.count * H);  // XROWS: [level][h_l][W]
    if (mode == 2) {
        // area: the adaptive-average window [floor(i in / out), ceil((i + 1) in / out)) of every (level, x) and (level, y), tabulated once
        // per workgroup (i0 = start, i1 = end): the tile loop otherwise divides four times per value and level
        for (int l = 0; l < lv.count; ++l) {
            for (int i = threadIdx.x; i < W + H; i += kPyrBlock) {
                const bool isx = i < W;
                const int64_t d = isx ? i : i - W, in = isx ? lv.w[l] : lv.h[l], on = isx ? W : H;
                Lin a;
                a.i0 = (int)((d * in) / on);
                a.i1 = (int)(((d + 1) * in + on - 1) / on);
                a.w0 = (float)(a.i1 - a.i0);
                a.w1 = 0.0f;
                (isx ? xtab[l * W + d] : ytab[l * H + d]) = a;
            }
        }
    }
    // The parts other than the first reach their share of a tile's burst by stepping their generators over the iterations before it: a
    // dependent chain (~160 cycles per word at two waves per SIMD, 5 k cycles for half a burst -- those waves left the tile loop 2.2 us
    // after part 0's).  For the workgroup's first plane they do it HERE, while part 0 builds the coordinate tables alone.
    constexpr int kIters = kTileIters / kPyrParts;
    const int64_t t_ahead = part != 0 ? (elem_offset + (int64_t)bid * HW) / kTileElems + wave % (kBlock / 64) : -1;
    TileRng rng_ahead, prng_ahead;
    if (threadIdx.x == 0) {
        int off = 0, item = 0, row = 0;
#pragma unroll
        for (int l = 0; l < kMaxLevels; ++l) {
            if (l < lv.count) {
                const int n = lv.h[l] * lv.w[l];
                lvl_row0[l] = row;  // XROWS: the level's first stretched row
                row += lv.h[l];
                lvl_h[l] = lv.h[l];
                lvl_w[l] = lv.w[l];
                lvl_weight[l] = lv.weight[l];
                lvl_stream[l] = lv.draw_stream[l];
                lvl_off[l] = off;
                lvl_item0[l] = item;
                off += n;
                item += lv.ptr[l] ? 0 : min(kBlock, (n + 3) / 4);  // drawn levels: one item per active generator slot
            }
        }
        lvl_off[lv.count] = off;
        lvl_item0[lv.count] = item;
    }
    __syncthreads();
    SONAR_NG_STAMP(13);               // (trace builds: the level parameters are in LDS)
    SONAR_NG_STAMP_T(kBlock, 14);
    if (part != 0 && bid < planes) {
        rng_ahead = rng_stream(seed, stream_id, (uint64_t)t_ahead, lane);
        if constexpr (PRE != 0) prng_ahead = rng_stream(pre.seed, pre.stream_id, (uint64_t)t_ahead, lane);
        for (int k = 0; k < part; ++k) {
            skip_words<kIters * 4>(rng_ahead);
            if constexpr (PRE != 0) skip_words<kIters * 4>(prng_ahead);
        }
    }
    SONAR_NG_STAMP_T(kBlock, 15);     // (trace builds: the second part's generators are stepped ahead)
    if (mode == 0 && part == 0) {
        const int per = W + H;
        for (int j = threadIdx.x; j < lv.count * per; j += kBlock) {
            const int l = j / per, r = j - l * per, h = lvl_h[l], w = lvl_w[l];
            if (r < W) {
                xtab[l * W + r] = lin_coord(r, (float)w / (float)W, w);
            } else {
                const int i = r - W;
                Lin ly = lin_coord(i, (float)h / (float)H, h);
                const int pitch = XROWS ? W : w;
                ly.i0 *= pitch;  // row offsets
                ly.i1 *= pitch;
                if constexpr (XROWS) {
                    ly.i0 += lvl_row0[l] * W;  // ... from the first level's first stretched row
                    ly.i1 += lvl_row0[l] * W;
                    const float wt = lvl_weight[l];
                    ly.w0 *= wt;
                    ly.w1 *= wt;
                }
                ytab[l * H + i] = ly;
            }
        }
    }
    SONAR_NG_STAMP(1);
    for (int64_t p = bid; p < planes; p += nblocks) {
        __syncthreads();
        SONAR_NG_STAMP(2);
        if (lv.supplied) {  // (generate mode has none: the walk below costs a wait for three scalar loads per level to find that out)
            int off = 0;
            for (int l = 0; l < lv.count; ++l) {  // supplied levels (replay mode): copied
                const int n = lv.h[l] * lv.w[l];
                if (lv.ptr[l]) {
                    const float* src = lv.ptr[l] + p * (int64_t)n;
                    for (int i = threadIdx.x; i < n; i += kPyrBlock) pyr_lds[off + i] = src[i];
                }
                off += n;
            }
        }
        // drawn levels: stream (level stream id, global plane, slot), slot t of 256 owns elements 4t.., 4(t + 256)..; the (level, slot)
        // items of all levels are dealt to the workgroup's threads in one go (per level, the small levels left most threads idle
        // behind a generator seeding each)
        for (int item = threadIdx.x; item < lvl_item0[lv.count]; item += kPyrBlock) {
            int l = 0;
            while (item >= lvl_item0[l + 1]) ++l;
            const int gslot = item - lvl_item0[l], off = lvl_off[l], n = lvl_off[l + 1] - off;
            TileRng rng = rng_stream(seed, lvl_stream[l], (uint64_t)(elem_offset / HW + p), gslot);
            for (int i = gslot * 4; i < n; i += kBlock * 4) {
                float z[4];
                rng.normal4(z);
                float* const g = pyr_lds + off + i;
                if (i + 3 < n) {  // (four plain stores: the guarded loop below is a vector loop with a select chain per value, ~80 instructions)
                    g[0] = z[0]; g[1] = z[1]; g[2] = z[2]; g[3] = z[3];
                } else {
                    for (int k = 0; k < 4 && i + k < n; ++k) g[k] = z[k];
                }
            }
        }
        SONAR_NG_STAMP(3);
        __syncthreads();
        SONAR_NG_STAMP(4);
        if constexpr (XROWS) {
            int go = 0, ro = 0;
            for (int l = 0; l < lv.count; ++l) {
                const int h = lv.h[l], w = lv.w[l];
                const float* g = pyr_lds + go;
                const Lin* xt = xtab + l * W;
                int yy = (int)threadIdx.x / W, xx = (int)threadIdx.x - yy * W;
                const int sy = kPyrBlock / W, sx = kPyrBlock - sy * W;
                // kPyrStretch items per thread and round: their table reads and gathers are in flight together (one at a time, a level of
                // 45 rows was 13 rounds of three dependent LDS latencies: 4.4-4.9 k cycles per plane, as long as half the tile loop)
                if (sx == 0) {
                    // the workgroup's threads cover whole rows (W divides 512: every power-of-two width): a thread keeps its column, so
                    // its table entry is read once per level and an item is two gathers at a fixed stride, a multiply-add and a store
                    if (yy < h) {
                        const Lin lx = xt[xx];
                        const float* c0 = g + lx.i0;
                        const float* c1 = g + lx.i1;
                        for (int r = yy; r < h; r += kPyrStretch * sy) {
                            float a[kPyrStretch], b[kPyrStretch];
#pragma unroll
                            for (int u = 0; u < kPyrStretch; ++u) {
                                const int ru = r + u * sy < h ? r + u * sy : r;
                                a[u] = c0[ru * w];
                                b[u] = c1[ru * w];
                            }
#pragma unroll
                            for (int u = 0; u < kPyrStretch; ++u)
                                if (r + u * sy < h) xrows[ro + (r + u * sy) * W + xx] = __builtin_fmaf(b[u], lx.w1, a[u] * lx.w0);
                        }
                    }
                    go += h * w;
                    ro += h * W;
                    continue;
                }
                for (int i = threadIdx.x; i < h * W; i += kPyrStretch * kPyrBlock) {
                    float a[kPyrStretch], b[kPyrStretch], wa[kPyrStretch], wb[kPyrStretch];
#pragma unroll
                    for (int u = 0; u < kPyrStretch; ++u) {
                        const bool on = i + u * kPyrBlock < h * W;
                        const Lin lx = xt[on ? xx : 0];
                        const float* row = g + (on ? yy : 0) * w;
                        a[u] = row[lx.i0];
                        b[u] = row[lx.i1];
                        wa[u] = lx.w0;
                        wb[u] = lx.w1;
                        yy += sy;
                        xx += sx;
                        if (xx >= W) {
                            xx -= W;
                            yy += 1;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kPyrStretch; ++u)
                        if (i + u * kPyrBlock < h * W) xrows[ro + i + u * kPyrBlock] = __builtin_fmaf(b[u], wb[u], a[u] * wa[u]);
                }
                go += h * w;
                ro += h * W;
            }
            SONAR_NG_STAMP(5);
            __syncthreads();
        }
        SONAR_NG_STAMP(6);
        float* const oplane = out + p * (int64_t)HW;
        const Accum pfold{fold.y, pre.ya, pre.f};
        const Divider pdiv(PRE == 2 ? pre.div_fac : 1.0f);
        // Perlin prefix: the plane's lattice vectors (the lattice repeats per latent of pre.chw = channels * H * W elements)
        const float* const tplane = PRE == 2 ? pre.terms + (int)(p % (pre.chw / HW)) * HW : nullptr;
        // RNG tiles overlapping this plane (a plane need not start or end on a tile boundary: the neighbours' workgroups draw
        // the shared tile too and each keeps its own part)
        const int64_t g0 = elem_offset + p * (int64_t)HW;
        const int64_t tile_first = g0 / kTileElems, tile_last = (g0 + HW - 1) / kTileElems;
#ifdef SONAR_PYR_SETUP_ONLY  // profiling builds: what the per-plane setup (grids, tables, stretched rows) costs
        if (oplane != nullptr) continue;
#endif
        // a tile's burst is split between the parts: part k runs iterations [k, k + 1) * kTileIters / kPyrParts after stepping its
        // generators over the iterations before them (8 plain instructions per word instead of a Box-Muller pair)
        for (int64_t t = tile_first + wave % (kBlock / 64); t <= tile_last; t += kBlock / 64) {
            TileRng rng, prng;
            if (t == t_ahead && p == bid) {  // wave-uniform: stepped ahead before the tables
                rng = rng_ahead;
                if constexpr (PRE != 0) prng = prng_ahead;
            } else {
                rng = rng_stream(seed, stream_id, (uint64_t)t, lane);
                if constexpr (PRE != 0) prng = rng_stream(pre.seed, pre.stream_id, (uint64_t)t, lane);
                for (int k = 0; k < part; ++k) {
                    skip_words<kIters * 4>(rng);
                    if constexpr (PRE != 0) skip_words<kIters * 4>(prng);
                }
            }
            SONAR_NG_STAMP(7);
            // element index inside the plane; < 0 or >= HW: not ours
            int e = (int)(t * kTileElems - g0) + (int)lane * 4 + part * kIters * 256;
            int y = e >= 0 ? e / W : -((-e + W - 1) / W);        // floor
            int x4 = e - y * W;
            if constexpr (XROWS) {
                // kPyrGroup burst steps at a time: a level's table entries, then its row pairs, are requested for the whole group before
                // the first is used.  Step by step a level was two dependent LDS latencies (table entry -> row addresses -> rows) and a
                // scalar load of its height: ~675 cycles per step with three levels at one or two workgroups per CU (trace build of round
                // 5), the arithmetic of a step being ~120 instructions.  Same draws, same operation order per value: the same bits.
                constexpr int G = PRE != 0 ? kPyrGroupPre : kPyrGroup;
                static_assert(kIters % G == 0, "groups tile a part's share of the burst");
                const float base_mul = lv.fullres ? lv.base_scale : 1.0f;  // (x 1 is exact: no select per value)
                // WHOLE: the tile lies inside the plane (always, when planes are whole tiles): no per-lane ownership test around the stores
                auto burst = [&](auto whole_c, const auto& dv) {  // dv: the hosted Perlin item's word-to-value converter (with_divider)
                    constexpr bool WHOLE = decltype(whole_c)::value;
                    // A hosted Perlin item's lattice vectors.  Launch-bound sizes (NT, one workgroup per CU): all of the part's share of
                    // the burst requested before the first step, the loop unrolled so that the steps index them statically -- a step that
                    // asks for its own waits for it with one other wave on its SIMD to cover: 7 k of the chain's 15.5 k cycles per tile
                    // (trace build of round 5), chain at batch 64 26.1 -> 25.2 us.  At batch 512 (four waves per SIMD cover the load) the
                    // same form is 3 % SLOWER (120 registers, eight copies of the step), and so is a request one step ahead in the rolled
                    // loop (+5 %): there a step loads its own.
                    constexpr bool LAT_AHEAD = PRE == 2 && NT;
                    auto lattice = [&](int ei) {
                        return PRE == 2 && (WHOLE || (ei >= 0 && ei < HW)) ? *reinterpret_cast<const float4*>(tplane + ei) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    };
                    constexpr int NTV = LAT_AHEAD ? kIters : 1;
                    float4 tva[NTV];
                    if constexpr (LAT_AHEAD) {
#pragma unroll
                        for (int i = 0; i < kIters; ++i) tva[i] = lattice(e + 256 * i);
                    }
#pragma unroll NTV
                    for (int it0 = 0; it0 < kIters; it0 += G) {
                        float4 tvs[G];
#pragma unroll
                        for (int j = 0; j < G; ++j) tvs[j] = LAT_AHEAD ? tva[LAT_AHEAD ? it0 + j : 0] : lattice(e + 256 * j);
                        float v[G][4], px[G][4];
                        int ee[G], yy[G], xx[G];
#pragma unroll
                        for (int j = 0; j < G; ++j) {
                            rng.normal4(v[j]);
                            const bool ours = WHOLE || (e >= 0 && e < HW);
                            if constexpr (PRE != 0) {  // the prefix's generator walks every iteration of the tile, ours or not
                                const float4 tv = tvs[j];
                                prefix_draw<PRE>(prng, dv, tv, px[j]);
                            }
                            ee[j] = e;
                            yy[j] = ours ? y : 0;  // (not ours: any entry of the tables, the values are dropped)
                            xx[j] = ours ? x4 : 0;
                            e += 256;
                            y += dy;
                            x4 += dx;
                            if (x4 >= W) {
                                x4 -= W;
                                y += 1;
                            }
#pragma unroll
                            for (int k = 0; k < 4; ++k) v[j][k] *= base_mul;
                        }
                        // the four values of a step as ONE 16-byte register value from the draw to the store: the level sums are packed
                        // multiply-adds on its halves, the store takes it as it is (as four scalars the compiler moved them between
                        // register pairs around the level loop and again before the store: 8-11 moves per step)
                        sonar_v4f vv[G];
#pragma unroll
                        for (int j = 0; j < G; ++j) vv[j] = sonar_v4f{v[j][0], v[j][1], v[j][2], v[j][3]};
                        const int nl = lv.count;
                        for (int l = 0; l < nl; ++l) {
                            Lin ly[G];
#pragma unroll
                            for (int j = 0; j < G; ++j) ly[j] = ytab[l * H + yy[j]];
                            sonar_v4f a[G], b[G];
#pragma unroll
                            for (int j = 0; j < G; ++j) {
                                a[j] = *reinterpret_cast<const sonar_v4f*>(xrows + ly[j].i0 + xx[j]);
                                b[j] = *reinterpret_cast<const sonar_v4f*>(xrows + ly[j].i1 + xx[j]);
                            }
#pragma unroll
                            for (int j = 0; j < G; ++j) {
                                const sonar_v4f w0 = {ly[j].w0, ly[j].w0, ly[j].w0, ly[j].w0}, w1 = {ly[j].w1, ly[j].w1, ly[j].w1, ly[j].w1};
                                vv[j] = __builtin_elementwise_fma(a[j], w0, __builtin_elementwise_fma(b[j], w1, vv[j]));
                            }
                        }
#pragma unroll
                        for (int j = 0; j < G; ++j) {
                            const int ej = ee[j];
                            if (!WHOLE && (ej < 0 || ej >= HW)) continue;
                            if constexpr (PRE != 0) {
                                float w[4] = {vv[j].x, vv[j].y, vv[j].z, vv[j].w};
                                prefix_fold(pre, pfold, fold, p * (int64_t)HW + ej, px[j], w);
                                vv[j] = sonar_v4f{w[0], w[1], w[2], w[3]};
                            } else if (fold.y) {
                                const float4 yv = *reinterpret_cast<const float4*>(fold.y + p * (int64_t)HW + ej);
                                vv[j] = sonar_v4f{fold(yv.x, vv[j].x), fold(yv.y, vv[j].y), fold(yv.z, vv[j].z), fold(yv.w, vv[j].w)};
                            }
                            if constexpr (ROLE == 1) vv[j] = sonar_v4f{enorm(vv[j].x), enorm(vv[j].y), enorm(vv[j].z), enorm(vv[j].w)};
                            if constexpr (ROLE != 2) {
                                if constexpr (NT) __builtin_nontemporal_store(vv[j], reinterpret_cast<sonar_v4f*>(oplane + ej));
                                else *reinterpret_cast<sonar_v4f*>(oplane + ej) = vv[j];
                            }
                            if constexpr (STATS || ROLE == 2) {
                                float s01 = vv[j].x + vv[j].y;
                                asm volatile("" : "+v"(s01));  // (two scalar adds: paired into one packed add they cost three moves)
                                const float ps = s01 + (vv[j].z + vv[j].w);
                                const float pq = __builtin_fmaf(vv[j].x, vv[j].x, __builtin_fmaf(vv[j].y, vv[j].y, __builtin_fmaf(vv[j].z, vv[j].z, vv[j].w * vv[j].w)));
                                s += (double)ps;
                                q += (double)pq;
                            }
                        }
                    }
                };
                const bool whole = t * kTileElems >= g0 && (t + 1) * kTileElems <= g0 + HW;
                auto run = [&](const auto& dv) {
                    if (whole) burst(std::true_type{}, dv);
                    else burst(std::false_type{}, dv);
                };
                if constexpr (PRE == 2) with_divider(pdiv, run);
                else run(pdiv);
                continue;
            }
SONAR_PYR_UNROLL
            for (int it = 0; it < kIters; ++it, e += 256, y += dy, x4 += dx, y += x4 >= W, x4 -= x4 >= W ? W : 0) {
                float v[4];
                rng.normal4(v);
                float px[4];
                if constexpr (PRE != 0) {  // the prefix's generator walks every iteration of the tile, ours or not
                    const bool ours = e >= 0 && e < HW;
                    const float4 tv = PRE == 2 && ours ? *reinterpret_cast<const float4*>(tplane + e) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    prefix_draw<PRE>(prng, pdiv, tv, px);
                }
                if (e < 0 || e >= HW) continue;
                if (lv.fullres) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] *= lv.base_scale;
                }
                int lo = 0;
                for (int l = 0; l < lv.count; ++l) {
                    const int h = lv.h[l], w = lv.w[l];
                    const float* plane = pyr_lds + lo;
                    lo += h * w;
                    const float wt = lv.weight[l];
                    if (mode == 0) {
                        const Lin ly = ytab[l * H + y];
                        const float* r0 = plane + ly.i0;
                        const float* r1 = plane + ly.i1;
                        const Lin* xt = xtab + l * W + x4;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const Lin lx = xt[k];
                            const float t0 = r0[lx.i0] * lx.w0 + r0[lx.i1] * lx.w1;
                            const float t1 = r1[lx.i0] * lx.w0 + r1[lx.i1] * lx.w1;
                            v[k] += (t0 * ly.w0 + t1 * ly.w1) * wt;
                        }
                    } else if (mode == 1) {
                        const float sy = (float)h / (float)H, sx = (float)w / (float)W;
                        const float* row = plane + nearest_exact_idx(y, sy, h) * w;
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] += row[nearest_exact_idx(x4 + k, sx, w)] * wt;
                    } else {
                        const Lin ly = ytab[l * H + y];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const Lin lx = xtab[l * W + x4 + k];
                            float acc = 0.0f;  // same order and rounding as area_sample
                            for (int yy = ly.i0; yy < ly.i1; ++yy)
                                for (int xx = lx.i0; xx < lx.i1; ++xx) acc += plane[yy * w + xx];
                            v[k] += acc / (ly.w0 * lx.w0) * wt;
                        }
                    }
                }
                if constexpr (PRE != 0) {
                    prefix_fold(pre, pfold, fold, p * (int64_t)HW + e, px, v);
                } else if (fold.y) {
                    const float4 yv = *reinterpret_cast<const float4*>(fold.y + p * (int64_t)HW + e);
                    v[0] = fold(yv.x, v[0]); v[1] = fold(yv.y, v[1]); v[2] = fold(yv.z, v[2]); v[3] = fold(yv.w, v[3]);
                }
                store4<NT>(oplane + e, v[0], v[1], v[2], v[3]);
                if constexpr (STATS) {
                    const float ps = (v[0] + v[1]) + (v[2] + v[3]);
                    const float pq = __builtin_fmaf(v[0], v[0], __builtin_fmaf(v[1], v[1], __builtin_fmaf(v[2], v[2], v[3] * v[3])));
                    s += (double)ps;
                    q += (double)pq;
                }
            }
        }
    }
    SONAR_NG_STAMP(8);
#ifdef SONAR_NG_TRACE  // when each of the waves 1..6 left the tile loop (wave 0 is slot 8)
#ifdef SONAR_NG_TRACE_PART1  // (the second part's waves 4..6 instead)
    if (lane == 0 && wave >= 4 && wave <= 6 && blockIdx.x < 256) g_ng_trace[blockIdx.x * 16 + 6 + wave] = __builtin_readcyclecounter();
#else
    if (lane == 0 && wave >= 1 && wave <= 3 && blockIdx.x < 256) g_ng_trace[blockIdx.x * 16 + 9 + wave] = __builtin_readcyclecounter();
#endif
#endif
    if constexpr (STATS) write_partial_at<kPyrBlock>(s, q, partials, red, bid, nblocks);
    if constexpr (ROLE == 2) write_partial_at<kPyrBlock>(s, q, ah.partials, red, bid, nblocks);
    SONAR_NG_STAMP(9);
}

template <bool STATS, bool XROWS, int PRE = 0, bool NT = false /* common.h store4: launch-bound sizes */, bool AHEAD = false>
__global__ void __launch_bounds__(kPyrBlock) pyramid_plane_kernel(float* out, int64_t planes, int H, int W, PyramidLevels lv, int mode,
                                                               uint64_t seed, uint64_t stream_id, int64_t elem_offset,
                                                               double* partials, int grid_floats, Accum fold, Prefix pre, LatticeJob lat,
                                                               PyrAhead ah) {
    kernarg_touch_for(out, planes, H, W, lv, mode, seed, stream_id, elem_offset, partials, grid_floats, fold, pre, lat, ah);
    if ((int)blockIdx.x < lat.blocks) {
        perlin_lattice_cells<kPyrBlock>(lat.out, lat.iters, lat.C, H, W, lat.blend_mode, lat.seed, lat.stream_id, blockIdx.x, lat.blocks);
        return;
    }
    const int bid = (int)blockIdx.x - lat.blocks;  // this workgroup among those that own planes
    if constexpr (!AHEAD) {
        pyramid_plane_body<STATS, XROWS, PRE, NT, 0>(out, planes, H, W, lv, mode, seed, stream_id, elem_offset, partials, grid_floats, fold, pre, ah, bid,
                                                     (int)gridDim.x - lat.blocks);
    } else if (bid < ah.main_blocks) {  // (workgroup-uniform)
        pyramid_plane_body<false, true, 0, NT, 1>(out, planes, H, W, lv, mode, seed, stream_id, elem_offset, nullptr, grid_floats, fold, pre, ah, bid,
                                                  ah.main_blocks);
    } else {
        pyramid_plane_body<false, true, 0, NT, 2>(out, planes, H, W, ah.lv, mode, seed, ah.stream_id, elem_offset, nullptr, ah.grid_floats, fold, pre, ah,
                                                  bid - ah.main_blocks, (int)gridDim.x - lat.blocks - ah.main_blocks);
    }
}

// true if the plane kernel was launched; *slots (optional): the number of partial pairs its workgroups own (the rest are zeroed)
static bool launch_pyramid_plane(float* out, int64_t planes, int64_t H, int64_t W, const PyramidLevels& lv, int mode, uint64_t seed,
                                 uint64_t stream_id, int64_t elem_offset, double* partials, hipStream_t st, Accum fold = kNoAccum,
                                 int pre_kind = 0, Prefix pre = Prefix{1.0f, 1.0f, 1.0f, 0, 0, nullptr, 1, 0}, LatticeJob lat = kNoLattice,
                                 int* slots = nullptr) {
    size_t grid_floats = 0, rows = 0;
    for (int l = 0; l < lv.count; ++l) {
        grid_floats += (size_t)lv.h[l] * lv.w[l];
        rows += (size_t)lv.h[l];
    }
    grid_floats = (grid_floats + 3) & ~(size_t)3;  // the coordinate tables that follow are 16-byte entries
    const size_t lds = grid_floats * sizeof(float) + (mode == 0 || mode == 2 ? (size_t)lv.count * (H + W) * sizeof(Lin) : 0);
    const size_t lds_x = lds + rows * W * sizeof(float);
    if (W % 4 != 0 || elem_offset % (H * W) != 0 || lds > kPyramidLdsBudget) return false;
    const bool xrows = mode == 0 && W % 4 == 0 && lds_x <= kPyramidLdsBudget;
    static const int grid_cap = [] { const char* e = getenv("SONAR_PYR_GRID"); return e ? atoi(e) : kNPart; }();
    const int g = (int)std::min<int64_t>(planes, std::min(grid_cap, kNPart));
    if (slots) *slots = g;
    const bool nt = nt_stores_host(planes * H * W);
#define SONAR_PPN(ST, XR, P, N) \
    hipLaunchKernelGGL((pyramid_plane_kernel<ST, XR, P, N>), dim3(g + lat.blocks), dim3(kPyrBlock), XR ? lds_x : lds, st, out, planes, (int)H, (int)W, lv, mode, \
                       seed, stream_id, elem_offset, partials, (int)grid_floats, fold, pre, lat, PyrAhead{})
#define SONAR_PP(ST, XR, P) \
    do { \
        if (nt) SONAR_PPN(ST, XR, P, true); else SONAR_PPN(ST, XR, P, false); \
    } while (0)
#define SONAR_PPK(ST, XR) \
    do { \
        if (pre_kind == SONAR_PREFIX_NORMAL) SONAR_PP(ST, XR, 1); \
        else if (pre_kind == SONAR_PREFIX_PERLIN) SONAR_PP(ST, XR, 2); \
        else SONAR_PP(ST, XR, 0); \
    } while (0)
    if (partials) {
        if (xrows) SONAR_PPK(true, true); else SONAR_PPK(true, false);
    } else {
        if (xrows) SONAR_PPK(false, true); else SONAR_PPK(false, false);
    }
#undef SONAR_PPK
#undef SONAR_PP
#undef SONAR_PPN
    return true;
}

// LDS of the stretched-rows form for one level table; 0: the form does not take it
static size_t pyramid_xrows_lds(const PyramidLevels& lv, int64_t H, int64_t W, int* grid_floats) {
    size_t gf = 0, rows = 0;
    for (int l = 0; l < lv.count; ++l) {
        gf += (size_t)lv.h[l] * lv.w[l];
        rows += (size_t)lv.h[l];
    }
    gf = (gf + 3) & ~(size_t)3;
    *grid_floats = (int)gf;
    const size_t lds = gf * sizeof(float) + (size_t)lv.count * (H + W) * sizeof(Lin) + rows * W * sizeof(float);
    return lds <= kPyramidLdsBudget ? lds : 0;
}
// The look-ahead launch: `now` planes of this call (null: none -- the statistics of `next` alone, for a call that found none left), `next`
// the call whose statistics go to partials_next.  false: a level table is beyond the stretched-rows form (nothing launched).
static bool launch_pyramid_ahead(float* out, int64_t planes, int64_t H, int64_t W, const PyramidLevels* now, uint64_t stream_now,
                                 const PyramidLevels& next, uint64_t stream_next, uint64_t seed, int64_t elem_offset, NormArgs na, int npart,
                                 double* partials_next, hipStream_t st) {
    int gf_now = 0, gf_next = 0;
    const size_t lds_now = now ? pyramid_xrows_lds(*now, H, W, &gf_now) : 1, lds_next = pyramid_xrows_lds(next, H, W, &gf_next);
    if (W % 4 != 0 || elem_offset % (H * W) != 0 || !lds_now || !lds_next) return false;
    static const int grid_cap = [] { const char* e = getenv("SONAR_PYR_GRID"); return e ? atoi(e) : kNPart; }();
    const int g = (int)std::min<int64_t>(planes, std::min(grid_cap, kNPart));  // launch_pyramid_plane's grid: the partials' grouping
    PyrAhead ah{};
    ah.na = na;
    ah.npart = npart;
    ah.lv = next;
    ah.stream_id = stream_next;
    ah.grid_floats = gf_next;
    ah.main_blocks = now ? g : 0;
    ah.partials = partials_next;
    const PyramidLevels& first = now ? *now : next;
    const size_t lds = std::max(now ? lds_now : 0, lds_next);
    const Prefix nopre{1.0f, 1.0f, 1.0f, 0, 0, nullptr, 1, 0};
    if (nt_stores_host(planes * H * W))
        hipLaunchKernelGGL((pyramid_plane_kernel<false, true, 0, true, true>), dim3(ah.main_blocks + g), dim3(kPyrBlock), lds, st, out, planes, (int)H, (int)W, first,
                           0, seed, stream_now, elem_offset, (double*)nullptr, gf_now, kNoAccum, nopre, kNoLattice, ah);
    else
        hipLaunchKernelGGL((pyramid_plane_kernel<false, true, 0, false, true>), dim3(ah.main_blocks + g), dim3(kPyrBlock), lds, st, out, planes, (int)H, (int)W, first,
                           0, seed, stream_now, elem_offset, (double*)nullptr, gf_now, kNoAccum, nopre, kNoLattice, ah);
    return true;
}

// ------------------------------------------------------------------------------------------------
// Levels that are drawn only to be shrunk.  PyramidOld (py/noise_generation.py:567-606): noise = sum_i discount^i * interpolate(normal(std =
// 0.5^i) at (2^(i+1) H) x (2^(i+1) W), size = (H, W)); HighresPyramid (:517-564): levels of up to 15 x the latent's sides.  The reference (and the replay path here) materialises every level -- the last of five is 32 x 32 times the latent:
// 1 GiB for four SDXL latents -- to keep, with the default nearest-exact mode, ONE value of each 2^(i+1) x 2^(i+1) block.  On-device
// draws need no such tensor: the level value at (plane, ys, xs) is a counter-based normal keyed by its global element index
// (Philox4x32-10 of group e / 4, Box-Muller, slot e % 4), so the kernel draws exactly the taps the shrinking interpolation reads --
// nearest-exact / nearest 1, bilinear 2 x 2, bicubic 4 x 4 per level and output (the ratio is an exact power of two: the source
// coordinate sits half way between two samples), any ratio through the resampler's own index rules.  sonar_level_normal_f32 writes a
// whole level from the same keys: the definition the sampled kernel is tested against.  Area mode averages whole blocks of independent normals: the
// block mean IS a normal of std 0.5^i / 2^(i+1), drawn directly (same joint distribution as drawing the level and pooling it).
__device__ __forceinline__ float level_normal(uint64_t seed, uint64_t stream, int64_t e) {
    float z[4];
    philox_normal4(seed, stream, (uint64_t)(e >> 2), z);
    const int slot = (int)(e & 3);
    return slot == 0 ? z[0] : slot == 1 ? z[1] : slot == 2 ? z[2] : z[3];
}

// MODE: RESAMPLE ids 0 bilinear, 1 nearest-exact, 2 area (whole-block windows only), 3 nearest, 4 bicubic -- the resampler's own index and
// weight rules (lin_coord / nearest_*_idx / cubic_coord above) with the plane read replaced by the keyed draw.
constexpr int kMaxSampledLevels = 16;
struct SampledLevels {
    int count;
    int h[kMaxSampledLevels], w[kMaxSampledLevels];
    float weight[kMaxSampledLevels], sd[kMaxSampledLevels];
};

template <int MODE>
__global__ void __launch_bounds__(kBlock) levels_sampled_kernel(float* out, int64_t planes, int H, int W, SampledLevels lv, uint64_t seed,
                                                                 uint64_t stream0, int64_t plane_offset, int accumulate) {
    kernarg_touch_for(out, planes, H, W, lv, seed, stream0, plane_offset, accumulate);
    const int64_t total = planes * H * W;
    for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kBlock) {
        const int64_t p = idx / ((int64_t)H * W);
        const int rem = (int)(idx - p * H * W), y = rem / W, x = rem - y * W;
        float acc = accumulate ? out[idx] : 0.0f;
        for (int l = 0; l < lv.count; ++l) {
            const int h = lv.h[l], w = lv.w[l];
            const float sd = lv.sd[l];
            const int64_t base = (plane_offset + p) * (int64_t)h * w;
            const uint64_t stream = stream0 + (uint64_t)l;
            auto at = [&](int yy, int xx) { return level_normal(seed, stream, base + (int64_t)yy * w + xx) * sd; };
            const float sy = (float)h / (float)H, sx = (float)w / (float)W;
            float v;
            if constexpr (MODE == 2) {
                // area over whole r x r blocks of independent N(0, sd^2) values: the block mean is exactly one N(0, (sd / r)^2) value,
                // independent from block to block -- drawn as such, keyed by the OUTPUT element (no level value is defined for this mode)
                v = level_normal(seed, stream, (plane_offset + p) * (int64_t)H * W + rem) * (sd / sqrtf((float)(h / H) * (float)(w / W)));
            } else if constexpr (MODE == 1) {
                v = at(nearest_exact_idx(y, sy, h), nearest_exact_idx(x, sx, w));
            } else if constexpr (MODE == 3) {
                v = at(nearest_idx(y, sy, h), nearest_idx(x, sx, w));
            } else if constexpr (MODE == 0) {
                const Lin ly = lin_coord(y, sy, h), lx = lin_coord(x, sx, w);
                const float t0 = at(ly.i0, lx.i0) * lx.w0 + at(ly.i0, lx.i1) * lx.w1;
                const float t1 = at(ly.i1, lx.i0) * lx.w0 + at(ly.i1, lx.i1) * lx.w1;
                v = t0 * ly.w0 + t1 * ly.w1;
            } else {
                const Cubic cy = cubic_coord(y, sy, h, false), cx = cubic_coord(x, sx, w, false);
                float rows[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    rows[k] = at(cy.i[k], cx.i[0]) * cx.c[0] + at(cy.i[k], cx.i[1]) * cx.c[1] + at(cy.i[k], cx.i[2]) * cx.c[2] + at(cy.i[k], cx.i[3]) * cx.c[3];
                v = rows[0] * cy.c[0] + rows[1] * cy.c[1] + rows[2] * cy.c[2] + rows[3] * cy.c[3];
            }
            v = v * lv.weight[l];  // the resampler's order: scale, then add to the running sum
            acc = acc + v;
        }
        out[idx] = acc;
    }
}

__global__ void __launch_bounds__(kBlock) level_normal_kernel(float* level, int64_t n, float sd, uint64_t seed, uint64_t stream,
                                                               int64_t elem_offset) {
    kernarg_touch_for(level, n, sd, seed, stream, elem_offset);
    // a thread per Philox group of four consecutive elements (the ends of the range may cut a group)
    const int64_t g0 = elem_offset >> 2, g1 = (elem_offset + n + 3) >> 2;
    for (int64_t g = g0 + (int64_t)blockIdx.x * kBlock + threadIdx.x; g < g1; g += (int64_t)gridDim.x * kBlock) {
        float z[4];
        philox_normal4(seed, stream, (uint64_t)g, z);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = 4 * g + k - elem_offset;
            if (i >= 0 && i < n) level[i] = z[k] * sd;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Brownian-interval noise (the reference wraps ComfyUI's BrownianTreeNoiseSampler -> torchsde, un-vendored:
// py/noise_generation.py:262-286, py/nodes/powernoise.py:383-393).  One Brownian path per element over [t_lo, t_hi], defined
// point by point as the sampler asks for times (torchsde's BrownianInterval grows its tree the same way): a new time t between
// the nearest known times a < t < b is the Brownian bridge
//   W(t) = ((b - t) W(a) + (t - a) W(b)) / (b - a) + sqrt((t - a)(b - t) / (b - a)) z(node_t),
// z(node, element) a counter-based N(0,1) keyed by (seed, node id, global element index).  So every W(t) is a LINEAR combination
// of node normals (the host keeps the coefficients in fp64) and can be evaluated two ways with the same value up to fp32
// rounding: from the cached tensors W(a), W(b) plus ONE fresh normal per element (`base_a` / `base_b` below: the sampler's
// case, a step starts where the last one ended), or from its expansion out[e] = sum_k coef[k] * z(node[k], e).  Values depend on
// (seed, node, global element index) only: repeated / nested / abutting intervals are consistent, batches shard bit-identically.
constexpr int kMaxBrownianNodes = 96;
struct BrownianTerms {
    unsigned long long node[kMaxBrownianNodes];
    float coef[kMaxBrownianNodes];
    int count;
};

// Burst variant (latents of a multiple of kTileElems elements, one seed): a wave owns a sub-tile of 4 steps x 64 lanes x 4 = 1024
// elements and keeps its 4 x 4 partial sums in registers over the nodes; z(node, .) on the sub-tile is a multiply-with-carry burst
// (common.h, Mwc) of 12 words per lane.
// Round 6 (the virtual Brownian tree made a call 25 of these bursts per element, 354 us on cfg5's shard -- 22 ns of SIMD time per
// wave-normal, a fifth of it the Philox block that seeded every (node, sub-tile, lane) burst):
//  * ONE Philox block per (sub-tile, lane) and call -- the sub-tile's base state, keyed by (seed, kBrownStream, sub-tile, lane) -- and per
//    node a burst seeded by hashing the base with the node: x = fmix32(base.x ^ hx(node)), c = fmix32(base.c ^ hc(node)), where
//    (hx, hc) = splitmix64(node id) comes from the host and fmix32 is MurmurHash3's 32-bit finaliser (full avalanche: bursts of different
//    nodes start at unrelated points of the generator's one cycle, as Philox-seeded ones do): 16 instructions instead of ~60;
//  * the conversions of the power-law draw (common.h): 23 radius bits through the mantissa of a float in [1, 2), 16 angle bits per value
//    as a fraction of a revolution, one angle word for two Box-Muller pairs: 12 words and 20 conversion instructions per 16 normals
//    instead of 16 and 56;
//  * the coefficient rides under the radius' square root (c r = sqrt(-2 ln2 c^2 log2 u)) and its sign on the accumulating FMA's operand:
//    one multiply per pair instead of three.
// 230 -> ~150 instructions per node and sub-tile.  Every Brownian value changed with it (they were never pinned: torchsde is absent).
constexpr int kBrownIters = 4;
constexpr int kBrownTile = kBrownIters * 256;
// eight waves per workgroup where the expansion is long: a launch that reduces statistics has at most kNPart workgroups (one partial
// slot each), and with four waves each that left a compute-bound kernel at four waves per SIMD where it has the registers for eight
// (tree call on cfg5's shard 264 -> 244 us); the bridge route (one term, HBM-bound) keeps four (89 us; 101 with eight)
constexpr int kBrownBlock = 512;
constexpr int kBrownLongTerms = 4;
constexpr uint64_t kBrownStream = 0xB0B000000001ull;  // the base states' stream id (48 bits; node ids no longer are stream ids)
struct BrownianBurstTerms {
    uint32_t hx[kMaxBrownianNodes], hc[kMaxBrownianNodes];
    float coef[kMaxBrownianNodes];
    int count;
};
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}
// acc[it][j] += c z(node)[4 it + j] for the sub-tile's 16 values of one lane: w = -2 ln2 c^2, NEG = (c < 0)
template <bool NEG>
__device__ __forceinline__ void brownian_burst_add(Mwc rng, float w, float (&acc)[kBrownIters][4]) {
#pragma unroll
    for (int it = 0; it < kBrownIters; ++it) {
        const uint32_t ra = rng.next(), rb = rng.next(), t = rng.next();
        float r0 = __builtin_amdgcn_sqrtf(w * __builtin_amdgcn_logf(2.0f - unit_mantissa(ra)));
        float r1 = __builtin_amdgcn_sqrtf(w * __builtin_amdgcn_logf(2.0f - unit_mantissa(rb)));
        if constexpr (NEG) {
            r0 = -r0;  // (folds into the FMAs as an operand modifier)
            r1 = -r1;
        }
        const float a0 = angle_lo(t), a1 = angle_hi(t);
        acc[it][0] = __builtin_fmaf(r0, __builtin_amdgcn_cosf(a0), acc[it][0]);
        acc[it][1] = __builtin_fmaf(r0, __builtin_amdgcn_sinf(a0), acc[it][1]);
        acc[it][2] = __builtin_fmaf(r1, __builtin_amdgcn_cosf(a1), acc[it][2]);
        acc[it][3] = __builtin_fmaf(r1, __builtin_amdgcn_sinf(a1), acc[it][3]);
    }
}
// Both kernels: acc = fa base_a + fb base_b + sum_k coef_k z(node_k) (either base may be null); out = scale * (acc - prev) (prev
// may be null), w_out = acc (may be null) -- ONE path point W(t), differenced against a cached W(t') (sonar_brownian_point_f32 /
// sonar_brownian_bridge_f32).  No __restrict__ on the inputs: prev is usually one of the bases.
struct BrownianBase {
    const float* a;
    const float* b;
    float fa, fb;
};
// PRE: 0 none, 1 Gaussian draw, 2 Perlin (summed lattice, tile-aligned latents).  With a prefix a wave walks the kBrownPerTile
// consecutive Brownian tiles of one generator tile, carrying the prefix's generator state across them.
constexpr int kBrownPerTile = kTileElems / (4 * 256);
template <int PRE, int BLOCK>
__global__ void __launch_bounds__(BLOCK) brownian_burst_kernel(float* out, int64_t n, int64_t elem_offset, BrownianBurstTerms terms,
                                                                uint64_t seed, const float* prev, float* w_out, float scale,
                                                                BrownianBase base, Accum fold, double* partials, Prefix pre) {
    kernarg_touch_for(out, n, elem_offset, terms, seed, prev, w_out, scale, base, fold, partials, pre);
    __shared__ double red[2 * BLOCK / 64];
    double s = 0.0, q = 0.0;
    const uint32_t lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
    const int64_t first = elem_offset / kBrownTile, tiles = n / kBrownTile;  // both aligned (launcher)
    constexpr int SUB = PRE ? kBrownPerTile : 1;
    const Accum pfold{fold.y, pre.ya, pre.f};
    const Divider pdiv(PRE == 2 ? pre.div_fac : 1.0f);
    auto tiles_loop = [&](const auto& dv) {  // dv: a hosted Perlin item's word-to-value converter (with_divider: its kind decided once)
    for (int64_t T = wave; T < tiles / SUB; T += nwaves) {
        TileRng prng;
        const float4* trow = nullptr;
        if constexpr (PRE != 0) {
            prng = rng_stream(pre.seed, pre.stream_id, (uint64_t)(elem_offset / kTileElems + T), lane);
            // shards start on a latent boundary and latents are whole tiles (launcher): the tile's vectors sit at trow[64 * iteration]
            if constexpr (PRE == 2) trow = reinterpret_cast<const float4*>(pre.terms + (int)((T * kTileElems + (int64_t)lane * 4) % pre.chw));
        }
#pragma unroll 1
        for (int sub = 0; sub < SUB; ++sub) {
            const int64_t t = T * SUB + sub;
            float acc[kBrownIters][4];
            const int64_t o = t * kBrownTile + (int64_t)lane * 4;
#pragma unroll
            for (int it = 0; it < kBrownIters; ++it) {
                float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (base.a) {
                    const float4 p = *reinterpret_cast<const float4*>(base.a + o + it * 256);
                    v = make_float4(base.fa * p.x, base.fa * p.y, base.fa * p.z, base.fa * p.w);
                }
                if (base.b) {
                    const float4 p = *reinterpret_cast<const float4*>(base.b + o + it * 256);
                    v = make_float4(__builtin_fmaf(base.fb, p.x, v.x), __builtin_fmaf(base.fb, p.y, v.y), __builtin_fmaf(base.fb, p.z, v.z),
                                    __builtin_fmaf(base.fb, p.w, v.w));
                }
                acc[it][0] = v.x; acc[it][1] = v.y; acc[it][2] = v.z; acc[it][3] = v.w;
            }
            if (terms.count > 0) {
                const Mwc seedpoint = rng_stream(seed, kBrownStream, (uint64_t)(first + t), lane);
                for (int k = 0; k < terms.count; ++k) {
                    const Mwc rng = Mwc::seeded(fmix32(seedpoint.x ^ terms.hx[k]), fmix32(seedpoint.c ^ terms.hc[k]));
                    const float c = terms.coef[k];
                    const float w = -1.3862943611198906f * (c * c);
                    if (c < 0.0f) brownian_burst_add<true>(rng, w, acc);  // (uniform: a kernel argument)
                    else brownian_burst_add<false>(rng, w, acc);
                }
            }
#pragma unroll
            for (int it = 0; it < kBrownIters; ++it) {
                float4 a = make_float4(acc[it][0], acc[it][1], acc[it][2], acc[it][3]);
                if (w_out) *reinterpret_cast<float4*>(w_out + o + it * 256) = a;
                if (out) {
                    if (prev) {
                        const float4 p = *reinterpret_cast<const float4*>(prev + o + it * 256);
                        a = make_float4(a.x - p.x, a.y - p.y, a.z - p.z, a.w - p.w);
                    }
                    float v[4] = {a.x * scale, a.y * scale, a.z * scale, a.w * scale};
                    if constexpr (PRE != 0) {
                        // the previous item's values for these four elements, folded into y first: y1 = y * ya + x * f (its own kernel's
                        // arithmetic), then this item's fold on y1
                        float x[4];
                        float4 tv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        if constexpr (PRE == 2) tv = trow[64 * (sub * kBrownIters + it)];
                        prefix_draw<PRE>(prng, dv, tv, x);
                        prefix_fold(pre, pfold, fold, o + it * 256, x, v);
                    } else {
                        accumulate_group<true>(fold, n, o + it * 256, v);
                    }
                    store_group<true>(out, n, o + it * 256, v, s, q, partials != nullptr);
                }
            }
        }
    }
    };
    if constexpr (PRE == 2) with_divider(pdiv, tiles_loop);
    else tiles_loop(pdiv);
    if (partials) write_partial<BLOCK>(s, q, partials, red);  // uniform branch (kernel argument)
}

__global__ void __launch_bounds__(kBlock) brownian_kernel(float* out, int64_t n, int64_t elem_offset, BrownianTerms terms,
                                                          uint64_t seed, const unsigned long long* __restrict__ latent_seeds,
                                                          int64_t latent_elems, const float* prev, float* w_out, float scale,
                                                          BrownianBase base, Accum fold, double* partials) {
    kernarg_touch_for(out, n, elem_offset, terms, seed, latent_seeds, latent_elems, prev, w_out, scale, base, fold, partials);
    __shared__ double red[2 * kBlock / 64];
    double s = 0.0, q = 0.0;
    // a thread owns one GLOBAL group of four elements (the counter of its draws): a shard whose first element is not a multiple of
    // four (odd latent sizes) starts and ends inside a group and keeps its part of it
    const int shift = latent_seeds ? 0 : (int)(elem_offset & 3);
    const int64_t groups = (n + shift + 3) / 4;
    for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < groups; g += (int64_t)gridDim.x * kBlock) {
        const int64_t e = g * 4 - shift;               // local element index of the group's first element (< 0: before the shard)
        uint64_t key = seed;
        uint64_t ctr = (uint64_t)(elem_offset + e) >> 2;  // global 4-group
        if (latent_seeds) {                            // one seed per latent: counters restart inside each latent
            const int64_t lat = e / latent_elems;
            key = latent_seeds[lat];
            ctr = (uint64_t)(e - lat * latent_elems) >> 2;
        }
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int j = 0; j < 4; ++j) {
            if (e + j < 0 || e + j >= n) continue;
            if (base.a) acc[j] = base.fa * base.a[e + j];
            if (base.b) acc[j] = __builtin_fmaf(base.fb, base.b[e + j], acc[j]);
        }
        for (int k = 0; k < terms.count; ++k) {
            const unsigned long long node = terms.node[k];
            const Philox4 p = philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)node, (uint32_t)(node >> 32), (uint32_t)key,
                                            (uint32_t)(key >> 32));
            float z[4];
            box_muller(p.v[0], p.v[1], z[0], z[1]);
            box_muller(p.v[2], p.v[3], z[2], z[3]);
            const float c = terms.coef[k];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(c, z[j], acc[j]);
        }
        for (int j = 0; j < 4; ++j) {
            if (e + j < 0 || e + j >= n) continue;
            if (w_out) w_out[e + j] = acc[j];
            if (out) {
                float v = (prev ? acc[j] - prev[e + j] : acc[j]) * scale;
                if (fold.y) v = fold(fold.y[e + j], v);
                out[e + j] = v;
                s += (double)v;
                q += (double)v * (double)v;
            }
        }
    }
    if (partials) write_partial<kBlock>(s, q, partials, red);
}

static int fill_levels(PyramidLevels& lv, int64_t H, int64_t W, int64_t nlevels, const float* const* level_ptrs,
                       const int64_t* level_h, const int64_t* level_w, const float* level_weight, uint64_t stream_id, bool* drawn,
                       const char* what) {
    *drawn = false;
    lv = PyramidLevels{};
    lv.fullres_weight = 1.0f;
    lv.base_scale = 1.0f;
    SONAR_REQUIRE(nlevels == 0 || (level_ptrs && level_h && level_w && level_weight), SONAR_ERR_ARG, "%s: level arrays missing", what);
    for (int64_t l = 0; l < nlevels; ++l) {
        if (level_ptrs[l] == nullptr && level_h[l] == H && level_w[l] == W) {
            // a NULL pointer at the latent's own size: the level is folded into the base draw
            SONAR_REQUIRE(lv.fullres == 0, SONAR_ERR_ARG, "%s: only one in-kernel full-resolution level", what);
            lv.fullres = 1;
            lv.fullres_weight = level_weight[l];
            lv.base_scale = sqrtf(1.0f + level_weight[l] * level_weight[l]);
            continue;
        }
        SONAR_REQUIRE(lv.count < kMaxLevels, SONAR_ERR_UNSUPPORTED, "%s: too many levels", what);
        lv.ptr[lv.count] = level_ptrs[l];
        lv.draw_stream[lv.count] = stream_id + 2 + (uint64_t)l;  // used when ptr is NULL: the plane kernel draws the grid
        if (!level_ptrs[l]) *drawn = true;
        else ++lv.supplied;
        lv.h[lv.count] = (int)level_h[l];
        lv.w[lv.count] = (int)level_w[l];
        lv.weight[lv.count] = level_weight[l];
        ++lv.count;
    }
    return SONAR_OK;
}

}  // namespace sonar

using namespace sonar;

// ================================================================================================
extern "C" int sonar_philox_normal_f32(float* out, int64_t n, uint64_t seed, uint64_t stream_id, int64_t elem_offset,
                                       double* partials, void* stream) {
    SONAR_REQUIRE(out && n >= 0 && elem_offset >= 0, SONAR_ERR_ARG, "sonar_philox_normal_f32: bad argument");
    return launch_fill<Dist::Normal>(out, n, seed, stream_id, elem_offset, Affine{0.f, 1.f, 0.f, 0}, partials,
                                     (hipStream_t)stream, "sonar_philox_normal_f32");
}

extern "C" int sonar_philox_uniform_f32(float* out, int64_t n, uint64_t seed, uint64_t stream_id, int64_t elem_offset,
                                        float sub, float mul, float add, double* partials, void* stream) {
    SONAR_REQUIRE(out && n >= 0 && elem_offset >= 0, SONAR_ERR_ARG, "sonar_philox_uniform_f32: bad argument");
    const int active = !(sub == 0.0f && mul == 1.0f && add == 0.0f);
    return launch_fill<Dist::Uniform>(out, n, seed, stream_id, elem_offset, Affine{sub, mul, add, active}, partials,
                                      (hipStream_t)stream, "sonar_philox_uniform_f32");
}

extern "C" int sonar_philox_noise_f32(int uniform, float* out, int64_t n, uint64_t seed, uint64_t stream_id, int64_t elem_offset,
                                      float sub, float mul, float add, float factor, float threshold_std_devs, double* partials,
                                      void* stream) {
    SONAR_REQUIRE(out && partials && n >= 0 && elem_offset >= 0, SONAR_ERR_ARG, "sonar_philox_noise_f32: bad argument");
    const int active = !(sub == 0.0f && mul == 1.0f && add == 0.0f);
    const Affine aff{sub, mul, add, uniform ? active : 0};
    if (uniform)
        return launch_fill_norm<Dist::Uniform>(out, n, seed, stream_id, elem_offset, aff, factor, threshold_std_devs, partials,
                                               (hipStream_t)stream, "sonar_philox_noise_f32");
    if (factor == 1.0f) {
        // N(0,1) draws with factor 1 almost never need the normalisation: the thresholds sit at 2.5 standard errors of the mean and
        // 3.5 of the standard deviation (98.7 % of tensors pass both).  So: ONE pass that stores the draws and reduces their
        // statistics, then scale_noise's own kernel, which takes the decision on the device and returns at once when there is
        // nothing to do -- instead of drawing everything twice (statistics pass + final pass: 26 + 32 us per 512 SDXL latents).
        const int rc = launch_fill<Dist::Normal>(out, n, seed, stream_id, elem_offset, aff, partials, (hipStream_t)stream, "sonar_philox_noise_f32");
        if (rc != SONAR_OK || n == 0) return rc;
        return sonar_scale_noise_f32(out, n, 1.0f, 1, threshold_std_devs, partials, kNPart, n, stream);
    }
    return launch_fill_norm<Dist::Normal>(out, n, seed, stream_id, elem_offset, aff, factor, threshold_std_devs, partials,
                                          (hipStream_t)stream, "sonar_philox_noise_f32");
}

extern "C" int sonar_philox_noise_ahead_ok(int uniform, int64_t n, float factor) {
    // N(0,1) with factor 1 has a one-pass route (draw once, store with statistics, a no-op scale_noise launch): drawing every normal twice
    // only pays while the call is launch-bound (batch 64: 12.4 -> 9.7 us; batch 512: 31.3 -> 35.8 us, the vector ALUs become the bound)
    if (!uniform && factor == 1.0f) return n > 0 && n <= kNtMaxElems ? 1 : 0;
    return n > 0 ? 1 : 0;
}

extern "C" int sonar_philox_noise_ahead_f32(int uniform, float* out, int64_t n, uint64_t seed, uint64_t stream_id, int64_t elem_offset,
                                            float sub, float mul, float add, float factor, float threshold_std_devs, double* partials,
                                            int have_stats, uint64_t next_stream_id, double* partials_next, void* stream) {
    SONAR_REQUIRE(out && partials && partials_next && partials != partials_next && n >= 0 && elem_offset >= 0, SONAR_ERR_ARG,
                  "sonar_philox_noise_ahead_f32: bad argument");
    const int active = !(sub == 0.0f && mul == 1.0f && add == 0.0f);
    const Affine aff{sub, mul, add, uniform ? active : 0};
    if (uniform)
        return launch_fill_ahead<Dist::Uniform>(out, n, seed, stream_id, elem_offset, aff, factor, threshold_std_devs, partials, have_stats,
                                                next_stream_id, partials_next, (hipStream_t)stream, "sonar_philox_noise_ahead_f32");
    return launch_fill_ahead<Dist::Normal>(out, n, seed, stream_id, elem_offset, aff, factor, threshold_std_devs, partials, have_stats,
                                           next_stream_id, partials_next, (hipStream_t)stream, "sonar_philox_noise_ahead_f32");
}

static int brownian_launch(float* out, float* w_out, const float* prev, float scale, int64_t n, int64_t elem_offset,
                           const uint64_t* node_ids, const float* coefs, int nnodes, uint64_t seed, const uint64_t* latent_seeds,
                           int64_t latent_elems, void* stream, const char* what, BrownianBase base = BrownianBase{nullptr, nullptr, 0.0f, 0.0f},
                           Accum acc = kNoAccum, double* partials = nullptr, const sonar_fold_prefix* pre = nullptr) {
    SONAR_REQUIRE((out || w_out) && n >= 0 && elem_offset >= 0 && nnodes >= 0 && (nnodes == 0 || (node_ids && coefs)), SONAR_ERR_ARG,
                  "%s: bad argument", what);
    SONAR_REQUIRE(!latent_seeds || (elem_offset & 3) == 0, SONAR_ERR_ARG, "%s: per-latent seeds need an element offset of whole 4-element groups", what);
    SONAR_REQUIRE(nnodes <= kMaxBrownianNodes, SONAR_ERR_UNSUPPORTED, "%s: more than %d path nodes", what, kMaxBrownianNodes);
    SONAR_REQUIRE(!latent_seeds || (latent_elems > 0 && latent_elems % 4 == 0 && n % latent_elems == 0), SONAR_ERR_ARG,
                  "%s: per-latent seeds need whole latents of a multiple of 4 elements", what);
    for (int k = 0; k < nnodes; ++k) SONAR_REQUIRE((node_ids[k] >> 48) == 0, SONAR_ERR_ARG, "%s: node ids are 48-bit", what);
    if (n == 0) return SONAR_OK;
    BrownianTerms t;
    BrownianBurstTerms bt;
    t.count = bt.count = nnodes;
    for (int k = 0; k < nnodes; ++k) {
        t.node[k] = node_ids[k];
        t.coef[k] = bt.coef[k] = coefs[k];
        uint64_t h = node_ids[k] + 0x9E3779B97F4A7C15ull;  // splitmix64 of the node id
        h = (h ^ (h >> 30)) * 0xBF58476D1CE4E5B9ull;
        h = (h ^ (h >> 27)) * 0x94D049BB133111EBull;
        h ^= h >> 31;
        bt.hx[k] = (uint32_t)h;
        bt.hc[k] = (uint32_t)(h >> 32);
    }
    auto al = [](const void* p) { return !p || (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    // the variant is a function of the latent size and the seed kind only, so every shard of a batch picks the same one
    const bool burst = !latent_seeds && latent_elems > 0 && latent_elems % kTileElems == 0 && n % latent_elems == 0 &&
                       elem_offset % latent_elems == 0 && al(out) && al(w_out) && al(prev) && al(base.a) && al(base.b) && al(acc.y);
    SONAR_REQUIRE(!partials || out, SONAR_ERR_ARG, "%s: statistics need an output tensor", what);
    // statistics: one (sum, sumsq) pair per block, at most kNPart blocks
    const int cap = partials ? kNPart : kMaxGrid;
    Prefix px{1.0f, 1.0f, 1.0f, 0, 0, nullptr, 1, 0};
    if (pre) {
        // the previous chain item rides along only in the tile kernel, on the same running sum, with whole generator tiles per latent
        const int prc = make_prefix(pre, px, what);
        if (prc != SONAR_OK) return prc;
        SONAR_REQUIRE(burst && acc.y && out == acc.y && (pre->kind != SONAR_PREFIX_PERLIN || pre->chw == latent_elems), SONAR_ERR_UNSUPPORTED,
                      "%s: this shape cannot host a fold prefix (apply it with its own entry point)", what);
    }
#define SONAR_BBL(P, BL) \
    hipLaunchKernelGGL((brownian_burst_kernel<P, BL>), dim3(std::min(cap, grid_for(n / (kBrownTile * (P ? kBrownPerTile : 1)), BL / 64))), dim3(BL), 0, \
                       (hipStream_t)stream, out, n, elem_offset, bt, seed, prev, w_out, scale, base, acc, partials, px)
#define SONAR_BB(P) \
    do { \
        if (nnodes >= kBrownLongTerms) SONAR_BBL(P, kBrownBlock); \
        else SONAR_BBL(P, kBlock); \
    } while (0)
    if (burst && pre && pre->kind == SONAR_PREFIX_NORMAL)
        SONAR_BB(1);
    else if (burst && pre)
        SONAR_BB(2);
    else if (burst)
        SONAR_BB(0);
#undef SONAR_BBL
#undef SONAR_BB
    else
        hipLaunchKernelGGL(brownian_kernel, dim3(std::min(cap, grid_for((n + 6) / 4, kBlock))), dim3(kBlock), 0, (hipStream_t)stream, out, n,
                           elem_offset, t, seed, reinterpret_cast<const unsigned long long*>(latent_seeds), latent_elems, prev, w_out, scale,
                           base, acc, partials);
    return check_launch(what);
}

extern "C" int sonar_brownian_f32(float* out, int64_t n, int64_t elem_offset, const uint64_t* node_ids, const float* coefs,
                                  int nnodes, uint64_t seed, const uint64_t* latent_seeds, int64_t latent_elems, void* stream) {
    SONAR_REQUIRE(out, SONAR_ERR_ARG, "sonar_brownian_f32: bad argument");
    return brownian_launch(out, nullptr, nullptr, 1.0f, n, elem_offset, node_ids, coefs, nnodes, seed, latent_seeds, latent_elems, stream,
                           "sonar_brownian_f32");
}

extern "C" int sonar_brownian_point_f32(float* out, float* w_out, const float* prev, float scale, int64_t n, int64_t elem_offset,
                                        const uint64_t* node_ids, const float* coefs, int nnodes, uint64_t seed,
                                        const uint64_t* latent_seeds, int64_t latent_elems, void* stream) {
    return brownian_launch(out, w_out, prev, scale, n, elem_offset, node_ids, coefs, nnodes, seed, latent_seeds, latent_elems, stream,
                           "sonar_brownian_point_f32");
}

extern "C" int sonar_brownian_bridge_f32(float* out, float* w_out, const float* prev, float scale, const float* base_a, float fa,
                                         const float* base_b, float fb, int64_t n, int64_t elem_offset, const uint64_t* node_ids,
                                         const float* coefs, int nnodes, uint64_t seed, const uint64_t* latent_seeds,
                                         int64_t latent_elems, double* partials, void* stream) {
    return brownian_launch(out, w_out, prev, scale, n, elem_offset, node_ids, coefs, nnodes, seed, latent_seeds, latent_elems, stream,
                           "sonar_brownian_bridge_f32", BrownianBase{base_a, base_b, fa, fb}, kNoAccum, partials);
}

// accumulating forms: y <- y * y_mul + x * x_mul with x the generator's values (never stored), partials (nullable) <- statistics of y
static bool accum_ok(const sonar_accumulate* a) { return a && a->y; }

extern "C" int sonar_brownian_bridge_acc_f32(const sonar_accumulate* acc, float* w_out, const float* prev, float scale, const float* base_a,
                                             float fa, const float* base_b, float fb, int64_t n, int64_t elem_offset,
                                             const uint64_t* node_ids, const float* coefs, int nnodes, uint64_t seed,
                                             const uint64_t* latent_seeds, int64_t latent_elems, void* stream) {
    SONAR_REQUIRE(accum_ok(acc), SONAR_ERR_ARG, "sonar_brownian_bridge_acc_f32: bad argument");
    return brownian_launch(acc->y, w_out, prev, scale, n, elem_offset, node_ids, coefs, nnodes, seed, latent_seeds, latent_elems, stream,
                           "sonar_brownian_bridge_acc_f32", BrownianBase{base_a, base_b, fa, fb}, Accum{acc->y, acc->y_mul, acc->x_mul},
                           acc->partials);
}

extern "C" int sonar_brownian_bridge_chain_f32(const sonar_accumulate* acc, const sonar_fold_prefix* pre, float* w_out, const float* prev,
                                               float scale, const float* base_a, float fa, const float* base_b, float fb, int64_t n,
                                               int64_t elem_offset, const uint64_t* node_ids, const float* coefs, int nnodes, uint64_t seed,
                                               int64_t latent_elems, void* stream) {
    SONAR_REQUIRE(accum_ok(acc) && pre, SONAR_ERR_ARG, "sonar_brownian_bridge_chain_f32: bad argument");
    return brownian_launch(acc->y, w_out, prev, scale, n, elem_offset, node_ids, coefs, nnodes, seed, nullptr, latent_elems, stream,
                           "sonar_brownian_bridge_chain_f32", BrownianBase{base_a, base_b, fa, fb}, Accum{acc->y, acc->y_mul, acc->x_mul},
                           acc->partials, pre);
}

extern "C" int sonar_philox_normal_acc_f32(const sonar_accumulate* acc, int64_t n, uint64_t seed, uint64_t stream_id, int64_t elem_offset,
                                           void* stream) {
    SONAR_REQUIRE(accum_ok(acc) && n >= 0 && elem_offset >= 0, SONAR_ERR_ARG, "sonar_philox_normal_acc_f32: bad argument");
    return launch_fill<Dist::Normal>(acc->y, n, seed, stream_id, elem_offset, Affine{0.f, 1.f, 0.f, 0}, acc->partials, (hipStream_t)stream,
                                     "sonar_philox_normal_acc_f32", Accum{acc->y, acc->y_mul, acc->x_mul});
}

extern "C" int sonar_perlin_generate_acc_f32(const sonar_accumulate* acc, const float* terms, int64_t B, int64_t chw, int64_t iters,
                                             float div_fac, uint64_t seed, uint64_t stream_id, int64_t elem_offset, void* stream) {
    SONAR_REQUIRE(accum_ok(acc) && (terms || iters == 0) && B >= 0 && chw > 0 && chw < (1LL << 31) && iters >= 0 && elem_offset >= 0,
                  SONAR_ERR_ARG, "sonar_perlin_generate_acc_f32: bad argument");
    return launch_perlin_generate<0>(terms, acc->y, B, chw, iters, div_fac, seed, stream_id, elem_offset, acc->partials, NormArgs{},
                                     (hipStream_t)stream, "sonar_perlin_generate_acc_f32", Accum{acc->y, acc->y_mul, acc->x_mul});
}

// the accumulating forms with the chain's previous item riding along (sonar_fold_prefix, see sonar_hip.h)
extern "C" int sonar_philox_normal_chain_f32(const sonar_accumulate* acc, const sonar_fold_prefix* pre, int64_t n, uint64_t seed,
                                             uint64_t stream_id, int64_t elem_offset, void* stream) {
    return launch_pair_fold(acc, pre, SONAR_PREFIX_NORMAL, Prefix{1.0f, 1.0f, 1.0f, seed, stream_id, nullptr, 1, 0}, n, elem_offset,
                            (hipStream_t)stream, "sonar_philox_normal_chain_f32");
}

extern "C" int sonar_perlin_generate_chain_f32(const sonar_accumulate* acc, const sonar_fold_prefix* pre, const float* terms, int64_t B,
                                               int64_t chw, float div_fac, uint64_t seed, uint64_t stream_id, int64_t elem_offset,
                                               void* stream) {
    SONAR_REQUIRE(terms && aligned16(terms) && B >= 0 && chw > 0 && chw < (1LL << 31), SONAR_ERR_ARG,
                  "sonar_perlin_generate_chain_f32: bad argument (the summed, 16-byte aligned lattice is required)");
    return launch_pair_fold(acc, pre, SONAR_PREFIX_PERLIN, Prefix{1.0f, 1.0f, div_fac, seed, stream_id, terms, (int)chw, 0}, B * chw, elem_offset,
                            (hipStream_t)stream, "sonar_perlin_generate_chain_f32");
}

extern "C" int sonar_perlin_terms_f32(const float* angles, float* terms, int64_t iters, int64_t C, int64_t H, int64_t W,
                                      int blend_mode, void* stream) {
    SONAR_REQUIRE(angles && terms && iters >= 0 && C > 0 && H > 0 && W > 0 && blend_mode >= 0 && blend_mode <= 2,
                  SONAR_ERR_ARG, "sonar_perlin_terms_f32: bad argument");
    SONAR_REQUIRE(H < (1 << 20) && W < (1 << 20), SONAR_ERR_UNSUPPORTED, "sonar_perlin_terms_f32: plane too large");
    if (iters == 0) return SONAR_OK;
    const int64_t total = iters * C * H * W;
    hipLaunchKernelGGL(perlin_terms_kernel, dim3(grid_for(total, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, angles,
                       terms, iters * C, (int)H, (int)W, blend_mode);
    return check_launch("sonar_perlin_terms_f32");
}

extern "C" int sonar_perlin_lattice_f32(float* terms_sum, int64_t iters, int64_t C, int64_t H, int64_t W, int blend_mode, uint64_t seed,
                                        uint64_t stream_id, void* stream) {
    SONAR_REQUIRE(terms_sum && iters >= 0 && iters < (1 << 20) && C > 0 && H > 0 && W > 0 && blend_mode >= 0 && blend_mode <= 2,
                  SONAR_ERR_ARG, "sonar_perlin_lattice_f32: bad argument");
    SONAR_REQUIRE(H < (1 << 20) && W < (1 << 20), SONAR_ERR_UNSUPPORTED, "sonar_perlin_lattice_f32: plane too large");
    hipLaunchKernelGGL(perlin_lattice_kernel, dim3(grid_for(C * H * W, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, terms_sum,
                       (int)iters, C, (int)H, (int)W, blend_mode, seed, stream_id);
    return check_launch("sonar_perlin_lattice_f32");
}

extern "C" int sonar_perlin_apply_f32(const float* base, const float* terms, float* out, int64_t B, int64_t chw,
                                      int64_t iters, float div_fac, double* partials, void* stream) {
    SONAR_REQUIRE(base && out && (terms || iters == 0) && B >= 0 && chw > 0 && iters >= 0, SONAR_ERR_ARG,
                  "sonar_perlin_apply_f32: bad argument");
    return launch_perlin_apply(base, terms, out, B, chw, iters, div_fac, partials, (hipStream_t)stream, "sonar_perlin_apply_f32");
}

extern "C" int sonar_perlin_generate_f32(const float* terms, float* out, int64_t B, int64_t chw, int64_t iters,
                                         float div_fac, uint64_t seed, uint64_t stream_id, int64_t elem_offset,
                                         double* partials, void* stream) {
    SONAR_REQUIRE(out && (terms || iters == 0) && B >= 0 && chw > 0 && chw < (1LL << 31) && iters >= 0 && elem_offset >= 0, SONAR_ERR_ARG,
                  "sonar_perlin_generate_f32: bad argument");
    return launch_perlin_generate<0>(terms, out, B, chw, iters, div_fac, seed, stream_id, elem_offset, partials, NormArgs{},
                                     (hipStream_t)stream, "sonar_perlin_generate_f32");
}

extern "C" int sonar_perlin_noise_f32(const float* terms, float* out, int64_t B, int64_t chw, int64_t iters, float div_fac,
                                      uint64_t seed, uint64_t stream_id, int64_t elem_offset, float factor,
                                      float threshold_std_devs, double* partials, void* stream) {
    SONAR_REQUIRE(out && partials && (terms || iters == 0) && B >= 0 && chw > 0 && chw < (1LL << 31) && iters >= 0 && elem_offset >= 0, SONAR_ERR_ARG,
                  "sonar_perlin_noise_f32: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const NormArgs na{partials, B * chw, factor, threshold_std_devs};
    int rc = launch_perlin_generate<1>(terms, out, B, chw, iters, div_fac, seed, stream_id, elem_offset, partials, NormArgs{}, st,
                                       "sonar_perlin_noise_f32(stats)");
    if (rc != SONAR_OK) return rc;
    return launch_perlin_generate<2>(terms, out, B, chw, iters, div_fac, seed, stream_id, elem_offset, nullptr, na, st,
                                     "sonar_perlin_noise_f32(write)");
}

extern "C" int sonar_perlin_noise_ahead_ok(int64_t B, int64_t chw, int64_t elem_offset) {
    // the fast, tile-aligned shape of the device-drawn call (whole RNG tiles per latent, shards on tile boundaries).  (Up to round 4 also
    // "at most 4096 tiles": the launch-bound sizes only.  At 512 SDXL latents the statistics of the next call ride under the final pass's
    // stores: 41.9 -> see profiles/r05_fill_rates.txt.)
    const int64_t n = B * chw;
    return (n > 0 && chw >= 256 && chw < (1LL << 31) && chw % kTileElems == 0 && elem_offset >= 0 && elem_offset % kTileElems == 0 &&
            n / kTileElems <= (1 << 20)) ? 1 : 0;
}

extern "C" int sonar_perlin_noise_ahead_f32(const float* terms, float* out, int64_t B, int64_t chw, float div_fac, uint64_t seed, uint64_t stream_id,
                                            int64_t elem_offset, float factor, float threshold_std_devs, double* partials, int have_stats,
                                            uint64_t next_stream_id, const float* terms_next, double* partials_next, float* lattice_out,
                                            int64_t lattice_iters, int64_t C, int64_t H, int64_t W, int blend_mode, uint64_t lattice_stream_id,
                                            void* stream) {
    const char* what = "sonar_perlin_noise_ahead_f32";
    SONAR_REQUIRE(terms && out && partials && B >= 0 && chw > 0 && elem_offset >= 0 && (!terms_next || partials_next) &&
                      partials_next != partials, SONAR_ERR_ARG, "%s: bad argument", what);
    SONAR_REQUIRE(!lattice_out || (lattice_iters >= 0 && lattice_iters < (1 << 20) && C > 0 && H > 0 && W > 0 && H < (1 << 20) && W < (1 << 20) &&
                                   C * H * W == chw && blend_mode >= 0 && blend_mode <= 2 && lattice_out != terms && lattice_out != terms_next),
                  SONAR_ERR_ARG, "%s: bad lattice request", what);
    SONAR_REQUIRE(sonar_perlin_noise_ahead_ok(B, chw, elem_offset) && aligned16(out) && aligned16(terms) && (!terms_next || aligned16(terms_next)),
                  SONAR_ERR_UNSUPPORTED, "%s: whole 4096-element tiles per latent, 16-byte aligned tensors", what);
    if (B == 0) return SONAR_OK;
    hipStream_t st = (hipStream_t)stream;
    if (!have_stats) {
        const int rc = launch_perlin_generate<1>(terms, out, B, chw, 1, div_fac, seed, stream_id, elem_offset, partials, NormArgs{}, st, what);
        if (rc != SONAR_OK) return rc;
    }
    PerlinAhead a{};
    a.terms = terms;
    a.out = out;
    a.n = B * chw;
    a.chw = chw;
    a.div_fac = div_fac;
    a.seed = seed;
    a.stream_id = stream_id;
    a.elem_offset = elem_offset;
    a.na = NormArgs{partials, B * chw, factor, threshold_std_devs};
    a.next_stream_id = next_stream_id;
    a.terms_next = terms_next;
    a.partials_next = partials_next;
    a.lattice_out = lattice_out;
    a.lat_blocks = lattice_out ? grid_for(chw, kBlock) : 0;
    a.lat_iters = (int)lattice_iters;
    a.blend_mode = blend_mode;
    a.C = C;
    a.H = (int)H;
    a.W = (int)W;
    a.lattice_stream_id = lattice_stream_id;
    a.tile_blocks = tile_grid(a.n, elem_offset);
    // a wave with a tile or more of its own to store has stores in flight to hide the next call's statistics behind (batch >= 256 SDXL
    // latents: the capped grid); below that the two passes go to separate blocks, twice as many waves with one short chain each
    static const int fuse_env = [] { const char* e = getenv("SONAR_PERLIN_FUSED"); return e ? atoi(e) : -1; }();  // (A/B)
    a.fused = terms_next && (fuse_env >= 0 ? fuse_env != 0 : a.tile_blocks == kNPart) ? 1 : 0;
    hipLaunchKernelGGL(perlin_ahead_kernel, dim3(a.lat_blocks + a.tile_blocks * (terms_next && !a.fused ? 2 : 1)), dim3(kBlock), 0, st, a);
    return check_launch(what);
}

extern "C" int sonar_resample_acc_f32(float* dst, const float* src, int64_t planes, int64_t H, int64_t W, int64_t h,
                                      int64_t w, float scale, int mode, int accumulate, double* partials, void* stream) {
    SONAR_REQUIRE(dst && src && planes >= 0 && H > 0 && W > 0 && h > 0 && w > 0 && mode >= 0 && mode <= 6, SONAR_ERR_ARG,
                  "sonar_resample_acc_f32: bad argument");
    SONAR_REQUIRE(H < (1 << 24) && W < (1 << 24) && h < (1 << 24) && w < (1 << 24), SONAR_ERR_UNSUPPORTED,
                  "sonar_resample_acc_f32: plane too large");
    if (planes == 0) return SONAR_OK;
    const int g = (int)std::min<int64_t>(kNPart, grid_for(planes * H * W, kBlock * 2));
    if (partials)
        hipLaunchKernelGGL((resample_acc_kernel<true>), dim3(g), dim3(kBlock), 0, (hipStream_t)stream, dst, src, planes,
                           (int)H, (int)W, (int)h, (int)w, scale, mode, accumulate, partials);
    else
        hipLaunchKernelGGL((resample_acc_kernel<false>), dim3(g), dim3(kBlock), 0, (hipStream_t)stream, dst, src, planes,
                           (int)H, (int)W, (int)h, (int)w, scale, mode, accumulate, partials);
    return check_launch("sonar_resample_acc_f32");
}

static int pyramid_common(const char* what, float* out, int64_t planes, int64_t H, int64_t W, int mode, int64_t elem_offset) {
    SONAR_REQUIRE(out && planes >= 0 && H > 0 && W > 0 && mode >= 0 && mode <= 2 && elem_offset >= 0, SONAR_ERR_ARG,
                  "%s: bad argument", what);
    SONAR_REQUIRE(W % 4 == 0 && aligned16(out) && elem_offset % 4 == 0, SONAR_ERR_UNSUPPORTED,
                  "%s: needs W %% 4 == 0 and 16-byte aligned output", what);
    return SONAR_OK;
}

extern "C" int sonar_pyramid_generate_f32(float* out, int64_t planes, int64_t H, int64_t W, int64_t nlevels,
                                          const float* const* level_ptrs, const int64_t* level_h, const int64_t* level_w,
                                          const float* level_weight, int mode, uint64_t seed, uint64_t stream_id,
                                          int64_t elem_offset, double* partials, void* stream) {
    int rc = pyramid_common("sonar_pyramid_generate_f32", out, planes, H, W, mode, elem_offset);
    if (rc != SONAR_OK) return rc;
    PyramidLevels lv;
    bool drawn = false;
    rc = fill_levels(lv, H, W, nlevels, level_ptrs, level_h, level_w, level_weight, stream_id, &drawn, "sonar_pyramid_generate_f32");
    if (rc != SONAR_OK) return rc;
    if (planes == 0) return SONAR_OK;
    if (launch_pyramid_plane(out, planes, H, W, lv, mode, seed, stream_id, elem_offset, partials, (hipStream_t)stream))
        return check_launch("sonar_pyramid_generate_f32");
    SONAR_REQUIRE(!drawn, SONAR_ERR_UNSUPPORTED, "sonar_pyramid_generate_f32: in-kernel level grids need the plane kernel (H*W %% 4096 == 0, "
                  "whole planes, grids within the LDS budget): pass the grids explicitly");
    const int g = tile_grid(planes * H * W, elem_offset);
    if (partials)
        hipLaunchKernelGGL((pyramid_generate_kernel<0, true>), dim3(g), dim3(kBlock), 0, (hipStream_t)stream, out, planes,
                           (int)H, (int)W, lv, mode, seed, stream_id, elem_offset, partials, NormArgs{});
    else
        hipLaunchKernelGGL((pyramid_generate_kernel<0, false>), dim3(g), dim3(kBlock), 0, (hipStream_t)stream, out, planes,
                           (int)H, (int)W, lv, mode, seed, stream_id, elem_offset, partials, NormArgs{});
    return check_launch("sonar_pyramid_generate_f32");
}

// The same values folded into a chain's running sum (y <- y * y_mul + pyramid * x_mul, sonar_accumulate), optionally with the chain's
// previous item riding along (sonar_fold_prefix).  Plane kernel only: SONAR_ERR_UNSUPPORTED when it cannot run this shape.
static int pyramid_generate_acc(const char* what, const sonar_accumulate* acc, const sonar_fold_prefix* pre, int64_t planes, int64_t H, int64_t W,
                                int64_t nlevels, const float* const* level_ptrs, const int64_t* level_h, const int64_t* level_w,
                                const float* level_weight, int mode, uint64_t seed, uint64_t stream_id, int64_t elem_offset, LatticeJob lat,
                                void* stream) {
    SONAR_REQUIRE(acc && acc->y, SONAR_ERR_ARG, "%s: bad argument", what);
    int rc = pyramid_common(what, acc->y, planes, H, W, mode, elem_offset);
    if (rc != SONAR_OK) return rc;
    PyramidLevels lv;
    bool drawn = false;
    rc = fill_levels(lv, H, W, nlevels, level_ptrs, level_h, level_w, level_weight, stream_id, &drawn, what);
    if (rc != SONAR_OK) return rc;
    Prefix px{1.0f, 1.0f, 1.0f, 0, 0, nullptr, 1, 0};
    if (pre) {
        rc = make_prefix(pre, px, what);
        if (rc != SONAR_OK) return rc;
        SONAR_REQUIRE(pre->kind != SONAR_PREFIX_PERLIN || pre->chw % (H * W) == 0, SONAR_ERR_UNSUPPORTED,
                      "%s: the Perlin prefix's latent is not a whole number of planes", what);
    }
    if (planes == 0) return SONAR_OK;
    SONAR_REQUIRE(launch_pyramid_plane(acc->y, planes, H, W, lv, mode, seed, stream_id, elem_offset, acc->partials, (hipStream_t)stream,
                                       Accum{acc->y, acc->y_mul, acc->x_mul}, pre ? pre->kind : 0, px, lat),
                  SONAR_ERR_UNSUPPORTED, "%s: the plane kernel cannot run this shape (whole planes, grids within the LDS budget)", what);
    return check_launch(what);
}

extern "C" int sonar_pyramid_generate_acc_f32(const sonar_accumulate* acc, const sonar_fold_prefix* pre, int64_t planes, int64_t H, int64_t W,
                                              int64_t nlevels, const float* const* level_ptrs, const int64_t* level_h,
                                              const int64_t* level_w, const float* level_weight, int mode, uint64_t seed,
                                              uint64_t stream_id, int64_t elem_offset, void* stream) {
    return pyramid_generate_acc("sonar_pyramid_generate_acc_f32", acc, pre, planes, H, W, nlevels, level_ptrs, level_h, level_w, level_weight, mode,
                                seed, stream_id, elem_offset, kNoLattice, stream);
}

extern "C" int sonar_pyramid_generate_acc_ahead_f32(const sonar_accumulate* acc, const sonar_fold_prefix* pre, int64_t planes, int64_t H, int64_t W,
                                                    int64_t nlevels, const float* const* level_ptrs, const int64_t* level_h,
                                                    const int64_t* level_w, const float* level_weight, int mode, uint64_t seed,
                                                    uint64_t stream_id, int64_t elem_offset, float* lattice_out, int64_t lattice_iters,
                                                    int64_t lattice_channels, int blend_mode, uint64_t lattice_stream_id, void* stream) {
    const char* what = "sonar_pyramid_generate_acc_ahead_f32";
    SONAR_REQUIRE(pre && pre->kind == SONAR_PREFIX_PERLIN && lattice_out && lattice_out != pre->terms && lattice_iters >= 0 &&
                      lattice_iters < (1 << 20) && lattice_channels > 0 && blend_mode >= 0 && blend_mode <= 2 && H > 0 && W > 0 &&
                      lattice_channels * H * W == pre->chw,
                  SONAR_ERR_ARG, "%s: a hosted Perlin item and a lattice buffer for the later call's [C, H, W] table are required", what);
    LatticeJob lat{lattice_out, grid_for(pre->chw, kPyrBlock), (int)lattice_iters, blend_mode, lattice_channels, seed, lattice_stream_id};
    return pyramid_generate_acc(what, acc, pre, planes, H, W, nlevels, level_ptrs, level_h, level_w, level_weight, mode, seed, stream_id, elem_offset,
                                lat, stream);
}

extern "C" int sonar_levels_sampled_f32(float* out, int64_t planes, int64_t H, int64_t W, int nlevels, const int64_t* level_h,
                                       const int64_t* level_w, const float* level_weight, const float* level_sd, int mode, uint64_t seed,
                                       uint64_t stream_id, int64_t plane_offset, int accumulate, void* stream) {
    SONAR_REQUIRE(out && planes >= 0 && H > 0 && W > 0 && nlevels >= 0 && nlevels <= kMaxSampledLevels && plane_offset >= 0 &&
                      (nlevels == 0 || (level_h && level_w && level_weight && level_sd)),
                  SONAR_ERR_ARG, "sonar_levels_sampled_f32: bad argument (at most %d levels)", kMaxSampledLevels);
    SONAR_REQUIRE(mode >= 0 && mode <= 4, SONAR_ERR_UNSUPPORTED, "sonar_levels_sampled_f32: mode %d", mode);
    SampledLevels lv;
    lv.count = nlevels;
    for (int l = 0; l < nlevels; ++l) {
        SONAR_REQUIRE(level_h[l] > 0 && level_w[l] > 0 && level_h[l] < (1ll << 30) && level_w[l] < (1ll << 30) &&
                          (double)(plane_offset + planes) * (double)level_h[l] * (double)level_w[l] < 9.0e18,
                      SONAR_ERR_UNSUPPORTED, "sonar_levels_sampled_f32: level %d is too large for its element keys", l);
        // area: only whole blocks average independent values (any other ratio's windows overlap: draw the level and pool it)
        SONAR_REQUIRE(mode != 2 || (level_h[l] % H == 0 && level_w[l] % W == 0), SONAR_ERR_UNSUPPORTED,
                      "sonar_levels_sampled_f32: area mode needs levels of whole multiples of the output size");
        lv.h[l] = (int)level_h[l];
        lv.w[l] = (int)level_w[l];
        lv.weight[l] = level_weight[l];
        lv.sd[l] = level_sd[l];
    }
    if (planes == 0) return SONAR_OK;
    const int g = grid_for(planes * H * W, kBlock);
    hipStream_t st = (hipStream_t)stream;
#define SONAR_LS(M) hipLaunchKernelGGL(levels_sampled_kernel<M>, dim3(g), dim3(kBlock), 0, st, out, planes, (int)H, (int)W, lv, seed, stream_id, plane_offset, accumulate)
    if (mode == 0) SONAR_LS(0); else if (mode == 1) SONAR_LS(1); else if (mode == 2) SONAR_LS(2); else if (mode == 3) SONAR_LS(3); else SONAR_LS(4);
#undef SONAR_LS
    return check_launch("sonar_levels_sampled_f32");
}

extern "C" int sonar_level_normal_f32(float* level, int64_t planes, int64_t h, int64_t w, float sd, uint64_t seed, uint64_t stream_id,
                                      int64_t plane_offset, void* stream) {
    SONAR_REQUIRE(level && planes >= 0 && h > 0 && w > 0 && plane_offset >= 0, SONAR_ERR_ARG, "sonar_level_normal_f32: bad argument");
    const int64_t n = planes * h * w;
    if (n == 0) return SONAR_OK;
    hipLaunchKernelGGL(level_normal_kernel, dim3(grid_for(n / 4 + 2, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, level, n, sd, seed, stream_id,
                       plane_offset * h * w);
    return check_launch("sonar_level_normal_f32");
}

extern "C" int sonar_pyramid_noise_f32(float* out, int64_t planes, int64_t H, int64_t W, int64_t nlevels,
                                       const float* const* level_ptrs, const int64_t* level_h, const int64_t* level_w,
                                       const float* level_weight, int mode, uint64_t seed, uint64_t stream_id,
                                       int64_t elem_offset, float factor, float threshold_std_devs, double* partials,
                                       void* stream) {
    int rc = pyramid_common("sonar_pyramid_noise_f32", out, planes, H, W, mode, elem_offset);
    if (rc != SONAR_OK) return rc;
    SONAR_REQUIRE(partials, SONAR_ERR_ARG, "sonar_pyramid_noise_f32: partials workspace required");
    PyramidLevels lv;
    bool drawn = false;
    rc = fill_levels(lv, H, W, nlevels, level_ptrs, level_h, level_w, level_weight, stream_id, &drawn, "sonar_pyramid_noise_f32");
    if (rc != SONAR_OK) return rc;
    if (planes == 0) return SONAR_OK;
    // The level gathers make a re-draw cost more than a sweep: generate once (with statistics), then normalise in place
    int slots = kNPart;
    if (launch_pyramid_plane(out, planes, H, W, lv, mode, seed, stream_id, elem_offset, partials, (hipStream_t)stream, kNoAccum, 0,
                             Prefix{1.0f, 1.0f, 1.0f, 0, 0, nullptr, 1, 0}, kNoLattice, &slots)) {
        rc = check_launch("sonar_pyramid_noise_f32");
        if (rc != SONAR_OK) return rc;
        // the normalising pass reduces only the pairs the plane kernel's workgroups own (the others are zeros: the same sums, bit for bit,
        // from a quarter of the reads at batch 64, where every one of its 2048 workgroups reduced 16 KB of partials for 8 KB of values)
        return sonar_scale_noise_f32(out, planes * H * W, factor, 1, threshold_std_devs, partials, slots, planes * H * W, stream);
    }
    SONAR_REQUIRE(!drawn, SONAR_ERR_UNSUPPORTED, "sonar_pyramid_noise_f32: in-kernel level grids need the plane kernel: pass the grids explicitly");
    const int g = tile_grid(planes * H * W, elem_offset);
    const NormArgs na{partials, planes * H * W, factor, threshold_std_devs};
    hipLaunchKernelGGL((pyramid_generate_kernel<1, false>), dim3(g), dim3(kBlock), 0, (hipStream_t)stream, out, planes, (int)H,
                       (int)W, lv, mode, seed, stream_id, elem_offset, partials, NormArgs{});
    hipLaunchKernelGGL((pyramid_generate_kernel<2, false>), dim3(g), dim3(kBlock), 0, (hipStream_t)stream, out, planes, (int)H,
                       (int)W, lv, mode, seed, stream_id, elem_offset, nullptr, na);
    return check_launch("sonar_pyramid_noise_f32");
}

extern "C" int sonar_pyramid_noise_ahead_f32(float* out, int64_t planes, int64_t H, int64_t W, int64_t nlevels, const int64_t* level_h,
                                             const int64_t* level_w, const float* level_weight, int mode, uint64_t seed, uint64_t stream_id,
                                             int64_t elem_offset, float factor, float threshold_std_devs, double* partials, int have_stats,
                                             uint64_t next_stream_id, int64_t next_nlevels, const int64_t* next_level_h,
                                             const int64_t* next_level_w, const float* next_level_weight, double* partials_next, void* stream) {
    const char* what = "sonar_pyramid_noise_ahead_f32";
    int rc = pyramid_common(what, out, planes, H, W, mode, elem_offset);
    if (rc != SONAR_OK) return rc;
    SONAR_REQUIRE(partials && partials_next && partials != partials_next, SONAR_ERR_ARG, "%s: two statistics workspaces required", what);
    SONAR_REQUIRE(mode == 0 && nlevels <= kMaxLevels && next_nlevels <= kMaxLevels, SONAR_ERR_UNSUPPORTED,
                  "%s: bilinear levels drawn in the kernel only (sonar_pyramid_noise_f32 takes the rest)", what);
    const float* none[kMaxLevels] = {};
    PyramidLevels now, next;
    bool drawn = false;
    rc = fill_levels(now, H, W, nlevels, none, level_h, level_w, level_weight, stream_id, &drawn, what);
    if (rc != SONAR_OK) return rc;
    rc = fill_levels(next, H, W, next_nlevels, none, next_level_h, next_level_w, next_level_weight, next_stream_id, &drawn, what);
    if (rc != SONAR_OK) return rc;
    if (planes == 0) return SONAR_OK;
    int gf = 0;
    SONAR_REQUIRE(W % 4 == 0 && pyramid_xrows_lds(now, H, W, &gf) && pyramid_xrows_lds(next, H, W, &gf), SONAR_ERR_UNSUPPORTED,
                  "%s: a level table is beyond the plane kernel's stretched-rows form (nothing was launched)", what);
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = planes * H * W;
    static const int grid_cap = [] { const char* e = getenv("SONAR_PYR_GRID"); return e ? atoi(e) : kNPart; }();
    const int slots = (int)std::min<int64_t>(planes, std::min(grid_cap, kNPart));
    if (!have_stats) {
        // nobody left this call's statistics: its planes once without stores (the same partials the ordinary generating launch leaves)
        SONAR_REQUIRE(launch_pyramid_ahead(out, planes, H, W, nullptr, stream_id, now, stream_id, seed, elem_offset, NormArgs{}, 0, partials, st),
                      SONAR_ERR_UNSUPPORTED, "%s: shape not taken", what);
    }
    SONAR_REQUIRE(launch_pyramid_ahead(out, planes, H, W, &now, stream_id, next, next_stream_id, seed, elem_offset,
                                       NormArgs{partials, n, factor, threshold_std_devs}, slots, partials_next, st),
                  SONAR_ERR_UNSUPPORTED, "%s: shape not taken", what);
    return check_launch(what);
}

#ifdef SONAR_NG_TRACE
extern "C" int sonar_debug_ng_trace(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(sonar::g_ng_trace), sizeof(sonar::g_ng_trace));
}
#endif
