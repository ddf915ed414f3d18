// WaveletCFG's transform-domain step with ANY per-level, per-orientation band scales, in ONE launch with the coefficients in LDS
// (py/wavelet_cfg.py:750-791, py/wavelet_functions.py:193-238: yl *= l; yh[j][:, :, k] *= s_jk; inverse).
//
// The scaled reconstruction of one tensor v is linear, and with perfect reconstruction along each axis (S_lo A_lo + S_hi A_hi = I,
// the same wavelet both ways) two of a level's three detail bands never have to be formed.  With X = LL_{j-1} (X = v for j = 1),
// lowW = A^W_lo X, LL_j = A^H_lo lowW, cH = A^H_hi lowW, cV = A^H_lo A^W_hi X, cD = A^H_hi A^W_hi X and scales (a_h, a_v, a_d):
//     S^H_hi cH = lowW - S^H_lo LL_j,   S^H_hi cD = highW - S^H_lo cV,   S^W_hi highW = X - S^W_lo lowW
//     R_{j-1} = a_d X + S^W_lo[ (a_h - a_d) lowW + S^H_lo (R_j - a_h LL_j) ] + S^W_hi[ (a_v - a_d) S^H_lo cV ],   R_J = l LL_J
// so a level keeps LL_j and ONE band, cV_j (nothing when a_v == a_d: with one scale per level this is dwt_lowpass.h's pyramid), and
// lowW is one filter pass over a row of the plane below, recomputed on the way up.  For a 128 x 128 plane and db4 that is 2 (67^2 + 37^2 +
// 22^2 + 14^2 + 10^2) values: 53 KB in fp32 (two workgroups per CU), 106 KB in fp64 (one 1024-thread workgroup per CU).  The bands never
// cross HBM: the tensors are read (cond and uncond twice: the level-1 analysis, then the output rows) and the result is written --
// where the three band kernels of dwt_tile.h moved 2.5-3.85 x the 16N bytes of the step.
//   difference-only rules (cond / uncond / final scales all 1):  out = x - (ku u + kt Phi_D(c - u))           one launch
//   any other linear rule: blend(s_u U, s_d (s_c C - s_u U), t) s_f = A C + B U per band:                      two launches,
//       out1 = x - Phi_B(u), out = out1 - Phi_A(c)   (b == nullptr: v = a)
#pragma once
#include "dwt_lowpass.h"

namespace sonar {

template <typename T>
struct BandsArgs {
    int64_t planes;
    int levels;
    int H[kLowMaxLevels + 1], W[kLowMaxLevels + 1];  // [0]: the latent plane; [j]: coefficient plane of level j
    int off_ll[kLowMaxLevels + 1];                    // LDS offsets (elements of T) of LL_j
    int off_v[kLowMaxLevels + 1];                     // ... of cV_j (low H, high W); unused when av == ad at that level
    int off_tmp;                                      // scratch (elements of T)
    int off_tmp1, rows1;                              // level-1 analysis scratch (overlays the deeper levels and the scratch), its tile height
    int rows_up[kLowMaxLevels + 1];                   // rows of level j - 1 per tile of the way up from level j (even)
    int rows_out;                                     // output rows per tile of the last stage (even)
    int off_cu;                                       // byte offset of the staged (a, b) rows of the last stage
    int off_maps;                                     // byte offset of the extension tables
    int map_h[kLowMaxLevels + 1], map_w[kLowMaxLevels + 1];
    T ah[kLowMaxLevels + 1], av[kLowMaxLevels + 1], ad[kLowMaxLevels + 1];  // [1 .. J]: scales of cH, cV, cD
    T yl;                                             // scale of the approximation
    T ku, kt;                                         // result = ku b + kt Phi(v)
    int subtract_from_x, mode_fwd, mode_inv;
    int vec2;                                         // x and out are aligned for pair accesses
    TapsSmall<T> dec, rec;
};

#ifndef SONAR_BANDS_STAGE_AHEAD
#define SONAR_BANDS_STAGE_AHEAD 4  // values per thread of the last stage's NEXT tile requested a tile ahead (0: a tile's loads at its own start)
#endif
#ifndef SONAR_BANDS_STAGE_AHEAD_ONE
#define SONAR_BANDS_STAGE_AHEAD_ONE 6  // ... when the stage reads ONE tensor in fp32 arithmetic (the two launches of a cond / uncond rule): 6 values 187 us per rule,
                                       // 4 values 194; with two tensors staged (difference rules) 4 values 170 us, 6 values 177; in fp64 arithmetic (the tile route's
                                       // deeper-levels call is this form) 6 values cost 12-29 us per rule (same-box sweeps, round 5)
#endif
#ifndef SONAR_BANDS_ROWS_AHEAD
#define SONAR_BANDS_ROWS_AHEAD 1  // level 1 down: the next item's rows requested an item ahead (0: every item waits for its own loads)
#endif
#ifdef SONAR_BANDS_TRACE  // profiling builds (scratch/bands_trace.py): thread 0's cycle stamps of the first plane of every workgroup
__device__ unsigned long long g_bands_trace[512 * 32];
#define SONAR_BANDS_STAMP(slot) do { if (threadIdx.x == 0 && blockIdx.x < 512 && p == (int64_t)blockIdx.x) g_bands_trace[blockIdx.x * 32 + (slot)] = __builtin_readcyclecounter(); } while (0)
// the last stage's three phases, summed over its tiles into slots 20 .. 22
#define SONAR_BANDS_LAP(slot) do { if (threadIdx.x == 0 && blockIdx.x < 512 && p == (int64_t)blockIdx.x) { const unsigned long long now_ = __builtin_readcyclecounter(); g_bands_trace[blockIdx.x * 32 + (slot)] = (y0 == 0 ? 0ull : g_bands_trace[blockIdx.x * 32 + (slot)]) + (now_ - lap_); lap_ = now_; } } while (0)
#define SONAR_BANDS_LAP_BEGIN() unsigned long long lap_ = __builtin_readcyclecounter()
#else
#define SONAR_BANDS_STAMP(slot) do { } while (0)
#define SONAR_BANDS_LAP(slot) do { } while (0)
#define SONAR_BANDS_LAP_BEGIN() do { } while (0)
#endif
template <int NT>
struct WalkN {
    int r, c, dr, dc;
    __device__ __forceinline__ WalkN(int tid, int cols) : r(tid / cols), c(tid - (tid / cols) * cols), dr(NT / cols), dc(NT - (NT / cols) * cols) {}
    __device__ __forceinline__ void next(int cols) {
        r += dr;
        c += dc;
        if (c >= cols) {
            c -= cols;
            r += 1;
        }
    }
};

// TIO: the tensors' element type -- float (the latents) or T (a coefficient plane in the band kernels' workspace: the deeper levels of
// sonar_wcfg_fused_* run through this kernel with the level-1 approximation as their "latent")
template <typename TIO>
struct alignas(2 * sizeof(TIO)) Pair2 {
    TIO x, y;
};
// what the last stage keeps of a tile's rows: (a, b), or a alone when there is no second tensor (b reads as zero)
template <typename TIO>
struct One1 {
    TIO x;
    static constexpr TIO y = TIO(0);
};

// ZERO: the analysis extension is zero padding -- the only mode whose tables hold "no source" entries (-1); every other mode reads a real
// sample for every tap, and its taps carry no clamp and no select (the kernel is instruction-bound: a third of a tap's instructions)
// AHEAD: level 1 down requests an item's rows an item ahead (2 NRS more registers per thread: chosen by the launcher when LDS, not
// registers, decides how many workgroups a CU holds)
// TV: storage type of the cV planes (float under fp64 arithmetic when the process stores detail bands in fp32: sonar_wcfg_hi_storage, the
// deeper levels' call of the tile route).  HASB: a second tensor b (v = a - b, result = ku b + ...); without one the last stage stages single
// values instead of (a, b) pairs.  Both halve an LDS area: the fp64 deeper-levels call fits a CU four times instead of three (round 5).
template <typename T, typename TIO, int FT, int NT, bool ZERO, bool AHEAD = false, typename TV = T, bool HASB = true>
__global__ void __launch_bounds__(NT) wcfg_bands_kernel(const TIO* __restrict__ ta, const TIO* __restrict__ tb, const TIO* xin, TIO* out,
                                                        BandsArgs<T> a) {
    kernarg_touch_for(ta, tb, xin, out, a);
    auto at0 = [](int s) { return ZERO ? max(s, 0) : s; };                       // index of a tap's sample
    auto live = [](int s, T v) { return ZERO ? (s >= 0 ? v : T(0)) : v; };       // its value
    extern __shared__ __align__(16) unsigned char bands_smem[];
    T* const lds = reinterpret_cast<T*>(bands_smem);
    int* const maps = reinterpret_cast<int*>(bands_smem + a.off_maps);
    using In2 = Pair2<TIO>;
    using St = std::conditional_t<HASB, Pair2<TIO>, One1<TIO>>;
    St* const cu = reinterpret_cast<St*>(bands_smem + a.off_cu);
    T* const tmp = lds + a.off_tmp;
    const int tid = threadIdx.x;
    const int J = a.levels;
    // extension tables, the same for every plane: tap j of output i reads table[2 i + F - 1 - j] (source index, -1 = implicit zero)
    for (int j = 1; j <= J; ++j) {
        const int off = a.mode_fwd == kPeriodization ? FT / 2 : 1;
        const int Hp = a.H[j - 1], Wp = a.W[j - 1];
        const int He = (a.mode_fwd == kPeriodization && (Hp & 1)) ? Hp + 1 : Hp, We = (a.mode_fwd == kPeriodization && (Wp & 1)) ? Wp + 1 : Wp;
        for (int i = tid; i < 2 * a.H[j] + FT - 2; i += NT) maps[a.map_h[j] + i] = src_index(i + off - (FT - 1), Hp, He, a.mode_fwd);
        for (int i = tid; i < 2 * a.W[j] + FT - 2; i += NT) maps[a.map_w[j] + i] = src_index(i + off - (FT - 1), Wp, We, a.mode_fwd);
    }
    const int H = a.H[0], W = a.W[0], h1 = a.H[1], w1 = a.W[1];
    const int Wh = (W + 1) >> 1, Ws = 2 * Wh;  // parity-split row of the level-1 scratch: slot(x) = (x & 1) Wh + x / 2
    // fp64 taps: three filters of FT doubles are 6 FT scalar registers; loaded where the compiler likes (once, in front of the plane loop)
    // they do not fit beside the stages' other scalars and come back from vector-register lanes -- one v_readlane and its hazard wait in
    // front of every use (100-170 of them in the loops of the db4 instantiations).  An opaque zero added to the tap index at the top of
    // every stage keeps each stage's loads inside it: two or three 64-byte scalar loads per stage that hit the scalar cache.
    for (int64_t p = blockIdx.x; p < a.planes; p += gridDim.x) {
        const TIO* pa = ta + p * (int64_t)H * W;
        const TIO* pb = tb ? tb + p * (int64_t)H * W : nullptr;
        __syncthreads();  // tables are built; the previous plane's readers are done
        SONAR_BANDS_STAMP(0);
        // ---------------------------------------------------------------- level 1 down: v = a - b from global, along H in registers, along W out of LDS
        {
            T* const ll1 = lds + a.off_ll[1];
            TV* const cv1 = reinterpret_cast<TV*>(lds + a.off_v[1]);
            const bool want_v = a.av[1] != a.ad[1];
            const int* const ymap = maps + a.map_h[1];
            const int* const xmap = maps + a.map_w[1];
            constexpr int THS = 4, NRS = 2 * THS + FT - 2;
            T* const tmp1 = lds + a.off_tmp1;
            for (int y0 = 0; y0 < h1; y0 += a.rows1) {
                const int tz = tap_zero<T, FT>();
                const int th = min(a.rows1, h1 - y0);
                // An item's NRS rows come straight from global memory; a thread walks ~4 items per tile and used to wait for each item's loads in
                // turn (`scratch/bands_trace.py`: this stage is a third of a plane's time in the single-launch kernel).  Round 5: the NEXT item's
                // values are requested before this one's are filtered -- unconditionally (a thread's last item asks for itself again), so
                // that the loaded registers never merge with old values in a waiting copy.
                // (2 NRS more registers per thread: an instantiation of its own, launched when at most two workgroups fit a CU's LDS anyway --
                // with the smaller footprints of rules without a cV band, or of the tile route's deeper levels, they cost a resident workgroup:
                // 208 -> 277 us on the pair rule)
                constexpr bool kRowsAhead = AHEAD;
                TIO na[kRowsAhead ? NRS : 1], nb[kRowsAhead ? NRS : 1];
                auto request_rows = [&](int sub, int x) {
#pragma unroll
                    for (int r = 0; r < NRS; ++r) {
                        const int sy = ymap[min(2 * (y0 + sub * THS) + r, 2 * h1 + FT - 3)];
                        const int at = at0(sy) * W + x;
                        na[r] = pa[at];
                        nb[r] = pb ? pb[at] : TIO(0);
                    }
                };
                WalkN<NT> wk(tid, W);
                if (kRowsAhead && wk.r * THS < th) request_rows(wk.r, wk.c);
                for (; wk.r * THS < th;) {
                    const int sub = wk.r, x = wk.c;
                    T v[NRS];
#pragma unroll
                    for (int r = 0; r < NRS; ++r) {
                        const int sy = ymap[min(2 * (y0 + sub * THS) + r, 2 * h1 + FT - 3)];
                        if constexpr (kRowsAhead) {
                            v[r] = live(sy, (T)na[r] - (T)nb[r]);
                        } else {
                            const int at = at0(sy) * W + x;
                            const T d = pb ? (T)pa[at] - (T)pb[at] : (T)pa[at];
                            v[r] = live(sy, d);
                        }
                    }
                    wk.next(W);
                    if constexpr (kRowsAhead) {
                        const bool more = wk.r * THS < th;
                        request_rows(more ? wk.r : sub, more ? wk.c : x);
                    }
                    T* dst = tmp1 + (x & 1) * Wh + (x >> 1);
#pragma unroll
                    for (int yl = 0; yl < THS; ++yl) {
                        T lo = T(0);
#pragma unroll
                        for (int j = 0; j < FT; ++j) lo = fma_t(a.dec.lo[tz + j], v[2 * yl + FT - 1 - j], lo);
                        if (sub * THS + yl < th) dst[(sub * THS + yl) * Ws] = lo;
                    }
                }
                __syncthreads();
                for (WalkN<NT> wk(tid, w1); wk.r < th; wk.next(w1)) {
                    const int yl = wk.r, xo = wk.c;
                    const T* row = tmp1 + yl * Ws;
                    const int* xm = xmap + 2 * xo + (FT - 1);
                    T lo = T(0), hi = T(0);
#pragma unroll
                    for (int j = 0; j < FT; ++j) {
                        const int sx = xm[-j];
                        const int sc = at0(sx);
                        const T q = live(sx, row[(sc & 1) * Wh + (sc >> 1)]);
                        lo = fma_t(a.dec.lo[tz + j], q, lo);
                        hi = fma_t(a.dec.hi[tz + j], q, hi);
                    }
                    ll1[(y0 + yl) * w1 + xo] = lo;
                    if (want_v) cv1[(y0 + yl) * w1 + xo] = (TV)hi;
                }
                __syncthreads();
            }
        }
        SONAR_BANDS_STAMP(1);
        // ---------------------------------------------------------------- deeper levels down: LL_j, cV_j from LL_{j-1}, all in LDS
        for (int j = 2; j <= J; ++j) {
            const int tz = tap_zero<T, FT>();
            const int Wp = a.W[j - 1], h = a.H[j], w = a.W[j];
            const T* const src = lds + a.off_ll[j - 1];
            T* const dll = lds + a.off_ll[j];
            TV* const dcv = reinterpret_cast<TV*>(lds + a.off_v[j]);
            const bool want_v = a.av[j] != a.ad[j];
            const int* const ymap = maps + a.map_h[j];
            const int* const xmap = maps + a.map_w[j];
            for (WalkN<NT> wk(tid, Wp); wk.r < h; wk.next(Wp)) {   // along H (low)
                const int yo = wk.r, x = wk.c;
                const int* ym = ymap + 2 * yo + (FT - 1);
                T acc = T(0);
#pragma unroll
                for (int t = 0; t < FT; ++t) {
                    const int sy = ym[-t];
                    const T q = src[at0(sy) * Wp + x];
                    acc = fma_t(a.dec.lo[tz + t], live(sy, q), acc);
                }
                tmp[yo * Wp + x] = acc;
            }
            __syncthreads();
            for (WalkN<NT> wk(tid, w); wk.r < h; wk.next(w)) {    // along W (low and high)
                const int yo = wk.r, xo = wk.c;
                const int* xm = xmap + 2 * xo + (FT - 1);
                const T* row = tmp + yo * Wp;
                T lo = T(0), hi = T(0);
#pragma unroll
                for (int t = 0; t < FT; ++t) {
                    const int sx = xm[-t];
                    const T q = live(sx, row[at0(sx)]);
                    lo = fma_t(a.dec.lo[tz + t], q, lo);
                    hi = fma_t(a.dec.hi[tz + t], q, hi);
                }
                dll[yo * w + xo] = lo;
                if (want_v) dcv[yo * w + xo] = (TV)hi;
            }
            __syncthreads();
            SONAR_BANDS_STAMP(j);
        }
        // ---------------------------------------------------------------- top: B_J = (l - a_h^J) LL_J
        {
            T* const top = lds + a.off_ll[J];
            const T gJ = a.yl - a.ah[J];
            for (int it = tid; it < a.H[J] * a.W[J]; it += NT) top[it] *= gJ;
            __syncthreads();
            SONAR_BANDS_STAMP(8);
        }
        // ---------------------------------------------------------------- way up: B_{j-1} = (a_d^j - a_h^{j-1}) LL_{j-1} + S^W_lo[(a_h - a_d) lowW + S^H_lo B_j] + S^W_hi[(a_v - a_d) S^H_lo cV_j]
        for (int j = J; j >= 2; --j) {
            const int h = a.H[j], w = a.W[j], Ho = a.H[j - 1], Wo = a.W[j - 1];
            const T* const B = lds + a.off_ll[j];
            const TV* const V = reinterpret_cast<const TV*>(lds + a.off_v[j]);
            T* const dst = lds + a.off_ll[j - 1];
            const int* const xmap = maps + a.map_w[j];
            const T c_low = a.ah[j] - a.ad[j], c_v = a.av[j] - a.ad[j], c_x = a.ad[j] - a.ah[j - 1];
            const bool want_v = a.av[j] != a.ad[j];
            const int wp = (Wo + 1) >> 1;
            T* const tA = tmp;
            T* const tB = tmp + a.rows_up[j] * w;
            for (int ya = 0; ya < Ho; ya += a.rows_up[j]) {
                const int tz = tap_zero<T, FT>();
                const int th = min(a.rows_up[j], Ho - ya);
                for (WalkN<NT> wk(tid, w); 2 * wk.r < th; wk.next(w)) {   // along H: rows (2m, 2m + 1) of column xo, plus lowW of those rows
                    const int mp = wk.r, xo = wk.c, m = (ya >> 1) + mp;
                    T e, o, e2 = T(0), o2 = T(0);
                    synth_low_pair<T, FT>(m, h, a.mode_inv, a.rec.lo + tz, [&](int i) { return B[i * w + xo]; }, e, o);
                    if (want_v) synth_low_pair<T, FT>(m, h, a.mode_inv, a.rec.lo + tz, [&](int i) { return (T)V[i * w + xo]; }, e2, o2);
                    const int* xm = xmap + 2 * xo + (FT - 1);
                    const T* r0 = dst + (ya + 2 * mp) * Wo;
                    const bool two = 2 * mp + 1 < th;
                    const T* r1 = two ? r0 + Wo : r0;
                    T l0 = T(0), l1 = T(0);
                    if (c_low != T(0)) {
#pragma unroll
                        for (int t = 0; t < FT; ++t) {
                            const int sx = xm[-t];
                            const int at = at0(sx);
                            l0 = fma_t(a.dec.lo[tz + t], live(sx, r0[at]), l0);
                            l1 = fma_t(a.dec.lo[tz + t], live(sx, r1[at]), l1);
                        }
                    }
                    tA[(2 * mp) * w + xo] = fma_t(c_low, l0, e);
                    tB[(2 * mp) * w + xo] = c_v * e2;
                    if (two) {
                        tA[(2 * mp + 1) * w + xo] = fma_t(c_low, l1, o);
                        tB[(2 * mp + 1) * w + xo] = c_v * o2;
                    }
                }
                __syncthreads();
                for (WalkN<NT> wk(tid, wp); wk.r < th; wk.next(wp)) {    // along W: both channels, accumulate into the plane below
                    const int yl = wk.r, m = wk.c;
                    const T* ra = tA + yl * w;
                    const T* rb = tB + yl * w;
                    T e, o;
                    if (want_v) SynthPair<T, FT>::run(m, w, a.mode_inv, TapView<T>{a.rec.lo + tz, a.rec.hi + tz}, [&](int i) { return ra[i]; }, [&](int i) { return rb[i]; }, e, o);
                    else synth_low_pair<T, FT>(m, w, a.mode_inv, a.rec.lo + tz, [&](int i) { return ra[i]; }, e, o);
                    T* d = dst + (ya + yl) * Wo + 2 * m;
                    d[0] = fma_t(c_x, d[0], e);
                    if (2 * m + 1 < Wo) d[1] = fma_t(c_x, d[1], o);
                }
                __syncthreads();
            }
            SONAR_BANDS_STAMP(8 + (J - j + 1));
        }
        // ---------------------------------------------------------------- level 1 up + the elementwise tail, straight to global
        {
            const T* const B = lds + a.off_ll[1];
            const TV* const V = reinterpret_cast<const TV*>(lds + a.off_v[1]);
            const TIO* px = a.subtract_from_x ? xin + p * (int64_t)H * W : nullptr;
            TIO* po = out + p * (int64_t)H * W;
            const int* const xmap = maps + a.map_w[1];
            const T c_low = a.ah[1] - a.ad[1], c_v = a.av[1] - a.ad[1], a_d = a.ad[1];
            const bool want_v = a.av[1] != a.ad[1];
            const int wp = (W + 1) >> 1;
            T* const tA = tmp;
            T* const tB = tmp + a.rows_out * w1;
            // The tile's rows of (a, b) are read once, used for lowW, for a_d v and for ku b.  Round 5: a thread's first kStageAhead values of
            // the NEXT tile are requested while this one is synthesised -- the stage sat behind its own loads at the top of every tile
            // (`scratch/bands_trace.py`: this stage is 46 % of a plane's time in the single-launch kernel, 42 % in the deeper levels' call).
            // The request is unconditional (the last tile asks for itself again, from L2): under a condition the loaded registers would
            // merge with their old values in a copy that waits for the loads on the spot (see spectral_filter128_kernel).
            constexpr int KP = (!HASB && sizeof(T) == 4) ? SONAR_BANDS_STAGE_AHEAD_ONE : SONAR_BANDS_STAGE_AHEAD;
            [[maybe_unused]] St pre[KP > 0 ? KP : 1];
            // (Measured and dropped, twice: the x pairs the tail subtracts from, requested a tile ahead -- selected by item number inside the
            // tail's loop they keep 70 more registers alive and a CU holds one workgroup instead of two: 198 -> 310 us -- or a phase ahead
            // with the first two items peeled off the loop: 30-40 more registers, 171 -> 240 us.)
            auto request = [&](int y0n) {
                const int lim = min(a.rows_out, H - y0n) * W;
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    const int it = tid + k * NT, at = y0n * W + (it < lim ? it : 0);
                    if constexpr (HASB) pre[k] = St{pa[at], pb ? pb[at] : TIO(0)};
                    else pre[k] = St{pa[at]};
                }
            };
            if constexpr (KP > 0) request(0);
            SONAR_BANDS_LAP_BEGIN();
            for (int y0 = 0; y0 < H; y0 += a.rows_out) {
                const int tz = tap_zero<T, FT>();
                const int th = min(a.rows_out, H - y0);
                if constexpr (KP > 0) {
#pragma unroll
                    for (int k = 0; k < KP; ++k) {
                        const int it = tid + k * NT;
                        if (it < th * W) cu[it] = pre[k];
                    }
                }
                for (int it = tid + KP * NT; it < th * W; it += NT) {
                    const int at = y0 * W + it;
                    if constexpr (HASB) cu[it] = St{pa[at], pb ? pb[at] : TIO(0)};
                    else cu[it] = St{pa[at]};
                }
                __syncthreads();
                SONAR_BANDS_LAP(20);
                if constexpr (KP > 0) request(y0 + a.rows_out < H ? y0 + a.rows_out : y0);
                for (WalkN<NT> wk(tid, w1); 2 * wk.r < th; wk.next(w1)) {
                    const int mp = wk.r, xo = wk.c, m = (y0 >> 1) + mp;
                    T e, o, e2 = T(0), o2 = T(0);
                    synth_low_pair<T, FT>(m, h1, a.mode_inv, a.rec.lo + tz, [&](int i) { return B[i * w1 + xo]; }, e, o);
                    if (want_v) synth_low_pair<T, FT>(m, h1, a.mode_inv, a.rec.lo + tz, [&](int i) { return (T)V[i * w1 + xo]; }, e2, o2);
                    const int* xm = xmap + 2 * xo + (FT - 1);
                    const bool two = 2 * mp + 1 < th;
                    const St* r0 = cu + (2 * mp) * W;
                    const St* r1 = two ? r0 + W : r0;
                    T l0 = T(0), l1 = T(0);
                    if (c_low != T(0)) {
#pragma unroll
                        for (int t = 0; t < FT; ++t) {
                            const int sx = xm[-t];
                            const St q0 = r0[at0(sx)], q1 = r1[at0(sx)];
                            l0 = fma_t(a.dec.lo[tz + t], live(sx, (T)q0.x - (T)q0.y), l0);
                            l1 = fma_t(a.dec.lo[tz + t], live(sx, (T)q1.x - (T)q1.y), l1);
                        }
                    }
                    tA[(2 * mp) * w1 + xo] = fma_t(c_low, l0, e);
                    tB[(2 * mp) * w1 + xo] = c_v * e2;
                    if (two) {
                        tA[(2 * mp + 1) * w1 + xo] = fma_t(c_low, l1, o);
                        tB[(2 * mp + 1) * w1 + xo] = c_v * o2;
                    }
                }
                __syncthreads();
                SONAR_BANDS_LAP(21);
                for (WalkN<NT> wk(tid, wp); wk.r < th; wk.next(wp)) {
                    const int yl = wk.r, m = wk.c;
                    const T* ra = tA + yl * w1;
                    const T* rb = tB + yl * w1;
                    T e, o;
                    if (want_v) SynthPair<T, FT>::run(m, w1, a.mode_inv, TapView<T>{a.rec.lo + tz, a.rec.hi + tz}, [&](int i) { return ra[i]; }, [&](int i) { return rb[i]; }, e, o);
                    else synth_low_pair<T, FT>(m, w1, a.mode_inv, a.rec.lo + tz, [&](int i) { return ra[i]; }, e, o);
                    const int at = (y0 + yl) * W + 2 * m;
                    const bool pair = 2 * m + 1 < W;
                    const St q0 = cu[yl * W + 2 * m];
                    const St q1 = pair ? cu[yl * W + 2 * m + 1] : q0;
                    const T r0 = fma_t(a.ku, (T)q0.y, a.kt * fma_t(a_d, (T)q0.x - (T)q0.y, e));
                    const T r1 = fma_t(a.ku, (T)q1.y, a.kt * fma_t(a_d, (T)q1.x - (T)q1.y, o));
                    if (pair && (W & 1) == 0 && a.vec2) {
                        In2 res{(TIO)r0, (TIO)r1};
                        if (px) {
                            const In2 x2 = *reinterpret_cast<const In2*>(px + at);
                            res = In2{x2.x - res.x, x2.y - res.y};
                        }
                        *reinterpret_cast<In2*>(po + at) = res;
                    } else {
                        po[at] = px ? px[at] - (TIO)r0 : (TIO)r0;
                        if (pair) po[at + 1] = px ? px[at + 1] - (TIO)r1 : (TIO)r1;
                    }
                }
                __syncthreads();
                SONAR_BANDS_LAP(22);
            }
        }
        SONAR_BANDS_STAMP(16);
    }
}

// LDS plan; false when the plane's coefficients do not fit one workgroup (the caller takes the band-by-band kernels)
template <typename T>
static bool bands_plan(BandsArgs<T>& a, size_t& lds_bytes, int64_t H, int64_t W, int levels, int flen, int mode_fwd, int mode_inv, bool any_v,
                       size_t io_size = sizeof(float), size_t v_size = sizeof(T) /* storage of a cV value */, bool hasb = true /* (a, b) pairs staged */) {
    if (levels < 1 || levels > kLowMaxLevels || !tile_taps_ok(flen) || flen > kDeepTaps || !dims_ok(H, W) || H > 4096 || W > 4096) return false;
    if (flen > 2 && (mode_fwd == kPeriodization) != (mode_inv == kPeriodization)) return false;  // a shifted reconstruction: not the identity used here
    a.levels = levels;
    a.H[0] = (int)H;
    a.W[0] = (int)W;
    int at = 0, ints = 0;
    for (int j = 1; j <= levels; ++j) {
        a.H[j] = (int)dwt_len(a.H[j - 1], flen, mode_fwd);
        a.W[j] = (int)dwt_len(a.W[j - 1], flen, mode_fwd);
        const int Hr = mode_inv == kPeriodization ? 2 * a.H[j] : 2 * a.H[j] - flen + 2;
        const int Wr = mode_inv == kPeriodization ? 2 * a.W[j] : 2 * a.W[j] - flen + 2;
        if (Hr < a.H[j - 1] || Wr < a.W[j - 1]) return false;  // the inverse cannot cover the level below
        a.off_ll[j] = at;
        at += a.H[j] * a.W[j];
        a.off_v[j] = at;
        if (any_v) at += (int)(((size_t)a.H[j] * a.W[j] * v_size + sizeof(T) - 1) / sizeof(T));
        a.map_h[j] = ints;
        ints += 2 * a.H[j] + flen;
        a.map_w[j] = ints;
        ints += 2 * a.W[j] + flen;
    }
    // scratch: the deeper levels' H pass (h_j x W_{j-1}), the way up's two channel planes per row tile, the last stage's two channel planes
    const size_t budget = 158 * 1024;
    const int resident = at;
    // lean: without the room for a level-1 analysis tile in the scratch proper -- that tile lies over the deeper levels' planes (off_tmp1) and
    // `end` below grows when they do not hold it; the shorter scratch means shorter tiles on the way up, so only when it buys a workgroup
    auto layout = [&](int rows_out, bool lean = false) {
        int tmp = lean ? rows_out * 2 * a.W[1] : std::max(rows_out * 2 * a.W[1], kLowRows * 2 * (((int)W + 1) / 2));
        for (int j = 2; j <= levels; ++j) tmp = std::max(tmp, a.H[j] * a.W[j - 1]);
        for (int j = 2; j <= levels; ++j) {  // as many rows of the plane below per tile as the scratch the other phases need anyway holds
            const int full = (a.H[j - 1] + 1) / 2 * 2;
            a.rows_up[j] = std::max(2, std::min(full, tmp / (2 * a.W[j]) / 2 * 2));
            tmp = std::max(tmp, a.rows_up[j] * 2 * a.W[j]);
        }
        a.rows_out = rows_out;
        a.off_tmp = resident;
        int end = resident + tmp;
        a.off_tmp1 = levels >= 2 ? a.off_ll[2] : a.off_tmp;
        const int ws1 = 2 * (((int)W + 1) / 2);
        a.rows1 = std::max(kLowRows, std::min((a.H[1] + 3) / 4 * 4, (end - a.off_tmp1) / ws1 / 4 * 4));
        end = std::max(end, a.off_tmp1 + a.rows1 * ws1);
        a.off_cu = (int)(((size_t)end * sizeof(T) + 15) / 16 * 16);
        a.off_maps = (int)((a.off_cu + (size_t)rows_out * (size_t)W * (hasb ? 2 : 1) * io_size + 15) / 16 * 16);
        return (size_t)a.off_maps + (size_t)ints * sizeof(int);
    };
    auto per_cu = [](size_t bytes) { return (160 * 1024) / (bytes + 512); };
    static const int forced_rows = [] { const char* e = getenv("SONAR_BANDS_ROWS_OUT"); return e ? atoi(e) : 0; }();  // (experiments)
    if (forced_rows > 0) {
        lds_bytes = layout(forced_rows);
        return lds_bytes <= budget;
    }
    // Output rows per tile of the last stage.  First what LDS allows: shorter tiles when they buy another resident workgroup.  Then, among
    // the heights that keep that many workgroups resident, the one whose item counts waste the fewest rounds of the workgroup's threads
    // (round 5): the stage's second phase has (rows / 2) x W1 items, its third rows x ceil(W / 2), each a dependent chain of LDS reads
    // -- 16 rows of a 128 x 128 plane (db4: W1 = 67) are 536 items for 512 threads, TWO rounds for 24 items' sake, where 14 rows are one:
    // single-launch rules 180 -> 169 us, 209 -> 199 us; the deeper levels of the tile route (67-row planes, W1 = 37) take 24 rows.
    size_t want_cu = 0;
    for (int lean = 0; lean < 2; ++lean)
        for (int r : {kLowRows / 2, kLowRows}) want_cu = std::max(want_cu, (size_t)per_cu(layout(r, lean != 0)));
    // ... and what the registers allow: the 512-thread instantiations take 67-77 of them (six or seven waves per SIMD: three workgroups);
    // a plan for four pays for them with short tiles and gets three (fp64 deeper levels: 231 against 217 us on the difference rule)
    static const int cap_cu = [] { const char* e = getenv("SONAR_BANDS_WANT_CU"); return e ? atoi(e) : 3; }();  // (the variable: experiments)
    if (cap_cu > 0) want_cu = std::min(want_cu, (size_t)cap_cu);
    const int nt = sizeof(T) == 8 && io_size == 4 && want_cu <= 1 ? 1024 : 512;  // (wcfg_bands: whole latent planes in fp64 take 1024 threads)
    int best_rows = kLowRows / 2;
    bool best_lean = true;
    double best_cost = 1e30;
    for (int lean = 0; lean < 2; ++lean) {
        for (int r = 8; r <= 40; r += 2) {
            const size_t bytes = layout(r, lean != 0);
            if (bytes > budget || (size_t)per_cu(bytes) < want_cu) continue;
            const int tiles = ((int)H + r - 1) / r, items2 = (r / 2) * a.W[1], items3 = r * (((int)W + 1) / 2);
            double cost = tiles * (2.0 * ((items2 + nt - 1) / nt) + 1.0 * ((items3 + nt - 1) / nt) + 1.5);
            for (int j = 2; j <= levels; ++j) cost += 2.0 * ((a.H[j - 1] + a.rows_up[j] - 1) / a.rows_up[j]);  // the way up's tiles (two phases each)
            if (cost < best_cost - 1e-9) {
                best_cost = cost;
                best_rows = r;
                best_lean = lean != 0;
            }
        }
    }
    if (best_cost > 1e29) return false;
    lds_bytes = layout(best_rows, best_lean);
    static const bool plan_debug = getenv("SONAR_BANDS_PLAN_DEBUG") != nullptr;  // (experiments)
    if (plan_debug)
        fprintf(stderr, "bands_plan %dx%d T%zu io%zu v%zu hasb%d: rows_out %d lean %d cost %.1f lds %zu per_cu %zu want %zu rows1 %d rows_up %d %d %d\n", (int)H, (int)W,
                sizeof(T), io_size, v_size, (int)hasb, best_rows, (int)best_lean, best_cost, lds_bytes, (size_t)per_cu(lds_bytes), want_cu, a.rows1, a.rows_up[2],
                a.rows_up[3], a.rows_up[4]);
    return lds_bytes <= budget;
}

constexpr bool bands_rows_ahead_ok(size_t io_size, int ft) { return SONAR_BANDS_ROWS_AHEAD && io_size == 4 && ft <= 10; }  // (12 taps spill at the 1024-thread instantiation's 128 registers)
// what the plan assumed about the kernel's two LDS-halving parameters: the cV planes in fp32 (fp64 arithmetic, coefficient-plane I/O: the
// tile route's deeper levels, when the process keeps detail bands in fp32), single staged values (no second tensor; coefficient-plane or
// fp32 I/O -- the fp64 latent kernel holds one workgroup per CU either way and keeps the pair form)
template <typename T, typename TIO>
struct BandsForm {
    bool v_float, hasb;
    static BandsForm of(const TIO* tb) {
        constexpr bool same = std::is_same<T, TIO>::value, dbl = std::is_same<T, double>::value;
        return BandsForm{same && dbl && wcfg_hi_fp32_switch() != 0, !(same && tb == nullptr)};
    }
    size_t v_size() const { return v_float ? sizeof(float) : sizeof(T); }
};
template <typename T, typename TIO, int FT, int NT, bool ZERO, bool AHEAD, typename TV, bool HASB>
static void launch_bands_zav(int grid, size_t lds, hipStream_t st, const TIO* ta, const TIO* tb, const TIO* x, TIO* out, const BandsArgs<T>& a) {
    auto kern = wcfg_bands_kernel<T, TIO, FT, NT, ZERO, AHEAD, TV, HASB>;
    if (lds > 64 * 1024) lds_attr(reinterpret_cast<const void*>(kern), 160 * 1024);  // dynamic LDS above the 64 KB default: once per kernel and device
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st, ta, tb, x, out, a);
}
template <typename T, typename TIO, int FT, int NT, bool ZERO, bool AHEAD = false>
static void launch_bands_za(int grid, size_t lds, hipStream_t st, const TIO* ta, const TIO* tb, const TIO* x, TIO* out, const BandsArgs<T>& a) {
    const BandsForm<T, TIO> form = BandsForm<T, TIO>::of(tb);
    if constexpr (std::is_same<T, TIO>::value && std::is_same<T, double>::value) {  // coefficient planes in fp64: never a second tensor
        if (form.hasb) return;  // (wcfg_bands refuses it)
        if (form.v_float) launch_bands_zav<T, TIO, FT, NT, ZERO, AHEAD, float, false>(grid, lds, st, ta, tb, x, out, a);
        else launch_bands_zav<T, TIO, FT, NT, ZERO, AHEAD, T, false>(grid, lds, st, ta, tb, x, out, a);
        return;
    } else if constexpr (std::is_same<T, TIO>::value) {
        if (form.hasb) launch_bands_zav<T, TIO, FT, NT, ZERO, AHEAD, T, true>(grid, lds, st, ta, tb, x, out, a);
        else launch_bands_zav<T, TIO, FT, NT, ZERO, AHEAD, T, false>(grid, lds, st, ta, tb, x, out, a);
        return;
    }
    auto kern = wcfg_bands_kernel<T, TIO, FT, NT, ZERO, AHEAD>;
    if (lds > 64 * 1024) lds_attr(reinterpret_cast<const void*>(kern), 160 * 1024);  // dynamic LDS above the 64 KB default: once per kernel and device
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st, ta, tb, x, out, a);
}
template <typename T, typename TIO, int FT, int NT, bool ZERO>
static void launch_bands_z(int grid, size_t lds, hipStream_t st, const TIO* ta, const TIO* tb, const TIO* x, TIO* out, const BandsArgs<T>& a) {
    if constexpr (bands_rows_ahead_ok(sizeof(TIO), FT)) {
        if (3 * (lds + 512) > 160 * 1024) {  // at most two workgroups fit a CU's LDS: eight waves per SIMD at most, 128 registers each
            launch_bands_za<T, TIO, FT, NT, ZERO, true>(grid, lds, st, ta, tb, x, out, a);
            return;
        }
    }
    launch_bands_za<T, TIO, FT, NT, ZERO, false>(grid, lds, st, ta, tb, x, out, a);
}
template <typename T, typename TIO, int FT, int NT>
static void launch_bands(int grid, size_t lds, hipStream_t st, const TIO* ta, const TIO* tb, const TIO* x, TIO* out, const BandsArgs<T>& a) {
    if (a.mode_fwd == kZero) launch_bands_z<T, TIO, FT, NT, true>(grid, lds, st, ta, tb, x, out, a);
    else launch_bands_z<T, TIO, FT, NT, false>(grid, lds, st, ta, tb, x, out, a);
}

template <typename T, typename TIO>
static int wcfg_bands(const TIO* ta, const TIO* tb, const TIO* x, TIO* out, int64_t planes, int64_t H, int64_t W, int levels,
                      const double* dec_lo, const double* dec_hi, const double* rec_lo, const double* rec_hi, int flen, int mode_fwd, int mode_inv,
                      const double* yh_scales, double yl_scale, double ku, double kt, int subtract_from_x, hipStream_t st, const char* what) {
    SONAR_REQUIRE(ta && out && (x || !subtract_from_x) && dec_lo && dec_hi && rec_lo && rec_hi && yh_scales && planes >= 0 && mode_fwd >= 0 &&
                      mode_fwd <= 5 && mode_inv >= 0 && mode_inv <= 5,
                  SONAR_ERR_ARG, "%s: bad argument", what);
    SONAR_REQUIRE(levels >= 1 && levels <= kLowMaxLevels, SONAR_ERR_UNSUPPORTED, "%s: 1 .. %d levels", what, kLowMaxLevels);
    BandsArgs<T> a{};
    bool any_v = false;
    for (int j = 1; j <= levels; ++j) {
        a.ah[j] = (T)yh_scales[3 * (j - 1) + 0];
        a.av[j] = (T)yh_scales[3 * (j - 1) + 1];
        a.ad[j] = (T)yh_scales[3 * (j - 1) + 2];
        any_v = any_v || a.av[j] != a.ad[j];
    }
    size_t lds = 0;
    const BandsForm<T, TIO> form = BandsForm<T, TIO>::of(tb);
    SONAR_REQUIRE(!(std::is_same<T, TIO>::value && std::is_same<T, double>::value && tb != nullptr), SONAR_ERR_UNSUPPORTED,
                  "%s: no second tensor with fp64 coefficient planes", what);
    SONAR_REQUIRE(bands_plan(a, lds, H, W, levels, flen, mode_fwd, mode_inv, any_v, sizeof(TIO), form.v_size(), form.hasb), SONAR_ERR_UNSUPPORTED,
                  "%s: the plane's coefficients do not fit in LDS (or unsupported filter length / extension pair)", what);
    if (planes == 0) return SONAR_OK;
    a.planes = planes;
    a.yl = (T)yl_scale;
    a.ku = (T)ku;
    a.kt = (T)kt;
    a.subtract_from_x = subtract_from_x;
    a.mode_fwd = mode_fwd;
    a.mode_inv = mode_inv;
    a.vec2 = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & (2 * sizeof(TIO) - 1)) == 0;
    for (int i = 0; i < kDeepTaps; ++i) {
        a.dec.lo[i] = i < flen ? (T)dec_lo[i] : T(0);
        a.dec.hi[i] = i < flen ? (T)dec_hi[i] : T(0);
        a.rec.lo[i] = i < flen ? (T)rec_lo[i] : T(0);
        a.rec.hi[i] = i < flen ? (T)rec_hi[i] : T(0);
    }
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / (lds + 512)));
    // one workgroup per CU leaves half of the SIMDs' wave slots empty at 512 threads: such planes (fp64 at SDXL size) take 1024
    const bool wide = per_cu == 1;
    (void)wide;
    const int grid = (int)std::min<int64_t>(planes, (int64_t)256 * per_cu);
    with_taps(flen, [&](auto ft) {
        constexpr int FT = decltype(ft)::value;
        if constexpr (std::is_same<T, double>::value && std::is_same<TIO, float>::value) {
            if (wide) {  // whole latent planes in fp64: the only shape that leaves a CU with one workgroup
                launch_bands<T, TIO, FT, 1024>(grid, lds, st, ta, tb, x, out, a);
                return;
            }
        }
        launch_bands<T, TIO, FT, 512>(grid, lds, st, ta, tb, x, out, a);
    });
    return check_launch(what);
}

}  // namespace sonar
