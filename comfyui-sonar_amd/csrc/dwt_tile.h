// LDS-staged 2-D DWT / IDWT: one launch per level, for even filter lengths 2..20 (longer / odd filters use the per-pass
// kernels in dwt.hip).  A workgroup owns a tile of output rows of one plane:
//   analysis   pass 1 filters along H straight from global memory: one thread per (tensor, column) keeps the 2 TH + F - 2
//              input rows of its column in registers (every row is loaded once, coalesced across the wave) and slides the
//              filter down them; the (low, high)-along-H rows go to LDS.  Pass 2 filters along W out of LDS, applies the
//              WaveletCFG band arithmetic (py/wavelet_cfg.py:765-787) and stores: the intermediate never touches HBM, and
//              cond / uncond are transformed together so only ONE detail tensor per level is written.
//   synthesis  pass 1 along H from global, pass 2 along W out of LDS; a thread produces the (even, odd) output pair that
//              shares one set of coefficients, so there is no per-term parity select or bounds test.
// Multiply-adds are fused (fma): the reference's transform is a library convolution with no fixed summation order, so
// the parity bar is the tolerance stated in tests/test_gpu_wavelet.py, not bit equality with the per-pass kernels.
#pragma once
#include <cstdlib>
#include <type_traits>

#include "dwt_common.h"

namespace sonar {
// fp64 taps are two scalar registers each; a kernel that loads its filters once, in front of its job loop, gets them back from
// vector-register lanes (a v_readlane and its hazard wait in front of every use: tools/asm_loop_loads.py --lanes).  A view of the
// argument's tap arrays behind an opaque zero, made per stage, keeps the loads inside it (they hit the scalar cache; used by the band kernel,
// dwt_bands.h -- in the tile kernels below the lane moves are not the taps: no change there).  Up to 14
// taps: beyond, the compiler answers the run-time index with a private copy of the argument block (scratch: never).
#ifndef SONAR_TAPS_LOCAL
#define SONAR_TAPS_LOCAL 1
#endif
template <typename T, int FT>
__device__ __forceinline__ int tap_zero() {
    int z = 0;
    if constexpr (sizeof(T) == 8 && FT <= 14 && SONAR_TAPS_LOCAL) asm volatile("" : "+s"(z));
    return z;
}
template <typename T>
struct TapView {
    const T* lo;
    const T* hi;
};

constexpr int kTileThreads = 256;
constexpr int kFwdRows = 8;    // output rows per analysis tile (register window = 2 * kFwdRows + F - 2 input rows)
constexpr int kInvRows = 16;   // output rows per synthesis tile (even)
constexpr size_t kTileLdsLimit = 64 * 1024;

// source index of extended-signal position `pos` (periodization pads odd lengths with a repeat of the last sample)
__device__ __forceinline__ int src_index(int pos, int n, int ne, int mode) {
    if (mode == kPeriodization) {
        int p = pos % ne;
        if (p < 0) p += ne;
        return p < n ? p : n - 1;
    }
    return ext_index(pos, n, mode);
}

static inline bool tile_taps_ok(int F) { return F >= 2 && F <= 20 && (F & 1) == 0; }

// analysis LDS: [kFwdRows][2 ceil(W/2) slots][2 tensors' (lowH, highH)] of T, then xmap[2w + F], ymap[2h + F]
static inline size_t fwd_lds_bytes(int W, int h, int w, int F, size_t elem, int tensors, int rows = kFwdRows) {
    return (size_t)tensors * rows * 2 * (2 * ((W + 1) / 2)) * elem + (size_t)(2 * w + 2 * h + 2 * F) * sizeof(int);
}
// synthesis LDS: [kInvRows][lo_w | hi_w][w] of T
static inline size_t inv_lds_bytes(int w, size_t elem, int rows = kInvRows) { return (size_t)rows * 2 * w * elem; }

// compact on purpose: these live in scalar registers next to the filter taps (fp64: 2 SGPRs per value)
template <typename T>
struct BandArgs {
    T hc[3], hu[3], hd[3], hf[3];  // cond / uncond / diff / final scale per orientation (cH, cV, cD)
    T lc, lu, ld, lf;              // the same for the approximation band
    int combine_ll;                // last level: the approximation band is combined too
    int blend_mode;
    T strength;
};

// How a forward tile gets its input and what it does before storing:
//   kFwdPlain  one tensor, coefficients stored as they are                      (Wavelet.forward)
//   kFwdPair   cond and uncond side by side, band arithmetic before the store   (any WaveletCFG rule)
//   kFwdDiff   ONE tensor v = cond - uncond formed on load, bands scaled by the difference scales on store -- difference-only rules:
//              IDWT(blend(U, D (C - U), t)) = ku u + kt IDWT(D DWT(c - u)), so uncond's transform is never computed
//   kFwdScale  one tensor (a deeper level of kFwdDiff), bands scaled on store
enum FwdMode { kFwdPlain = 0, kFwdPair = 1, kFwdDiff = 2, kFwdScale = 3 };

// blend(u * s_u, (c * s_c - u * s_u) * s_d, strength) * s_f  (py/wavelet_cfg.py:765-787); scales of 1 are skipped
template <typename T>
__device__ __forceinline__ T band_combine4(T c, T u, T sc, T su, T sd, T sf, int blend_mode, T strength) {
    if (sc != T(1)) c = c * sc;
    if (su != T(1)) u = u * su;
    T d = c - u;
    if (sd != T(1)) d = d * sd;
    T r = blend<T>(blend_mode, u, d, strength);
    if (sf != T(1)) r = r * sf;
    return r;
}

// LDS tile layout: tmp[row][slot(x)][NV] with NV = 2 (lowH, highH) or 4 (cond lowH, cond highH, uncond lowH, uncond highH)
// and slot(x) = (x & 1) * ceil(W / 2) + x / 2: a tap reads one parity class, so consecutive lanes hit consecutive slots
// (conflict-free vector reads) and one read fetches every value a tap needs.
template <typename T, int NV>
struct alignas(sizeof(T) * NV > 16 ? 16 : sizeof(T) * NV) TileVec {
    T v[NV];
};

// LDS geometry of one analysis tile: tmp[kFwdRows][2 ceil(W/2)] vectors, then xmap[2w + FT], ymap[2h + FT]
template <typename T, int NV, int FT>
struct FwdLds {
    using Vec = TileVec<T, NV>;
    Vec* tmp;
    int* xmap;
    int* ymap;
    int Wh, Ws;
    __device__ __forceinline__ FwdLds(unsigned char* smem, int W, int w, int rows = kFwdRows) {
        Wh = (W + 1) >> 1;
        Ws = 2 * Wh;
        tmp = reinterpret_cast<Vec*>(smem);
        xmap = reinterpret_cast<int*>(tmp + rows * Ws);
        ymap = xmap + (2 * w + FT);
    }
};

// extension tables for extended position i + off - (F - 1): xmap = LDS slot (or -1), ymap = source row (or -1)
template <int FT>
__device__ __forceinline__ void fwd_build_maps(int* xmap, int* ymap, int H, int W, int h, int w, int Wh, int mode) {
    const int off = mode == kPeriodization ? FT / 2 : 1;
    const int He = (mode == kPeriodization && (H & 1)) ? H + 1 : H;
    const int We = (mode == kPeriodization && (W & 1)) ? W + 1 : W;
    for (int i = threadIdx.x; i < 2 * w + FT - 2; i += kTileThreads) {
        const int sx = src_index(i + off - (FT - 1), W, We, mode);
        xmap[i] = sx < 0 ? -1 : (sx & 1) * Wh + (sx >> 1);
    }
    for (int i = threadIdx.x; i < 2 * h + FT - 2; i += kTileThreads) ymap[i] = src_index(i + off - (FT - 1), H, He, mode);
}

// one analysis tile: output rows [y0, y0 + th) of one plane (pc / pu, ollc / ollu / ohi are plane pointers)
// THi: storage type of the three detail bands (WaveletCFG's level 1 in fp64 mode keeps them in fp32: see wcfg_fused)
template <typename T, typename TIn, int MODE, int FT, bool ZERO, typename TP, bool SPLIT = false, typename THi = T>
__device__ __forceinline__ void fwd_tile_job(const TIn* __restrict__ pc, const TIn* __restrict__ pu, T* __restrict__ ollc,
                                             T* __restrict__ ollu, THi* __restrict__ ohi, int W, int h, int w, int y0, int th,
                                             const TP& tp, const BandArgs<T>& ba, const FwdLds<T, MODE == kFwdPair ? 4 : 2, FT>& lds) {
    constexpr bool PAIR = MODE == kFwdPair, DIFF = MODE == kFwdDiff, SCALE = MODE == kFwdDiff || MODE == kFwdScale;
    constexpr int TH = kFwdRows, NR = 2 * TH + FT - 2, NT = PAIR ? 2 : 1, NV = 2 * NT;
    using Vec = TileVec<T, NV>;
    using Half = TileVec<T, 2>;
    Vec* const tmp = lds.tmp;
    const int* const xmap = lds.xmap;
    const int* const ymap = lds.ymap;
    const int Wh = lds.Wh, Ws = lds.Ws;
    const int hw = h * w;
    const int dq = kTileThreads / w, dr = kTileThreads - dq * w;  // pass-2 item stride as (rows, columns)
    const int q0 = threadIdx.x / w, r0 = threadIdx.x - q0 * w;
    {
        __syncthreads();
        // ---- pass 1: analysis along H, one thread per (row group, tensor, column); window = extended input rows of the group.
        // SUB row groups per tile: a narrow level (the deep WaveletCFG levels: 2 x 67, 2 x 37 ... columns) would leave most of
        // the workgroup idle with one thread per column, so the tile's TH output rows are split over SUB threads per column.
        auto pass1 = [&](auto sub_c) {
            constexpr int SUB = decltype(sub_c)::value, THS = TH / SUB, NRS = 2 * THS + FT - 2;
            for (int it = threadIdx.x; it < SUB * NT * W; it += kTileThreads) {
                const int sub = it / (NT * W), rest = it - sub * (NT * W);
                const int ten = rest >= W ? 1 : 0;
                const int x = rest - ten * W;
                const TIn* col = (ten ? pu : pc) + x;
                T v[NRS];
#pragma unroll
                for (int r = 0; r < NRS; ++r) {
                    // rows past the tile's need (last tile) re-read a valid row; their outputs are never consumed
                    const int sy = ymap[min(2 * (y0 + sub * THS) + r, 2 * h + FT - 3)];
                    if constexpr (ZERO) {
                        T g = (T)col[(sy >= 0 ? sy : 0) * W];
                        if constexpr (DIFF) g -= (T)pu[(sy >= 0 ? sy : 0) * W + x];
                        v[r] = sy >= 0 ? g : T(0);
                    } else {
                        v[r] = (T)col[sy * W];
                        if constexpr (DIFF) v[r] -= (T)pu[sy * W + x];
                    }
                }
                Half* dst = reinterpret_cast<Half*>(tmp + ((x & 1) * Wh + (x >> 1))) + ten;
#pragma unroll
                for (int yl = 0; yl < THS; ++yl) {
                    T lo = T(0), hq = T(0);
                    if constexpr (std::is_same<T, float>::value) {
                        typedef float pk2 __attribute__((ext_vector_type(2)));
                        pk2 acc = {0.0f, 0.0f};
#pragma unroll
                        for (int j = 0; j < FT; ++j) acc = __builtin_elementwise_fma(pk2{tp.lo[j], tp.hi[j]}, pk2{v[2 * yl + FT - 1 - j], v[2 * yl + FT - 1 - j]}, acc);
                        lo = acc.x;
                        hq = acc.y;
                    } else {
#pragma unroll
                        for (int j = 0; j < FT; ++j) {  // tap j reads window row 2 yl + F - 1 - j
                            lo = fma_t(tp.lo[j], v[2 * yl + FT - 1 - j], lo);
                            hq = fma_t(tp.hi[j], v[2 * yl + FT - 1 - j], hq);
                        }
                    }
                    Half o2;
                    o2.v[0] = lo;
                    o2.v[1] = hq;
                    dst[(sub * THS + yl) * Ws * NT] = o2;
                }
            }
        };
        if constexpr (SPLIT) {
            // a job of `th` (any number of) rows: row groups of TH / 2, items (group, tensor, column) spread over the whole workgroup
            constexpr int THS = TH / 2, NRS = 2 * THS + FT - 2;
            const int groups = (th + THS - 1) / THS;
            for (int it = threadIdx.x; it < groups * NT * W; it += kTileThreads) {
                const int sub = it / (NT * W), rest = it - sub * (NT * W);
                const int ten = rest >= W ? 1 : 0;
                const int x = rest - ten * W;
                const TIn* col = (ten ? pu : pc) + x;
                T v[NRS];
#pragma unroll
                for (int r = 0; r < NRS; ++r) {
                    const int sy = ymap[min(2 * (y0 + sub * THS) + r, 2 * h + FT - 3)];
                    if constexpr (ZERO) {
                        T g = (T)col[(sy >= 0 ? sy : 0) * W];
                        if constexpr (DIFF) g -= (T)pu[(sy >= 0 ? sy : 0) * W + x];
                        v[r] = sy >= 0 ? g : T(0);
                    } else {
                        v[r] = (T)col[sy * W];
                        if constexpr (DIFF) v[r] -= (T)pu[sy * W + x];
                    }
                }
                Half* dst = reinterpret_cast<Half*>(tmp + ((x & 1) * Wh + (x >> 1))) + ten;
#pragma unroll
                for (int yl = 0; yl < THS; ++yl) {
                    T lo = T(0), hq = T(0);
                    if constexpr (std::is_same<T, float>::value) {
                        typedef float pk2 __attribute__((ext_vector_type(2)));
                        pk2 acc = {0.0f, 0.0f};
#pragma unroll
                        for (int j = 0; j < FT; ++j) acc = __builtin_elementwise_fma(pk2{tp.lo[j], tp.hi[j]}, pk2{v[2 * yl + FT - 1 - j], v[2 * yl + FT - 1 - j]}, acc);
                        lo = acc.x;
                        hq = acc.y;
                    } else {
#pragma unroll
                        for (int j = 0; j < FT; ++j) {
                            lo = fma_t(tp.lo[j], v[2 * yl + FT - 1 - j], lo);
                            hq = fma_t(tp.hi[j], v[2 * yl + FT - 1 - j], hq);
                        }
                    }
                    Half o2;
                    o2.v[0] = lo;
                    o2.v[1] = hq;
                    if (sub * THS + yl < th) dst[(sub * THS + yl) * Ws * NT] = o2;
                }
            }
        } else {
            pass1(std::integral_constant<int, 1>{});
        }
        __syncthreads();
        // ---- pass 2: analysis along W out of LDS, band arithmetic, store
        for (int yl = q0, xo = r0; yl < th;) {
            const Vec* row = tmp + yl * Ws;
            const int* xm = xmap + 2 * xo + (FT - 1);  // tap j reads slot xm[-j]
            T c_ll = T(0), c_h = T(0), c_v = T(0), c_d = T(0), u_ll = T(0), u_h = T(0), u_v = T(0), u_d = T(0);
            if constexpr (std::is_same<T, float>::value) {
                // fp32: (low W, high W) of one value share a packed FMA -- (lo[j], hi[j]) * q + (acc_lo, acc_hi)
                typedef float pk2 __attribute__((ext_vector_type(2)));
                pk2 a0 = {0.0f, 0.0f}, a1 = a0, a2 = a0, a3 = a0;
#pragma unroll
                for (int j = 0; j < FT; ++j) {
                    const int sx = xm[-j];
                    Vec q = row[ZERO ? max(sx, 0) : sx];
                    if constexpr (ZERO) {
                        if (sx < 0) {
#pragma unroll
                            for (int e = 0; e < NV; ++e) q.v[e] = T(0);
                        }
                    }
                    const pk2 t = {tp.lo[j], tp.hi[j]};
                    a0 = __builtin_elementwise_fma(t, pk2{q.v[0], q.v[0]}, a0);
                    a1 = __builtin_elementwise_fma(t, pk2{q.v[1], q.v[1]}, a1);
                    if constexpr (PAIR) {
                        a2 = __builtin_elementwise_fma(t, pk2{q.v[2], q.v[2]}, a2);
                        a3 = __builtin_elementwise_fma(t, pk2{q.v[3], q.v[3]}, a3);
                    }
                }
                c_ll = a0.x; c_v = a0.y; c_h = a1.x; c_d = a1.y;
                u_ll = a2.x; u_v = a2.y; u_h = a3.x; u_d = a3.y;
            } else {
#pragma unroll
            for (int j = 0; j < FT; ++j) {
                const int sx = xm[-j];
                Vec q = row[ZERO ? max(sx, 0) : sx];
                if constexpr (ZERO) {
                    if (sx < 0) {
#pragma unroll
                        for (int e = 0; e < NV; ++e) q.v[e] = T(0);
                    }
                }
                c_ll = fma_t(tp.lo[j], q.v[0], c_ll);  // low H, low W
                c_v = fma_t(tp.hi[j], q.v[0], c_v);    // low H, high W  (cV)
                c_h = fma_t(tp.lo[j], q.v[1], c_h);    // high H, low W  (cH)
                c_d = fma_t(tp.hi[j], q.v[1], c_d);
                if constexpr (PAIR) {
                    u_ll = fma_t(tp.lo[j], q.v[2], u_ll);
                    u_v = fma_t(tp.hi[j], q.v[2], u_v);
                    u_h = fma_t(tp.lo[j], q.v[3], u_h);
                    u_d = fma_t(tp.hi[j], q.v[3], u_d);
                }
            }
            }
            const int o = (y0 + yl) * w + xo;
            if constexpr (PAIR) {
                ohi[o] = (THi)band_combine4<T>(c_h, u_h, ba.hc[0], ba.hu[0], ba.hd[0], ba.hf[0], ba.blend_mode, ba.strength);
                ohi[hw + o] = (THi)band_combine4<T>(c_v, u_v, ba.hc[1], ba.hu[1], ba.hd[1], ba.hf[1], ba.blend_mode, ba.strength);
                ohi[2 * hw + o] = (THi)band_combine4<T>(c_d, u_d, ba.hc[2], ba.hu[2], ba.hd[2], ba.hf[2], ba.blend_mode, ba.strength);
                if (ba.combine_ll) {
                    ollc[o] = band_combine4<T>(c_ll, u_ll, ba.lc, ba.lu, ba.ld, ba.lf, ba.blend_mode, ba.strength);
                } else {
                    ollc[o] = c_ll;
                    ollu[o] = u_ll;
                }
            } else if constexpr (SCALE) {
                ollc[o] = ba.combine_ll ? c_ll * ba.ld : c_ll;
                ohi[o] = (THi)(c_h * ba.hd[0]);
                ohi[hw + o] = (THi)(c_v * ba.hd[1]);
                ohi[2 * hw + o] = (THi)(c_d * ba.hd[2]);
            } else {
                ollc[o] = c_ll;
                ohi[o] = (THi)c_h;
                ohi[hw + o] = (THi)c_v;
                ohi[2 * hw + o] = (THi)c_d;
            }
            yl += dq;
            xo += dr;
            if (xo >= w) {
                xo -= w;
                yl += 1;
            }
        }
    }
}

// PAIR = false: plain DWT of xc -> (llc, hi).  PAIR = true: DWT of xc and xu, hi = band(xc, xu) per orientation;
// llc / llu separate, or llc = band(ll_c, ll_u) when combine_ll.  ZERO: zero-extension mode (the only mode with "no
// source" positions; the others never need a select).
template <typename T, typename TIn, int MODE, int FT, bool ZERO, typename THi = T>
__global__ void __launch_bounds__(kTileThreads) dwt2_tile_kernel(const TIn* __restrict__ xc, const TIn* __restrict__ xu,
                                                                 T* __restrict__ llc, T* __restrict__ llu, THi* __restrict__ hi,
                                                                 int64_t planes, int H, int W, int h, int w, int tiles, Taps<T> tp,
                                                                 int mode, BandArgs<T> ba) {
    kernarg_touch_for(xc, xu, llc, llu, hi, planes, H, W, h, w, tiles, tp, mode, ba);
    extern __shared__ __align__(16) unsigned char tile_smem[];
    constexpr bool PAIR = MODE == kFwdPair, TWO = PAIR || MODE == kFwdDiff;
    const FwdLds<T, PAIR ? 4 : 2, FT> lds(tile_smem, W, w);
    fwd_build_maps<FT>(lds.xmap, lds.ymap, H, W, h, w, lds.Wh, mode);
    const int64_t hw = (int64_t)h * w;
    for (int64_t job = blockIdx.x; job < planes * tiles; job += gridDim.x) {
        const int64_t p = job / tiles;
        const int y0 = (int)(job - p * tiles) * kFwdRows;
        fwd_tile_job<T, TIn, MODE, FT, ZERO, Taps<T>, false, THi>(xc + p * (int64_t)H * W, TWO ? xu + p * (int64_t)H * W : nullptr, llc + p * hw,
                                             (PAIR && !ba.combine_ll) ? llu + p * hw : nullptr, hi + p * 3 * hw, W, h, w, y0,
                                             min(kFwdRows, h - y0), tp, ba, lds);
    }
}

// The outputs o = 2m and 2m + 1 of a synthesis share their coefficients (K = F / 2 terms each):
//   non-periodization    x[2m]   = sum_k a[i_k] lo[2k]   + d[i_k] hi[2k],     i_k = m + K - 1 - k   (0 <= i_k < n for valid o)
//                        x[2m+1] = sum_k a[i_k] lo[2k+1] + d[i_k] hi[2k+1]
//   periodization, K odd same taps, i_k = (m + (K-1)/2 - k) mod n
//   periodization, K even x[2m] pairs tap 2k+1 with i = (m + K/2 - 1 - k) mod n, x[2m+1] pairs tap 2k with i = (m + K/2 - k) mod n
// (from x[o] = sum over 2i + j == o + F - 2, resp. == o + F/2 - 1 mod 2n).  Terms are added with i ascending.
template <typename T, int FT>
struct SynthPair {
    static constexpr int K = FT / 2;
    template <typename TP, typename LoadA, typename LoadD>
    static __device__ __forceinline__ void run(int m, int n, int mode, const TP& tp, LoadA&& la, LoadD&& ld, T& even, T& odd) {
        even = T(0);
        odd = T(0);
        if (mode != kPeriodization || (K & 1) == 1) {
            T va[K], vd[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                int i;
                if (mode != kPeriodization) {
                    i = m + K - 1 - k;
                    i = i < n ? i : n - 1;  // only beyond the valid output length (the caller discards those outputs)
                } else {
                    i = (m + (K - 1) / 2 - k) % n;
                    if (i < 0) i += n;
                }
                va[k] = la(i);
                vd[k] = ld(i);
            }
#pragma unroll
            for (int k = K - 1; k >= 0; --k) {
                even = fma_t(va[k], tp.lo[2 * k], fma_t(vd[k], tp.hi[2 * k], even));
                odd = fma_t(va[k], tp.lo[2 * k + 1], fma_t(vd[k], tp.hi[2 * k + 1], odd));
            }
        } else {
            T va[K + 1], vd[K + 1];
#pragma unroll
            for (int q = 0; q <= K; ++q) {  // entry q holds i = m + K/2 - q
                int i = (m + K / 2 - q) % n;
                if (i < 0) i += n;
                va[q] = la(i);
                vd[q] = ld(i);
            }
#pragma unroll
            for (int k = K - 1; k >= 0; --k) {
                even = fma_t(va[k + 1], tp.lo[2 * k + 1], fma_t(vd[k + 1], tp.hi[2 * k + 1], even));
                odd = fma_t(va[k], tp.lo[2 * k], fma_t(vd[k], tp.hi[2 * k], odd));
            }
        }
    }
};

// one synthesis tile: output rows [y0, y0 + th) (y0 even) of one plane.  pll / phi: the plane's approximation (row stride
// ll_w) and detail bands; out / xsub / outf point at the plane's output.  FINAL = false: out (type T).  FINAL = true:
// outf = xsub - (float)rec (or (float)rec), the cast + crop + `x - result` of py/wavelet_cfg.py:729-748.  FINAL = 2 (difference-only
// rules, kFwdDiff): rec is the synthesis of the scaled bands of cond - uncond and the result is ku * uncond + kt * rec.
struct FinalMix {
    const float* usub = nullptr;
    double ku = 0.0, kt = 1.0;
};
template <typename T, int FINAL, int FT, typename TP, typename THi = T>
__device__ __forceinline__ void inv_tile_job(const T* __restrict__ pll, int ll_w, const THi* __restrict__ phi, T* __restrict__ out,
                                             const float* __restrict__ xsub, float* __restrict__ outf, int h, int w, int Wo, int y0,
                                             int th, const TP& tp, int mode, int subtract, T* tmp, const FinalMix& mix = FinalMix{}) {
    const int hw = h * w, w2 = 2 * w;
    const int wp = (Wo + 1) >> 1;                                     // output pairs per row
    const int dq = kTileThreads / w, dr = kTileThreads - dq * w;     // pass 1 items: (row pair, column)
    const int q0 = threadIdx.x / w, r0 = threadIdx.x - q0 * w;
    const int DQ = kTileThreads / wp, DR = kTileThreads - DQ * wp;   // pass 2 items: (row, column pair)
    const int Q0 = threadIdx.x / wp, R0 = threadIdx.x - Q0 * wp;
    __syncthreads();
    // ---- pass 1: synthesis along H (lanes along x: coalesced); rows (y0 + 2 mp, y0 + 2 mp + 1) -> LDS
    for (int mp = q0, xo = r0; 2 * mp < th;) {
        const int m = (y0 >> 1) + mp;
        T e0, o0, e1, o1;
        SynthPair<T, FT>::run(m, h, mode, tp, [&](int i) { return pll[i * ll_w + xo]; }, [&](int i) { return (T)phi[i * w + xo]; }, e0, o0);
        SynthPair<T, FT>::run(m, h, mode, tp, [&](int i) { return (T)phi[hw + i * w + xo]; }, [&](int i) { return (T)phi[2 * hw + i * w + xo]; },
                              e1, o1);
        tmp[(2 * mp) * w2 + xo] = e0;
        tmp[(2 * mp) * w2 + w + xo] = e1;
        if (2 * mp + 1 < th) {
            tmp[(2 * mp + 1) * w2 + xo] = o0;
            tmp[(2 * mp + 1) * w2 + w + xo] = o1;
        }
        mp += dq;
        xo += dr;
        if (xo >= w) {
            xo -= w;
            mp += 1;
        }
    }
    __syncthreads();
    // ---- pass 2: synthesis along W out of LDS -> global, two outputs per thread
    const int obase = y0 * Wo;
    for (int yl = Q0, m = R0; yl < th;) {
        const T* lo_w = tmp + yl * w2;
        const T* hi_w = lo_w + w;
        T e, o;
        SynthPair<T, FT>::run(m, w, mode, tp, [&](int i) { return lo_w[i]; }, [&](int i) { return hi_w[i]; }, e, o);
        const int at = obase + yl * Wo + 2 * m;
        const bool has_odd = 2 * m + 1 < Wo;
        if constexpr (FINAL != 0) {
            if constexpr (FINAL == 2) {
                e = fma_t((T)mix.ku, (T)mix.usub[at], (T)mix.kt * e);
                if (has_odd) o = fma_t((T)mix.ku, (T)mix.usub[at + 1], (T)mix.kt * o);
            }
            outf[at] = subtract ? xsub[at] - (float)e : (float)e;
            if (has_odd) outf[at + 1] = subtract ? xsub[at + 1] - (float)o : (float)o;
        } else {
            out[at] = e;
            if (has_odd) out[at + 1] = o;
        }
        yl += DQ;
        m += DR;
        if (m >= wp) {
            m -= wp;
            yl += 1;
        }
    }
}

template <typename T, int FINAL, int FT, typename THi = T>
__global__ void __launch_bounds__(kTileThreads) idwt2_tile_kernel(const T* __restrict__ ll, int ll_h, int ll_w,
                                                                  const THi* __restrict__ hi, T* __restrict__ out,
                                                                  const float* __restrict__ xsub, float* __restrict__ outf,
                                                                  int64_t planes, int h, int w, int Ho, int Wo, int tiles, Taps<T> tp,
                                                                  int mode, int subtract, FinalMix mix) {
    kernarg_touch_for(ll, ll_h, ll_w, hi, out, xsub, outf, planes, h, w, Ho, Wo, tiles, tp, mode, subtract, mix);
    extern __shared__ __align__(16) unsigned char tile_smem[];
    T* const tmp = reinterpret_cast<T*>(tile_smem);  // [kInvRows][lo_w | hi_w]
    const int64_t hw = (int64_t)h * w, ohw = (int64_t)Ho * Wo;
    for (int64_t job = blockIdx.x; job < planes * tiles; job += gridDim.x) {
        const int64_t p = job / tiles;
        const int y0 = (int)(job - p * tiles) * kInvRows;  // even
        FinalMix m = mix;
        if constexpr (FINAL == 2) m.usub += p * ohw;
        inv_tile_job<T, FINAL, FT, Taps<T>, THi>(ll + p * (int64_t)ll_h * ll_w, ll_w, hi + p * 3 * hw, FINAL ? nullptr : out + p * ohw,
                                   (FINAL && xsub) ? xsub + p * ohw : nullptr, FINAL ? outf + p * ohw : nullptr, h, w, Wo, y0,
                                   min(kInvRows, Ho - y0), tp, mode, subtract, tmp, m);
    }
}

// ---- levels >= 2 of the WaveletCFG step in ONE launch: a workgroup owns a plane and walks it down (analysis of cond and
// uncond, band arithmetic) and back up (synthesis) through the same global workspace the per-level launches use; every
// dependency is inside the workgroup, so a device-scope fence + barrier per level replaces 2 (levels - 1) launches of
// kernels that are mostly launch latency (a 10 x 10 plane per workgroup).
constexpr int kDeepMaxLevels = 8;
constexpr int kDeepTaps = 20;
template <typename T>
struct TapsSmall {
    T lo[kDeepTaps], hi[kDeepTaps];
};
template <typename T>
struct DeepArgs {
    int64_t planes;
    int levels;                                        // deep levels k = 1 .. levels
    int H[kDeepMaxLevels + 1], W[kDeepMaxLevels + 1];    // [0]: the input approximation plane; [k]: coefficient plane of level k
    int Hr[kDeepMaxLevels + 1], Wr[kDeepMaxLevels + 1];  // reconstruction produced FROM level k
    int64_t off_in_c, off_in_u;                        // element offsets into the workspace
    int64_t off_d[kDeepMaxLevels + 1], off_c[kDeepMaxLevels + 1], off_u[kDeepMaxLevels + 1], off_r[kDeepMaxLevels + 1];
    T hi_scales[kDeepMaxLevels + 1][12];               // per level: cond[3], uncond[3], diff[3], final[3]
    T ll_scales[4];
    T strength;
    int blend_mode, mode_fwd, mode_inv;
    int fwd_rows[kDeepMaxLevels + 1], inv_rows[kDeepMaxLevels + 1];  // rows per job (multiples of the tile heights; whole levels when LDS allows)
    TapsSmall<T> dec, rec;
};

template <typename T, int FT, bool ZERO, bool DIFF>
__global__ void __launch_bounds__(kTileThreads) wcfg_deep_kernel(T* base, DeepArgs<T> a) {
    kernarg_touch_for(base, a);
    extern __shared__ __align__(16) unsigned char tile_smem[];
    for (int64_t p = blockIdx.x; p < a.planes; p += gridDim.x) {
        // ---- analysis, finest deep level first
        for (int k = 1; k <= a.levels; ++k) {
            const int H = a.H[k - 1], W = a.W[k - 1], h = a.H[k], w = a.W[k];
            const int frows = a.fwd_rows[k];
            const FwdLds<T, DIFF ? 2 : 4, FT> lds(tile_smem, W, w, frows);
            __syncthreads();
            fwd_build_maps<FT>(lds.xmap, lds.ymap, H, W, h, w, lds.Wh, a.mode_fwd);
            BandArgs<T> ba;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                ba.hc[g] = a.hi_scales[k][g];
                ba.hu[g] = a.hi_scales[k][3 + g];
                ba.hd[g] = a.hi_scales[k][6 + g];
                ba.hf[g] = a.hi_scales[k][9 + g];
            }
            ba.lc = a.ll_scales[0];
            ba.lu = a.ll_scales[1];
            ba.ld = a.ll_scales[2];
            ba.lf = a.ll_scales[3];
            ba.combine_ll = k == a.levels;
            ba.blend_mode = a.blend_mode;
            ba.strength = a.strength;
            const int64_t in_plane = (int64_t)H * W, hw = (int64_t)h * w;
            const T* pc = base + (k == 1 ? a.off_in_c : a.off_c[k - 1]) + p * in_plane;
            const T* pu = DIFF ? nullptr : base + (k == 1 ? a.off_in_u : a.off_u[k - 1]) + p * in_plane;
            T* oc = base + a.off_c[k] + p * hw;
            T* ou = (DIFF || ba.combine_ll) ? nullptr : base + a.off_u[k] + p * hw;
            T* od = base + a.off_d[k] + p * 3 * hw;
            for (int y0 = 0; y0 < h; y0 += frows)
                fwd_tile_job<T, T, DIFF ? kFwdScale : kFwdPair, FT, ZERO, TapsSmall<T>, true>(pc, pu, oc, ou, od, W, h, w, y0, min(frows, h - y0), a.dec, ba, lds);
            // the workgroup re-reads what it just stored: workgroup scope is enough (one CU, one L1); a device-scope fence
            // would write back / invalidate L2 once per level per plane
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __syncthreads();
        }
        // ---- synthesis, coarsest level first
        const T* ll = base + a.off_c[a.levels] + p * (int64_t)a.H[a.levels] * a.W[a.levels];
        int ll_w = a.W[a.levels];
        T* const tmp = reinterpret_cast<T*>(tile_smem);
        for (int k = a.levels; k >= 1; --k) {
            const int h = a.H[k], w = a.W[k], Ho = a.Hr[k], Wo = a.Wr[k];
            const T* d = base + a.off_d[k] + p * 3 * (int64_t)h * w;
            T* r = base + a.off_r[k] + p * (int64_t)Ho * Wo;
            for (int y0 = 0; y0 < Ho; y0 += a.inv_rows[k])
                inv_tile_job<T, 0, FT>(ll, ll_w, d, r, nullptr, nullptr, h, w, Wo, y0, min(a.inv_rows[k], Ho - y0), a.rec, a.mode_inv, 0, tmp);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __syncthreads();
            ll = r;
            ll_w = Wo;
        }
    }
}

// run `body(std::integral_constant<int, FT>)` for the supported filter lengths
template <typename Body>
static inline void with_taps(int F, Body&& body) {
    switch (F) {
#define SONAR_FT(N) case N: body(std::integral_constant<int, N>{}); break;
        SONAR_FT(2) SONAR_FT(4) SONAR_FT(6) SONAR_FT(8) SONAR_FT(10) SONAR_FT(12) SONAR_FT(14) SONAR_FT(16) SONAR_FT(18) SONAR_FT(20)
#undef SONAR_FT
        default: break;
    }
}

static inline int tile_grid(int64_t jobs) { return (int)std::min<int64_t>(jobs, 1 << 20); }

// plain single-level forward through the tile kernel; false if unsupported here (caller falls back to the per-pass kernels)
template <typename T>
static bool dwt2_fwd_tiled(const T* x, T* ll, T* hi, int64_t planes, int H, int W, int h, int w, const Taps<T>& tp, int mode,
                           hipStream_t st) {
    if (!tile_taps_ok(tp.len)) return false;
    const size_t lds = fwd_lds_bytes(W, h, w, tp.len, sizeof(T), 1);
    if (lds > kTileLdsLimit) return false;
    const int tiles = (h + kFwdRows - 1) / kFwdRows;
    with_taps(tp.len, [&](auto ft) {
        constexpr int FT = decltype(ft)::value;
        if (mode == kZero)
            hipLaunchKernelGGL((dwt2_tile_kernel<T, T, kFwdPlain, FT, true>), dim3(tile_grid(planes * tiles)), dim3(kTileThreads), lds, st, x,
                               (const T*)nullptr, ll, (T*)nullptr, hi, planes, H, W, h, w, tiles, tp, mode, BandArgs<T>{});
        else
            hipLaunchKernelGGL((dwt2_tile_kernel<T, T, kFwdPlain, FT, false>), dim3(tile_grid(planes * tiles)), dim3(kTileThreads), lds, st, x,
                               (const T*)nullptr, ll, (T*)nullptr, hi, planes, H, W, h, w, tiles, tp, mode, BandArgs<T>{});
    });
    return true;
}

template <typename T>
static bool dwt2_inv_tiled(const T* ll, int ll_h, int ll_w, const T* hi, T* out, int64_t planes, int h, int w, int Ho, int Wo,
                           const Taps<T>& tp, int mode, hipStream_t st) {
    if (!tile_taps_ok(tp.len)) return false;
    const size_t lds = inv_lds_bytes(w, sizeof(T));
    if (lds > kTileLdsLimit) return false;
    const int tiles = (Ho + kInvRows - 1) / kInvRows;
    with_taps(tp.len, [&](auto ft) {
        hipLaunchKernelGGL((idwt2_tile_kernel<T, 0, decltype(ft)::value>), dim3(tile_grid(planes * tiles)), dim3(kTileThreads), lds, st,
                           ll, ll_h, ll_w, hi, out, (const float*)nullptr, (float*)nullptr, planes, h, w, Ho, Wo, tiles, tp, mode, 0, FinalMix{});
    });
    return true;
}

// dwt_bands.hip: levels 2 .. J with their coefficients resident in LDS (dwt_bands.h), on the level-1 approximation planes `ll`
// [planes][H1][W1]: out <- Phi_D(ll), or acc + Phi_D(ll) when `acc` is given.  False (nothing launched): not taken.
bool bands_deep(const float* ll, float* acc, float* out, int64_t planes, int H1, int W1, int levels, const double* dec_lo, const double* dec_hi,
                const double* rec_lo, const double* rec_hi, int flen, int mode_fwd, int mode_inv, const double* yh_scales, double yl_scale, hipStream_t st);
bool bands_deep(const double* ll, double* acc, double* out, int64_t planes, int H1, int W1, int levels, const double* dec_lo, const double* dec_hi,
                const double* rec_lo, const double* rec_hi, int flen, int mode_fwd, int mode_inv, const double* yh_scales, double yl_scale, hipStream_t st);

// ---- whole WaveletCFG transform-domain step (py/wavelet_cfg.py:750-791 + 729-748), `levels` launches each way
struct WcfgPlan {
    int levels;
    int H[kMaxLevels + 1], W[kMaxLevels + 1];   // [0] = input plane, [j] = coefficient plane of level j
    int Hr[kMaxLevels + 1], Wr[kMaxLevels + 1]; // reconstruction produced FROM level j (size of level j-1 input, maybe +1)
    int64_t off_d[kMaxLevels + 1], off_c[kMaxLevels + 1], off_u[kMaxLevels + 1], off_r[kMaxLevels + 1];  // elements
    int64_t total;  // elements
};

static inline bool wcfg_plan(WcfgPlan& pl, int64_t planes, int64_t H, int64_t W, int levels, int dec_len, int mode_fwd, int rec_len,
                             int mode_inv, bool diff_only = false) {
    if (levels < 1 || levels > kMaxLevels) return false;
    pl.levels = levels;
    pl.H[0] = (int)H;
    pl.W[0] = (int)W;
    int64_t at = 0;
    for (int j = 1; j <= levels; ++j) {
        pl.H[j] = (int)dwt_len(pl.H[j - 1], dec_len, mode_fwd);
        pl.W[j] = (int)dwt_len(pl.W[j - 1], dec_len, mode_fwd);
        pl.Hr[j] = mode_inv == kPeriodization ? 2 * pl.H[j] : 2 * pl.H[j] - rec_len + 2;
        pl.Wr[j] = mode_inv == kPeriodization ? 2 * pl.W[j] : 2 * pl.W[j] - rec_len + 2;
        if (pl.Hr[j] < pl.H[j - 1] || pl.Wr[j] < pl.W[j - 1]) return false;  // the inverse cannot cover the level below
        const int64_t n = planes * pl.H[j] * pl.W[j];
        pl.off_d[j] = at; at += 3 * n;
        pl.off_c[j] = at; at += n;
        pl.off_u[j] = at; at += (j < levels && !diff_only ? n : 0);
        pl.off_r[j] = at; at += (j > 1 ? planes * (int64_t)pl.Hr[j] * pl.Wr[j] : 0);
    }
    pl.total = at;
    return true;
}

// process-wide: WaveletCFG's level-1 detail bands stored in fp32 in fp64 mode (default on; sonar_wcfg_hi_storage)
inline int& wcfg_hi_fp32_switch() {
    static int on = 1;
    return on;
}

template <typename T>
static int wcfg_fused(const float* cond, const float* uncond, const float* x, float* out, int64_t planes, int64_t H, int64_t W,
                      int levels, const double* dec_lo, const double* dec_hi, int dec_len, int mode_fwd, const double* rec_lo,
                      const double* rec_hi, int rec_len, int mode_inv, const double* yl_scales, const double* yh_scales,
                      int blend_mode, double strength, int subtract_from_x, void* ws, int64_t ws_bytes, hipStream_t st,
                      const char* what, bool perfect_reconstruction = false) {
    SONAR_REQUIRE(cond && uncond && out && (x || !subtract_from_x) && ws && yl_scales && yh_scales && planes >= 0 && mode_fwd >= 0 &&
                      mode_fwd <= 5 && mode_inv >= 0 && mode_inv <= 5 && blend_mode >= 0 && blend_mode <= 2,
                  SONAR_ERR_ARG, "%s: bad argument", what);
    SONAR_REQUIRE(dims_ok(H, W), SONAR_ERR_UNSUPPORTED, "%s: bad plane size", what);
    Taps<T> dec, rec;
    SONAR_REQUIRE(make_taps(dec, dec_lo, dec_hi, dec_len) && make_taps(rec, rec_lo, rec_hi, rec_len), SONAR_ERR_ARG,
                  "%s: 1..%d filter taps required", what, kMaxTaps);
    SONAR_REQUIRE(tile_taps_ok(dec_len) && tile_taps_ok(rec_len), SONAR_ERR_UNSUPPORTED, "%s: even filter lengths 2..20 only", what);
    // difference-only rule over a perfect-reconstruction pair (the caller vouches for the pair: one wavelet both ways): every cond /
    // uncond / final scale is 1, the blend is linear in (uncond band, difference band), so
    //   IDWT(blend(U, D (C - U), t)) = ku u + kt IDWT(D DWT(c - u))
    // and ONE tensor is transformed instead of two (py/wavelet_cfg.py:765-787 with the identities of csrc/dwt_lowpass.h)
    bool diff_only = perfect_reconstruction && levels >= 1 && (dec_len == 2 || (mode_fwd == kPeriodization) == (mode_inv == kPeriodization));
    for (int i = 0; diff_only && i < 4; ++i)
        if (i != 2 && yl_scales[i] != 1.0) diff_only = false;
    for (int j = 0; diff_only && j < levels; ++j)
        for (int i = 0; i < 12; ++i)
            if (i / 3 != 2 && yh_scales[(int64_t)j * 12 + i] != 1.0) diff_only = false;
    FinalMix mix;
    if (diff_only) {
        mix.usub = uncond;
        mix.ku = blend_mode == SONAR_BLEND_LERP ? 1.0 - strength : 1.0;
        mix.kt = blend_mode == SONAR_BLEND_SUBTRACT_B ? -strength : strength;
    }
    WcfgPlan pl;
    SONAR_REQUIRE(wcfg_plan(pl, planes, H, W, levels, dec_len, mode_fwd, rec_len, mode_inv, diff_only), SONAR_ERR_UNSUPPORTED,
                  "%s: 1..%d levels and an inverse filter that covers every level are required", what, kMaxLevels);
    SONAR_REQUIRE(ws_bytes >= pl.total * (int64_t)sizeof(T), SONAR_ERR_ARG, "%s: workspace too small (%lld < %lld bytes)", what,
                  (long long)ws_bytes, (long long)(pl.total * (int64_t)sizeof(T)));
    if (planes == 0) return SONAR_OK;
    T* base = (T*)ws;
    // every level must fit its tile in LDS (checked before the first launch so a refusal has no side effects)
    for (int j = 1; j <= levels; ++j)
        SONAR_REQUIRE(fwd_lds_bytes(pl.W[j - 1], pl.H[j], pl.W[j], dec_len, sizeof(T), diff_only ? 1 : 2) <= kTileLdsLimit &&
                          inv_lds_bytes(pl.W[j], sizeof(T)) <= kTileLdsLimit,
                      SONAR_ERR_UNSUPPORTED, "%s: level %d (%d x %d) does not fit the LDS tile", what, j, pl.H[j], pl.W[j]);
    auto band_args = [&](int j) {
        BandArgs<T> ba;
        const double* row = yh_scales + (int64_t)(j - 1) * 12;
        for (int g = 0; g < 3; ++g) {
            ba.hc[g] = (T)row[0 + g];
            ba.hu[g] = (T)row[3 + g];
            ba.hd[g] = (T)row[6 + g];
            ba.hf[g] = (T)row[9 + g];
        }
        ba.lc = (T)yl_scales[0];
        ba.lu = (T)yl_scales[1];
        ba.ld = (T)yl_scales[2];
        ba.lf = (T)yl_scales[3];
        ba.combine_ll = j == levels;
        ba.blend_mode = blend_mode;
        ba.strength = (T)strength;
        return ba;
    };
    // Level 1's three detail bands cross the workspace once each way and are 3/4 of that level's coefficients: in fp64 mode they are
    // STORED in fp32 (the arithmetic on both sides stays fp64, and the result is an fp32 tensor: a detail coefficient rounded to 2^-24
    // relative moves an output value by a fraction of its own final rounding; the fixtures' 2e-6 tolerance has a 20 x margin) -- 110 MB
    // less HBM traffic per 256 SDXL latents.  The approximation band, which feeds the deeper levels, and everything above level 1 stay fp64.
    // sonar_wcfg_hi_storage(0) keeps them in T (the test that pins the fused route to the per-pass route at 1e-11 runs that way).
    using Hi1 = std::conditional_t<std::is_same<T, double>::value, float, T>;
    const bool hi32 = std::is_same<T, double>::value && wcfg_hi_fp32_switch() != 0;
    // levels >= 2 in one launch (a workgroup per plane) when the filters have one length and there are enough levels to matter
    const bool deep = levels >= 3 && dec_len == rec_len && levels - 1 <= kDeepMaxLevels && dec_len <= kDeepTaps;
    auto launch_fwd = [&](int j) {
        const int tiles = (pl.H[j] + kFwdRows - 1) / kFwdRows;
        const int nten = diff_only ? 1 : 2;
        const size_t lds = fwd_lds_bytes(pl.W[j - 1], pl.H[j], pl.W[j], dec_len, sizeof(T), nten);
        const dim3 grid(tile_grid(planes * tiles)), blk(kTileThreads);
        T* d = base + pl.off_d[j];
        T* c = base + pl.off_c[j];
        T* u = base + pl.off_u[j];
        const BandArgs<T> ba = band_args(j);
        with_taps(dec_len, [&](auto ft) {
            constexpr int FT = decltype(ft)::value;
            auto go = [&](auto zero) {
                constexpr bool Z = decltype(zero)::value;
                if (diff_only && j == 1 && hi32)
                    hipLaunchKernelGGL((dwt2_tile_kernel<T, float, kFwdDiff, FT, Z, Hi1>), grid, blk, lds, st, cond, uncond, c, (T*)nullptr,
                                       reinterpret_cast<Hi1*>(d), planes, pl.H[0], pl.W[0], pl.H[1], pl.W[1], tiles, dec, mode_fwd, ba);
                else if (diff_only && j == 1)
                    hipLaunchKernelGGL((dwt2_tile_kernel<T, float, kFwdDiff, FT, Z>), grid, blk, lds, st, cond, uncond, c, (T*)nullptr, d,
                                       planes, pl.H[0], pl.W[0], pl.H[1], pl.W[1], tiles, dec, mode_fwd, ba);
                else if (diff_only)
                    hipLaunchKernelGGL((dwt2_tile_kernel<T, T, kFwdScale, FT, Z>), grid, blk, lds, st, (const T*)(base + pl.off_c[j - 1]),
                                       (const T*)nullptr, c, (T*)nullptr, d, planes, pl.H[j - 1], pl.W[j - 1], pl.H[j], pl.W[j], tiles, dec,
                                       mode_fwd, ba);
                else if (j == 1 && hi32)
                    hipLaunchKernelGGL((dwt2_tile_kernel<T, float, kFwdPair, FT, Z, Hi1>), grid, blk, lds, st, cond, uncond, c, u,
                                       reinterpret_cast<Hi1*>(d), planes, pl.H[0], pl.W[0], pl.H[1], pl.W[1], tiles, dec, mode_fwd, ba);
                else if (j == 1)
                    hipLaunchKernelGGL((dwt2_tile_kernel<T, float, kFwdPair, FT, Z>), grid, blk, lds, st, cond, uncond, c, u, d, planes,
                                       pl.H[0], pl.W[0], pl.H[1], pl.W[1], tiles, dec, mode_fwd, ba);
                else
                    hipLaunchKernelGGL((dwt2_tile_kernel<T, T, kFwdPair, FT, Z>), grid, blk, lds, st, (const T*)(base + pl.off_c[j - 1]),
                                       (const T*)(base + pl.off_u[j - 1]), c, u, d, planes, pl.H[j - 1], pl.W[j - 1], pl.H[j], pl.W[j],
                                       tiles, dec, mode_fwd, ba);
            };
            if (mode_fwd == kZero) go(std::true_type{}); else go(std::false_type{});
        });
    };
    // one synthesis level; returns the (pointer, rows, cols) of what it produced for the level above
    const T* ll = base + pl.off_c[levels];
    int ll_h = pl.H[levels], ll_w = pl.W[levels];
    auto launch_inv = [&](int j) {
        const int Ho = j > 1 ? pl.Hr[j] : (int)H, Wo = j > 1 ? pl.Wr[j] : (int)W;
        const int tiles = (Ho + kInvRows - 1) / kInvRows;
        const size_t lds = inv_lds_bytes(pl.W[j], sizeof(T));
        const dim3 grid(tile_grid(planes * tiles)), blk(kTileThreads);
        const T* d = base + pl.off_d[j];
        if (j > 1) {
            T* r = base + pl.off_r[j];
            with_taps(rec_len, [&](auto ft) {
                hipLaunchKernelGGL((idwt2_tile_kernel<T, 0, decltype(ft)::value>), grid, blk, lds, st, ll, ll_h, ll_w, d, r,
                                   (const float*)nullptr, (float*)nullptr, planes, pl.H[j], pl.W[j], Ho, Wo, tiles, rec, mode_inv, 0, FinalMix{});
            });
            ll = r;
            ll_h = Ho;
            ll_w = Wo;
        } else {
            with_taps(rec_len, [&](auto ft) {
                if (diff_only && hi32)
                    hipLaunchKernelGGL((idwt2_tile_kernel<T, 2, decltype(ft)::value, Hi1>), grid, blk, lds, st, ll, ll_h, ll_w,
                                       reinterpret_cast<const Hi1*>(d), (T*)nullptr, x, out, planes, pl.H[j], pl.W[j], Ho, Wo, tiles, rec, mode_inv,
                                       subtract_from_x, mix);
                else if (diff_only)
                    hipLaunchKernelGGL((idwt2_tile_kernel<T, 2, decltype(ft)::value>), grid, blk, lds, st, ll, ll_h, ll_w, d, (T*)nullptr, x,
                                       out, planes, pl.H[j], pl.W[j], Ho, Wo, tiles, rec, mode_inv, subtract_from_x, mix);
                else if (hi32)
                    hipLaunchKernelGGL((idwt2_tile_kernel<T, 1, decltype(ft)::value, Hi1>), grid, blk, lds, st, ll, ll_h, ll_w,
                                       reinterpret_cast<const Hi1*>(d), (T*)nullptr, x, out, planes, pl.H[j], pl.W[j], Ho, Wo, tiles, rec, mode_inv,
                                       subtract_from_x, FinalMix{});
                else
                    hipLaunchKernelGGL((idwt2_tile_kernel<T, 1, decltype(ft)::value>), grid, blk, lds, st, ll, ll_h, ll_w, d, (T*)nullptr, x,
                                       out, planes, pl.H[j], pl.W[j], Ho, Wo, tiles, rec, mode_inv, subtract_from_x, FinalMix{});
            });
        }
    };
    // With a perfect-reconstruction pair the deeper levels are a LINEAR map of the level-1 approximation(s) with per-band coefficients
    // (blend(s_u U, s_d (s_c C - s_u U), t) s_f = A C + B U), and dwt_bands.h evaluates it with the coefficients resident in LDS: no
    // workspace round trips between the levels, a quarter of the old deep kernel's time.
    const bool pr_pair = perfect_reconstruction && dec_len == rec_len && (dec_len == 2 || (mode_fwd == kPeriodization) == (mode_inv == kPeriodization));
    bool deep_done = false;
    if (deep && pr_pair && levels - 1 <= kDeepMaxLevels) {
        const double t = strength, sign = blend_mode == SONAR_BLEND_SUBTRACT_B ? -1.0 : 1.0, keep = blend_mode == SONAR_BLEND_LERP ? 1.0 - t : 1.0;
        double sa[3 * kMaxLevels], sb[3 * kMaxLevels];
        for (int j = 2; j <= levels; ++j) {
            const double* row = yh_scales + (int64_t)(j - 1) * 12;
            for (int g = 0; g < 3; ++g) {
                const double s_c = row[g], s_u = row[3 + g], s_d = row[6 + g], s_f = row[9 + g];
                sa[3 * (j - 2) + g] = diff_only ? s_d : sign * t * s_d * s_c * s_f;
                sb[3 * (j - 2) + g] = (keep - sign * t * s_d) * s_u * s_f;
            }
        }
        const double yl_a = diff_only ? yl_scales[2] : sign * t * yl_scales[2] * yl_scales[0] * yl_scales[3];
        const double yl_b = (keep - sign * t * yl_scales[2]) * yl_scales[1] * yl_scales[3];
        T* const c1 = base + pl.off_c[1];
        T* const r1 = base + pl.off_r[2];  // [planes][Hr[2]][Wr[2]] holds a [planes][H[1]][W[1]] image
        // probe with the first launch; a refusal (LDS, ...) leaves the workspace untouched apart from level 1, which the old route redoes
        launch_fwd(1);
        if (diff_only) {
            deep_done = bands_deep(c1, (T*)nullptr, r1, planes, pl.H[1], pl.W[1], levels - 1, dec_lo, dec_hi, rec_lo, rec_hi, dec_len, mode_fwd, mode_inv,
                                   sa, yl_a, st);
        } else {
            T* const u1 = base + pl.off_u[1];
            deep_done = bands_deep(u1, (T*)nullptr, r1, planes, pl.H[1], pl.W[1], levels - 1, dec_lo, dec_hi, rec_lo, rec_hi, dec_len, mode_fwd, mode_inv,
                                   sb, yl_b, st) &&
                        bands_deep(c1, r1, r1, planes, pl.H[1], pl.W[1], levels - 1, dec_lo, dec_hi, rec_lo, rec_hi, dec_len, mode_fwd, mode_inv, sa,
                                   yl_a, st);
        }
        if (deep_done) {
            ll = r1;
            ll_h = pl.H[1];
            ll_w = pl.W[1];
            launch_inv(1);
        }
    }
    if (deep_done) {
        // issued above
    } else if (deep) {
        launch_fwd(1);  // reads the fp32 inputs (cast in registers)
        DeepArgs<T> a{};
        a.planes = planes;
        a.levels = levels - 1;
        a.off_in_c = pl.off_c[1];
        a.off_in_u = pl.off_u[1];
        size_t lds = 0;
        for (int k = 0; k <= a.levels; ++k) {
            a.H[k] = pl.H[k + 1];
            a.W[k] = pl.W[k + 1];
            if (k == 0) continue;
            a.Hr[k] = pl.Hr[k + 1];
            a.Wr[k] = pl.Wr[k + 1];
            a.off_d[k] = pl.off_d[k + 1];
            a.off_c[k] = pl.off_c[k + 1];
            a.off_u[k] = pl.off_u[k + 1];
            a.off_r[k] = pl.off_r[k + 1];
            const BandArgs<T> ba = band_args(k + 1);
            for (int g = 0; g < 3; ++g) {
                a.hi_scales[k][g] = ba.hc[g];
                a.hi_scales[k][3 + g] = ba.hu[g];
                a.hi_scales[k][6 + g] = ba.hd[g];
                a.hi_scales[k][9 + g] = ba.hf[g];
            }
            // rows per job: as many as fit the per-workgroup LDS budget (whole levels for the small ones) -- the kernel is VALU-bound
            // and a job of 8 x 37 outputs leaves half the lanes of its second sweep idle
            constexpr size_t kDeepLdsBudget = 39 * 1024;  // four workgroups per CU
            int fr = kFwdRows, ir = kInvRows;
            while (fr < a.H[k] && fwd_lds_bytes(a.W[k - 1], a.H[k], a.W[k], dec_len, sizeof(T), diff_only ? 1 : 2, fr + kFwdRows) <= kDeepLdsBudget) fr += kFwdRows;
            while (ir < a.Hr[k] && inv_lds_bytes(a.W[k], sizeof(T), ir + kInvRows) <= kDeepLdsBudget) ir += kInvRows;
            a.fwd_rows[k] = fr;
            a.inv_rows[k] = ir;
            lds = std::max(lds, std::max(fwd_lds_bytes(a.W[k - 1], a.H[k], a.W[k], dec_len, sizeof(T), diff_only ? 1 : 2, fr), inv_lds_bytes(a.W[k], sizeof(T), ir)));
        }
        a.ll_scales[0] = (T)yl_scales[0];
        a.ll_scales[1] = (T)yl_scales[1];
        a.ll_scales[2] = (T)yl_scales[2];
        a.ll_scales[3] = (T)yl_scales[3];
        a.strength = (T)strength;
        a.blend_mode = blend_mode;
        a.mode_fwd = mode_fwd;
        a.mode_inv = mode_inv;
        for (int i = 0; i < kDeepTaps; ++i) {
            a.dec.lo[i] = i < dec_len ? dec.lo[i] : T(0);
            a.dec.hi[i] = i < dec_len ? dec.hi[i] : T(0);
            a.rec.lo[i] = i < rec_len ? rec.lo[i] : T(0);
            a.rec.hi[i] = i < rec_len ? rec.hi[i] : T(0);
        }
        with_taps(dec_len, [&](auto ft) {
            // zero extension is the only mode with "no source" positions: the others skip its per-tap selects (the kernel is VALU-bound)
            constexpr int FT = decltype(ft)::value;
            const dim3 grid((int)std::min<int64_t>(planes, 1 << 20)), blk(kTileThreads);
            if (mode_fwd == kZero && diff_only)
                hipLaunchKernelGGL((wcfg_deep_kernel<T, FT, true, true>), grid, blk, lds, st, base, a);
            else if (mode_fwd == kZero)
                hipLaunchKernelGGL((wcfg_deep_kernel<T, FT, true, false>), grid, blk, lds, st, base, a);
            else if (diff_only)
                hipLaunchKernelGGL((wcfg_deep_kernel<T, FT, false, true>), grid, blk, lds, st, base, a);
            else
                hipLaunchKernelGGL((wcfg_deep_kernel<T, FT, false, false>), grid, blk, lds, st, base, a);
        });
        ll = base + pl.off_r[2];  // what the deep kernel reconstructed for level 1
        ll_h = pl.Hr[2];
        ll_w = pl.Wr[2];
        launch_inv(1);
    } else {
        for (int j = 1; j <= levels; ++j) launch_fwd(j);
        for (int j = levels; j >= 1; --j) launch_inv(j);
    }
    return check_launch(what);
}

}  // namespace sonar
