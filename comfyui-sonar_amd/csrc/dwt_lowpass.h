// WaveletCFG for rules that only scale the DIFFERENCE bands, with one scale per level (py/wavelet_cfg.py:750-791 with `cond`,
// `uncond`, `final` absent or unit and `difference.yh_scales` scalar per level -- the node's placeholder rule, BASELINE cfg4).
//
// The transform-domain step is linear:  IDWT(blend(U, D (C - U), t)) = k_u u + k_t IDWT(D DWT(c - u)), U = DWT(u), C = DWT(c),
// because IDWT(DWT(u)) = u (perfect reconstruction, same wavelet both ways).  With ONE scale d_j for the three detail bands of level
// j and l for the approximation, the band-scaled reconstruction telescopes through the same identity, level by level
// (details_j = LL_{j-1} - Up_j(LL_j)):
//     IDWT(D DWT(v)) = d_1 v + Up_1( (d_2 - d_1) LL_1 + Up_2( (d_3 - d_2) LL_2 + ... Up_J( (l - d_J) LL_J ) ) )
// LL_j = low-pass analysis chain of v, Up_j = synthesis of level j with zero detail bands.  No detail band is ever formed: a
// workgroup owns a plane, keeps the LL pyramid (67^2 + 37^2 + ... values for a 128 x 128 plane) in LDS, and the tensors cross HBM
// once: read cond, uncond, x, write out = x - result (16N bytes per latent, the figure SURVEY.md 8d prices).  fp64 (the reference's
// high_precision_mode default) or fp32 arithmetic; results agree with the band-by-band path to rounding (tests compare both with
// the reference-run fixtures).
#pragma once
#ifndef SONAR_LOWPASS_NT
#define SONAR_LOWPASS_NT 0  // profiling builds: the output phase's 16-byte stores with the non-temporal hint
#endif
#include "dwt_tile.h"

namespace sonar {

#ifndef SONAR_LOW_THREADS
#define SONAR_LOW_THREADS 512  // (A/B, round 6: 768 / 1024 threads per plane are SLOWER -- 122.9 / 116.2 us against 95.5 fp64, profiles/r06_experiments.md)
#endif
constexpr int kLowThreads = SONAR_LOW_THREADS;   // 8 waves per plane: more than the 4 of round 2 hide more of the phases' latency; beyond 8 the phases' fixed cost wins
constexpr int kLowRows = 16;       // output rows per level-1 analysis tile / final synthesis tile
constexpr int kLowMaxLevels = 8;

template <typename T>
struct LowArgs {
    int64_t planes;
    int levels;
    int H[kLowMaxLevels + 1], W[kLowMaxLevels + 1];  // [0]: the latent plane; [j]: LL plane of level j
    int off_ll[kLowMaxLevels + 1];                    // LDS offsets (elements of T) of LL_1 .. LL_J
    int off_tmp;                                      // LDS scratch (elements of T)
    int off_tmp1, rows1;                              // level-1 analysis scratch (overlays LL_2 .. and the scratch) and its tile height
    int rows_out;                                     // output rows per tile of the final synthesis
    int off_maps;                                     // LDS byte offset of the extension tables
    int map_h[kLowMaxLevels + 1], map_w[kLowMaxLevels + 1];  // table offsets (ints) per level, rows / columns
    T g[kLowMaxLevels + 1];                           // g[0] = d_1; g[j] = d_{j+1} - d_j; g[J] = l - d_J
    T ku, kt;                                         // result = ku u + kt (g[0] v + Up_1(...))
    int subtract_from_x, mode_fwd, mode_inv;
    int vec2;                                         // ... 8-byte aligned (a view at an odd storage offset is not): pairs as one access
    int vec4;                                         // cond, uncond, x and out are 16-byte aligned: the output phase may use 16-byte accesses
    T dlo[kDeepTaps], rlo[kDeepTaps];
};

// base[elem] with the BYTE offset formed in 32 bits: "uniform base + zero-extended 32-bit offset" is the one form the compiler turns into a
// scalar-base global access (`global_load_dword v, v_off, s[base:base+1]`); indexed with an int -- or an unsigned element index, whose
// scaling by four may leave 32 bits as far as the compiler knows -- it builds a 64-bit vector address per access: six instructions per row
// and tensor in the level-1 analysis (round 6: 155 -> ~100 instructions per item).  A plane is far below 4 GiB.
template <typename V>
__device__ __forceinline__ const V& at_u32(const float* base, uint32_t elem) {
    return *reinterpret_cast<const V*>(reinterpret_cast<const char*>(base) + (uint32_t)(elem * 4u));
}
template <typename V>
__device__ __forceinline__ V& at_u32(float* base, uint32_t elem) {
    return *reinterpret_cast<V*>(reinterpret_cast<char*>(base) + (uint32_t)(elem * 4u));
}

// items (row, col) of a rows x cols grid dealt to the workgroup's threads in flat order, without a division per item
struct Walk2 {
    int r, c, dr, dc;
    __device__ __forceinline__ Walk2(int tid, int cols) : r(tid / cols), c(tid - (tid / cols) * cols), dr(kLowThreads / cols), dc(kLowThreads - (kLowThreads / cols) * cols) {}
    __device__ __forceinline__ void next(int cols) {
        r += dr;
        c += dc;
        if (c >= cols) {
            c -= cols;
            r += 1;
        }
    }
};

// low-pass synthesis of the output pair (2m, 2m + 1) from one coefficient sequence (SynthPair of dwt_tile.h without the detail terms)
template <typename T, int FT, typename Load>
__device__ __forceinline__ void synth_low_pair(int m, int n, int mode, const T* __restrict__ rlo, Load&& la, T& even, T& odd) {
    constexpr int K = FT / 2;
    even = T(0);
    odd = T(0);
    if (mode != kPeriodization || (K & 1) == 1) {
#pragma unroll
        for (int k = K - 1; k >= 0; --k) {
            int i;
            if (mode != kPeriodization) {
                i = m + K - 1 - k;
                i = i < n ? i : n - 1;  // only beyond the valid output length (never consumed)
            } else {
                i = (m + (K - 1) / 2 - k) % n;
                if (i < 0) i += n;
            }
            const T a = la(i);
            even = fma_t(a, rlo[2 * k], even);
            odd = fma_t(a, rlo[2 * k + 1], odd);
        }
    } else {
#pragma unroll
        for (int k = K - 1; k >= 0; --k) {
            int ie = (m + K / 2 - (k + 1)) % n, io = (m + K / 2 - k) % n;
            if (ie < 0) ie += n;
            if (io < 0) io += n;
            even = fma_t(la(ie), rlo[2 * k + 1], even);
            odd = fma_t(la(io), rlo[2 * k], odd);
        }
    }
}

// ZERO: zero padding on the way down -- the only extension whose tables hold "no source" entries; the other modes' taps carry no clamp
// and no select
template <typename T, int FT, bool ZERO>
__global__ void __launch_bounds__(kLowThreads, (kLowThreads > 512 ? 8 : 1)) wcfg_lowpass_kernel(const float* __restrict__ cond, const float* __restrict__ uncond,
                                                                    const float* __restrict__ xin, float* __restrict__ out, LowArgs<T> a) {
    kernarg_touch_for(cond, uncond, xin, out, a);
    auto at0 = [](int s) { return ZERO ? max(s, 0) : s; };
    auto live = [](int s, T v) { return ZERO ? (s >= 0 ? v : T(0)) : v; };
    extern __shared__ __align__(16) unsigned char low_smem[];
    T* const lds = reinterpret_cast<T*>(low_smem);
    int* const maps = reinterpret_cast<int*>(low_smem + a.off_maps);
    T* const tmp = lds + a.off_tmp;
    const int tid = threadIdx.x;
    const int J = a.levels;
    // extension tables, the same for every plane: tap j of output i reads table[2 i + F - 1 - j] (source index, -1 = implicit zero)
    for (int j = 1; j <= J; ++j) {
        const int off = a.mode_fwd == kPeriodization ? FT / 2 : 1;
        const int Hp = a.H[j - 1], Wp = a.W[j - 1];
        const int He = (a.mode_fwd == kPeriodization && (Hp & 1)) ? Hp + 1 : Hp, We = (a.mode_fwd == kPeriodization && (Wp & 1)) ? Wp + 1 : Wp;
        for (int i = tid; i < 2 * a.H[j] + FT - 2; i += kLowThreads) maps[a.map_h[j] + i] = src_index(i + off - (FT - 1), Hp, He, a.mode_fwd);
        for (int i = tid; i < 2 * a.W[j] + FT - 2; i += kLowThreads) {
            const int sx = src_index(i + off - (FT - 1), Wp, We, a.mode_fwd);
            // level 1 reads its scratch rows in the parity-split layout: the table holds the slot
            maps[a.map_w[j] + i] = (j == 1 && sx >= 0) ? (sx & 1) * ((Wp + 1) >> 1) + (sx >> 1) : sx;
        }
    }
#ifdef SONAR_LOW_STAGGER  // profiling builds: the k-th resident workgroup of a CU starts k * SONAR_LOW_STAGGER (x 0.85 us) late
    for (int i = 0; i < (int)(blockIdx.x / 256) * SONAR_LOW_STAGGER; ++i) __builtin_amdgcn_s_sleep(32);
#endif
    const int H = a.H[0], W = a.W[0], h1 = a.H[1], w1 = a.W[1];
    const int Wh = (W + 1) >> 1, Ws = 2 * Wh;  // parity-split row of the level-1 scratch: slot(x) = (x & 1) Wh + x / 2
    for (int64_t p = blockIdx.x; p < a.planes; p += gridDim.x) {
        const float* pc = cond + p * (int64_t)H * W;
        const float* pu = uncond + p * (int64_t)H * W;
        __syncthreads();  // tables are built; the previous plane's readers are done
        // ---------------------------------------------------------------- level 1: low-pass analysis of v = cond - uncond (from global)
        {
            T* const ll1 = lds + a.off_ll[1];
            const int* const ymap = maps + a.map_h[1];
            const int* const xmap = maps + a.map_w[1];
            constexpr int THS = 4, NRS = 2 * THS + FT - 2;
            T* const tmp1 = lds + a.off_tmp1;
#ifdef SONAR_LOW_NOANALYSIS
            if (false)
#endif
            for (int y0 = 0; y0 < h1; y0 += a.rows1) {
                const int th = min(a.rows1, h1 - y0);
                // along H: one thread per (row group of 4, column); each input row of the group's window is read once.  (Two columns per
                // thread with 8-byte loads: 118 / 90 us instead of 99 / 85 -- the second window costs 25 registers and a resident workgroup.)
                for (Walk2 wk(tid, W); wk.r * THS < th; wk.next(W)) {
                    const int sub = wk.r, x = wk.c;
                    {
                        T v[NRS];
#pragma unroll
                        for (int r = 0; r < NRS; ++r) {
                            const int sy = ymap[min(2 * (y0 + sub * THS) + r, 2 * h1 + FT - 3)];
                            // (an UNSIGNED 32-bit element offset from the plane's uniform base: one scalar-base load per tensor; as a signed int
                            // the compiler sign-extended it and built two 64-bit addresses per row: eight instructions where two do)
                            const uint32_t at = (uint32_t)(at0(sy) * W + x);
                            const T d = (T)at_u32<float>(pc, at) - (T)at_u32<float>(pu, at);
                            v[r] = live(sy, d);
                        }
                        T* dst = tmp1 + (x & 1) * Wh + (x >> 1);
#pragma unroll
                        for (int yl = 0; yl < THS; ++yl) {
                            T lo = T(0);
#pragma unroll
                            for (int j = 0; j < FT; ++j) lo = fma_t(a.dlo[j], v[2 * yl + FT - 1 - j], lo);
                            if (sub * THS + yl < th) dst[(sub * THS + yl) * Ws] = lo;
                        }
                    }
                }
                __syncthreads();
                // along W out of LDS
                for (Walk2 wk(tid, w1); wk.r < th; wk.next(w1)) {
                    const int yl = wk.r, xo = wk.c;
                    const T* row = tmp1 + yl * Ws;
                    const int* xm = xmap + 2 * xo + (FT - 1);
                    T acc = T(0);
#pragma unroll
                    for (int j = 0; j < FT; ++j) {
                        const int slot = xm[-j];
                        const T q = row[at0(slot)];
                        acc = fma_t(a.dlo[j], live(slot, q), acc);
                    }
                    ll1[(y0 + yl) * w1 + xo] = acc;
                }
                __syncthreads();
            }
        }
        // ---------------------------------------------------------------- deeper levels: LL_j from LL_{j-1}, all in LDS
#ifdef SONAR_LOW_NODEEP
        if (false)
#endif
        for (int j = 2; j <= J; ++j) {
            const int Hp = a.H[j - 1], Wp = a.W[j - 1], h = a.H[j], w = a.W[j];
            const T* const src = lds + a.off_ll[j - 1];
            T* const dst = lds + a.off_ll[j];
            const int* const ymap = maps + a.map_h[j];
            const int* const xmap = maps + a.map_w[j];
            for (Walk2 wk(tid, Wp); wk.r < h; wk.next(Wp)) {   // along H
                const int yo = wk.r, x = wk.c;
                const int* ym = ymap + 2 * yo + (FT - 1);
                T acc = T(0);
#pragma unroll
                for (int t = 0; t < FT; ++t) {
                    const int sy = ym[-t];
                    const T q = src[at0(sy) * Wp + x];
                    acc = fma_t(a.dlo[t], live(sy, q), acc);
                }
                tmp[yo * Wp + x] = acc;
            }
            __syncthreads();
            for (Walk2 wk(tid, w); wk.r < h; wk.next(w)) {    // along W
                const int yo = wk.r, xo = wk.c;
                const int* xm = xmap + 2 * xo + (FT - 1);
                const T* row = tmp + yo * Wp;
                T acc = T(0);
#pragma unroll
                for (int t = 0; t < FT; ++t) {
                    const int sx = xm[-t];
                    const T q = row[at0(sx)];
                    acc = fma_t(a.dlo[t], live(sx, q), acc);
                }
                dst[yo * w + xo] = acc;
            }
            __syncthreads();
        }
        // ---------------------------------------------------------------- back up: B_J = g_J LL_J; B_{j-1} = g_{j-1} LL_{j-1} + Up_j(B_j)
        {
            T* const top = lds + a.off_ll[J];
            const T gJ = a.g[J];
            for (int it = tid; it < a.H[J] * a.W[J]; it += kLowThreads) top[it] *= gJ;
            __syncthreads();
        }
#ifdef SONAR_LOW_NODEEP
        if (false)
#endif
        for (int j = J; j >= 2; --j) {
            const int h = a.H[j], w = a.W[j], Ho = a.H[j - 1], Wo = a.W[j - 1];  // only the rows / columns the level below keeps
            const T* const B = lds + a.off_ll[j];
            T* const dst = lds + a.off_ll[j - 1];
            const int hp = (Ho + 1) >> 1, wp = (Wo + 1) >> 1;
            for (Walk2 wk(tid, w); wk.r < hp; wk.next(w)) {   // along H: rows (2m, 2m + 1) of column xo
                const int m = wk.r, xo = wk.c;
                T e, o;
                synth_low_pair<T, FT>(m, h, a.mode_inv, a.rlo, [&](int i) { return B[i * w + xo]; }, e, o);
                tmp[(2 * m) * w + xo] = e;
                if (2 * m + 1 < Ho) tmp[(2 * m + 1) * w + xo] = o;
            }
            __syncthreads();
            const T gp = a.g[j - 1];
            for (Walk2 wk(tid, wp); wk.r < Ho; wk.next(wp)) {  // along W, accumulate into g_{j-1} LL_{j-1}
                const int y = wk.r, m = wk.c;
                const T* row = tmp + y * w;
                T e, o;
                synth_low_pair<T, FT>(m, w, a.mode_inv, a.rlo, [&](int i) { return row[i]; }, e, o);
                T* d = dst + y * Wo + 2 * m;
                d[0] = fma_t(gp, d[0], e);
                if (2 * m + 1 < Wo) d[1] = fma_t(gp, d[1], o);
            }
            __syncthreads();
        }
        // ---------------------------------------------------------------- level 1 synthesis + the elementwise tail, straight to global
        {
            const T* const B = lds + a.off_ll[1];
            const float* px = a.subtract_from_x ? xin + p * (int64_t)H * W : nullptr;
            float* po = out + p * (int64_t)H * W;
            const int wp = (W + 1) >> 1;
            const T g0 = a.g[0];
            for (int y0 = 0; y0 < H; y0 += a.rows_out) {
                const int th = min(a.rows_out, H - y0);
                for (Walk2 wk(tid, w1); wk.r < ((th + 1) >> 1); wk.next(w1)) {
                    const int mp = wk.r, xo = wk.c;
                    T e, o;
                    synth_low_pair<T, FT>((y0 >> 1) + mp, h1, a.mode_inv, a.rlo, [&](int i) { return B[i * w1 + xo]; }, e, o);
                    tmp[(2 * mp) * w1 + xo] = e;
                    if (2 * mp + 1 < th) tmp[(2 * mp + 1) * w1 + xo] = o;
                }
                __syncthreads();
                if ((W & 3) == 0 && a.vec4 && a.mode_inv != kPeriodization) {
                    // four consecutive outputs per item: one 16-byte access per tensor instead of two 8-byte ones (this phase is bound by
                    // the number of memory instructions, like every store phase on this chip), and the two output pairs share all but one of
                    // their K coefficients
                    constexpr int K = FT / 2;
                    const int wq = W >> 2;
                    for (Walk2 wk(tid, wq); wk.r < th; wk.next(wq)) {
                        const int yl = wk.r, m = 2 * wk.c;
                        const T* row = tmp + yl * w1;
                        const uint32_t at = (uint32_t)((y0 + yl) * W + 4 * wk.c);
#ifdef SONAR_LOW_NOREREAD  // profiling builds: what the second read of cond / uncond costs
                        float4 c4 = make_float4(1.0f, 1.0f, 1.0f, 1.0f), u4 = make_float4(2.0f, 2.0f, 2.0f, 2.0f);
#else
                        float4 c4 = at_u32<float4>(pc, at), u4 = at_u32<float4>(pu, at);
#endif
                        float4 x4 = px ? at_u32<float4>(px, at) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        T cf[K + 1];  // coefficients m .. m + K (clamped as synth_low_pair clamps: beyond the valid length only)
#pragma unroll
                        for (int i = 0; i <= K; ++i) cf[i] = row[min(m + i, w1 - 1)];
                        T e0 = T(0), o0 = T(0), e1 = T(0), o1 = T(0);
#pragma unroll
                        for (int k = K - 1; k >= 0; --k) {  // pair m reads index m + K - 1 - k, pair m + 1 the next one
                            e0 = fma_t(cf[K - 1 - k], a.rlo[2 * k], e0);
                            o0 = fma_t(cf[K - 1 - k], a.rlo[2 * k + 1], o0);
                            e1 = fma_t(cf[K - k], a.rlo[2 * k], e1);
                            o1 = fma_t(cf[K - k], a.rlo[2 * k + 1], o1);
                        }
                        const T r0 = fma_t(a.ku, (T)u4.x, a.kt * fma_t(g0, (T)c4.x - (T)u4.x, e0));
                        const T r1 = fma_t(a.ku, (T)u4.y, a.kt * fma_t(g0, (T)c4.y - (T)u4.y, o0));
                        const T r2 = fma_t(a.ku, (T)u4.z, a.kt * fma_t(g0, (T)c4.z - (T)u4.z, e1));
                        const T r3 = fma_t(a.ku, (T)u4.w, a.kt * fma_t(g0, (T)c4.w - (T)u4.w, o1));
                        float4 res = make_float4((float)r0, (float)r1, (float)r2, (float)r3);
                        if (px) res = make_float4(x4.x - res.x, x4.y - res.y, x4.z - res.z, x4.w - res.w);
                        store4<(SONAR_LOWPASS_NT != 0)>(&at_u32<float>(po, at), res.x, res.y, res.z, res.w);
                    }
                } else
                for (Walk2 wk(tid, wp); wk.r < th; wk.next(wp)) {
                    const int yl = wk.r, m = wk.c;
                    const T* row = tmp + yl * w1;
                    const int at = (y0 + yl) * W + 2 * m;
                    const bool pair = 2 * m + 1 < W;
                    const bool vec = pair && (W & 1) == 0 && a.vec2;  // rows are 8-byte aligned: one vector access per tensor
                    // the three global reads are requested BEFORE the synthesis out of LDS, not after it: the compiler keeps source order
                    // here and otherwise waits for them with nothing left to overlap (-2..3 % on the kernel; requesting them a whole
                    // item ahead measured the same for fp32 and slower for fp64)
                    float2 c2 = make_float2(0.0f, 0.0f), u2 = c2, x2 = c2;
                    if (vec) {
#ifdef SONAR_LOW_NOREREAD  // profiling builds: what the second read of cond / uncond costs
                        c2 = make_float2(1.0f, 1.0f);
                        u2 = make_float2(2.0f, 2.0f);
#else
                        c2 = *reinterpret_cast<const float2*>(pc + at);
                        u2 = *reinterpret_cast<const float2*>(pu + at);
#endif
                        if (px) x2 = *reinterpret_cast<const float2*>(px + at);
                    }
                    T e, o;
                    synth_low_pair<T, FT>(m, w1, a.mode_inv, a.rlo, [&](int i) { return row[i]; }, e, o);
                    if (vec) {
                        const T r0 = fma_t(a.ku, (T)u2.x, a.kt * fma_t(g0, (T)c2.x - (T)u2.x, e));
                        const T r1 = fma_t(a.ku, (T)u2.y, a.kt * fma_t(g0, (T)c2.y - (T)u2.y, o));
                        float2 res = make_float2((float)r0, (float)r1);
                        if (px) res = make_float2(x2.x - res.x, x2.y - res.y);
                        *reinterpret_cast<float2*>(po + at) = res;
                    } else {
                        const T r0 = fma_t(a.ku, (T)pu[at], a.kt * fma_t(g0, (T)pc[at] - (T)pu[at], e));
                        po[at] = px ? px[at] - (float)r0 : (float)r0;
                        if (pair) {
                            const T r1 = fma_t(a.ku, (T)pu[at + 1], a.kt * fma_t(g0, (T)pc[at + 1] - (T)pu[at + 1], o));
                            po[at + 1] = px ? px[at + 1] - (float)r1 : (float)r1;
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
}

// LDS plan; false when the plane's LL pyramid does not fit one workgroup's share (the caller takes the band-by-band path)
template <typename T>
static bool lowpass_plan(LowArgs<T>& a, size_t& lds_bytes, int64_t H, int64_t W, int levels, int flen, int mode_fwd, int mode_inv) {
    if (levels < 1 || levels > kLowMaxLevels || !tile_taps_ok(flen) || flen > kDeepTaps || !dims_ok(H, W) || H > 4096 || W > 4096) return false;
    // a periodised transform paired with any other extension reconstructs a plane shifted by flen / 2 - 1 samples: not the identity
    // this path is built on (py/wavelet_functions.py:81-105 runs both transforms for real; the band-by-band path does the same)
    if (flen > 2 && (mode_fwd == kPeriodization) != (mode_inv == kPeriodization)) return false;
    a.levels = levels;
    a.H[0] = (int)H;
    a.W[0] = (int)W;
    int at = 0;
    int tmp = kLowRows * 2 * (((int)W + 1) / 2);  // level-1 analysis tile (parity-split rows)
    int ints = 0;
    for (int j = 1; j <= levels; ++j) {
        a.H[j] = (int)dwt_len(a.H[j - 1], flen, mode_fwd);
        a.W[j] = (int)dwt_len(a.W[j - 1], flen, mode_fwd);
        const int Hr = mode_inv == kPeriodization ? 2 * a.H[j] : 2 * a.H[j] - flen + 2;
        const int Wr = mode_inv == kPeriodization ? 2 * a.W[j] : 2 * a.W[j] - flen + 2;
        if (Hr < a.H[j - 1] || Wr < a.W[j - 1]) return false;  // the inverse cannot cover the level below
        a.off_ll[j] = at;
        at += a.H[j] * a.W[j];
        if (j >= 2) tmp = std::max(tmp, std::max(a.H[j] * a.W[j - 1], (a.H[j - 1] + 1) * a.W[j]));
        a.map_h[j] = ints;
        ints += 2 * a.H[j] + flen;
        a.map_w[j] = ints;
        ints += 2 * a.W[j] + flen;
    }
    tmp = std::max(tmp, (kLowRows + 1) * a.W[1]);  // final synthesis tile
    a.off_tmp = at;
    at += tmp;
    // level 1 runs before the deeper LL planes exist: its scratch may lie over them -> taller tiles, fewer barrier-separated phases
    a.off_tmp1 = levels >= 2 ? a.off_ll[2] : a.off_tmp;
    const int ws1 = 2 * (((int)W + 1) / 2);
    a.rows1 = std::max(kLowRows, std::min((a.H[1] + 3) / 4 * 4, (at - a.off_tmp1) / ws1 / 4 * 4));
    a.rows_out = std::max(kLowRows, std::min(((int)H + 1) / 2 * 2, (tmp / a.W[1] - 1) / 2 * 2));
    // ... and of the heights the scratch holds, the one whose phases waste the fewest rounds of the workgroup's threads (round 5, as in
    // dwt_bands.h's plan): the synthesis along H has (rows / 2) x W1 items, the pass along W rows x W / 4 -- 36 rows of a 128 x 128 plane
    // (db4) are 1206 and 1152 items for 512 threads, three rounds each; 30 rows are two (fp32 85.3 -> 83.3 us, fp64 92.5 -> 89.7 us)
    static const int forced_rows = [] { const char* e = getenv("SONAR_LOW_ROWS_OUT"); return e ? atoi(e) : 0; }();  // (experiments: at most the scratch's)
    if (forced_rows > 0) {
        a.rows_out = std::min(a.rows_out, std::max(2, forced_rows / 2 * 2));
    } else {
        int best = a.rows_out;
        long best_cost = -1;
        for (int r = a.rows_out; r >= kLowRows; r -= 2) {
            const long tiles = ((int)H + r - 1) / r, items2 = (long)(r / 2) * a.W[1], items3 = (long)r * (((int)W + 3) / 4);
            const long cost = tiles * ((items2 + kLowThreads - 1) / kLowThreads + (items3 + kLowThreads - 1) / kLowThreads + 1);
            if (best_cost < 0 || cost < best_cost) {
                best_cost = cost;
                best = r;
            }
        }
        a.rows_out = best;
    }
    a.off_maps = (int)(((size_t)at * sizeof(T) + 15) / 16 * 16);
    lds_bytes = (size_t)a.off_maps + (size_t)ints * sizeof(int);
    return lds_bytes <= 80 * 1024;  // two workgroups per CU
}

template <typename T>
static int wcfg_lowpass(const float* cond, const float* uncond, const float* x, float* out, int64_t planes, int64_t H, int64_t W, int levels,
                        const double* dec_lo, const double* rec_lo, int flen, int mode_fwd, int mode_inv, const double* g, double ku, double kt,
                        int subtract_from_x, hipStream_t st, const char* what) {
    SONAR_REQUIRE(cond && uncond && out && (x || !subtract_from_x) && dec_lo && rec_lo && g && planes >= 0 && mode_fwd >= 0 && mode_fwd <= 5 &&
                      mode_inv >= 0 && mode_inv <= 5,
                  SONAR_ERR_ARG, "%s: bad argument", what);
    LowArgs<T> a{};
    size_t lds = 0;
    SONAR_REQUIRE(lowpass_plan(a, lds, H, W, levels, flen, mode_fwd, mode_inv), SONAR_ERR_UNSUPPORTED,
                  "%s: the plane's low-pass pyramid does not fit in LDS (or unsupported filter length / level count)", what);
    if (planes == 0) return SONAR_OK;
    a.planes = planes;
    for (int j = 0; j <= levels; ++j) a.g[j] = (T)g[j];
    a.ku = (T)ku;
    a.kt = (T)kt;
    a.subtract_from_x = subtract_from_x;
    a.mode_fwd = mode_fwd;
    a.mode_inv = mode_inv;
    a.vec2 = ((reinterpret_cast<uintptr_t>(cond) | reinterpret_cast<uintptr_t>(uncond) | reinterpret_cast<uintptr_t>(x) |
               reinterpret_cast<uintptr_t>(out)) & 7u) == 0;
    a.vec4 = ((reinterpret_cast<uintptr_t>(cond) | reinterpret_cast<uintptr_t>(uncond) | reinterpret_cast<uintptr_t>(x) |
               reinterpret_cast<uintptr_t>(out)) & 15u) == 0;  // a view at an odd storage offset, or any direct caller of the C ABI
    for (int i = 0; i < kDeepTaps; ++i) {
        a.dlo[i] = i < flen ? (T)dec_lo[i] : T(0);
        a.rlo[i] = i < flen ? (T)rec_lo[i] : T(0);
    }
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / (lds + 512)));
    const int grid = (int)std::min<int64_t>(planes, (int64_t)256 * per_cu);
    with_taps(flen, [&](auto ft) {
        constexpr int FT = decltype(ft)::value;
        auto go = [&](auto zero) {
            auto kern = wcfg_lowpass_kernel<T, FT, decltype(zero)::value>;
            if (lds > 64 * 1024) lds_attr(reinterpret_cast<const void*>(kern), 80 * 1024);  // dynamic LDS above the 64 KB default: once per kernel and device
            hipLaunchKernelGGL(kern, dim3(grid), dim3(kLowThreads), lds, st, cond, uncond, x, out, a);
        };
        if (mode_fwd == kZero) go(std::true_type{}); else go(std::false_type{});
    });
    return check_launch(what);
}

}  // namespace sonar
