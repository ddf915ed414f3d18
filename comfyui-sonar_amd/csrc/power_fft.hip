// Power-law (coloured) rFFT noise: out = irfft2(z * filter, s=(H,W), norm="ortho") per plane,
// py/nodes/powernoise.py:366-377, with z either supplied (replay) or drawn on device (Philox).
//
// One 256-thread workgroup owns one H x (W/2+1) half-spectrum in LDS at a time (persistent loop
// over planes).  The plane never touches HBM between the draw and the final real output:
//   fill     z*filter -> LDS A[ky][kx] (kx < M = W/2); the kx = M column goes to a side buffer
//   fix-up   columns 0 and M only contribute their REAL part after the column transform (c2r
//            drops Im of DC/Nyquist), so both are Hermitian-symmetrised and packed into one complex
//            column:  Q = sym(Z[:,0]) + i*sym(Z[:,M])  ->  Re/Im of its transform are the two columns
//   columns  length-H inverse DFT per column, four-step N = N1*N2, in place in LDS
//            (lanes = consecutive columns -> row-contiguous, conflict-free ds_read/ds_write_b64)
//   rows     c2r of length W via one length-M complex inverse DFT of
//            G[k] = (X[k] + conj X[M-k]) + i (X[k] - conj X[M-k]) e^{2 pi i k / W}
//            (lanes = consecutive rows, odd LDS row stride -> conflict-free), second pass stores
//            straight to global as float2 (x[2m], x[2m+1]) in 64-B runs.
// The normaliser's (sum, sumsq) partials are accumulated from the stored values.
#include "power_core.h"

namespace sonar {

// Waves per SIMD the register allocator has to leave room for (the second launch bound).  Four (128 registers) where two workgroups
// share a CU; the 256-row / 256-column strips and the 64 x 32 plane take what their general passes need instead of spilling (round 5:
// NO kernel of the library may use scratch memory -- a private segment costs the process a 50-90 ms queue stall; these
// shapes spilled 2-50 registers at 128): two waves per SIMD, one workgroup per CU.
template <int H, int W>
constexpr int plane_min_waves() {
    return (H * W >= 32768 || (H == 256 && W == 64) || (H == 64 && W == 256) || (H == 64 && W == 32)) ? 2 : SONAR_FFT_WAVES;
}

// SRC: 0 = spectrum `z` supplied (replay), 1 = spectrum drawn on device, 2 = `z` is a REAL H x W plane: forward r2c FFT in
// LDS, x filter, then the same inverse (spectral filter: out = irfft2(rfft2(x) * filter), py/nodes/powernoise.py:356-366)
template <int H, int W, int SRC, bool STATS, bool NORM>
__global__ void __launch_bounds__((plane_threads<H, W>()), (plane_min_waves<H, W>())) power_irfft2_kernel(const float* __restrict__ z,
                                                                       const float* __restrict__ filter, float* out,
                                                                       int64_t planes, uint64_t seed, uint64_t stream_id,
                                                                       int64_t plane_offset, int group, int split, double* partials,
                                                                       NormArgs na, StatsAhead sa) {
    kernarg_touch_for(z, filter, out, planes, seed, stream_id, plane_offset, group, split, partials, na, sa);
    using C = PlaneCfg<H, W>;
    constexpr int NT = plane_threads<H, W>();
    constexpr int M = C::M, S = C::S;
    constexpr int RN1 = C::RN1, RN2 = C::RN2;
    // FAST shapes (W = 128, H = 64 / 128, 8 waves): every wave owns ONE residue n2 in both twiddled passes, so all
    // twiddles are wave-uniform AND loop-invariant -> loaded once into scalar registers before the plane loop
    // (no s_load / lgkmcnt(0) stall inside the passes); the column split is H = (H/8) x 8 instead of 8 x (H/8).
    constexpr bool FAST = (W == 128) && (H == 128 || H == 64) && (NT == 512) && !SONAR_FFT_TW_LDS;
    constexpr int CN1 = FAST ? H / 8 : C::CN1, CN2 = FAST ? 8 : C::CN2;
    // (Measured dead end, round 3: fusing the spectral filter's last forward column pass, the filter and the inverse's first column pass
    // in registers -- the same 16 rows of a column per thread -- spills at the 128-register cap: 107 us instead of 94 per 512 latents.)
    __shared__ c32 A[C::kLdsComplex];
    c32* const T0 = A + H * S;      // raw column kx = 0
    c32* const TM = T0 + H;         // raw column kx = M
    c32* const TW = TM + H;         // e^{2 pi i j / 256}
    __shared__ double red[2 * NT / 64];
    __shared__ NormDecision shd;
    __shared__ int edge_seq;  // FAST generate path: edge-column waves drawn so far (two per plane), see the fill
    constexpr bool kSumsInLds = SRC == 2 && STATS && FAST && H == 128;
    __shared__ double sums_lds[kSumsInLds ? 2 * NT : 1];
    const int tid = threadIdx.x;
    if constexpr (kSumsInLds) {
        sums_lds[tid] = 0.0;
        sums_lds[NT + tid] = 0.0;
    }
    int64_t bid = blockIdx.x, nblk = gridDim.x;
    if constexpr (SRC == 1 && NORM && !STATS) {
        static_assert(kStatsBatch * 2 * H <= C::kLdsComplex, "the statistics' edge columns borrow the plane's LDS");
        if (sa.partials) {
            if ((int)blockIdx.x >= sa.main_blocks) {
                power_stats_body<H, W>(filter, planes, seed, sa.stream_id, plane_offset, group, split, sa.partials, bid - sa.main_blocks,
                                       nblk - sa.main_blocks, reinterpret_cast<c32(*)[2][H]>(A), red);
                return;
            }
            nblk = sa.main_blocks;
        }
    }
    constexpr bool GEN = SRC == 1;
    // norm="ortho" on the inverse; the spectral filter also carries the forward transform's 1/sqrt(HW)
    float scale = SRC == 2 ? 1.0f / ((float)H * (float)W) : 1.0f / sqrtf((float)H * (float)W);
    // normalised output = (v * scale - mean) / std * factor folded into one multiply-add per value: v * nm - nc
    float nm = scale, nc = 0.0f;
    // The normalisation decision needs the statistics pass's partials: they are requested here and first USED right before the first
    // plane's stores, so their latency (every workgroup reads all kNPart pairs) hides behind that plane's draw and transforms.
    constexpr int NPRE = (NORM && kNPart % NT == 0) ? kNPart / NT : 0;
    [[maybe_unused]] double pre_s[NPRE > 0 ? NPRE : 1], pre_q[NPRE > 0 ? NPRE : 1];
    [[maybe_unused]] bool norm_pending = NORM;
    if constexpr (NORM) {
        if constexpr (NPRE > 0) {
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                pre_s[i] = na.partials[2 * (tid + i * NT)];
                pre_q[i] = na.partials[2 * (tid + i * NT) + 1];
            }
        } else {
            const NormDecision dec = decide_norm<NT>(na.partials, kNPart, na.n_total, na.thr_sd, red, &shd);
            const float g = (dec.do_div ? 1.0f / dec.stdv : 1.0f) * na.factor;
            nm = scale * g;
            nc = dec.do_sub ? dec.mean * g : 0.0f;
            norm_pending = false;
        }
    }
    double s = 0.0, q = 0.0;
    for (int j = tid; j < 256; j += NT) TW[j] = c_tw256[j];
#if SONAR_FFT_TW_LDS
    auto tw = [&](int idx, int n, bool) -> c32 { return TW[(idx * (256 / n)) & 255]; };
#else
    // `uni`: the index is wave-uniform (lanes = consecutive columns / rows of one n2) -> scalar load
    // (a per-lane index reads the LDS copy: vector loads from constant memory cost 64-bit address registers)
    // (__device__ on the lambdas that name c_tw256: an unmarked lambda is host-callable as far as the compiler knows, a static table it names
    // becomes an external symbol, and every kernel of this file then fetched the table's ADDRESS through the global offset table -- one more
    // scalar-cache miss in front of the first twiddle, 576 sites; spectral filter 62.4 -> 61.8 us, 13.05 -> 12.7 us at batch 64)
    auto tw = [&] __device__(int idx, int n, bool uni) -> c32 {
        if (uni) return c_tw256[(__builtin_amdgcn_readfirstlane(idx) * (256 / n)) & 255];
        return TW[(idx * (256 / n)) & 255];
    };
#endif
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    c32 ctw[CN1], gtw[RN1], ptw[RN1];
    auto load_twiddles = [&] __device__(int w) {
#pragma unroll
        for (int k1 = 0; k1 < CN1; ++k1) ctw[k1] = c_tw256[((w * k1) * (256 / H)) & 255];
#pragma unroll
        for (int n1 = 0; n1 < RN1; ++n1) gtw[n1] = c_tw256[((RN2 * n1 + w) * (256 / W)) & 255];
#pragma unroll
        for (int k1 = 0; k1 < RN1; ++k1) ptw[k1] = c_tw256[((w * k1) * (256 / M)) & 255];
    };
    if constexpr (FAST) load_twiddles(wv);

#ifdef SONAR_PW_DESYNC  // profiling builds: the second resident workgroup of every CU starts late (out of phase with the first)
    if (blockIdx.x >= gridDim.x / 2)
        for (int i = 0; i < SONAR_PW_DESYNC; ++i) __builtin_amdgcn_s_sleep(32);
#endif
    [[maybe_unused]] int pidx = 0;  // planes this workgroup has started (trace builds)
    [[maybe_unused]] int edge_want = 0;
    if (tid == 0) edge_seq = 0;  // visible after the first plane's top-of-loop barrier
    // one workgroup draws the `group` planes of an RNG group back to back (group = 1 unless generating)
    for (int64_t unit = bid; unit < (split ? planes : planes / group); unit += nblk) {
    const GroupWalk gw(unit, group, split);
    SpectrumRng rng;
    if constexpr (GEN) {
        rng = spectrum_rng<H, true>(seed, stream_id, plane_offset / group + gw.grp, tid);
        for (int i = 0; i < gw.first; ++i) skip_plane<H, W, true>(rng, tid);
    }
    for (int gp = gw.first; gp < gw.first + gw.count; ++gp) {
        const int64_t plane = gw.grp * group + gp;
        __syncthreads();  // previous plane's LDS reads are done (and TW is visible)
        SONAR_STAMP(0);
        // The general passes index by item = thread + k * NT: every LDS address of a plane is loop-invariant, the optimiser hoists them all
        // out of the plane loop (the 64 x 64 kernel then wants 184 registers, spills 47 at its 128 and reloads them plane after plane).
        // An opaque copy of the thread index per plane keeps the address arithmetic -- a few integer operations -- inside the loop
        // (64 x 64: 106 -> 78 us per 33.5 M values, 256 x 64: 185 -> 91; the planes below 4096 values fit their budget as they are, or lose by it: 64 x 32 89 -> 98 us).
        int ptid = tid;
        if constexpr (!FAST && H * W >= 4096) asm volatile("" : "+v"(ptid));
        if constexpr (SRC < 2) {
        // ---------------------------------------------------------------- fill: z * filter
        auto sink = [&](int ky, int kx, c32 v) { (kx == 0 ? T0[ky] : kx == M ? TM[ky] : A[ky * S + kx]) = v; };
        // the draw is one long vector-ALU stream, the transform passes are short bursts between LDS round trips and barriers: with the
        // passes at a higher issue priority the co-resident workgroup's draw fills their gaps instead of delaying them (-2.7 us per launch)
        __builtin_amdgcn_s_setprio(0);
        if constexpr (GEN) {
            if constexpr (!(SONAR_PW_SKIP & 1)) fill_plane_gen<H, W, S>(filter, rng, tid, A, T0, TM, (FAST && H == 2 * 64) ? &edge_seq : nullptr);
            if constexpr (FAST && H == 2 * 64 && !(SONAR_PW_SKIP & 1)) {
                // The edge columns were drawn first, by waves 0-1, which then announced themselves in edge_seq; the LAST two waves build
                // the packed column 0 from them at the end of their own draws (Q[ky] = sym(Z0)[ky] + i sym(ZM)[ky]): no separate
                // fix-up phase and barrier, and the edge work is not stacked on the waves that already drew it.
                edge_want += 2;
                if (tid >= NT - H) {
                    while (__hip_atomic_load(&edge_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < edge_want) __builtin_amdgcn_s_sleep(1);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    const int ky = tid - (NT - H), kn = (H - ky) & (H - 1);
                    const c32 a = T0[ky], an = T0[kn], b = TM[ky], bn = TM[kn];
                    A[ky * S] = make_float2(0.5f * (a.x + an.x) - 0.5f * (b.y - bn.y), 0.5f * (a.y - an.y) + 0.5f * (b.x + bn.x));
                }
            }
        } else fill_plane<H, W>(z, filter, plane, tid, sink);
        __builtin_amdgcn_s_setprio(3);
        SONAR_STAMP(1);
        __syncthreads();
        SONAR_STAMP(2);
        } else {
        // ---------------------------------------------------------------- forward r2c of a real plane (mirror of the inverse:
        // the same four-step splits run backwards, so the spectrum lands in natural (ky, kx) order where the inverse reads it)
        constexpr int Wh = M + 1;
        const float* const xin = z + plane * (int64_t)H * W;
        // rows, pass b': spatial row y sits in LDS row r with y = r / CN2 + CN1 * (r % CN2) (what the column passes expect);
        // complex element m = k1 + RN1 * k2 is (x[2m], x[2m+1]); DFT over k2 -> n2, twiddle
        // two items per trip: the second item's eight row loads are in flight while the first is transformed
        // (two adjacent k1 per item with 16-byte loads, half the memory instructions: 89 us instead of 85 per 512 latents -- the second set
        // of eight values costs the pass its overlap)
#pragma unroll 2
        for (int item = ptid; item < RN1 * H; item += NT) {
            const int k1 = item % RN1, r = item / RN1;
            const int y = (r / CN2) + CN1 * (r % CN2);
            const float* xrow = xin + (int64_t)y * W;
            c32 u[RN2];
#pragma unroll
            for (int k2 = 0; k2 < RN2; ++k2) u[k2] = *reinterpret_cast<const float2*>(xrow + 2 * (k1 + RN1 * k2));
            fdft<RN2>(u);
#pragma unroll
            for (int n2 = 1; n2 < RN2; ++n2) u[n2] = cmulc(u[n2], tw(n2 * k1, M, false));
#pragma unroll
            for (int n2 = 0; n2 < RN2; ++n2) A[r * S + C::rpos(k1, n2)] = u[n2];
        }
        __syncthreads();
        // rows, pass a': DFT over k1 -> n1: C[k = RN2 n1 + n2]; reads the swizzled columns, writes the natural ones (what the split
        // and the column passes index), so every read of the pass is done before the first write
        {
            constexpr int ITEMS = (RN2 * H + NT - 1) / NT;
            c32 v[ITEMS][RN1];
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int item = ptid + it * NT;
                if (item < RN2 * H) {
                    const int r = item % H, n2 = item / H;
#pragma unroll
                    for (int k1 = 0; k1 < RN1; ++k1) v[it][k1] = A[r * S + C::rpos(k1, n2)];
                    fdft<RN1>(v[it]);
                }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int item = ptid + it * NT;
                if (item < RN2 * H) {
                    const int r = item % H, n2 = item / H;
#pragma unroll
                    for (int n1 = 0; n1 < RN1; ++n1) A[r * S + RN2 * n1 + n2] = v[it][n1];
                }
            }
        }
        __syncthreads();
        // r2c split: X[k] = E + w^k O, X[M-k] = conj(E - w^k O), E = (C[k] + conj C[M-k]) / 2, O = (C[k] - conj C[M-k]) / 2i,
        // w = e^{-2 pi i / W}; the two real columns X[0], X[M] are packed into column 0 as X[0] + i X[M]
        for (int item = ptid; item < (M / 2 + 1) * H; item += NT) {
            const int r = item % H, k = item / H;
            c32* row = A + r * S;
            if (k == 0) {
                const c32 c0 = row[0];
                row[0] = make_float2(c0.x + c0.y, c0.x - c0.y);
            } else if (k == M / 2) {
                const c32 c = row[k];
                row[k] = make_float2(c.x, -c.y);
            } else {
                const c32 a = row[k], b = row[M - k];
                const c32 e = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
                const c32 o = make_float2(0.5f * (a.y + b.y), -0.5f * (a.x - b.x));
                const c32 t = cmulc(o, tw(k, W, SONAR_FWD_UNI && H % 64 == 0));
                row[k] = make_float2(e.x + t.x, e.y + t.y);
                row[M - k] = make_float2(e.x - t.x, -(e.y - t.y));
            }
        }
        __syncthreads();
        // columns, pass b': DFT over k2 -> n2 for fixed k1 (rows CN2 k1 + .), twiddle
        for (int item = ptid; item < CN1 * M; item += NT) {
            const int c = item % M, k1 = item / M;
            c32 u[CN2];
#pragma unroll
            for (int k2 = 0; k2 < CN2; ++k2) u[k2] = A[(CN2 * k1 + k2) * S + c];
            fdft<CN2>(u);
#pragma unroll
            for (int n2 = 1; n2 < CN2; ++n2) u[n2] = cmulc(u[n2], tw(n2 * k1, H, SONAR_FWD_UNI && M % 64 == 0));
#pragma unroll
            for (int n2 = 0; n2 < CN2; ++n2) A[(CN2 * k1 + n2) * S + c] = u[n2];
        }
        __syncthreads();
        // columns, pass a': DFT over k1 -> n1: Z[ky = CN2 n1 + n2][c], in place
        for (int item = ptid; item < CN2 * M; item += NT) {
            const int c = item % M, n2 = item / M;
            c32 v[CN1];
#pragma unroll
            for (int k1 = 0; k1 < CN1; ++k1) v[k1] = A[(CN2 * k1 + n2) * S + c];
            fdft<CN1>(v);
#pragma unroll
            for (int n1 = 0; n1 < CN1; ++n1) A[(CN2 * n1 + n2) * S + c] = v[n1];
        }
        __syncthreads();
        // unpack column 0 (P = DFT(X0 + i XM): Z0 = (P[ky] + conj P[-ky]) / 2, ZM = (P[ky] - conj P[-ky]) / 2i), x filter
        for (int ky = ptid; ky < H; ky += NT) {
            const int kn = (H - ky) & (H - 1);
            const c32 p = A[ky * S], pn = A[kn * S];
            const float f0 = SRC == 3 ? 1.0f : filter[ky * Wh], fm = SRC == 3 ? 1.0f : filter[ky * Wh + M];
            T0[ky] = make_float2(0.5f * (p.x + pn.x) * f0, 0.5f * (p.y - pn.y) * f0);
            TM[ky] = make_float2(0.5f * (p.y + pn.y) * fm, -0.5f * (p.x - pn.x) * fm);
        }
        if constexpr (SRC != 3 && !(FAST && SONAR_FILTER_IN_PASS)) {
        // unrolled: the filter values are global loads (L2 hits) -- eight in flight instead of a wait per element
#pragma unroll 8
        for (int j = ptid; j < H * M; j += NT) {
            const int ky = j / M, c = j - ky * M;
            if (c != 0) {
                const float f = filter[ky * Wh + c];
                c32 v = A[ky * S + c];
                v.x *= f; v.y *= f;
                A[ky * S + c] = v;
            }
        }
        }
        __syncthreads();
        if constexpr (SRC == 3) {
            // forward only: the unscaled half-spectrum rfft2(x)[ky][kx], kx = 0 .. W/2, to global (complex64) and on to the next plane
            c32* const zp = reinterpret_cast<c32*>(out) + plane * (int64_t)H * Wh;
            for (int j = ptid; j < H * Wh; j += NT) {
                const int ky = j / Wh, kx = j - ky * Wh;
                zp[j] = kx == 0 ? T0[ky] : kx == M ? TM[ky] : A[ky * S + kx];
            }
            continue;
        }
        }
        if constexpr (FAST) {
            if constexpr (!(GEN && H == 2 * 64) || (SONAR_PW_SKIP & 1)) {
                // fix-up: Q[ky] = sym(Z0)[ky] + i sym(ZM)[ky] -> column 0 of the plane (the 128-row generate path did it inside the fill)
                if (tid < H) {
                    const int ky = tid, kn = (H - ky) & (H - 1);
                    const c32 a = T0[ky], an = T0[kn], b = TM[ky], bn = TM[kn];
                    A[ky * S] = make_float2(0.5f * (a.x + an.x) - 0.5f * (b.y - bn.y), 0.5f * (a.y - an.y) + 0.5f * (b.x + bn.x));
                }
                __syncthreads();
            }
            SONAR_STAMP(3);
            // ------------------------------------------------------------ columns, pass a: radix CN1, one item per thread
            if constexpr (!(SONAR_PW_SKIP & 2)) {
                const int c = lane, n2 = wv;
                c32 v[CN1];
                if constexpr (SRC == 2 && SONAR_FILTER_IN_PASS) {
                    // the spectral filter's multiply rides on this pass's loads instead of being a pass over the plane of its own (column 0
                    // is the packed edge pair, filtered when it was unpacked).  The filter values are the same for every plane: an opaque
                    // lane index keeps their loads (L1 hits) in the plane loop -- hoisted they would hold CN1 registers the passes need.
                    int fl = lane;
                    asm volatile("" : "+v"(fl));
                    float f[CN1];
#pragma unroll
                    for (int n1 = 0; n1 < CN1; ++n1) f[n1] = filter[(CN2 * n1 + n2) * (M + 1) + fl];
#pragma unroll
                    for (int n1 = 0; n1 < CN1; ++n1) v[n1] = A[(CN2 * n1 + n2) * S + c];
#pragma unroll
                    for (int n1 = 0; n1 < CN1; ++n1) v[n1] = cscale(v[n1], c == 0 ? 1.0f : f[n1]);
                } else {
#pragma unroll
                for (int n1 = 0; n1 < CN1; ++n1) v[n1] = A[(CN2 * n1 + n2) * S + c];
                }
                idft<CN1>(v);
#pragma unroll
                for (int k1 = 1; k1 < CN1; ++k1) v[k1] = cmul(v[k1], ctw[k1]);
#pragma unroll
                for (int k1 = 0; k1 < CN1; ++k1) A[(CN2 * k1 + n2) * S + c] = v[k1];
            }
            SONAR_STAMP(4);
            __syncthreads();
            SONAR_STAMP(5);
            if constexpr (H == 128) {
            // ------------------------------------------------------------ the pipelined kernel's passes: all LDS operands requested up
            // front, row pass a leaves element (k1, n2) at column 8 n2 + k1 for the 16-byte stores of pass b
            if constexpr (!(SONAR_PW_SKIP & 2)) pipe_col_b<H, W, 8>(A, A, wv, lane);  // in place: an item reads and writes the same 8 rows of its column
            __syncthreads();
            if constexpr (!(SONAR_PW_SKIP & 4)) pipe_row_a<H, W, 8, kRowsInPlace>(A, A, wv, lane);
            __syncthreads();
            } else {
            // ------------------------------------------------------------ columns, pass b: radix 8, rows 8 k1 .. 8 k1 + 7
#pragma unroll
            for (int it = 0; it < ((SONAR_PW_SKIP & 2) ? 0 : CN1 / 8); ++it) {
                const int c = lane, k1 = wv + 8 * it;
                c32 u[CN2];
#pragma unroll
                for (int n2 = 0; n2 < CN2; ++n2) u[n2] = A[(CN2 * k1 + n2) * S + c];
                idft<CN2>(u);
#pragma unroll
                for (int k2 = 0; k2 < CN2; ++k2) A[(CN2 * k1 + k2) * S + c] = u[k2];
            }
            SONAR_STAMP(6);
            __syncthreads();
            SONAR_STAMP(7);
            // ------------------------------------------------------------ rows, pass a: n2 = wave, rows lane + 64 it
            if constexpr (!(SONAR_PW_SKIP & 4)) {
                constexpr int ITEMS = H / 64;
                const int n2 = wv;
                c32 g[ITEMS][RN1];
#pragma unroll
                for (int it = 0; it < ITEMS; ++it) {
                    const c32* row = A + (lane + 64 * it) * S;
#pragma unroll
                    for (int n1 = 0; n1 < RN1; ++n1) {
                        const int k = RN2 * n1 + n2;
                        c32 xa, xb;
                        if (k == 0) {  // uniform: wave 0, n1 = 0
                            const c32 p = row[0];
                            xa = make_float2(p.x, 0.0f);
                            xb = make_float2(p.y, 0.0f);
                        } else {
                            xa = row[k];
                            xb = row[M - k];
                        }
                        const c32 xc = make_float2(xb.x, -xb.y);  // conj
                        g[it][n1] = cadd_i(cadd(xa, xc), cmul(csub(xa, xc), gtw[n1]));
                    }
                    idft<RN1>(g[it]);
#pragma unroll
                    for (int k1 = 1; k1 < RN1; ++k1) g[it][k1] = cmul(g[it][k1], ptw[k1]);
                }
                SONAR_STAMP(8);
                __syncthreads();  // every mirrored read is done before anyone overwrites
#pragma unroll
                for (int it = 0; it < ITEMS; ++it) {
#pragma unroll
                    for (int k1 = 0; k1 < RN1; ++k1) A[(lane + 64 * it) * S + C::rpos(k1, n2)] = g[it][k1];
                }
            }
            SONAR_STAMP(9);
            __syncthreads();
            SONAR_STAMP(10);
            }  // H != 128
        } else {
        // ---------------------------------------------------------------- columns, pass a
        // Column 0 is built on the fly from the raw kx = 0 / kx = M columns:
        //   Q[ky] = sym(Z0)[ky] + i sym(ZM)[ky],  sym(Z)[ky] = (Z[ky] + conj Z[-ky]) / 2
SONAR_UNROLL_ITEMS
        for (int item = ptid; item < CN2 * M; item += NT) {
            const int c = item % M;
            const int n2 = item / M;
            c32 v[CN1];
            if (c != 0) {
#pragma unroll
                for (int n1 = 0; n1 < CN1; ++n1) v[n1] = A[(CN2 * n1 + n2) * S + c];
            } else {
#pragma unroll
                for (int n1 = 0; n1 < CN1; ++n1) {
                    const int ky = CN2 * n1 + n2, kn = (H - ky) & (H - 1);
                    const c32 a = T0[ky], an = T0[kn], b = TM[ky], bn = TM[kn];
                    v[n1] = make_float2(0.5f * (a.x + an.x) - 0.5f * (b.y - bn.y), 0.5f * (a.y - an.y) + 0.5f * (b.x + bn.x));
                }
            }
            idft<CN1>(v);
#pragma unroll
            for (int k1 = 1; k1 < CN1; ++k1) v[k1] = cmul(v[k1], tw(n2 * k1, H, M % 64 == 0));
#pragma unroll
            for (int k1 = 0; k1 < CN1; ++k1) A[(CN2 * k1 + n2) * S + c] = v[k1];
        }
        __syncthreads();
        // ---------------------------------------------------------------- columns, pass b
        // LDS row r = CN2*k1 + k2 afterwards holds spatial row y = k1 + CN1*k2
SONAR_UNROLL_ITEMS
        for (int item = ptid; item < CN1 * M; item += NT) {
            const int c = item % M, k1 = item / M;
            c32 u[CN2];
#pragma unroll
            for (int n2 = 0; n2 < CN2; ++n2) u[n2] = A[(CN2 * k1 + n2) * S + c];
            idft<CN2>(u);
#pragma unroll
            for (int k2 = 0; k2 < CN2; ++k2) A[(CN2 * k1 + k2) * S + c] = u[k2];
        }
        __syncthreads();
        // ---------------------------------------------------------------- rows, pass a (c2r pre-twiddle fused)
        {
            constexpr int ITEMS = (RN2 * H + NT - 1) / NT;
            c32 g[ITEMS][RN1];
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int item = ptid + it * NT;
                if (item < RN2 * H) {
                    const int r = item % H;
                    const int n2 = item / H;
                    const c32* row = A + r * S;
#pragma unroll
                    for (int n1 = 0; n1 < RN1; ++n1) {
                        const int k = RN2 * n1 + n2;
                        c32 xa, xb;
                        if (k == 0) {
                            const c32 p = row[0];
                            xa = make_float2(p.x, 0.0f);
                            xb = make_float2(p.y, 0.0f);
                        } else {
                            xa = row[k];
                            xb = row[M - k];
                        }
                        const c32 e = make_float2(xa.x + xb.x, xa.y - xb.y);
                        const c32 d = make_float2(xa.x - xb.x, xa.y + xb.y);
                        const c32 o = cmul(d, tw(k, W, H % 64 == 0));
                        g[it][n1] = make_float2(e.x - o.y, e.y + o.x);
                    }
                    idft<RN1>(g[it]);
#pragma unroll
                    for (int k1 = 1; k1 < RN1; ++k1) g[it][k1] = cmul(g[it][k1], tw(n2 * k1, M, H % 64 == 0));
                }
                __builtin_amdgcn_sched_barrier(0);  // keep item it+1's loads from being hoisted over item it (VGPR budget)
            }
            __syncthreads();  // every mirrored read is done before anyone overwrites
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int item = ptid + it * NT;
                if (item < RN2 * H) {
                    const int r = item % H;
                    const int n2 = item / H;
#pragma unroll
                    for (int k1 = 0; k1 < RN1; ++k1) A[r * S + C::rpos(k1, n2)] = g[it][k1];
                }
            }
        }
        __syncthreads();
        }  // !FAST
        if constexpr (NPRE > 0) {
            if (norm_pending) {  // uniform: the first plane of every workgroup
                double ps0 = 0.0, pq0 = 0.0;
#pragma unroll
                for (int i = 0; i < NPRE; ++i) {
                    ps0 += pre_s[i];
                    pq0 += pre_q[i];
                }
                const NormDecision dec = decide_from_sums<NT>(ps0, pq0, na.n_total, na.thr_sd, red, &shd);
                const float g = (dec.do_div ? 1.0f / dec.stdv : 1.0f) * na.factor;
                nm = scale * g;
                nc = dec.do_sub ? dec.mean * g : 0.0f;
                norm_pending = false;
            }
        }
        // ---------------------------------------------------------------- rows, pass b -> global
        float* const oplane = out + plane * (int64_t)H * W;
        if constexpr (FAST && H == 128) {
            if constexpr (kSumsInLds) {
                // the spectral filter's statistics variant is four registers over its 128: the thread's two running fp64 sums wait in LDS
                // between planes (same additions, same order, same bits)
                double ps_ = 0.0, pq_ = 0.0;
                pipe_row_b<H, W, 8, STATS, NORM>(A, oplane, wv, lane, scale, nm, nc, ps_, pq_);
                sums_lds[tid] += ps_;
                sums_lds[NT + tid] += pq_;
            } else {
                pipe_row_b<H, W, 8, STATS, NORM>(A, oplane, wv, lane, scale, nm, nc, s, q);
            }
            SONAR_STAMP(11);
            ++pidx;
            continue;
        }
        float ps = 0.0f, pq = 0.0f;  // per-plane fp32 partials (<= 64 values per thread), folded into fp64 below
SONAR_UNROLL_ITEMS
        for (int item = ptid; item < RN1 * H; item += NT) {
            const int k1 = item % RN1, r = item / RN1;
            const int y = (r / CN2) + CN1 * (r % CN2);
            c32 u[RN2];
#pragma unroll
            for (int n2 = 0; n2 < RN2; ++n2) u[n2] = (SONAR_PW_SKIP & 8) ? make_float2((float)(item + n2), 1.0f) : A[r * S + C::rpos(k1, n2)];
            if constexpr (!(SONAR_PW_SKIP & 8)) idft<RN2>(u);
            float* orow = oplane + (int64_t)y * W;
#pragma unroll
            for (int k2 = 0; k2 < RN2; ++k2) {
                float a, b;
                if constexpr (NORM) {
                    a = __builtin_fmaf(u[k2].x, nm, -nc);
                    b = __builtin_fmaf(u[k2].y, nm, -nc);
                } else {
                    a = u[k2].x * scale;
                    b = u[k2].y * scale;
                }
                if constexpr (!(SONAR_PW_SKIP & 16)) *reinterpret_cast<float2*>(orow + 2 * (k1 + RN1 * k2)) = make_float2(a, b);
                else if (a == 123.456f && b == 654.321f) *reinterpret_cast<float2*>(orow) = make_float2(a, b);  // keeps the arithmetic alive
                if constexpr (STATS) {
                    ps += a + b;
                    pq = __builtin_fmaf(a, a, __builtin_fmaf(b, b, pq));
                }
            }
        }
        if constexpr (STATS) {
            s += (double)ps;
            q += (double)pq;
        }
        SONAR_STAMP(11);
        ++pidx;
    }
    }
    if constexpr (kSumsInLds) {
        s = sums_lds[tid];
        q = sums_lds[NT + tid];
    }
    if constexpr (STATS) write_partial<NT>(s, q, partials, red);
}

// ---- pipelined generate path ---------------------------------------------------------------------------------------------------
// power_irfft2_kernel runs draw -> columns -> rows -> store as barrier-separated phases of ONE 8-wave team; with two such workgroups
// per CU the vector ALUs idle whenever both sit in a latency-bound transform phase (profiles/r02_power_kernel.md: 27 us of issue in
// a 50 us pass, 37 % of a wave's time at barriers).  Here ONE 16-wave workgroup per CU runs two teams over two plane buffers:
//   drawing team (waves 8-15)   draws plane j -- thread slot (wave n2, lane) draws rows n2, n2 + 8, ... of ONE column, which are
//                               exactly the 16 inputs of its radix-16 item of column pass a -- keeps them in registers, transforms
//                               them and writes pass a's OUTPUT to LDS: the spectrum itself never goes through LDS
//   transforming team (0-7)     meanwhile runs column pass b, the row passes and the stores of plane j - 1
// so every SIMD holds two drawing and two transforming waves at all times and the draw fills the transform's LDS round trips.
// While the drawing team works in registers the buffer of plane j is free: the transforming team ping-pongs between the buffers
// (pass b: X -> Y, row pass a: Y -> X, no read-before-write barrier inside row pass a), and the drawing team writes plane j into Y in
// the last phase, when nothing reads Y any more.  The hardware barrier counts all 16 waves: both teams run the SAME three barriers
// per plane; the draw is cut into chunks of whole pair iterations that follow the transform phases' durations (SONAR_PIPE_CHUNKS =
// iterations in phases 1 and 2; the rest, with column pass a, in phase 3).  Thread slot = thread within the drawing team: streams,
// draw order and arithmetic -- therefore every output bit -- are those of power_irfft2_kernel<H, W, 1, ...>.
#ifndef SONAR_PIPE_NT
#define SONAR_PIPE_NT 0  // profiling builds: the row pass's 16-byte stores with the non-temporal hint (64-byte runs per four lanes: slower, common.h)
#endif
#ifndef SONAR_PIPE_CHUNKS
#define SONAR_PIPE_CHUNKS 0, 4  // round 5: 0 + 4 + 4 iterations per phase.  Re-swept at the round's end (`scratch/pipe_ab.py`, same box, us per call): 2 + 4 + 2 (the
                                // setting while the draw was being shortened) 39.6; 1 + 4 + 3 39.5; 1 + 3 + 4 38.9; **0 + 4 + 4 38.6**; 0 + 5 + 3 39.2; 0 + 3 + 5 40.0; 0 + 6 + 2 39.8; 3 + 5 + 0 (round 4) 41.5+
#endif
// look-ahead statistics (TeamStats): planes of the unit whose radius words are drawn in the first / by the end of the second of the
// three phases; A, B drawing team (while the last plane is transformed), C, D transforming team (while the first plane is drawn)
#ifndef SONAR_AHEAD_SPLIT_A
#define SONAR_AHEAD_SPLIT_A 1
#define SONAR_AHEAD_SPLIT_B 3
#endif
#ifndef SONAR_AHEAD_SPLIT_C
#define SONAR_AHEAD_SPLIT_C 2
#define SONAR_AHEAD_SPLIT_D 3
#endif
#ifndef SONAR_AHEAD_PRIO
#define SONAR_AHEAD_PRIO 0  // the transforming team's issue priority while it computes look-ahead statistics beside the first draw
#endif
// Measured and left off (round 5, gpurun_out/dc, profiles/r05_power_kernel.md): with SONAR_PIPE_DECOUPLE the teams meet at ONE workgroup
// barrier per plane -- the transforming team's two exchanges inside an iteration become a team-only rendezvous (below), the drawing team
// draws its plane in one stretch and only looks at the other team's counter before it writes -- same bits, no hang, and no faster:
// 42.3-42.5 against 41.6-42.0 us per call, 39.4 = 39.4 us for the final pass alone (spinning at low priority, longer sleeps, equal
// priorities: 42.5-43.8).  The three lock-step phases are not what bounds the kernel; its instruction streams are.
#ifndef SONAR_PIPE_DECOUPLE
#define SONAR_PIPE_DECOUPLE 0
#endif
// A team's own exchange (SONAR_PIPE_DECOUPLE): the hardware barrier counts all sixteen waves, so a team-only rendezvous is a counter in
// LDS -- every wave's lane 0 adds one when the wave's LDS operations are complete (LDS executes a wave's operations in order, and the
// waves' in arrival order: who sees the count sees what was written in front of it), then the wave polls until the count reaches
// `target`.  team_wait alone is the other team looking at that counter.
__device__ __forceinline__ void team_wait(const int* ctr, int target) {
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < target) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void team_barrier(int* ctr, int target, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    team_wait(ctr, target);
}
#ifndef SONAR_PIPE_PRIO_DRAW
#define SONAR_PIPE_PRIO_DRAW 0
#endif
#ifndef SONAR_PIPE_PRIO_FFT
#define SONAR_PIPE_PRIO_FFT 3
#endif
#ifndef SONAR_PIPE_SKIP
#define SONAR_PIPE_SKIP 0  // profiling builds only: 1 no global stores, 2 no draw arithmetic (zeros), 4 no row pass b arithmetic, 8 no row pass a, 16 no column pass b
#endif
#ifdef SONAR_PW_TRACE
__device__ unsigned long long g_pipe_trace[256 * 16 * 10 * 4];
#define SONAR_PIPE_STAMP(slot) do { if (lane == 0 && j < 10 && blockIdx.x < 256) g_pipe_trace[((blockIdx.x * 16 + wv_all) * 10 + j) * 4 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define SONAR_PIPE_STAMP(slot) do { } while (0)
#endif

// iterations [IT0, IT1) of one thread slot's draw, kept in registers: iteration `it` is element (ky = n2 + 8 it, kx) and its partner
// 64 rows below = inputs n1 = it and n1 = it + H / 16 of the slot's column-pass-a item.  The slot meets the same 16 filter values in
// every plane: their weights wa / wb (filter_weight) stay in registers for the whole launch -- no filter loads, address registers or
// waits inside the draw (round 4 loaded the pair's two values one iteration ahead: sixteen 64-bit address registers and a vmcnt wait per
// iteration at two waves per SIMD).
template <int H, int W, int IT0, int IT1>
__device__ __forceinline__ void draw_chunk_regs(SpectrumRng& g, c32 (&v)[H / 8], const float (&wa)[H / 16], const float (&wb)[H / 16]) {
    constexpr int NT = plane_threads<H, W>(), M = W / 2, PAIRS = (H / 2) * M;
    static_assert(PAIRS % NT == 0 && IT0 <= IT1 && IT1 <= PAIRS / NT && NT == 8 * M && PAIRS / NT == H / 16, "slot = (wave n2, column)");
#pragma unroll
    for (int it = IT0; it < IT1; ++it) {
        const uint32_t ra = g.R.next();
        const uint32_t rb = g.R.next();
        const uint32_t t = g.T.next();
        if constexpr (SONAR_PIPE_SKIP & 2) {
            v[it] = make_float2(wa[it] + (float)ra, wb[it]);
            v[it + H / 16] = make_float2(wb[it] + (float)t, wa[it] + (float)rb);
        } else {
            v[it] = drawn_weighted(ra, angle_lo(t), wa[it]);
            v[it + H / 16] = drawn_weighted(rb, angle_hi(t), wb[it]);
        }
    }
}
// a filter with negative values (a wave that met one: uniform branch): the sign the weights dropped goes back on, read from the filter
template <int H, int W, int IT0, int IT1>
__device__ __forceinline__ void draw_chunk_signs(const float* __restrict__ filter, int tid, c32 (&v)[H / 8]) {
    constexpr int NT = plane_threads<H, W>(), M = W / 2, Wh = M + 1, LM = draw_shift<W>();
#pragma unroll
    for (int it = IT0; it < IT1; ++it) {
        const int p = tid + it * NT, pos = (p >> LM) * Wh + 1 + (p & (M - 1));
        if (filter[pos] < 0.0f) v[it] = make_float2(-v[it].x, -v[it].y);
        if (filter[pos + (H / 2) * Wh] < 0.0f) v[it + H / 16] = make_float2(-v[it + H / 16].x, -v[it + H / 16].y);
    }
}

// ---- the transforming team's passes.  NW waves share a pass (8 in the steady state; all 16 on the workgroup's last plane, when the
// drawing team has nothing left to draw); `w` = this wave's index among them.  Every pass requests ALL of its LDS operands before the
// first butterfly: scheduled item by item the loads come in batches of four with a wait behind each, and a team of two waves per SIMD
// has little else to run meanwhile.
// columns, pass b: radix 8 over rows 8 k1 .. 8 k1 + 7 of column `lane`, X -> Y
template <int H, int W, int NW>
__device__ __forceinline__ void pipe_col_b(const c32* X, c32* Y, int w, int lane) {
    constexpr int S = PlaneCfg<H, W>::S, CN1 = H / 8, CN2 = 8, ITEMS = CN1 / NW;
    static_assert(CN1 % NW == 0, "whole items per wave");
    c32 u[ITEMS][CN2];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
#pragma unroll
        for (int n2 = 0; n2 < CN2; ++n2) u[it][n2] = X[(CN2 * (w + NW * it) + n2) * S + lane];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        idft<CN2>(u[it]);
#pragma unroll
        for (int k2 = 0; k2 < CN2; ++k2) Y[(CN2 * (w + NW * it) + k2) * S + lane] = u[it][k2];
    }
}

// rows, pass a (c2r pre-twiddle fused), Y -> X: G[k] = (X[k] + conj X[M-k]) + i (X[k] - conj X[M-k]) w^k, w = e^{2 pi i / W}, then the
// radix-8 over n1 of k = 8 n1 + n2 and the twiddle e^{2 pi i n2 k1 / M}; element (k1, n2) of a row lands at column 8 n2 + k1.
// G[k] and G[M-k] are made of the SAME two operands: with S = X[k] + conj X[M-k], D = (X[k] - conj X[M-k]) w^k,
//     G[k] = S + i D,   G[M-k] = conj(S - i D)        (w^{M-k} = -conj w^k)
// and k -> M - k maps residue n2 to 8 - n2 (n1 to 7 - n1; 8 - n1 for residue 0).  Round 5: a thread therefore takes BOTH residues of
// one row -- half the LDS reads of the pass and 5 instead of 2 x 5 packed operations per pair of G -- where round 4 took one residue of
// two rows and read every operand twice.  The mirrored residue needs no twiddle table of its own: e^{2 pi i (8 - n2) k1 / M} =
// e^{2 pi i k1 / 8} conj(p[k1]), and the factor e^{2 pi i k1 / 8} is a circular shift of the radix-8's INPUT by one place (register naming).
// Waves 0-5: residue pairs (1,7) (2,6) (3,5), rows lane + 64 (w & 1).  Residues 0 and 4 mirror onto themselves (four operand pairs
// per item): waves 6 and 7, two rows each -- the same 133-140 packed operations for every wave.  Twiddles are wave-uniform (scalar registers).
// INPLACE (one plane buffer, X == Y: the phase-serial kernel): a workgroup barrier between the loads and the stores.
// The pass's sixteen twiddles come from a contiguous per-class table (c_rowa_tw128, twiddles256.h) with two scalar loads PER PASS, behind
// an opaque class index so that the loads stay inside the plane loop: held across the loop, 32 more long-lived scalar registers made the
// allocator spill the twiddles themselves into vector-register lanes and read them back one v_readlane (and its hazard nops) in front of
// every use -- 46 of the pass's 207 instructions; pinned in vector registers instead they spilled 64 of those to scratch.
template <int W>
__device__ __forceinline__ RowATw<W> row_a_twiddles(int w) {
    static_assert(W == 128, "c_rowa_tw128");
    int cls = w < 6 ? (w >> 1) : w - 3;
    asm volatile("" : "+s"(cls));
    const c32* t = c_rowa_tw128[cls];
    RowATw<W> r;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        r.g[i] = t[i];
        r.p[i] = t[8 + i];
    }
    return r;
}
__device__ __forceinline__ c32 cconj(c32 a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ c32 cmul_conj(c32 a, c32 b) {  // a * conj(b), packed like cmul
    const v2f A = vv(a), B = vv(b);
    return cc(__builtin_elementwise_fma(A.yx, v2f{B.y, -B.y}, A * B.xx));
}

template <int H, int W, int NW, int LAYOUT>
__device__ __forceinline__ void pipe_row_a(const c32* Y, c32* X, int w, int lane) {
    using C = PlaneCfg<H, W>;
    constexpr bool INPLACE = LAYOUT == kRowsInPlace, SWZ = LAYOUT == kRowsSwizzled;
    // column of spectrum element 8 n1 + res on the way in, of element (k1, res) on the way out (`res` is wave-uniform, not a constant)
    // (swizzled: the residue's offset goes through an opaque scalar per pass, see sf_row_a_split)
    auto swz_base = [](int res) {
        int b = (res & 1) + 16 * (res >> 1);
        if constexpr (SWZ) asm volatile("" : "+s"(b));
        return b;
    };
    auto cin = [&](int n1, int res, int base) { return SWZ ? 2 * n1 + base : 8 * n1 + res; };
    auto cout = [&](int k1, int res, int base) { return SWZ ? 2 * k1 + base : 8 * res + k1; };
    static_assert(!SWZ || (C::kRowSwizzle && C::rpos(3, 5) == 2 * 3 + 1 + 16 * 2), "cin / cout restate PlaneCfg::rpos");
    const RowATw<W> tw = row_a_twiddles<W>(__builtin_amdgcn_readfirstlane(w));
    constexpr int M = C::M, S = C::S, RN1 = C::RN1, RN2 = C::RN2;
    static_assert(H == 128 && RN1 == 8 && RN2 == 8 && NW == 8, "three residue pairs x two row halves + two self-mirrored residues x all rows");
    // S + i D and conj(S - i D) = (S.x + D.y, D.x - S.y) of one operand pair: six packed operations (the conjugates are operand modifiers)
    auto pair = [&](c32 xa, c32 xb, c32 g, c32& plus, c32& minus_conj) {
        const c32 xc = cconj(xb);
        const c32 s = cadd(xa, xc), d = cmul(csub(xa, xc), g);
        plus = cadd_i(s, d);
        minus_conj = cc(vv(d).yx + v2f{s.x, -s.y});
    };
    if (w < 6) {  // uniform
        const int a = 1 + (w >> 1), b = 8 - a, r = lane + 64 * (w & 1);
        const int ba = swz_base(a), bb = swz_base(b);
        const c32* row = Y + r * S;
        c32 xa[8], xb[8];
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) {
            xa[n1] = row[cin(n1, a, ba)];
            xb[n1] = row[cin(7 - n1, b, bb)];  // column M - (8 n1 + a)
        }
        __builtin_amdgcn_sched_barrier(0);
        c32 ga[8], tb[8];
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) {
            c32 m;
            pair(xa[n1], xb[n1], tw.g[n1], ga[n1], m);
            tb[(8 - n1) & 7] = m;  // G of residue b at n1' = 7 - n1, shifted by one place
        }
        idft<8>(ga);
        idft<8>(tb);
#pragma unroll
        for (int k1 = 1; k1 < 8; ++k1) ga[k1] = cmul(ga[k1], tw.p[k1]);
#pragma unroll
        for (int k1 = 1; k1 < 8; ++k1) tb[k1] = cmul_conj(tb[k1], tw.p[k1]);
        if constexpr (INPLACE) __syncthreads();  // every wave holds its results: all operands of all rows have been read
        c32* orow = X + r * S;
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) orow[cout(k1, a, ba)] = ga[k1];
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) orow[cout(k1, b, bb)] = tb[k1];
    } else {
        const bool zero = w == 6;  // uniform: residue 0 (its first element is the packed column: Re = column 0, Im = column M) or residue 4
        const int a = zero ? 0 : 4, ba = swz_base(a);
        c32 x[2][8];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const c32* row = Y + (lane + 64 * it) * S;
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) x[it][n1] = row[cin(n1, a, ba)];
        }
        __builtin_amdgcn_sched_barrier(0);
        c32 g[2][8];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            if (zero) {
                const c32 p = x[it][0];
                // k = 0: X[0] = Re p, X[M] = Im p, both real: G = (Re p + Im p) + i (Re p - Im p) -- through the general expression
                // (same bits as round 4's)
                {
                    const c32 xa = make_float2(p.x, 0.0f), xc = make_float2(p.y, -0.0f);
                    g[it][0] = cadd_i(cadd(xa, xc), cmul(csub(xa, xc), tw.g[0]));
                }
                {  // k = M / 2 mirrors onto itself
                    const c32 xa = x[it][4], xc = cconj(xa);
                    g[it][4] = cadd_i(cadd(xa, xc), cmul(csub(xa, xc), tw.g[4]));
                }
#pragma unroll
                for (int n1 = 1; n1 < 4; ++n1) {
                    c32 m;
                    pair(x[it][n1], x[it][8 - n1], tw.g[n1], g[it][n1], m);
                    g[it][8 - n1] = m;
                }
            } else {
#pragma unroll
                for (int n1 = 0; n1 < 4; ++n1) {
                    c32 m;
                    pair(x[it][n1], x[it][7 - n1], tw.g[n1], g[it][n1], m);
                    g[it][7 - n1] = m;
                }
            }
            idft<8>(g[it]);
#pragma unroll
            for (int k1 = 1; k1 < 8; ++k1) g[it][k1] = cmul(g[it][k1], tw.p[k1]);
        }
        if constexpr (INPLACE) __syncthreads();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
#pragma unroll
            for (int k1 = 0; k1 < 8; ++k1) X[(lane + 64 * it) * S + cout(k1, a, ba)] = g[it][k1];
        }
    }
}

// rows, pass b -> global.  A thread owns LDS row r and the output residues k1 = 2a, 2a + 1 (NW = 8): its two radix-8 items yield the
// complex outputs m = k1 + 8 k2, FOUR consecutive floats per k2 -> eight 16-byte stores (64-byte runs from the four lanes of a row)
// instead of sixteen 8-byte ones: the store phase is bound by store instructions, not bytes.  Pass a left element (k1, n2) at column
// 8 n2 + k1, so the thread's two inputs of an n2 sit side by side (one ds_read2_b64).  Lane -> row: a read2's access groups are 16
// consecutive lanes over 32 banks; they hold rows r0 + {0, 1, 8, 9} x a < 4, whose 2-dword windows start at 2 r + 4 a (mod 32) =
// {0, 2, 16, 18} + 4 a with the odd row stride -- all different.  NW = 16 (last plane): the second eight waves take k1 = 2a + 1.
// kRowsSwizzled: the inputs sit at rpos(k1, n2) = 2 k1 + n2 % 2 + 16 (n2 / 2) -- the thread's two inputs of an n2 are two places apart
// (still one ds_read2_b64), windows start at 2 r + 8 a (mod 32): the 16 lanes of an access group take rows r0 + {0, 1, 2, 3} instead.
template <int H, int W, int NW, bool STATS, bool NORM, int LAYOUT>
__device__ __forceinline__ void pipe_row_b(const c32* X, float* oplane, int w, int lane, float scale, float nm, float nc, double& s, double& q) {
    using C = PlaneCfg<H, W>;
    constexpr int S = C::S, RN1 = C::RN1, RN2 = C::RN2, CN1 = H / 8, CN2 = 8, NK = NW == 8 ? 2 : 1;
    constexpr bool SWZ = LAYOUT == kRowsSwizzled;
    static_assert(H == 128 && RN1 == 8 && RN2 == 8 && (NW == 8 || NW == 16), "16 rows x 4 residue pairs per wave");
    const int a = lane & 3, w8 = w & 7, k0 = 2 * a + (NW == 8 ? 0 : w >> 3);
    const int r = SWZ ? 16 * w8 + (lane >> 2) : 16 * w8 + 2 * (lane >> 4) + 8 * ((lane >> 3) & 1) + ((lane >> 2) & 1);
    const int y = (r / CN2) + CN1 * (r % CN2);
    c32 u[NK][RN2];
#pragma unroll
    for (int n2 = 0; n2 < RN2; ++n2) {
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            if constexpr (SONAR_PIPE_SKIP & 4) u[i][n2] = make_float2((float)(lane + n2), 1.0f + i);
            else u[i][n2] = X[r * S + (SWZ ? C::rpos(k0 + i, n2) : RN1 * n2 + k0 + i)];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(SONAR_PIPE_SKIP & 4)) {
#pragma unroll
        for (int i = 0; i < NK; ++i) idft<RN2>(u[i]);
    }
    float* const orow = oplane + (int64_t)y * W + 2 * k0;
    float ps = 0.0f, pq = 0.0f;
#pragma unroll
    for (int k2 = 0; k2 < RN2; ++k2) {
        float o[2 * NK];
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            if constexpr (NORM) {
                o[2 * i] = __builtin_fmaf(u[i][k2].x, nm, -nc);
                o[2 * i + 1] = __builtin_fmaf(u[i][k2].y, nm, -nc);
            } else {
                o[2 * i] = u[i][k2].x * scale;
                o[2 * i + 1] = u[i][k2].y * scale;
            }
            if constexpr (STATS) {
                ps += o[2 * i] + o[2 * i + 1];
                pq = __builtin_fmaf(o[2 * i], o[2 * i], __builtin_fmaf(o[2 * i + 1], o[2 * i + 1], pq));
            }
        }
        float* const dst = orow + 2 * RN1 * k2;
        if constexpr (SONAR_PIPE_SKIP & 1) {
            if (o[0] == 123.456f && o[1] == 654.321f) *reinterpret_cast<float2*>(dst) = make_float2(o[0], o[1]);  // keeps the arithmetic alive
        } else if constexpr (NK == 2) {
            store4<(SONAR_PIPE_NT != 0)>(dst, o[0], o[1], o[2], o[3]);
        } else {
            *reinterpret_cast<float2*>(dst) = make_float2(o[0], o[1]);
        }
    }
    if constexpr (STATS) {
        s += (double)ps;
        q += (double)pq;
    }
}

// ---- the spectral filter's forward half at 128 x 128 (round 5) --------------------------------------------------------------------
// Mirror images of the inverse passes above.  All values carry a factor 2 (the r2c split's halves are left to the final scale).
// rows, pass a' + r2c split: the radix-8 over k1 of residues a and 8 - a of one row, then
//     2 X[k] = s - i d conj(g),  2 X[M-k] = conj(s + i d conj(g)),   s = C[k] + conj C[M-k],  d = C[k] - conj C[M-k],  g = e^{2 pi i k / W}
// on the pair (k, M - k) = (8 n1 + a, 8 (7 - n1) + 8 - a) the thread holds -- round 4 ran the split as a pass of its own (a barrier, an LDS
// round trip) behind a pass a' that moved from swizzled to natural columns through a barrier of its own.  In place at rpos(., residue).
template <int H, int W>
__device__ __forceinline__ void sf_row_a_split(c32* A, int w, int lane) {
    using C = PlaneCfg<H, W>;
    constexpr int S = C::S;
    static_assert(H == 128 && W == 128 && C::RN1 == 8 && C::RN2 == 8 && C::kRowSwizzle, "three residue pairs x two row halves + two self-mirrored residues");
    const RowATw<W> tw = row_a_twiddles<W>(__builtin_amdgcn_readfirstlane(w));
    // PlaneCfg::rpos with a wave-uniform residue; its offset is recomputed per pass (hoisted out of the plane loop, the sixteen column
    // offsets of every pass become long-lived scalar registers and spill into vector-register lanes)
    auto res_base = [](int res) {
        int b = (res & 1) + 16 * (res >> 1);
        asm volatile("" : "+s"(b));
        return b;
    };
    auto split = [&](c32 ck, c32 cm, c32 g, c32& xk, c32& xm) {
        const c32 cj = cconj(cm);
        const c32 s = cadd(ck, cj), d = cmul_conj(csub(ck, cj), g);
        xk = csub_i(s, d);
        xm = cc(v2f{s.x, -s.y} - vv(d).yx);
    };
    if (w < 6) {  // uniform
        const int a = 1 + (w >> 1), b = 8 - a;
        c32* const row = A + (lane + 64 * (w & 1)) * S;
        c32* const rowa = row + res_base(a);
        c32* const rowb = row + res_base(b);
        c32 ua[8], ub[8];
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) {
            ua[k1] = rowa[2 * k1];
            ub[k1] = rowb[2 * k1];
        }
        __builtin_amdgcn_sched_barrier(0);
        fdft<8>(ua);
        fdft<8>(ub);
        c32 xa[8], xb[8];
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) split(ua[n1], ub[7 - n1], tw.g[n1], xa[n1], xb[7 - n1]);
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) rowa[2 * n1] = xa[n1];
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) rowb[2 * n1] = xb[n1];
    } else {
        const bool zero = w == 6;  // uniform: residue 0 (k = 0 becomes the packed pair X[0] + i X[M], both real; k = M / 2 mirrors onto itself) or 4
        c32* const row0 = A + lane * S + res_base(zero ? 0 : 4);
        c32 u[2][8];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
#pragma unroll
            for (int k1 = 0; k1 < 8; ++k1) u[it][k1] = row0[64 * it * S + 2 * k1];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            fdft<8>(u[it]);
            c32 x[8];
            if (zero) {
                const c32 c0 = u[it][0], c4 = u[it][4];
                x[0] = make_float2(2.0f * (c0.x + c0.y), 2.0f * (c0.x - c0.y));
                x[4] = make_float2(2.0f * c4.x, -2.0f * c4.y);
#pragma unroll
                for (int n1 = 1; n1 < 4; ++n1) split(u[it][n1], u[it][8 - n1], tw.g[n1], x[n1], x[8 - n1]);
            } else {
#pragma unroll
                for (int n1 = 0; n1 < 4; ++n1) split(u[it][n1], u[it][7 - n1], tw.g[n1], x[n1], x[7 - n1]);
            }
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) row0[64 * it * S + 2 * n1] = x[n1];
        }
    }
}

// columns, pass b': forward radix 8 over rows 8 k1 .. 8 k1 + 7 of column `lane` (k1 = w, w + 8), twiddle e^{-2 pi i n2 k1 / H}; in place
template <int H, int W>
__device__ __forceinline__ void sf_col_b(c32* A, int w, int lane) {
    constexpr int S = PlaneCfg<H, W>::S, ITEMS = H / 64;
    c32 u[ITEMS][8];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) u[it][k2] = A[(8 * (w + 8 * it) + k2) * S + lane];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        int k1 = __builtin_amdgcn_readfirstlane(w + 8 * it);
        asm volatile("" : "+s"(k1));  // the twiddles are fetched per pass (see row_a_twiddles)
        const c32* const t = c_colb_tw128[k1];
        fdft<8>(u[it]);
#pragma unroll
        for (int n2 = 1; n2 < 8; ++n2) u[it][n2] = cmul_conj(u[it][n2], t[n2]);
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) A[(8 * k1 + n2) * S + lane] = u[it][n2];
    }
}

// columns: forward pass a', x filter, inverse pass a -- the same sixteen rows 8 j + w of column `lane` go in and come out, so the
// spectrum itself only ever exists in registers.  (Round 3 measured this fusion as a dead end: it spilled.  What made it fit: offsets
// and twiddles that are made per pass instead of living across the plane loop, see row_a_twiddles.)
// Column 0 is the packed pair: P = DFT(X0 + i XM) needs its mirror P[-ky], which another wave holds -- lane 0 leaves P in `xch`,
// and after ONE workgroup barrier lanes 0-15 build what the inverse wants from it, one ky each,
//     Q[ky] = sym(Z0 f0)[ky] + i sym(ZM fM)[ky] = a1 P[ky] + a2 conj P[-ky],
//     a1, a2 = (g0 +- gM) / 2,  g.[ky] = (f.[ky] + f.[-ky]) / 2      (Z0, ZM are spectra of real columns: Hermitian already)
// and hand the sixteen values back to lane 0 through LDS (in order within a wave: no barrier) -- round 4 unpacked Z0 / ZM, filtered
// them and packed them again in three phases.  `xch`: [0] the (a1, a2) table, made once per workgroup, [1] P, [2] Q; each indexed
// [16 w + j] for ky = 8 j + w.  `f`: the thread's sixteen filter values, requested by the caller a pass ahead.
template <int H>
struct SfExchange {
    c32 wgt[H], p[H], q[H];
};
template <int H, int W>
__device__ __forceinline__ void sf_filter_values(const float* __restrict__ filter, int w, int lane, float (&f)[H / 8]) {
    constexpr int Wh = W / 2 + 1;
    // this lane's spectrum column: the inverse of rpos.  The values are the same for every plane: their loads (L2 hits) stay in the plane
    // loop behind an opaque index -- kept across it they would hold sixteen registers the row passes need.
    int kx = 8 * ((lane & 15) >> 1) + 2 * (lane >> 4) + (lane & 1);
    asm volatile("" : "+v"(kx));
    const float* const fcol = filter + w * Wh + kx;
#pragma unroll
    for (int n1 = 0; n1 < H / 8; ++n1) f[n1] = fcol[8 * n1 * Wh];
}
template <int H, int W>
__device__ __forceinline__ void sf_col_a_filter_col_a(c32* A, SfExchange<H>* xch, const float (&f)[H / 8], int w, int lane, int pidx) {
    using C = PlaneCfg<H, W>;
    constexpr int S = C::S, N = H / 8;
    static_assert(N == 16, "radix 16");
    [[maybe_unused]] const int tid = w * 64 + lane;  // (trace builds' stamps)
    c32 v[N];
#pragma unroll
    for (int k1 = 0; k1 < N; ++k1) v[k1] = A[(8 * k1 + w) * S + lane];
    fdft<N>(v);
    SONAR_STAMP(7);
    int wo = __builtin_amdgcn_readfirstlane(w);
    asm volatile("" : "+s"(wo));  // the exchange's offsets and the twiddles are made / fetched per pass (see row_a_twiddles)
    if (lane == 0) {
        float4* const dst = reinterpret_cast<float4*>(xch->p + 16 * wo);
#pragma unroll
        for (int n1 = 0; n1 < N; n1 += 2) dst[n1 / 2] = make_float4(v[n1].x, v[n1].y, v[n1 + 1].x, v[n1 + 1].y);
    }
#pragma unroll
    for (int n1 = 0; n1 < N; ++n1) v[n1] = cscale(v[n1], f[n1]);  // (lane 0's are replaced below)
    __syncthreads();
    SONAR_STAMP(8);
    if (lane < 16) {
        const int mirror = wo == 0 ? ((16 - lane) & 15) : 16 * (8 - wo) + 15 - lane;  // ky -> H - ky: residue 8 - w, j -> 15 - j (16 - j for residue 0)
        const c32 p = xch->p[16 * wo + lane], pn = xch->p[mirror], a = xch->wgt[16 * wo + lane];
        xch->q[16 * wo + lane] = make_float2(__builtin_fmaf(a.x, p.x, a.y * pn.x), __builtin_fmaf(a.x, p.y, -(a.y * pn.y)));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane == 0) {
        const float4* const src = reinterpret_cast<const float4*>(xch->q + 16 * wo);
#pragma unroll
        for (int n1 = 0; n1 < N; n1 += 2) {
            const float4 t = src[n1 / 2];
            v[n1] = make_float2(t.x, t.y);
            v[n1 + 1] = make_float2(t.z, t.w);
        }
    }
    SONAR_STAMP(9);
    idft<N>(v);
    const c32* const t = c_cola_tw128[wo];
#pragma unroll
    for (int k1 = 1; k1 < N; ++k1) v[k1] = cmul(v[k1], t[k1]);
    SONAR_STAMP(10);
#pragma unroll
    for (int k1 = 0; k1 < N; ++k1) A[(8 * k1 + w) * S + lane] = v[k1];
}

// The spectral filter at 128 x 128: out = irfft2(rfft2(x) * filter), py/nodes/powernoise.py:356-366.  Two 8-wave workgroups per CU, a
// plane in LDS each, SEVEN workgroup barriers per plane (round 4: twelve):
//   rows b'    global -> registers -> radix 8 over k2, twiddle -> LDS (the next plane's loads are requested three passes ahead: a
//              workgroup used to sit 5.4 of a plane's 19.7 us behind its own loads, profiles/r05_spectral_filter.md)
//   rows a'    + r2c split in registers (sf_row_a_split)                  cols b'   radix 8, twiddle (sf_col_b)
//   cols a' x filter, cols a   radix 16 forward, the filter, radix 16 inverse in registers (sf_col_a_filter_col_a; one barrier inside)
//   cols b, rows a, rows b     the pipelined generate kernel's passes on the swizzled row layout (kRowsSwizzled: row pass a in place
//              without the barrier between its loads and stores)
#ifndef SONAR_SF_FILTER_EARLY
#define SONAR_SF_FILTER_EARLY 1
#endif
#ifndef SONAR_SF_PREFETCH
#define SONAR_SF_PREFETCH 1  // 0: a plane's loads are requested at its own start (A/B)
#endif
#ifndef SONAR_SF_WIDE
#define SONAR_SF_WIDE 1  // rows b' takes items (k1 = 2 (t % 4) + i, LDS row t / 4) with eight 16-byte loads per thread; 0: (k1 = t % 8, LDS rows
#endif                   // t / 8 + 64 i) with sixteen 8-byte ones (+1 us per 512 latents).  (Global accesses need dword alignment, whatever their width.)
template <int H, int W, bool STATS>
__global__ void __launch_bounds__(512, 4) spectral_filter128_kernel(const float* __restrict__ x, const float* __restrict__ filter, float* out, int64_t planes,
                                                                     double* partials) {
    kernarg_touch_for(x, filter, out, planes, partials);
    using C = PlaneCfg<H, W>;
    constexpr int NT = 512, M = C::M, S = C::S, Wh = C::Wh, RN1 = C::RN1, RN2 = C::RN2, CN1 = H / 8, CN2 = 8;
    static_assert(H == 128 && W == 128 && RN1 == 8 && RN2 == 8 && plane_threads<H, W>() == NT, "8 waves, 8 x 8 rows, 16 x 8 columns");
    __shared__ __attribute__((aligned(16))) c32 A[H * S];
    __shared__ __attribute__((aligned(16))) SfExchange<H> xch;
    __shared__ c32 TW[M];  // e^{2 pi i j / M}: pass b''s lane-dependent twiddles
    __shared__ double red[2 * NT / 64];
    const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int j = tid; j < M; j += NT) TW[j] = c_tw256[j * (256 / M)];
    for (int i = tid; i < H; i += NT) {  // the packed column's weights, i = 16 w + j for ky = 8 j + w
        const int ky = 8 * (i & 15) + (i >> 4), kn = (H - ky) & (H - 1);
        const float g0 = 0.5f * (filter[ky * Wh] + filter[kn * Wh]), gm = 0.5f * (filter[ky * Wh + M] + filter[kn * Wh + M]);
        xch.wgt[i] = make_float2(0.5f * (g0 + gm), 0.5f * (g0 - gm));
    }
    // both transforms unscaled, the split's halves left out: 1 / (2 H W)
    const float scale = 0.5f / ((float)H * (float)W);
    __shared__ double sums[STATS ? 2 * NT : 1];
    if constexpr (STATS) {
        sums[tid] = 0.0;
        sums[NT + tid] = 0.0;
    }
    [[maybe_unused]] int pidx = 0;
    // rows b': item (k1 = tid % 8, LDS row r = tid / 8 + 64 it); LDS row r holds spatial row y = r / 8 + 16 (r % 8) (what the column passes
    // expect); complex element m = k1 + 8 k2 of a row is (x[2m], x[2m+1])
    c32 u[2][RN2];
    constexpr bool WIDE = SONAR_SF_WIDE != 0;
    auto request = [&](int64_t plane, int t) {
        if constexpr (WIDE) {
            const int r = t >> 2, y = (r / CN2) + CN1 * (r % CN2);
            const float* xrow = x + plane * (int64_t)H * W + (int64_t)y * W + 4 * (t & 3);
#pragma unroll
            for (int k2 = 0; k2 < RN2; ++k2) {
                const float4 v = *reinterpret_cast<const float4*>(xrow + 2 * RN1 * k2);
                u[0][k2] = make_float2(v.x, v.y);
                u[1][k2] = make_float2(v.z, v.w);
            }
        } else {
            const int k1 = t & 7;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int r = (t >> 3) + 64 * it, y = (r / CN2) + CN1 * (r % CN2);
                const float* xrow = x + plane * (int64_t)H * W + (int64_t)y * W;
#pragma unroll
                for (int k2 = 0; k2 < RN2; ++k2) u[it][k2] = *reinterpret_cast<const float2*>(xrow + 2 * (k1 + RN1 * k2));
            }
        }
    };
    // the inverse's last three passes (the next plane's loads are in flight through them)
    auto finish = [&](int64_t plane, int lane) {
        pipe_col_b<H, W, 8>(A, A, wv, lane);  // in place: an item reads and writes the same 8 rows of its column
        __syncthreads();
        SONAR_STAMP(5);
        pipe_row_a<H, W, 8, kRowsSwizzled>(A, A, wv, lane);
        __syncthreads();
        SONAR_STAMP(6);
        // (the statistics variant is four registers over its 128: the thread's two running fp64 sums wait in LDS between planes)
        double ps = 0.0, pq = 0.0;
        pipe_row_b<H, W, 8, STATS, false, kRowsSwizzled>(A, out + plane * (int64_t)H * W, wv, lane, scale, scale, 0.0f, ps, pq);
        if constexpr (STATS) {
            sums[tid] += ps;
            sums[NT + tid] += pq;
        }
        SONAR_STAMP(11);
        ++pidx;
    };
    int64_t plane = blockIdx.x;
    if (plane >= planes) return;  // (never: the grid is at most `planes`)
    if (SONAR_SF_PREFETCH) request(plane, tid);
    for (;;) {
        __syncthreads();  // the previous plane's LDS reads are done (and the tables are visible)
        SONAR_STAMP(0);
        int ptid = tid;  // per plane: every LDS address of a plane is loop-invariant, and hoisted they spill (power_irfft2_kernel)
        asm volatile("" : "+v"(ptid));
        const int lane = ptid & 63;
        if (!SONAR_SF_PREFETCH) request(plane, ptid);
        {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int k1 = WIDE ? 2 * (ptid & 3) + it : ptid & 7, r = WIDE ? ptid >> 2 : (ptid >> 3) + 64 * it;
                fdft<RN2>(u[it]);
#pragma unroll
                for (int n2 = 1; n2 < RN2; ++n2) u[it][n2] = cmulc(u[it][n2], TW[(n2 * k1) & (M - 1)]);
#pragma unroll
                for (int n2 = 0; n2 < RN2; ++n2) A[r * S + C::rpos(k1, n2)] = u[it][n2];
            }
        }
        __syncthreads();
        SONAR_STAMP(1);
        sf_row_a_split<H, W>(A, wv, lane);
        __syncthreads();
        SONAR_STAMP(2);
        float f[CN1];
        if (SONAR_SF_FILTER_EARLY) sf_filter_values<H, W>(filter, wv, lane, f);  // a pass ahead of their use
        sf_col_b<H, W>(A, wv, lane);
        __syncthreads();
        SONAR_STAMP(3);
        if (!SONAR_SF_FILTER_EARLY) sf_filter_values<H, W>(filter, wv, lane, f);
        sf_col_a_filter_col_a<H, W>(A, &xch, f, wv, lane, pidx);
        __syncthreads();
        SONAR_STAMP(4);
        // The loop is split at the prefetch, not closed behind the last pass: with the request under a condition the loaded registers
        // merge with their old values in a copy, and the copy waits for the loads right where they were issued.
        const int64_t next = plane + gridDim.x;
        if (next >= planes) {  // uniform
            finish(plane, lane);
            break;
        }
        if (SONAR_SF_PREFETCH) request(next, ptid);
        finish(plane, lane);
        plane = next;
    }
    if constexpr (STATS) write_partial<NT>(sums[tid], sums[NT + tid], partials, red);
}

// ---- statistics of the NEXT call, computed in the pipelined kernel's idle corners -------------------------------------------------
// A normalised call needs the Parseval statistics of all of its planes before its first store: power_stats_kernel, 10 us in front of
// a 41 us final pass.  A sampler calls the same generator step after step with consecutive stream ids, and the pipelined kernel has
// two idle corners per launch -- the transforming team while the first plane is drawn, the drawing team while the last plane is
// transformed -- so each team there computes the statistics of ONE unit of the call that follows (same seed, `next_stream`) and leaves
// them as that call's partials: its host wrapper then skips the statistics launch when the prediction held, and runs it when not.
// The work is the statistics kernel's, per unit, in the same order (same bits); it is spread over three barrier-separated steps
// because the workgroup barrier counts both teams.
template <int H, int W>
struct TeamStats {
    static constexpr int NT = plane_threads<H, W>(), M = W / 2, Wh = M + 1, LM = draw_shift<W>(), PAIRS = (H / 2) * M, ITER = draw_iters<H, W>();
    float wgt[2 * ITER], prod[2 * ITER];
    float f0, fm;
    SpectrumRng rng;
    double s, q;
    int planes_in_unit;

    // weights, stream states, and the unit's edge columns (all planes: the E stream is its own) into `edge` [plane][2][H]
    // `wa`, `wb`: the slot's weights when the caller holds them already (the drawing team), else read from the filter
    __device__ __forceinline__ void begin(const float* __restrict__ filter, uint64_t seed, uint64_t stream_id, int64_t plane_offset, int group,
                                          const GroupWalk& gw, int tid, c32* edge, const float* wa = nullptr, const float* wb = nullptr) {
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int p = tid + it * NT, ky = p >> LM, kx = 1 + (p & (M - 1));
            const bool live = p < PAIRS && kx < M;
            if (wa) {
                wgt[it] = live ? wa[it] : 0.0f;
                wgt[it + ITER] = live ? wb[it] : 0.0f;
            } else {
                wgt[it] = filter_weight(live ? filter[ky * Wh + kx] : 0.0f);
                wgt[it + ITER] = filter_weight(live ? filter[(ky + H / 2) * Wh + kx] : 0.0f);
            }
            prod[it] = 1.0f;
            prod[it + ITER] = 1.0f;
        }
        f0 = tid < H ? filter[tid * Wh] : 0.0f;
        fm = tid < H ? filter[tid * Wh + M] : 0.0f;
        s = 0.0;
        q = 0.0;
        planes_in_unit = gw.count;
        rng = spectrum_rng<H, false>(seed, stream_id, plane_offset / group + gw.grp, tid);
        for (int i = 0; i < gw.first; ++i) skip_plane<H, W, false>(rng, tid);
        if (tid < H) {
            for (int b = 0; b < gw.count; ++b) {
                const uint32_t r0 = rng.E.next();
                const uint32_t rm = rng.E.next();
                const uint32_t t = rng.E.next();
                edge[(2 * b) * H + tid] = drawn_elem(r0, angle_lo(t), f0);
                edge[(2 * b + 1) * H + tid] = drawn_elem(rm, angle_hi(t), fm);
            }
        }
    }
    // radius words of planes [b0, b1) of the unit
    __device__ __forceinline__ void radii(int b0, int b1, int tid) {
        for (int b = b0; b < min(b1, planes_in_unit); ++b) {
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const uint32_t ra = rng.R.next();
                const uint32_t rb = rng.R.next();
                prod[it] *= 2.0f - unit_mantissa(ra);
                prod[it + ITER] *= 2.0f - unit_mantissa(rb);
            }
        }
#pragma unroll
        for (int it = 0; it < 2 * ITER; ++it) asm volatile("" : "+v"(prod[it]));  // stays in this phase (registers only: see pin_chunk)
    }
    __device__ __forceinline__ void products() {
        float acc = 0.0f;
#pragma unroll
        for (int it = 0; it < 2 * ITER; ++it) acc = __builtin_fmaf(wgt[it], __builtin_amdgcn_logf(prod[it]), acc);
        q += 2.0 * (double)acc;
    }
    // the edge columns' terms (after a barrier behind begin()), then this wave's sums into `sred` [2][8]
    __device__ __forceinline__ void edges_and_wave_sums(const c32* edge, int tid, double* sred) {
        for (int b = 0; b < planes_in_unit; ++b) {
            float e = 0.0f;
            if (tid < H) {
                const int ky = tid, kn = (H - ky) & (H - 1);
                const c32 a = edge[(2 * b) * H + ky], an = edge[(2 * b) * H + kn], c = edge[(2 * b + 1) * H + ky], cn = edge[(2 * b + 1) * H + kn];
                const float ar = 0.5f * (a.x + an.x), ai = 0.5f * (a.y - an.y), br = 0.5f * (c.x + cn.x), bi = 0.5f * (c.y - cn.y);
                e += (ar * ar + ai * ai) + (br * br + bi * bi);
                if (ky == 0) s += (double)(sqrtf((float)H * (float)W) * ar);
            }
            q += (double)e;
        }
        const double ws = wave_sum(s), wq = wave_sum(q);
        if ((tid & 63) == 0) {
            sred[tid >> 6] = ws;
            sred[NT / 64 + (tid >> 6)] = wq;
        }
    }
    // (after a barrier) the unit's pair, summed over the team's waves in the statistics kernel's order
    static __device__ __forceinline__ void store(const double* sred, int tid, double* partials_next, int64_t slot) {
        if (tid == 0) {
            double ss = 0.0, qq = 0.0;
#pragma unroll
            for (int i = 0; i < NT / 64; ++i) {
                ss += sred[i];
                qq += sred[NT / 64 + i];
            }
            partials_next[2 * slot] = ss;
            partials_next[2 * slot + 1] = qq;
        }
    }
};

// The draw only touches registers, so nothing orders it against a barrier: the optimiser sinks every chunk to its first use, right in
// front of column pass a (phase 3), and the first two phases of the drawing team run empty.  An empty volatile asm that takes the
// chunk's results as read-write operands makes them opaque at that point: they must exist before it, and it stays before the barrier.
template <int IT0, int IT1, int N>
__device__ __forceinline__ void pin_chunk(c32 (&v)[N]) {
#pragma unroll
    for (int it = IT0; it < IT1; ++it) {
        asm volatile("" : "+v"(v[it].x), "+v"(v[it].y), "+v"(v[it + N / 2].x), "+v"(v[it + N / 2].y));
    }
}

#ifndef SONAR_PIPE_KERNARG_TOUCH
#define SONAR_PIPE_KERNARG_TOUCH 1
#endif
template <int H, int W, bool STATS, bool NORM>
__global__ void __launch_bounds__(1024) power_pipe_kernel(const float* __restrict__ filter, float* out, int64_t planes, uint64_t seed,
                                                          uint64_t stream_id, int64_t plane_offset, int group, int split, double* partials,
                                                          NormArgs na, uint64_t next_stream, double* partials_next) {
#if SONAR_PIPE_KERNARG_TOUCH
    kernarg_touch_for(filter, out, planes, seed, stream_id, plane_offset, group, split, partials, na, next_stream, partials_next);
#endif
    using C = PlaneCfg<H, W>;
    constexpr int NT = 512, NALL = 1024;
    static_assert(plane_threads<H, W>() == NT && W == 128 && H == 128, "one 8-wave team per plane, slot = one radix-16 column item");
    constexpr int M = C::M, S = C::S, Wh = M + 1, RN1 = C::RN1, RN2 = C::RN2, CN1 = H / 8, CN2 = 8, ITER = draw_iters<H, W>();
    constexpr int LM = draw_shift<W>();
    constexpr int kChunk[2] = {SONAR_PIPE_CHUNKS};
    constexpr int E0 = kChunk[0], E1 = E0 + kChunk[1];
    static_assert(E1 <= ITER && ITER == CN1 / 2, "chunks of whole pair iterations");
    constexpr int BUF = H * S;
    __shared__ c32 PLANES[2 * BUF];
    __shared__ c32 EDGE[3 * H];  // raw columns kx = 0 and kx = M of the plane being drawn, and the packed column built from them
    // stream states of this workgroup's SECOND unit, seeded by the transforming team while it waits for the first plane (the draw team
    // would spend ~0.8 us on the Philox rounds at the unit switch: 32-bit multiplies at a sixth of the plain rate)
    __shared__ uint4 SEED_RT[NT];  // (R, T) per slot; the drawing team's look-ahead edge columns borrow the area later (8 KB: four planes)
    static_assert(sizeof(uint4) * NT >= sizeof(c32) * 2 * H * kAheadMaxGroup, "edge columns of a look-ahead unit fit the seed area");
    __shared__ double sred[2 * NT / 64];  // wave sums of a team's look-ahead statistics (TeamStats)
    __shared__ double red[2 * NALL / 64];
    // SONAR_PIPE_DECOUPLE: arrivals at the transforming team's exchanges (two per iteration: behind column pass b, behind row pass a) and
    // at the drawing team's one (its look-ahead statistics)
    __shared__ int tbar, dbar;
    if constexpr (SONAR_PIPE_DECOUPLE) {
        if (threadIdx.x == 0) {
            tbar = 0;
            dbar = 0;
        }
        __syncthreads();
    }
    __shared__ NormDecision shd;
    const int wv_all = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const bool drawer = wv_all >= NT / 64;
    const int tid = threadIdx.x & (NT - 1), lane = threadIdx.x & 63, wv = wv_all & (NT / 64 - 1);
    // this workgroup's plane sequence: units blockIdx.x, blockIdx.x + gridDim.x, ...; `per_unit` planes each
    const int64_t nunits = split ? planes : planes / group;
    const int64_t group0 = plane_offset / group;  // the call's first RNG group (once: a 64-bit division per unit and stream otherwise)
    const int per_unit = split ? 1 : group;
    const int64_t my_units = (int64_t)blockIdx.x < nunits ? (nunits - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const int n = (int)(my_units * per_unit);
    // statistics of the call that follows (TeamStats): one unit per team, so only for workgroups with at most two units
    const bool ahead = partials_next != nullptr && nunits <= 2 * (int64_t)gridDim.x;
    const float scale = 1.0f / sqrtf((float)H * (float)W);
    float nm = scale, nc = 0.0f;
    // the statistics pass's partials: requested now, reduced by all 16 waves when the first plane has been drawn (one pair per thread)
    static_assert(!NORM || kNPart == NALL, "one partial pair per thread");
    [[maybe_unused]] double pre_s = 0.0, pre_q = 0.0;
    if constexpr (NORM) {
        pre_s = na.partials[2 * threadIdx.x];
        pre_q = na.partials[2 * threadIdx.x + 1];
    }
    // The decision in three steps that ride on barriers the pipeline has anyway (round 4 ran decide_from_sums at the top of iteration 1:
    // two extra workgroup barriers and a chain of fp64 divisions and square roots in ONE thread with fifteen waves waiting, 1.7 us per
    // launch): every wave leaves its sums in LDS before the last barrier of iteration 0; wave 0 of the transforming team adds them up
    // in wave order (the order of block_sum2: same bits) and decides during phase 1 of iteration 1; the transforming team picks the
    // decision up behind the second barrier, in front of its first stores.  The drawing team never needs it.
    auto leave_wave_sums = [&]() {
        if constexpr (NORM) {
            const double ws = wave_sum(pre_s), wq = wave_sum(pre_q);
            if (lane == 0) {
                red[wv_all] = ws;
                red[NALL / 64 + wv_all] = wq;
            }
        }
    };
    auto decide_in_wave0 = [&]() {
        if constexpr (NORM) {
            if (wv_all == 0) {
                double ss = 0.0, qq = 0.0;
#pragma unroll
                for (int i = 0; i < NALL / 64; ++i) {
                    ss += red[i];
                    qq += red[NALL / 64 + i];
                }
                const NormDecision dec = decision_from_totals(ss, qq, na.n_total, na.thr_sd);
                if (lane == 0) shd = dec;
            }
        }
    };
    auto pick_up_decision = [&]() {
        if constexpr (NORM) {
            const NormDecision dec = shd;
            const float g = (dec.do_div ? 1.0f / dec.stdv : 1.0f) * na.factor;
            nm = scale * g;
            nc = dec.do_sub ? dec.mean * g : 0.0f;
        }
    };
    double s = 0.0, q = 0.0;
    if (drawer) {
        // ------------------------------------------------------------------------------------------------ drawing team
        __builtin_amdgcn_s_setprio(SONAR_PIPE_PRIO_DRAW);
        const c32* const Q = EDGE + 2 * H;  // the packed column 0 of the plane being drawn, left by the transforming team in phase 2
        SpectrumRng rng;
        // the slot's 16 filter weights, for every plane of the launch (and the look-ahead statistics of the epilogue)
        float wa[ITER], wb[ITER];
        bool neg = false;
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int p = tid + it * NT, pos = (p >> LM) * Wh + 1 + (p & (M - 1));
            const float fa = filter[pos], fb = filter[pos + (H / 2) * Wh];
            wa[it] = filter_weight(fa);
            wb[it] = filter_weight(fb);
            neg = neg || fa < 0.0f || fb < 0.0f;
        }
        const bool wave_neg = __builtin_amdgcn_ballot_w64(neg) != 0;  // uniform: some slot of this wave has a negative filter value
        c32 ctw[CN1];  // e^{2 pi i n2 k1 / H}: wave-uniform, loop-invariant
#pragma unroll
        for (int k1 = 0; k1 < CN1; ++k1) ctw[k1] = c_tw256[((wv * k1) * (256 / H)) & 255];
        // lane L holds column kx = L + 1; the slot of kx = M (lane 63) is drawn and discarded, that lane transforms the packed column 0
        const int col = (lane + 1) & (M - 1);
        int64_t unit = blockIdx.x;
        int gp = 0;
        for (int j = 0; j < n; ++j) {
            c32* const A = PLANES + (j & 1) * BUF;
            c32 v[CN1];
            SONAR_PIPE_STAMP(0);
            if (gp == 0) {
                const GroupWalk gw(unit, group, split);
                if (j == per_unit) {  // the second unit: states left in LDS by the transforming team (visible: barriers since)
                    const uint4 rt = SEED_RT[tid];
                    rng.R = Mwc{rt.x, rt.y};
                    rng.T = Mwc{rt.z, rt.w};
                } else {
                    rng = spectrum_seed<true>(seed, stream_id, group0 + gw.grp, tid, false);  // the edge stream is the other team's
                }
                for (int i = 0; i < gw.first; ++i) skip_plane<H, W, true, false>(rng, tid);
            }
            if constexpr (SONAR_PIPE_DECOUPLE) {
                // the whole draw in one stretch: the only things this team needs from the other one are the free buffer (its row pass a
                // has read Y) and the packed column -- both behind the transforming team's second exchange of this iteration
                draw_chunk_regs<H, W, 0, ITER>(rng, v, wa, wb);
                if (wave_neg) draw_chunk_signs<H, W, 0, ITER>(filter, tid, v);
                pin_chunk<0, ITER>(v);
                SONAR_PIPE_STAMP(1);
                SONAR_PIPE_STAMP(2);
                team_wait(&tbar, 2 * (NT / 64) * (j + 1));
            } else {
            draw_chunk_regs<H, W, 0, E0>(rng, v, wa, wb);
            if (wave_neg) draw_chunk_signs<H, W, 0, E0>(filter, tid, v);
            pin_chunk<0, E0>(v);
            SONAR_PIPE_STAMP(1);
            __syncthreads();
            draw_chunk_regs<H, W, E0, E1>(rng, v, wa, wb);
            if (wave_neg) draw_chunk_signs<H, W, E0, E1>(filter, tid, v);
            pin_chunk<E0, E1>(v);
            SONAR_PIPE_STAMP(2);
            __syncthreads();
            draw_chunk_regs<H, W, E1, ITER>(rng, v, wa, wb);
            if (wave_neg) draw_chunk_signs<H, W, E1, ITER>(filter, tid, v);
            }
            if (lane == M - 1) {
#pragma unroll
                for (int n1 = 0; n1 < CN1; ++n1) v[n1] = Q[CN2 * n1 + wv];
            }
            // -------------------------------------------------------- columns, pass a: radix CN1 on the drawn registers (n2 = wave)
            idft<CN1>(v);
#pragma unroll
            for (int k1 = 1; k1 < CN1; ++k1) v[k1] = cmul(v[k1], ctw[k1]);
#pragma unroll
            for (int k1 = 0; k1 < CN1; ++k1) A[(CN2 * k1 + wv) * S + col] = v[k1];
            if (++gp == per_unit) {
                gp = 0;
                unit += gridDim.x;
            }
            if (j == 0) leave_wave_sums();
            SONAR_PIPE_STAMP(3);
            __syncthreads();
        }
        // nothing left to draw while the other team transforms the last plane (the same three barriers): the next call's statistics of
        // this workgroup's second unit; its edge columns wait in the seed area, which nobody reads any more
        [[maybe_unused]] const int j = n;  // (trace builds)
        SONAR_PIPE_STAMP(0);
        if (ahead && my_units == 2) {
            TeamStats<H, W> ts;
            c32* const edge = reinterpret_cast<c32*>(SEED_RT);
            const GroupWalk gw((int64_t)blockIdx.x + gridDim.x, group, split);
            ts.begin(filter, seed, next_stream, plane_offset, group, gw, tid, edge, wa, wb);
            ts.radii(0, SONAR_AHEAD_SPLIT_A, tid);
            SONAR_PIPE_STAMP(1);
            if constexpr (SONAR_PIPE_DECOUPLE) team_barrier(&dbar, NT / 64, lane);  // the unit's edge columns are complete
            else __syncthreads();
            ts.radii(SONAR_AHEAD_SPLIT_A, SONAR_AHEAD_SPLIT_B, tid);
            SONAR_PIPE_STAMP(2);
            if constexpr (!SONAR_PIPE_DECOUPLE) __syncthreads();
            ts.radii(SONAR_AHEAD_SPLIT_B, 4, tid);
            ts.products();
            ts.edges_and_wave_sums(edge, tid, sred);
            SONAR_PIPE_STAMP(3);
            __syncthreads();
        } else {
            SONAR_PIPE_STAMP(1);
            if constexpr (!SONAR_PIPE_DECOUPLE) __syncthreads();
            SONAR_PIPE_STAMP(2);
            if constexpr (!SONAR_PIPE_DECOUPLE) __syncthreads();
            SONAR_PIPE_STAMP(3);
            __syncthreads();
        }
    } else {
        // ------------------------------------------------------------------------------------------------ transforming team
        // wave wv owns residue n2 = wv in the twiddled row pass: every twiddle is wave-uniform and loop-invariant (scalar registers)
        __builtin_amdgcn_s_setprio(SONAR_PIPE_PRIO_FFT);
        int64_t unit = blockIdx.x;
        int gp = 0;
        // The two edge columns kx = 0, M of the plane being DRAWN are this team's (round 5; they were the drawing team's, which every
        // barrier waited for): rows ky = tid by waves 0-1 in phase 1, the packed column Q[ky] = sym(Z0)[ky] + i sym(ZM)[ky] from them
        // by waves 6-7 in phase 2 -- the drawing team's lane M - 1 reads Q in phase 3.  Same stream (slot ky's E), same arithmetic.
        c32* const T0 = EDGE;
        c32* const TM = EDGE + H;
        c32* const Q = EDGE + 2 * H;
        const float f0 = tid < H ? filter[tid * Wh] : 0.0f, fm = tid < H ? filter[tid * Wh + M] : 0.0f;
        Mwc rngE{0, 1}, rngE2{0, 1};
        int64_t eunit = blockIdx.x;
        int egp = 0;
        auto draw_edges = [&](int je) {  // plane je's edge rows (je < n)
            if (tid < H && je < n) {  // whole waves
                if (egp == 0) {
                    const GroupWalk gw(eunit, group, split);
                    rngE = je == per_unit ? rngE2 : spectrum_seed<false>(seed, stream_id, group0 + gw.grp, tid, true).E;
                    for (int i = 0; i < 3 * gw.first; ++i) rngE.next();
                }
                const uint32_t r0 = rngE.next();
                const uint32_t rm = rngE.next();
                const uint32_t t = rngE.next();
                T0[tid] = drawn_elem(r0, angle_lo(t), f0);
                TM[tid] = drawn_elem(rm, angle_hi(t), fm);
                if (++egp == per_unit) {
                    egp = 0;
                    eunit += gridDim.x;
                }
            }
        };
        auto pack_edges = [&](int je) {
            if (tid >= NT - H && je < n) {
                const int ky = tid - (NT - H), kn = (H - ky) & (H - 1);
                const c32 a = T0[ky], an = T0[kn], b = TM[ky], bn = TM[kn];
                Q[ky] = make_float2(0.5f * (a.x + an.x) - 0.5f * (b.y - bn.y), 0.5f * (a.y - an.y) + 0.5f * (b.x + bn.x));
            }
        };
        if (n > per_unit) {  // nothing to transform yet: seed the second unit's streams (R, T for the drawing team through LDS)
            const GroupWalk gw(unit + gridDim.x, group, split);
            const SpectrumRng g2 = spectrum_rng<H, true>(seed, stream_id, group0 + gw.grp, tid);
            SEED_RT[tid] = make_uint4(g2.R.x, g2.R.c, g2.T.x, g2.T.c);
            rngE2 = g2.E;
        }
        // iteration 0: the first plane is being drawn.  The next call's statistics of this workgroup's first unit (edge columns in the
        // second plane buffer, untouched until iteration 1), over the iteration's three barriers
        [[maybe_unused]] const int j = 0;  // (trace builds; shadowed by the loop below)
        SONAR_PIPE_STAMP(0);
        int tb = 0;  // this team's exchanges so far
        auto exchange = [&]() {
            if constexpr (SONAR_PIPE_DECOUPLE) team_barrier(&tbar, (NT / 64) * ++tb, lane);
            else __syncthreads();
        };
        if (ahead && my_units >= 1) {
            TeamStats<H, W> ts;
            c32* const edge = PLANES + BUF;
            const GroupWalk gw(unit, group, split);
            if (blockIdx.x == 0)
                for (int64_t slot = nunits + tid; slot < kNPart; slot += NT) {
                    partials_next[2 * slot] = 0.0;
                    partials_next[2 * slot + 1] = 0.0;
                }
            __builtin_amdgcn_s_setprio(SONAR_AHEAD_PRIO);
            draw_edges(0);
            ts.begin(filter, seed, next_stream, plane_offset, group, gw, tid, edge);
            ts.radii(0, SONAR_AHEAD_SPLIT_C, tid);
            SONAR_PIPE_STAMP(1);
            exchange();
            pack_edges(0);
            ts.radii(SONAR_AHEAD_SPLIT_C, SONAR_AHEAD_SPLIT_D, tid);
            SONAR_PIPE_STAMP(2);
            exchange();
            ts.radii(SONAR_AHEAD_SPLIT_D, 4, tid);
            ts.products();
            ts.edges_and_wave_sums(edge, tid, sred);
            leave_wave_sums();
            SONAR_PIPE_STAMP(3);
            __syncthreads();
            TeamStats<H, W>::store(sred, tid, partials_next, unit);
            __builtin_amdgcn_s_setprio(SONAR_PIPE_PRIO_FFT);
        } else {
            draw_edges(0);
            SONAR_PIPE_STAMP(1);
            exchange();
            pack_edges(0);
            SONAR_PIPE_STAMP(2);
            exchange();
            leave_wave_sums();
            SONAR_PIPE_STAMP(3);
            __syncthreads();
        }
        for (int j = 1; j <= n; ++j) {  // iteration j transforms plane j - 1 (and prepares the edge columns of plane j)
            constexpr int NW = 8;
            const bool work = j >= 1;
            c32* const X = PLANES + ((j + 1) & 1) * BUF;  // plane j - 1 (pass a's output)
            c32* const Y = PLANES + (j & 1) * BUF;        // free until the drawing team writes plane j in phase 3
            SONAR_PIPE_STAMP(0);
            if (work && !(SONAR_PIPE_SKIP & 16)) pipe_col_b<H, W, NW>(X, Y, wv, lane);
            draw_edges(j);
            if (j == 1) decide_in_wave0();
            SONAR_PIPE_STAMP(1);
            exchange();
            if (work && !(SONAR_PIPE_SKIP & 8)) pipe_row_a<H, W, NW>(Y, X, wv, lane);
            pack_edges(j);
            SONAR_PIPE_STAMP(2);
            exchange();
            if (work) {
                if (j == 1) pick_up_decision();
                const GroupWalk gw(unit, group, split);
                float* const oplane = out + (gw.grp * group + gw.first + gp) * (int64_t)H * W;
                pipe_row_b<H, W, NW, STATS, NORM>(X, oplane, wv, lane, scale, nm, nc, s, q);
                if (++gp == per_unit) {
                    gp = 0;
                    unit += gridDim.x;
                }
            }
            SONAR_PIPE_STAMP(3);
            __syncthreads();
        }
    }
    if (ahead) {
        __syncthreads();  // the drawing team's wave sums are in LDS
        if (drawer && my_units == 2) TeamStats<H, W>::store(sred, tid, partials_next, (int64_t)blockIdx.x + gridDim.x);
    }
    if constexpr (STATS) write_partial<NALL>(s, q, partials, red);
}

// Statistics of the output WITHOUT computing it (Parseval, ortho-normalised transform):
//   sum   x  = sqrt(H W) * Re(Zf[0][0])
//   sum x^2  = sum_ky ( |sym(Zf[:,0])[ky]|^2 + |sym(Zf[:,M])[ky]|^2 ) + 2 sum_ky sum_{0<kx<M} |Zf[ky][kx]|^2
// (only the Hermitian-symmetric part of the kx = 0 and kx = M columns survives the c2r stage).
// workgroup `bid` of `nb`: the statistics kernel's grid, or the trailing workgroups of a phase-serial generate launch that computes the
// NEXT call's statistics beside this call's planes (power_irfft2_kernel, look-ahead at the launch-bound batch sizes)
template <int H, int W>
__device__ __forceinline__ void power_stats_body(const float* __restrict__ filter, int64_t planes, uint64_t seed, uint64_t stream_id,
                                                 int64_t plane_offset, int group, int split, double* partials, int64_t bid, int64_t nb,
                                                 c32 (*EDGE)[2][H] /* [plane of the batch][kx = 0 | kx = M][ky] */, double* red) {
    constexpr int NT = plane_threads<H, W>(), M = W / 2, Wh = M + 1;
    constexpr int NB = kStatsBatch;
    const int tid = threadIdx.x;
    double s = 0.0, q = 0.0;
    auto edge_terms = [&](int p) {
        float edge = 0.0f;
        for (int ky = tid; ky < H; ky += NT) {
            const int kn = (H - ky) & (H - 1);
            const c32 a = EDGE[p][0][ky], an = EDGE[p][0][kn], b = EDGE[p][1][ky], bn = EDGE[p][1][kn];
            const float ar = 0.5f * (a.x + an.x), ai = 0.5f * (a.y - an.y), br = 0.5f * (b.x + bn.x), bi = 0.5f * (b.y - bn.y);
            edge += (ar * ar + ai * ai) + (br * br + bi * bi);
            if (ky == 0) s += (double)(sqrtf((float)H * (float)W) * ar);
        }
        q += (double)edge;
    };
    // A thread meets the same (ky, kx) in every plane: its weights -ln2 f^2 (|z f|^2 = f^2 rho^2 = -ln2 f^2 log2 u, the radius
    // word alone) live in registers; the discarded kx = M slots weigh 0.
    constexpr int LM = draw_shift<W>(), PAIRS = (H / 2) * M, ITER = draw_iters<H, W>();
    float wgt[2 * ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int p = tid + it * NT, ky = p >> LM, kx = 1 + (p & (M - 1));
        const bool live = p < PAIRS && kx < M;
        wgt[it] = filter_weight(live ? filter[ky * Wh + kx] : 0.0f);
        wgt[it + ITER] = filter_weight(live ? filter[(ky + H / 2) * Wh + kx] : 0.0f);
    }
    const float f0 = tid < H ? filter[tid * Wh] : 0.0f, fm = tid < H ? filter[tid * Wh + M] : 0.0f;
    for (int64_t unit = bid; unit < (split ? planes : planes / group); unit += nb) {
        const GroupWalk gw(unit, group, split);
        SpectrumRng rng = spectrum_rng<H, false>(seed, stream_id, plane_offset / group + gw.grp, tid);
        for (int i = 0; i < gw.first; ++i) skip_plane<H, W, false>(rng, tid);
        for (int g0 = 0; g0 < gw.count; g0 += NB) {
            const int cnt = min(NB, gw.count - g0);
            // The planes of a batch meet the same weight at the same slot, so sum_planes w log2 u = w log2(prod_planes u): ONE logarithm
            // per slot and batch instead of one per plane (the product of four u in [2^-23, 1] stays a normal float, and its rounding
            // error, 3 x 2^-24 relative, is below the logarithm's own).
            float prod[2 * ITER];
#pragma unroll
            for (int it = 0; it < 2 * ITER; ++it) prod[it] = 1.0f;
            for (int b = 0; b < cnt; ++b) {
                draw_plane<H, W, false>(
                    rng, tid,
                    [&](uint32_t r0, uint32_t rm, uint32_t t) {
                        EDGE[b][0][tid] = drawn_elem(r0, angle_lo(t), f0);
                        EDGE[b][1][tid] = drawn_elem(rm, angle_hi(t), fm);
                    },
                    [&](int it, int, uint32_t ra, uint32_t rb, uint32_t) {
                        prod[it] *= 2.0f - unit_mantissa(ra);
                        prod[it + ITER] *= 2.0f - unit_mantissa(rb);
                    });
            }
            float acc = 0.0f;
#pragma unroll
            for (int it = 0; it < 2 * ITER; ++it) acc = __builtin_fmaf(wgt[it], __builtin_amdgcn_logf(prod[it]), acc);
            q += 2.0 * (double)acc;
            __syncthreads();  // the batch's edge columns are complete
            for (int b = 0; b < cnt; ++b) edge_terms(b);
            __syncthreads();  // ... and read, before the next batch overwrites them
        }
    }
    write_partial_at<NT>(s, q, partials, red, (int)bid, (int)nb);
}

template <int H, int W>
__global__ void __launch_bounds__((plane_threads<H, W>())) power_stats_kernel(const float* __restrict__ filter, int64_t planes, uint64_t seed,
                                                                   uint64_t stream_id, int64_t plane_offset, int group, int split,
                                                                   double* partials) {
    kernarg_touch_for(filter, planes, seed, stream_id, plane_offset, group, split, partials);
    __shared__ c32 EDGE[kStatsBatch][2][H];
    __shared__ double red[2 * plane_threads<H, W>() / 64];
    power_stats_body<H, W>(filter, planes, seed, stream_id, plane_offset, group, split, partials, blockIdx.x, gridDim.x, EDGE, red);
}

// the spectrum draw_plane yields for (seed, stream_id, plane_offset, group), unit filter: zout[planes][H][W/2+1] complex64
template <int H, int W>
__global__ void __launch_bounds__((plane_threads<H, W>())) power_spectrum_kernel(float* zout, int64_t planes, uint64_t seed, uint64_t stream_id,
                                                                      int64_t plane_offset, int group, int split) {
    kernarg_touch_for(zout, planes, seed, stream_id, plane_offset, group, split);
    constexpr int M = W / 2, Wh = M + 1, NC = H * Wh;
    const int tid = threadIdx.x;
    for (int64_t unit = blockIdx.x; unit < (split ? planes : planes / group); unit += gridDim.x) {
        const GroupWalk gw(unit, group, split);
        SpectrumRng rng = spectrum_rng<H, true>(seed, stream_id, plane_offset / group + gw.grp, tid);
        for (int i = 0; i < gw.first; ++i) skip_plane<H, W, true>(rng, tid);
        for (int gp = gw.first; gp < gw.first + gw.count; ++gp) {
            c32* zp = reinterpret_cast<c32*>(zout) + (gw.grp * group + gp) * NC;
            draw_plane<H, W, true>(
                rng, tid,
                [&](uint32_t r0, uint32_t rm, uint32_t t) {
                    zp[tid * Wh] = unit_complex_normal(r0, angle_lo(t));
                    zp[tid * Wh + M] = unit_complex_normal(rm, angle_hi(t));
                },
                [&](int, int p, uint32_t ra, uint32_t rb, uint32_t t) {
                    const int ky = p >> draw_shift<W>(), kx = 1 + (p & (M - 1));
                    if (kx < M) {
                        zp[ky * Wh + kx] = unit_complex_normal(ra, angle_lo(t));
                        zp[(ky + H / 2) * Wh + kx] = unit_complex_normal(rb, angle_hi(t));
                    }
                });
        }
    }
}

template <int H, int W>
static int power_grid(int64_t planes, bool owns_partials = true) {
    using C = PlaneCfg<H, W>;
    static_assert(C::kLdsBytes + 256 <= 160 * 1024, "plane does not fit in LDS");
    // blocks/CU by LDS; persistent grid of resident blocks (<= kNPart when each owns a partial slot)
    // 16 waves per CU at the 128-row kernels' 128-VGPR budget; the smaller planes' kernels take 64 registers: 32 waves
    static const int waves = [] { const char* e = getenv("SONAR_POWER_WAVES"); return e ? atoi(e) : 0; }();
    const int threads_cu = waves > 0 ? waves * 64 : (H >= 128 ? 1024 : 2048);
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(threads_cu / plane_threads<H, W>(), (160 * 1024) / (C::kLdsBytes + 256)));
    const int64_t g = std::min<int64_t>(planes, (int64_t)256 * per_cu);
    return (int)(owns_partials ? std::min<int64_t>(g, kNPart) : g);
}

// process-wide: the pipelined generate kernel (default) or the phase-serial one for every batch size -- same streams, same bits;
// SONAR_POWER_PIPE=0 in the environment or sonar_power_pipeline(0) (A/B timing, and the test that the two kernels agree bit for bit)
static int& pipe_switch() {
    static int on = [] { const char* e = getenv("SONAR_POWER_PIPE"); return (e && e[0] == '0') ? 0 : 1; }();
    return on;
}
static bool pipe_enabled() { return pipe_switch() != 0; }
// the pipelined kernel's work units for `planes` planes in RNG groups of `group`
static int64_t pipe_units(int64_t planes, int group, int* psplit) {
    *psplit = group > 1 && planes / group < 256 ? 1 : 0;  // fewer RNG groups than CUs: single planes as units
    return *psplit ? planes : planes / group;
}

template <int H, int W>
static int launch_power(int what, const float* z, const float* filter, float* out, int64_t planes, uint64_t seed,
                        uint64_t stream_id, int64_t plane_offset, int group, double* partials, NormArgs na, hipStream_t st,
                        Ahead ah = Ahead()) {
    // Generated 128 x 128 planes, more than one per CU: the pipelined kernel (one 16-wave workgroup per CU, the draw of plane j + 1
    // under the transforms of plane j; same streams, same bits).  SONAR_POWER_PIPE=0 keeps the phase-serial kernel (A/B timing).
    if constexpr (H == 128 && W == 128) {
        if (pipe_enabled() && z == nullptr && (what == 0 || what == 1) && planes > 256) {
            int psplit;
            const int64_t units = pipe_units(planes, group, &psplit);
            const int pg = (int)std::min<int64_t>(units, 256);
            if (what == 1) {
                if (!ah.have_stats)
                    hipLaunchKernelGGL((power_stats_kernel<H, W>), dim3(std::min<int64_t>(units, kNPart)), dim3(plane_threads<H, W>()), 0, st, filter,
                                       planes, seed, stream_id, plane_offset, group, psplit, partials);
                hipLaunchKernelGGL((power_pipe_kernel<H, W, false, true>), dim3(pg), dim3(1024), 0, st, filter, out, planes, seed, stream_id,
                                   plane_offset, group, psplit, (double*)nullptr, na, ah.next_stream, ah.next);
            } else if (partials) {
                hipLaunchKernelGGL((power_pipe_kernel<H, W, true, false>), dim3(pg), dim3(1024), 0, st, filter, out, planes, seed, stream_id,
                                   plane_offset, group, psplit, partials, na, (uint64_t)0, (double*)nullptr);
            } else {
                hipLaunchKernelGGL((power_pipe_kernel<H, W, false, false>), dim3(pg), dim3(1024), 0, st, filter, out, planes, seed, stream_id,
                                   plane_offset, group, psplit, partials, na, (uint64_t)0, (double*)nullptr);
            }
            return check_launch("sonar_power_*");
        }
    }
    // too few RNG groups to fill the chip: one workgroup per plane (it fast-forwards the group's streams), same values
    const int split = group > 1 && planes / group < 2 * 256 ? 1 : 0;
    const int64_t ngroups = split ? planes : planes / group;  // work units
    const dim3 blk(plane_threads<H, W>());
#define SONAR_PW(G, ST, NM, PART) \
    hipLaunchKernelGGL((power_irfft2_kernel<H, W, G, ST, NM>), dim3(power_grid<H, W>(ngroups, ST)), blk, 0, st, z, filter, out, planes, seed, stream_id, plane_offset, group, split, PART, na, StatsAhead())
    if (what == 4) {
        SONAR_PW(3, false, false, nullptr);
    } else if (what == 3) {
        if constexpr (H == 128 && W == 128 && SONAR_SF_V2) {
            const dim3 grid(power_grid<H, W>(planes, partials != nullptr));
            if (partials) hipLaunchKernelGGL((spectral_filter128_kernel<H, W, true>), grid, blk, 0, st, z, filter, out, planes, partials);
            else hipLaunchKernelGGL((spectral_filter128_kernel<H, W, false>), grid, blk, 0, st, z, filter, out, planes, partials);
        } else if (partials) SONAR_PW(2, true, false, partials); else SONAR_PW(2, false, false, partials);
    } else if (what == 2) {
        hipLaunchKernelGGL((power_spectrum_kernel<H, W>), dim3(std::min<int64_t>(ngroups, 2048)), blk, 0, st, out, planes, seed, stream_id, plane_offset, group, split);
    } else if (what == 1) {
        if (!ah.have_stats)
            hipLaunchKernelGGL((power_stats_kernel<H, W>), dim3(std::min<int64_t>(ngroups, kNPart)), blk, 0, st, filter, planes, seed, stream_id,
                               plane_offset, group, split, partials);
        if (ah.next) {  // sonar_power_noise_ahead_ok: every workgroup resident at once -- the next call's statistics in the same launch
            StatsAhead sa;
            sa.partials = ah.next;
            sa.stream_id = ah.next_stream;
            sa.main_blocks = power_grid<H, W>(ngroups, false);
            hipLaunchKernelGGL((power_irfft2_kernel<H, W, 1, false, true>), dim3(sa.main_blocks + (int)std::min<int64_t>(ngroups, kNPart)), blk, 0, st, z,
                               filter, out, planes, seed, stream_id, plane_offset, group, split, (double*)nullptr, na, sa);
        } else {
            SONAR_PW(1, false, true, nullptr);
        }
    } else if (z == nullptr) {
        if (partials) SONAR_PW(1, true, false, partials); else SONAR_PW(1, false, false, partials);
    } else {
        if (partials) SONAR_PW(0, true, false, partials); else SONAR_PW(0, false, false, partials);
    }
#undef SONAR_PW
    return check_launch("sonar_power_*");
}

// C x C channel mixer (py/nodes/powernoise.py:96-101): out[b][i][p] = sum_j mixer[i][j] * in[b][j][p]
constexpr int kMaxMixC = 32;
template <bool STATS>
__global__ void __launch_bounds__(kBlock) channel_mix_kernel(const float* __restrict__ in,
                                                              const float* __restrict__ mixer, float* out, int64_t B,
                                                              int C, int64_t hw, double* partials) {
    kernarg_touch_for(in, mixer, out, B, C, hw, partials);
    __shared__ double red[2 * kBlock / 64];
    __shared__ float m[kMaxMixC * kMaxMixC];
    for (int i = threadIdx.x; i < C * C; i += kBlock) m[i] = mixer[i];
    __syncthreads();
    double s = 0.0, q = 0.0;
    const int64_t total = B * hw;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t b = i / hw, p = i - b * hw;
        const float* src = in + b * C * hw + p;
        float* dst = out + b * C * hw + p;
        for (int c = 0; c < C; ++c) {
            // same accumulation order as a row-times-column product: j ascending
            float acc = 0.0f;
            for (int j = 0; j < C; ++j) acc = __builtin_fmaf(m[c * C + j], src[(int64_t)j * hw], acc);
            dst[(int64_t)c * hw] = acc;
            if constexpr (STATS) {
                const double d = acc;
                s += d; q += d * d;
            }
        }
    }
    if constexpr (STATS) write_partial<kBlock>(s, q, partials, red);
}

}  // namespace sonar

#ifdef SONAR_PW_ONLY_128  // profiling builds (scratch/pw_build_variants.sh): the 128 x 128 kernels alone compile in under a minute
namespace sonar {
static bool any_plane_ok(int64_t, int64_t) { return false; }
static bool block_plane_ok(int64_t, int64_t) { return false; }
static bool any_ahead_ok(int64_t, int64_t, int64_t, int) { return false; }
static int launch_power_any(int, const float*, const float*, float*, int64_t, int64_t, int64_t, uint64_t, uint64_t, int64_t, int, double*, NormArgs, hipStream_t, Ahead) { return SONAR_ERR_UNSUPPORTED; }
static int launch_power_block(int, const float*, float*, float*, int64_t, int64_t, int64_t, uint64_t, uint64_t, int64_t, int, double*, NormArgs, hipStream_t) { return SONAR_ERR_UNSUPPORTED; }
}  // namespace sonar
bool sonar_lines_rows_r2c(const float*, float*, int64_t, int64_t, hipStream_t) { return false; }
bool sonar_lines_cols(const float*, const float*, float*, int64_t, int64_t, int64_t, int, hipStream_t) { return false; }
bool sonar_lines_rows_c2r(const float*, float*, int64_t, int64_t, float, double*, hipStream_t) { return false; }
bool sonar_lines_rows_c2r_norm(const float*, float*, int64_t, int64_t, float, double*, const sonar::NormArgs*, hipStream_t) { return false; }
#else
#include "power_any.h"
#include "power_block.h"
#endif

using namespace sonar;

static int power_dispatch(int what, const float* z, const float* filter, float* out, int64_t planes, int64_t H, int64_t W,
                          uint64_t seed, uint64_t stream_id, int64_t plane_offset, int group, double* partials, NormArgs na,
                          hipStream_t st, Ahead ah = Ahead()) {
    const bool gen = what == 1 || what == 2 || (what == 0 && z == nullptr);
    if (!gen) group = 1;
    SONAR_REQUIRE(group >= 1 && planes % group == 0 && plane_offset % group == 0, SONAR_ERR_ARG,
                  "sonar_power_*: planes (%lld) and plane_offset (%lld) must be multiples of the RNG group (%d)", (long long)planes,
                  (long long)plane_offset, group);
    SONAR_REQUIRE(group <= kMaxRngGroup, SONAR_ERR_UNSUPPORTED, "sonar_power_*: RNG groups of at most %d planes (got %d)", kMaxRngGroup, group);
#define SONAR_CASE(HH, WW) \
    if (H == HH && W == WW) return launch_power<HH, WW>(what, z, filter, out, planes, seed, stream_id, plane_offset, group, partials, na, st, ah)
    SONAR_CASE(128, 128);
#ifndef SONAR_PW_ONLY_128
    SONAR_CASE(64, 64);
    SONAR_CASE(32, 32);
    SONAR_CASE(16, 16);
    SONAR_CASE(256, 128);
    SONAR_CASE(128, 256);
    SONAR_CASE(128, 64);
    SONAR_CASE(64, 128);
    SONAR_CASE(64, 32);
    SONAR_CASE(32, 64);
    SONAR_CASE(256, 64);
    SONAR_CASE(64, 256);
#endif
#undef SONAR_CASE
    if (what == 4) {
        set_error("sonar_rfft2_f32: power-of-two planes from 16 x 16 to 256 x 128 only (got %lld x %lld)", (long long)H, (long long)W);
        return SONAR_ERR_UNSUPPORTED;
    }
    if (any_plane_ok(H, W)) return launch_power_any(what, z, filter, out, planes, H, W, seed, stream_id, plane_offset, group, partials, na, st, ah);
    set_error("sonar_power_*: unsupported plane %lld x %lld (even sizes whose half-spectrum fits in LDS: H <= 512, W <= 1024, about 19k complex values)",
              (long long)H, (long long)W);
    return SONAR_ERR_UNSUPPORTED;
}

#ifdef SONAR_PW_TRACE
extern "C" int sonar_debug_pw_trace(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(sonar::g_pw_trace), sizeof(sonar::g_pw_trace));
}
extern "C" int sonar_debug_pipe_trace(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(sonar::g_pipe_trace), sizeof(sonar::g_pipe_trace));
}
#endif

extern "C" int sonar_power_plane_kind(int64_t H, int64_t W) {
    static const int fast[][2] = {{128, 128}, {64, 64}, {32, 32}, {16, 16}, {256, 128}, {128, 256}, {128, 64}, {64, 128}, {64, 32}, {32, 64},
                                  {256, 64}, {64, 256}};
    for (const auto& f : fast)
        if (H == f[0] && W == f[1]) return 1;
    if (any_plane_ok(H, W)) return 2;
    return block_plane_ok(H, W) ? 4 : 0;  // 4: generated in column blocks through a workspace (sonar_power_block_f32); 3 is the host's name for the direct passes
}

extern "C" int64_t sonar_power_block_ws_bytes(int64_t planes, int64_t H, int64_t W) {
    if (planes < 0 || sonar_power_plane_kind(H, W) != 4) return -1;
    return planes * H * (W / 2 + 1) * (int64_t)sizeof(c32);
}

extern "C" int sonar_power_block_f32(const float* filter, float* ws, float* out, int64_t planes, int64_t H, int64_t W, uint64_t seed, uint64_t stream_id,
                                     int64_t plane_offset, int rng_group, int mode, float factor, float threshold_std_devs, double* partials,
                                     void* stream) {
    SONAR_REQUIRE(ws && planes >= 0 && plane_offset >= 0 && mode >= 0 && mode <= 2 && (mode == 2 || (filter && out)) && (mode != 1 || partials),
                  SONAR_ERR_ARG, "sonar_power_block_f32: bad argument");
    SONAR_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 7u) == 0 && (mode == 2 || (reinterpret_cast<uintptr_t>(out) & 7u) == 0), SONAR_ERR_ARG,
                  "sonar_power_block_f32: misaligned buffer");
    SONAR_REQUIRE(sonar_power_plane_kind(H, W) == 4, SONAR_ERR_UNSUPPORTED,
                  "sonar_power_block_f32: %lld x %lld is not a column-block plane (sonar_power_plane_kind != 4)", (long long)H, (long long)W);
    SONAR_REQUIRE(rng_group >= 1 && rng_group <= kMaxRngGroup && planes % rng_group == 0 && plane_offset % rng_group == 0, SONAR_ERR_ARG,
                  "sonar_power_block_f32: planes (%lld) and plane_offset (%lld) must be multiples of the RNG group (%d, at most %d)", (long long)planes,
                  (long long)plane_offset, rng_group, kMaxRngGroup);
    if (planes == 0) return SONAR_OK;
    return launch_power_block(mode, filter, ws, out, planes, H, W, seed, stream_id, plane_offset, rng_group, partials,
                              NormArgs{partials, planes * H * W, factor, threshold_std_devs}, (hipStream_t)stream);
}

extern "C" int sonar_power_irfft2_f32(const float* z, const float* filter, float* out, int64_t planes, int64_t H,
                                      int64_t W, uint64_t seed, uint64_t stream_id, int64_t plane_offset, int rng_group,
                                      double* partials, void* stream) {
    SONAR_REQUIRE(filter && out && planes >= 0 && H > 0 && W > 0 && plane_offset >= 0, SONAR_ERR_ARG,
                  "sonar_power_irfft2_f32: bad argument");
    SONAR_REQUIRE((reinterpret_cast<uintptr_t>(out) & 7u) == 0 && (z == nullptr || (reinterpret_cast<uintptr_t>(z) & 7u) == 0),
                  SONAR_ERR_ARG, "sonar_power_irfft2_f32: misaligned buffer");
    if (planes == 0) return SONAR_OK;
    return power_dispatch(0, z, filter, out, planes, H, W, seed, stream_id, plane_offset, rng_group, partials,
                          NormArgs{nullptr, 0, 1.0f, 0.0f}, (hipStream_t)stream);
}

extern "C" int sonar_power_noise_f32(const float* filter, float* out, int64_t planes, int64_t H, int64_t W, uint64_t seed,
                                     uint64_t stream_id, int64_t plane_offset, int rng_group, float factor,
                                     float threshold_std_devs, double* partials, void* stream) {
    SONAR_REQUIRE(filter && out && partials && planes >= 0 && H > 0 && W > 0 && plane_offset >= 0, SONAR_ERR_ARG,
                  "sonar_power_noise_f32: bad argument");
    SONAR_REQUIRE((reinterpret_cast<uintptr_t>(out) & 7u) == 0, SONAR_ERR_ARG, "sonar_power_noise_f32: misaligned buffer");
    if (planes == 0) return SONAR_OK;
    return power_dispatch(1, nullptr, filter, out, planes, H, W, seed, stream_id, plane_offset, rng_group, partials,
                          NormArgs{partials, planes * H * W, factor, threshold_std_devs}, (hipStream_t)stream);
}

extern "C" int sonar_power_pipeline(int enable) {
    const int before = pipe_switch();
    if (enable >= 0) pipe_switch() = enable != 0;
    return before;
}

extern "C" int sonar_power_noise_ahead_ok(int64_t planes, int64_t H, int64_t W, int rng_group) {
    if (planes < 1 || rng_group < 1 || rng_group > kMaxRngGroup || planes % rng_group) return 0;
    if (sonar_power_plane_kind(H, W) == 2) return any_ahead_ok(planes, H, W, rng_group) ? 1 : 0;  // general-size planes at launch-bound sizes
    if (sonar_power_plane_kind(H, W) != 1) return 0;
    if (H != 128 || W != 128 || !pipe_enabled() || planes <= 256) {
        // the phase-serial kernel (launch_power): the next call's statistics are extra workgroups of the launch -- while all of them are
        // resident at once (at most one plane and one statistics workgroup per CU; beyond that the two-launch form is as fast)
        const int split = rng_group > 1 && planes / rng_group < 2 * 256 ? 1 : 0;
        return (split ? planes : planes / rng_group) <= 256 ? 1 : 0;
    }
    int psplit;
    const int64_t units = pipe_units(planes, rng_group, &psplit);
    // the look-ahead statistics walk at most four planes per unit (TeamStats::radii, the edge columns' area): larger RNG groups only
    // when the units are single planes
    if (!psplit && rng_group > kAheadMaxGroup) return 0;
    return units <= 2 * 256 ? 1 : 0;  // one look-ahead unit per team: at most two units per workgroup
}

extern "C" int sonar_power_noise_ahead_f32(const float* filter, float* out, int64_t planes, int64_t H, int64_t W, uint64_t seed,
                                           uint64_t stream_id, int64_t plane_offset, int rng_group, float factor,
                                           float threshold_std_devs, double* partials, int have_stats, uint64_t next_stream_id,
                                           double* partials_next, void* stream) {
    SONAR_REQUIRE(filter && out && partials && planes >= 0 && H > 0 && W > 0 && plane_offset >= 0, SONAR_ERR_ARG,
                  "sonar_power_noise_ahead_f32: bad argument");
    SONAR_REQUIRE((reinterpret_cast<uintptr_t>(out) & 7u) == 0 && partials_next != partials, SONAR_ERR_ARG,
                  "sonar_power_noise_ahead_f32: misaligned buffer, or the same workspace for both calls' statistics");
    SONAR_REQUIRE(sonar_power_noise_ahead_ok(planes, H, W, rng_group), SONAR_ERR_UNSUPPORTED,
                  "sonar_power_noise_ahead_f32: no look-ahead for %lld planes of %lld x %lld (ask sonar_power_noise_ahead_ok first)", (long long)planes,
                  (long long)H, (long long)W);
    Ahead ah;
    ah.have_stats = have_stats != 0;
    ah.next_stream = next_stream_id;
    ah.next = partials_next;
    return power_dispatch(1, nullptr, filter, out, planes, H, W, seed, stream_id, plane_offset, rng_group, partials,
                          NormArgs{partials, planes * H * W, factor, threshold_std_devs}, (hipStream_t)stream, ah);
}

extern "C" int sonar_spectral_filter_f32(const float* x, const float* filter, float* out, int64_t planes, int64_t H, int64_t W,
                                         double* partials, void* stream) {
    SONAR_REQUIRE(x && filter && out && x != out && planes >= 0 && H > 0 && W > 0, SONAR_ERR_ARG,
                  "sonar_spectral_filter_f32: bad argument (in-place not supported)");
    SONAR_REQUIRE((reinterpret_cast<uintptr_t>(out) & 7u) == 0 && (reinterpret_cast<uintptr_t>(x) & 7u) == 0, SONAR_ERR_ARG,
                  "sonar_spectral_filter_f32: misaligned buffer");
    if (planes == 0) return SONAR_OK;
    return power_dispatch(3, x, filter, out, planes, H, W, 0, 0, 0, 1, partials, NormArgs{nullptr, 0, 1.0f, 0.0f}, (hipStream_t)stream);
}

extern "C" int sonar_rfft2_f32(const float* x, float* z_out, int64_t planes, int64_t H, int64_t W, void* stream) {
    SONAR_REQUIRE(x && z_out && planes >= 0 && H > 0 && W > 0, SONAR_ERR_ARG, "sonar_rfft2_f32: bad argument");
    SONAR_REQUIRE((reinterpret_cast<uintptr_t>(z_out) & 7u) == 0 && (reinterpret_cast<uintptr_t>(x) & 7u) == 0, SONAR_ERR_ARG,
                  "sonar_rfft2_f32: misaligned buffer");
    if (planes == 0) return SONAR_OK;
    return power_dispatch(4, x, nullptr, z_out, planes, H, W, 0, 0, 0, 1, nullptr, NormArgs{nullptr, 0, 1.0f, 0.0f}, (hipStream_t)stream);
}

extern "C" int sonar_power_spectrum_f32(float* z_out, int64_t planes, int64_t H, int64_t W, uint64_t seed, uint64_t stream_id,
                                        int64_t plane_offset, int rng_group, void* stream) {
    SONAR_REQUIRE(z_out && planes >= 0 && H > 0 && W > 0 && plane_offset >= 0, SONAR_ERR_ARG, "sonar_power_spectrum_f32: bad argument");
    if (planes == 0) return SONAR_OK;
    return power_dispatch(2, nullptr, nullptr, z_out, planes, H, W, seed, stream_id, plane_offset, rng_group, nullptr,
                          NormArgs{nullptr, 0, 1.0f, 0.0f}, (hipStream_t)stream);
}

extern "C" int sonar_channel_mix_f32(const float* in, const float* mixer, float* out, int64_t B, int64_t C, int64_t hw,
                                     double* partials, void* stream) {
    SONAR_REQUIRE(in && mixer && out && B >= 0 && C > 0 && hw > 0 && in != out, SONAR_ERR_ARG,
                  "sonar_channel_mix_f32: bad argument (in-place not supported)");
    SONAR_REQUIRE(C <= kMaxMixC, SONAR_ERR_UNSUPPORTED, "sonar_channel_mix_f32: more than %d channels", kMaxMixC);
    if (B == 0) return SONAR_OK;
    const int g = (int)std::min<int64_t>(kNPart, grid_for(B * hw, kBlock));
    if (partials)
        hipLaunchKernelGGL((channel_mix_kernel<true>), dim3(g), dim3(kBlock), 0, (hipStream_t)stream, in, mixer, out, B,
                           (int)C, hw, partials);
    else
        hipLaunchKernelGGL((channel_mix_kernel<false>), dim3(g), dim3(kBlock), 0, (hipStream_t)stream, in, mixer, out, B,
                           (int)C, hw, partials);
    return check_launch("sonar_channel_mix_f32");
}
