// Shared device helpers of the power-law (coloured) rFFT noise kernels: complex arithmetic on register pairs, the power-of-two DFT
// codelets, plane geometry, the spectrum's random streams and draws.  Included by power_fft.hip (fixed-size and pipelined kernels,
// dispatch, C ABI) and by the translation units that instantiate the general-size kernels for particular plane sizes
// (power_buckets_*.hip): templates, inline device functions and TU-local constant tables only.
#pragma once
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "twiddles256.h"

namespace sonar {

using c32 = float2;

// complex arithmetic on native 2-vectors: one register pair per complex value, packed adds / multiplies / FMAs, swaps and sign
// flips as operand modifiers (scalar .x / .y expressions let the vectoriser pair halves of DIFFERENT values and pay for it in moves)
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f vv(c32 a) { return v2f{a.x, a.y}; }
__device__ __forceinline__ c32 cc(v2f a) { return make_float2(a.x, a.y); }
__device__ __forceinline__ c32 cadd(c32 a, c32 b) { return cc(vv(a) + vv(b)); }
__device__ __forceinline__ c32 csub(c32 a, c32 b) { return cc(vv(a) - vv(b)); }
__device__ __forceinline__ c32 cmul(c32 a, c32 b) {
    const v2f A = vv(a), B = vv(b);
    return cc(__builtin_elementwise_fma(A.yy, v2f{-B.y, B.x}, A.xx * B));
}
// a +- i b as ONE packed FMA: swap(b) * (-+1, +-1) + a (the swap is an operand modifier, the sign pair a scalar register pair; exact, so
// the same bits as the add).  Written as `a + {-b.y, b.x}` the compiler builds the rotated vector with a v_xor and a v_mov first:
// three instructions for every multiply-by-i butterfly, 128 extra vector instructions per plane and thread in the transforms.
__device__ __forceinline__ c32 cadd_i(c32 a, c32 b) { return cc(__builtin_elementwise_fma(vv(b).yx, v2f{-1.0f, 1.0f}, vv(a))); }   // a + i b
__device__ __forceinline__ c32 csub_i(c32 a, c32 b) { return cc(__builtin_elementwise_fma(vv(b).yx, v2f{1.0f, -1.0f}, vv(a))); }   // a - i b
__device__ __forceinline__ c32 cmul_i(c32 a) { return cc(vv(a).yx * v2f{-1.0f, 1.0f}); }  // a * (+i)
__device__ __forceinline__ c32 cscale(c32 a, float r) { return cc(vv(a) * r); }

// ---- register codelets: in-place inverse (sign +) DFTs, natural order in and out -------------
template <int N>
__device__ __forceinline__ void idft(c32 (&v)[N]);

template <>
__device__ __forceinline__ void idft<1>(c32 (&)[1]) {}

template <>
__device__ __forceinline__ void idft<2>(c32 (&v)[2]) {
    const c32 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
}

template <>
__device__ __forceinline__ void idft<4>(c32 (&v)[4]) {
    const c32 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
    const c32 t2 = cadd(v[1], v[3]), t3 = csub(v[1], v[3]);
    v[0] = cadd(t0, t2);
    v[2] = csub(t0, t2);
    v[1] = cadd_i(t1, t3);
    v[3] = csub_i(t1, t3);
}

template <>
__device__ __forceinline__ void idft<8>(c32 (&v)[8]) {
    constexpr float r = 0.70710678118654752f;
    c32 e[4] = {v[0], v[2], v[4], v[6]};
    c32 o[4] = {v[1], v[3], v[5], v[7]};
    idft<4>(e);
    idft<4>(o);
    const c32 t0 = o[0];
    const c32 t1 = cscale(cadd_i(o[1], o[1]), r);              // * e^{i pi/4}  = r (o + i o)
    const c32 t3 = cscale(csub_i(o[3], o[3]), -r);             // * e^{3 i pi/4} = r (i o - o) = -r (o - i o)
    v[0] = cadd(e[0], t0); v[4] = csub(e[0], t0);
    v[1] = cadd(e[1], t1); v[5] = csub(e[1], t1);
    v[2] = cadd_i(e[2], o[2]); v[6] = csub_i(e[2], o[2]);      // * i
    v[3] = cadd(e[3], t3); v[7] = csub(e[3], t3);
}

template <>
__device__ __forceinline__ void idft<16>(c32 (&v)[16]) {
    constexpr float r = 0.70710678118654752f, c1 = 0.92387953251128674f, s1 = 0.38268343236508977f;
    c32 e[8], o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        e[i] = v[2 * i];
        o[i] = v[2 * i + 1];
    }
    idft<8>(e);
    idft<8>(o);
    c32 t[8];
    t[0] = o[0];
    t[1] = cmul(o[1], make_float2(c1, s1));
    t[2] = cscale(cadd_i(o[2], o[2]), r);
    t[3] = cmul(o[3], make_float2(s1, c1));
    t[5] = cmul(o[5], make_float2(-s1, c1));
    t[6] = cscale(csub_i(o[6], o[6]), -r);
    t[7] = cmul(o[7], make_float2(-c1, s1));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i == 4) {  // * i: folded into the butterfly
            v[4] = cadd_i(e[4], o[4]);
            v[12] = csub_i(e[4], o[4]);
        } else {
            v[i] = cadd(e[i], t[i]);
            v[i + 8] = csub(e[i], t[i]);
        }
    }
}

// forward (sign -) DFT through the inverse codelet: F(v) = swap(I(swap(v))), swap = exchange Re / Im (free in registers)
template <int N>
__device__ __forceinline__ void fdft(c32 (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = make_float2(v[i].y, v[i].x);
    idft<N>(v);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = make_float2(v[i].y, v[i].x);
}
__device__ __forceinline__ c32 cmulc(c32 a, c32 b) {  // a * conj(b)
    return make_float2(__builtin_fmaf(a.x, b.x, a.y * b.y), __builtin_fmaf(a.y, b.x, -(a.x * b.y)));
}

constexpr int split_n1(int n) { return n >= 256 ? 16 : n >= 64 ? 8 : n == 32 ? 4 : n == 16 ? 2 : 1; }

#ifndef SONAR_FILTER_IN_PASS
#define SONAR_FILTER_IN_PASS 1  // FAST spectral filter: multiply by the filter in the inverse column pass a
#endif
#ifndef SONAR_ROW32_SPLIT8
#define SONAR_ROW32_SPLIT8 1
#endif
#ifndef SONAR_FFT_THREADS
#define SONAR_FFT_THREADS 512
#endif
#ifndef SONAR_FFT_WAVES
#define SONAR_FFT_WAVES 4
#endif
#ifndef SONAR_FFT_UNROLL
#define SONAR_FFT_UNROLL 1
#endif
#define SONAR_PRAGMA(x) _Pragma(#x)
#define SONAR_UNROLL_ITEMS SONAR_PRAGMA(unroll SONAR_FFT_UNROLL)
#ifndef SONAR_DRAW_UNROLL
#define SONAR_DRAW_UNROLL 4  // the FFT kernel's draw loop: 1 / 2 / 4 measured 69.8 / 69 / 66.5 us per step at B=512; 8 spills (128-VGPR budget)
#endif
#ifndef SONAR_FFT_TW_LDS
#define SONAR_FFT_TW_LDS 0  // measured: constant-memory (scalar) twiddles 78 us vs LDS table 125 us at B=512
#endif
#ifndef SONAR_FWD_UNI
#define SONAR_FWD_UNI 1  // forward passes: wave-uniform twiddles through scalar loads (1) or as broadcast reads of the LDS table (0)
#endif
#ifndef SONAR_PW_SKIP
#define SONAR_PW_SKIP 0  // profiling builds only (scratch/pw_passes.py): 1 draw, 2 column passes, 4 rows pass a, 8 rows pass b arithmetic, 16 global stores
#endif
#ifdef SONAR_PW_TRACE  // profiling builds: per-phase s_memtime stamps of wave 0 of every workgroup (scratch/pw_trace.py)
__device__ unsigned long long g_pw_trace[1024 * 8 * 12];
#define SONAR_STAMP(slot) do { if (tid == 0 && pidx < 8 && blockIdx.x < 1024) g_pw_trace[(blockIdx.x * 8 + pidx) * 12 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define SONAR_STAMP(slot) do { } while (0)
#endif
constexpr int kFftThreads = SONAR_FFT_THREADS;  // waves per block x two blocks per CU (LDS-bound)
// threads (= RNG thread slots) of a fixed-size plane's workgroup: 512 from 8192 values up, smaller planes take fewer so that
// every pass has work for all of them (64 x 64: 256; 32 x 32: 128; 16 x 16: 64) and more workgroups share a CU
template <int H, int W>
constexpr int plane_threads() { return H * W >= 8192 ? kFftThreads : H * W >= 2048 ? 256 : H * W >= 1024 ? 128 : 64; }

template <int H, int W>
struct PlaneCfg {
    static constexpr int M = W / 2;       // complex length of the c2r stage
    static constexpr int Wh = M + 1;      // half-spectrum width
    static constexpr int S = M + 1;       // LDS row stride (complex): odd -> rows hit distinct banks
    static constexpr int CN1 = split_n1(H), CN2 = H / CN1;
    // 64 x 64: rows of 32 complex values split 8 x 4, not 4 x 8 -- row pass b's RN1 lanes of a row store side by side, 64-byte runs
    // instead of 32 (79 -> 77.5 us generated, 95 -> 92 us filtered per 33.5 M values; the other heights with W = 64 do not gain)
    static constexpr int RN1 = (H == 64 && M == 32 && SONAR_ROW32_SPLIT8) ? 8 : split_n1(M), RN2 = M / RN1;
    // Column of element (k1, n2) of a row BETWEEN the two row passes (inverse: written by pass a, read by pass b; forward: written by
    // pass b', read by pass a').  The natural k = RN2 k1 + n2 puts the RN1 lanes of a row that pass b runs side by side (k1 = lane %
    // RN1) RN2 complex values = 16 dwords apart: with S = 1 mod 16 the 16 lanes of an LDS access group (ds_read2_b64 / ds_write2_b64:
    // 16 contiguous lanes, 32 banks) fall on 2 * 16 / RN1 bank pairs -- a 4-way conflict at M = 64.  Spreading k1 with stride
    // G = 16 / RN1 (the rows of a group fill the gaps: bank pair = row + G k1 + n2 % G) makes the access conflict-free; lanes = rows
    // accesses (the other pass) only see a different constant offset.
    static constexpr bool kRowSwizzle = RN1 <= 16 && 16 % RN1 == 0 && RN2 % (16 / RN1) == 0 && S % 16 == 1;
    static __host__ __device__ constexpr int rpos(int k1, int n2) {
        constexpr int G = kRowSwizzle ? 16 / RN1 : 1;
        return kRowSwizzle ? G * k1 + (n2 % G) + 16 * (n2 / G) : RN2 * k1 + n2;
    }
    // plane + raw columns 0 and M (side buffers) + twiddle table
    static constexpr int kLdsComplex = H * S + 2 * H + 256;
    static constexpr size_t kLdsBytes = (size_t)kLdsComplex * sizeof(c32);
};

// ---- on-device spectrum draws (generate mode) ---------------------------------------------------------------------
// Streams are keyed by (seed, stream_id, plane group, thread slot): a group is `group` consecutive global planes (4 when the
// channel count is a multiple of 4, else 1 -- chosen by the host from C alone, so every shard of a batch agrees) that one
// workgroup draws back to back, so the Philox seeding cost is paid once per group instead of once per plane.
// Three streams per slot:  R = radius words and T = angle words of the interior columns 0 < kx < W/2,
// E = both for the two edge columns kx = 0, W/2 (slot ky).  The statistics pass (Parseval) needs only R and E:
// |z|^2 = -ln(u_R) for a unit complex normal, so it skips the angle words and all of sqrt / sin / cos.
// Round 5: multiply-with-carry streams (common.h, Mwc) instead of xoshiro128 -- a third of the generator's cost per word.
// (Measured and dropped: one radius stream per element of a pair -- two short multiply chains instead of one long one -- changes nothing.)
struct SpectrumRng {
    Mwc R, T, E;
};

// The three streams of a (group, thread slot) start from ONE Philox4x32 counter block: R and T from the four words after the standard
// 10 rounds (the counter of rng_stream's tile 4 * group), E from the first two words after two more rounds (a keyed bijection of an
// already mixed block).  Seeding costs 10 rounds per slot (12 for the H edge slots); 20 multiplies at the cost of a shift each.
struct PhiloxBlock {
    uint32_t c0, c1, c2, c3, k0, k1;
    template <int ROUNDS>
    __device__ __forceinline__ void rounds() {
        constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const uint64_t p0 = (uint64_t)M0 * (uint64_t)c0;
            const uint64_t p1 = (uint64_t)M1 * (uint64_t)c2;
            const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
            const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
            const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
            c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
            k0 += W0; k1 += W1;
        }
    }
};

template <bool NEED_T>
__device__ __forceinline__ SpectrumRng spectrum_seed(uint64_t seed, uint64_t stream_id, int64_t ggroup, int tid, bool edge_slot) {
    const uint64_t tile = (uint64_t)ggroup << 2;
    PhiloxBlock b{(uint32_t)tile, (uint32_t)(tile >> 32), (uint32_t)stream_id, (uint32_t)((stream_id >> 32) << 16) ^ (uint32_t)tid,
                  (uint32_t)seed, (uint32_t)(seed >> 32)};
    SpectrumRng g;
    b.rounds<10>();
    g.R = Mwc::seeded(b.c0, b.c1);
    g.T = Mwc::seeded(b.c2, b.c3);
    g.E = Mwc{0, 1};
    if (edge_slot) {  // whole waves: the edge slots are the first H threads, H a multiple of 64 on every fixed-size plane
        b.rounds<2>();
        g.E = Mwc::seeded(b.c0, b.c1);
    }
    return g;
}

template <int H, bool NEED_T>
__device__ __forceinline__ SpectrumRng spectrum_rng(uint64_t seed, uint64_t stream_id, int64_t ggroup, int tid) {
    return spectrum_seed<NEED_T>(seed, stream_id, ggroup, tid, tid < H);
}

// interior element order: the pairs p in [0, (H/2) M) walk rows of M slots, ky = p / M, kx = 1 + p % M (M a power of two: shifts);
// slot `tid` draws the pairs p = tid, tid + NT, ... -- element (ky, kx) and its partner H/2 rows below, same column.  Per pair:
// two radius words from R, ONE angle word from T (low / high half).  The last slot of a row (kx = M) is drawn and DISCARDED
// (the kx = M column comes from E): 1/M more generator steps buy addresses that are affine in the iteration -- no index
// arithmetic, LDS / filter offsets become constant strides.  Callbacks: edge(r0, rm, t) and pair(it, p, ra, rb, t) -- radius
// words of element (ky, kx) and of its partner, `t` = both angles (low half / high half; 0 when !NEED_T).
template <int W>
constexpr int draw_shift() { int l = 0; while ((1 << l) < W / 2) ++l; return l; }
template <int H, int W>
constexpr int draw_iters() { return ((H / 2) * (W / 2) + plane_threads<H, W>() - 1) / plane_threads<H, W>(); }

template <int H, int W, bool NEED_T, int UNROLL = 0, bool EDGES = true, typename Edge, typename Pair>
__device__ __forceinline__ void draw_plane(SpectrumRng& g, int tid, Edge&& edge, Pair&& pair) {
    constexpr int NT = plane_threads<H, W>(), M = W / 2, PAIRS = (H / 2) * M, ITER = draw_iters<H, W>();
    constexpr int UNR = UNROLL > 0 ? UNROLL : ITER;  // 0 = full (the statistics pass indexes registers by `it`)
    if (EDGES && tid < H) {  // row ky = tid of the edge columns: radius word of kx = 0, of kx = M, then one angle word for both
        const uint32_t r0 = g.E.next();
        const uint32_t rm = g.E.next();
        const uint32_t t = g.E.next();
        edge(r0, rm, t);
    }
#pragma unroll UNR
    for (int it = 0; it < ITER; ++it) {
        const int p = tid + it * NT;
        if (PAIRS % NT == 0 || p < PAIRS) {
            const uint32_t ra = g.R.next();  // radius words keep bits 31..9 only
            const uint32_t rb = g.R.next();
            const uint32_t t = NEED_T ? g.T.next() : 0u;
            pair(it, p, ra, rb, t);
        }
    }
}

// advance the streams past one plane's draws without using them (a workgroup that starts in the middle of an RNG group)
template <int H, int W, bool NEED_T, bool EDGES = true>
__device__ __forceinline__ void skip_plane(SpectrumRng& g, int tid) {
    draw_plane<H, W, NEED_T, 0, EDGES>(g, tid, [](uint32_t, uint32_t, uint32_t) {}, [](int, int, uint32_t, uint32_t, uint32_t) {});
}

// A kernel's work units: whole RNG groups (one workgroup draws the group's planes back to back; the seeding is paid once
// per group), or -- `split`, chosen by the launcher when there are too few groups to fill the chip -- single planes, the
// workgroup fast-forwarding the group's streams to its plane.  Same values either way.
// Work units of a launch over RNG groups of `group` planes.  split = 0: a unit is a whole group; 1: a single plane (the workgroup
// fast-forwards the group's streams to it); 2 + F (round 5, kernels that say so): the first F groups whole, the planes of the others
// singly -- F = the groups of the launch's FULL rounds of workgroups, so that a tail of a few groups spreads over all workgroups as planes
// instead of costing a few of them a whole group each (530 groups on 512 workgroups: eight plane-times became 5.3).
__host__ __device__ inline int64_t group_units(int64_t planes, int group, int split) {
    if (split == 0) return planes / group;
    if (split == 1) return planes;
    const int64_t full = (int64_t)split - 2;
    return full + (planes - full * group);
}
struct GroupWalk {
    int64_t grp;
    int first, count;
    __device__ __forceinline__ GroupWalk(int64_t unit, int group, int split)
        : grp(split ? unit / group : unit), first(split ? (int)(unit % group) : 0), count(split ? 1 : group) {
        if (split >= 2) {
            const int64_t full = (int64_t)split - 2;
            if (unit < full) {
                grp = unit, first = 0, count = group;
            } else {
                const int64_t pl = unit - full;
                grp = full + pl / group, first = (int)(pl % group), count = 1;
            }
        }
    }
};

// ---- unit complex normal z = rho e^{i theta}, E|z|^2 = 1, times a filter value, from raw generator bits (about 20 instruction slots) ----
// radius: 23 random bits become the mantissa of a float m in [1, 2) in ONE v_alignbit; u = 2 - m is uniform on (0, 1] and
//   rho^2 = -ln u (the 1/sqrt(2) of "(a + ib) / sqrt 2" folded into the radius), so rho <= sqrt(23 ln 2) = 3.99 (5.65 sigma
//   per component).  The filter value f rides UNDER the square root: |z f|^2 = f^2 rho^2 = w log2 u with the weight
//   w = -ln2 f^2 -- one multiply for "- ln 2", the logarithm's base and the filter together, and exactly the term the Parseval
//   statistics sum (power_stats_body, TeamStats): rho_f = sqrt(w log2 u); the sign of f goes back on with v_bfi (copysign).
//   Round 5: up to round 4 the element was (rho cos, rho sin) f with rho = sqrt(-ln2 log2 u) -- four multiplies more per value; the
//   two forms differ in the last bit, a break of the generate-mode seeds like round 3's.
// angle: 16 random bits per value, one 32-bit draw for two values.  v_sin / v_cos take revolutions, are periodic and accept
//   |x| <= 256, so the bits are dropped into the mantissa of a float in [128, 256) where the low 16 mantissa bits weigh
//   2^-1 .. 2^-16 revolutions and whatever sits above them whole revolutions: ONE instruction per angle (v_and_or for the low
//   half, its junk bits 22..16 being whole turns; v_alignbit for the high half) instead of mask / shift + or.  65536 directions x a
//   23-bit radius is far below fp32 output resolution after the 8192-term FFT sums.
constexpr float kNegLn2 = -0.6931471805599453f;
// (unit_mantissa, angle_lo, angle_hi: common.h -- the Brownian tile kernel draws its normals the same way since round 6)
__device__ __forceinline__ float log2_u(uint32_t r) { return __builtin_amdgcn_logf(2.0f - unit_mantissa(r)); }  // log2 u <= 0, u = 2 - m in (0, 1]
__device__ __forceinline__ float neg_ln_u(uint32_t r) { return kNegLn2 * log2_u(r); }                            // -ln u
__device__ __forceinline__ float filter_weight(float f) { return kNegLn2 * (f * f); }                            // w <= 0: |z f|^2 = w log2 u
// |f| z for the weight w = filter_weight(f): the magnitude part (the pipelined kernel keeps its slots' weights in registers)
__device__ __forceinline__ c32 drawn_weighted(uint32_t r, float angle, float w) {
    const float rho = __builtin_amdgcn_sqrtf(w * log2_u(r));
    return cc(v2f{__builtin_amdgcn_cosf(angle), __builtin_amdgcn_sinf(angle)} * rho);
}
// one drawn spectrum element times the filter value f (any sign)
__device__ __forceinline__ c32 drawn_elem(uint32_t r, float angle, float f) {
    const float rho = __builtin_copysignf(__builtin_amdgcn_sqrtf(filter_weight(f) * log2_u(r)), f);
    return cc(v2f{__builtin_amdgcn_cosf(angle), __builtin_amdgcn_sinf(angle)} * rho);
}
__device__ __forceinline__ c32 unit_complex_normal(uint32_t r, float angle) { return drawn_elem(r, angle, 1.0f); }
// filtered spectrum of one generated plane: interior straight into the LDS plane A (row stride S; the discarded kx = M slots
// land in A's never-read last column), edge columns into the side buffers T0 / TM.  The filter values of pair it + 1 are
// requested while pair it is drawn (the compiler otherwise issues each load right in front of its use).
template <int H, int W, int S>
__device__ __forceinline__ void fill_plane_gen(const float* __restrict__ filter, SpectrumRng& g, int tid, c32* A, c32* T0, c32* TM,
                                               int* edge_seq = nullptr) {
    constexpr int NT = plane_threads<H, W>(), M = W / 2, Wh = M + 1, LM = draw_shift<W>(), PAIRS = (H / 2) * M;
    auto fpos = [&](int p) { return (p >> LM) * Wh + 1 + (p & (M - 1)); };
    const int p0 = min(tid, PAIRS - 1);
    float fa = filter[fpos(p0)], fb = filter[fpos(p0) + (H / 2) * Wh];
    draw_plane<H, W, true, SONAR_DRAW_UNROLL>(
        g, tid,
        [&](uint32_t r0, uint32_t rm, uint32_t t) {
            T0[tid] = drawn_elem(r0, angle_lo(t), filter[tid * Wh]);
            TM[tid] = drawn_elem(rm, angle_hi(t), filter[tid * Wh + M]);
            if (edge_seq) {  // whole waves take this branch (H is a multiple of 64): publish "this wave's edge rows are in LDS"
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if ((tid & 63) == 0) __hip_atomic_fetch_add(edge_seq, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        },
        [&](int, int p, uint32_t ra, uint32_t rb, uint32_t t) {
            const int pn = min(p + NT, PAIRS - 1);
            const float na = filter[fpos(pn)], nb = filter[fpos(pn) + (H / 2) * Wh];
            c32* const a = A + (p >> LM) * S + 1 + (p & (M - 1));
            a[0] = drawn_elem(ra, angle_lo(t), fa);
            a[(H / 2) * S] = drawn_elem(rb, angle_hi(t), fb);
            fa = na;
            fb = nb;
        });
}

// One supplied plane's filtered half-spectrum (replay) -> sink(ky, kx, value); thread `tid` handles the complex pair at
// linear indices j and j + NC/2 (j = it*NT + tid), so consecutive lanes touch consecutive elements.
template <int H, int W, typename Sink>
__device__ __forceinline__ void fill_plane(const float* __restrict__ z, const float* __restrict__ filter, int64_t plane, int tid,
                                           Sink&& sink) {
    constexpr int NT = plane_threads<H, W>();
    constexpr int Wh = W / 2 + 1, NC = H * Wh, HALF = NC / 2;
    // (ky, kx) of linear index j, advanced incrementally (no per-element division); the partner element
    // j + HALF = j + (H/2) * Wh sits in the same column, H/2 rows below
    int ky = tid / Wh, kx = tid - ky * Wh;
    constexpr int DKY = NT / Wh, DKX = NT - DKY * Wh;
    const c32* zp = reinterpret_cast<const c32*>(z) + plane * NC;
#pragma unroll 2
    for (int j = tid; j < HALF; j += NT) {
        c32 za = zp[j], zb = zp[j + HALF];
        const float fa = filter[j], fb = filter[j + HALF];
        za.x *= fa; za.y *= fa;
        zb.x *= fb; zb.y *= fb;
        sink(ky, kx, za);
        sink(ky + H / 2, kx, zb);
        kx += DKX;
        ky += DKY;
        if (kx >= Wh) {
            kx -= Wh;
            ky += 1;
        }
    }
}

// the 8-wave passes of the pipelined kernel below, shared with the 128-row fast path here (defined with power_pipe_kernel)
template <int H, int W, int NW>
__device__ __forceinline__ void pipe_col_b(const c32* X, c32* Y, int w, int lane);
template <int W>
struct RowATw {  // row pass a's wave-uniform twiddles (scalar registers), see pipe_row_a
    c32 g[8], p[8];
};
// Layout of a row between / around the row passes.  kRowsTwoBuffers: pass a reads spectrum column k at k and leaves element (k1, n2) at
// 8 n2 + k1 in the other buffer (the pipelined kernel).  kRowsInPlace: the same in ONE buffer, a workgroup barrier between pass a's loads
// and stores (the phase-serial kernel's 128-row path).  kRowsSwizzled (the spectral filter, round 5): spectrum column k = 8 n1 + n2 lives
// at PlaneCfg::rpos(n1, n2) through every column pass and pass a leaves element (k1, n2) at rpos(k1, n2) -- a thread's reads and writes
// are the same sixteen places, no barrier inside the pass.
enum RowLayout { kRowsTwoBuffers = 0, kRowsInPlace = 1, kRowsSwizzled = 2 };
template <int H, int W, int NW, int LAYOUT = kRowsTwoBuffers>
__device__ __forceinline__ void pipe_row_a(const c32* Y, c32* X, int w, int lane);
template <int H, int W, int NW, bool STATS, bool NORM, int LAYOUT = kRowsTwoBuffers>
__device__ __forceinline__ void pipe_row_b(const c32* X, float* oplane, int w, int lane, float scale, float nm, float nc, double& s, double& q);
#ifndef SONAR_SF_V2
#define SONAR_SF_V2 1  // 0: the round-4 forward half (separate split / unpack / fix-up phases: 12 barriers per plane instead of 7)
#endif

// Look-ahead of the phase-serial generate kernel (launch-bound batch sizes: every workgroup of the launch is resident at once): the
// workgroups from `main_blocks` on compute the statistics of the NEXT call (stream `stream_id`, same seed / shape / filter) into
// `partials` while the first `main_blocks` produce this call's planes -- independent workgroups, no ordering between them.
constexpr int kAheadMaxGroup = 4;  // planes per unit the look-ahead statistics cover
constexpr int kMaxRngGroup = 8;    // planes per RNG group: the edge-column area of the generate kernels holds eight
constexpr int kStatsBatch = 4;  // statistics: planes per batch -- their edge columns wait in LDS for ONE barrier (a barrier per plane
                                // made the eight waves of a group wait for each other four times per group)
struct StatsAhead {
    double* partials = nullptr;
    uint64_t stream_id = 0;
    int main_blocks = 0;
};
template <int H, int W>
__device__ __forceinline__ void power_stats_body(const float* __restrict__ filter, int64_t planes, uint64_t seed, uint64_t stream_id,
                                                 int64_t plane_offset, int group, int split, double* partials, int64_t bid, int64_t nb,
                                                 c32 (*EDGE)[2][H], double* red);

// what: 0 = irfft2 (z given or drawn; optional statistics), 1 = normalised generate (stats pass + final pass),
//       2 = dump the drawn spectrum into `out`, 3 = spectral filter of the real planes `z`, 4 = forward rfft2 of the real planes `z` into `out`
// look-ahead of a normalised generate call (sonar_power_noise_ahead_f32): `have_stats` -- the partials already hold this call's
// statistics (left by the previous call's look-ahead), `next` -- where to leave those of the call with stream id `next_stream`
struct Ahead {
    int have_stats = 0;
    uint64_t next_stream = 0;
    double* next = nullptr;
};


}  // namespace sonar
