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
#include <math.h>

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
__device__ __forceinline__ c32 cmul_i(c32 a) { return make_float2(-a.y, a.x); }  // a * (+i)
__device__ __forceinline__ c32 cadd_i(c32 a, c32 b) { return cc(vv(a) + v2f{-b.y, b.x}); }   // a + i b
__device__ __forceinline__ c32 csub_i(c32 a, c32 b) { return cc(vv(a) - v2f{-b.y, b.x}); }   // a - i b
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
    const c32 t3 = cscale(csub(cmul_i(o[3]), o[3]), r);        // * e^{3 i pi/4} = r (i o - o)
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
    t[4] = cmul_i(o[4]);
    t[5] = cmul(o[5], make_float2(-s1, c1));
    t[6] = cscale(csub(cmul_i(o[6]), o[6]), r);
    t[7] = cmul(o[7], make_float2(-c1, s1));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v[i] = cadd(e[i], t[i]);
        v[i + 8] = csub(e[i], t[i]);
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
#ifndef SONAR_PW_SKIP
#define SONAR_PW_SKIP 0  // profiling builds only (scratch/pw_passes.py): 1 draw, 2 column passes, 4 rows pass a, 8 rows pass b arithmetic, 16 global stores
#endif
#ifdef SONAR_PW_TRACE  // profiling builds: per-phase s_memtime stamps of wave 0 of every workgroup (scratch/pw_trace.py)
__device__ unsigned long long g_pw_trace[1024 * 8 * 12];
#define SONAR_STAMP(slot) do { if (tid == 0 && pidx < 8) g_pw_trace[(blockIdx.x * 8 + pidx) * 12 + (slot)] = __builtin_readcyclecounter(); } while (0)
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
    static constexpr int RN1 = split_n1(M), RN2 = M / RN1;
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
// |z|^2 = -ln(u_R) for a unit complex normal, so it skips half of the generator steps and all of sqrt / sin / cos.
struct SpectrumRng {
    Xoshiro R, T, E;
};

template <int H, bool NEED_T>
__device__ __forceinline__ SpectrumRng spectrum_rng(uint64_t seed, uint64_t stream_id, int64_t ggroup, int tid) {
    SpectrumRng g;
    g.R = rng_stream(seed, stream_id, ((uint64_t)ggroup << 2) | 0u, (uint32_t)tid);
    if constexpr (NEED_T) g.T = rng_stream(seed, stream_id, ((uint64_t)ggroup << 2) | 1u, (uint32_t)tid);
    else g.T = Xoshiro{0, 0, 0, 1};
    g.E = Xoshiro{0, 0, 0, 1};
    if (tid < H) g.E = rng_stream(seed, stream_id, ((uint64_t)ggroup << 2) | 2u, (uint32_t)tid);
    return g;
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

template <int H, int W, bool NEED_T, int UNROLL = 0, typename Edge, typename Pair>
__device__ __forceinline__ void draw_plane(SpectrumRng& g, int tid, Edge&& edge, Pair&& pair) {
    constexpr int NT = plane_threads<H, W>(), M = W / 2, PAIRS = (H / 2) * M, ITER = draw_iters<H, W>();
    constexpr int UNR = UNROLL > 0 ? UNROLL : ITER;  // 0 = full (the statistics pass indexes registers by `it`)
    if (tid < H) {  // row ky = tid of the edge columns: radius word of kx = 0, of kx = M, then one angle word for both
        const uint32_t r0 = g.E.next_high();
        const uint32_t rm = g.E.next_high();
        const uint32_t t = g.E.next();
        edge(r0, rm, t);
    }
#pragma unroll UNR
    for (int it = 0; it < ITER; ++it) {
        const int p = tid + it * NT;
        if (PAIRS % NT == 0 || p < PAIRS) {
            const uint32_t ra = g.R.next_high();  // radius words keep bits 31..9 only
            const uint32_t rb = g.R.next_high();
            const uint32_t t = NEED_T ? g.T.next() : 0u;
            pair(it, p, ra, rb, t);
        }
    }
}

// advance the streams past one plane's draws without using them (a workgroup that starts in the middle of an RNG group)
template <int H, int W, bool NEED_T>
__device__ __forceinline__ void skip_plane(SpectrumRng& g, int tid) {
    draw_plane<H, W, NEED_T>(g, tid, [](uint32_t, uint32_t, uint32_t) {}, [](int, int, uint32_t, uint32_t, uint32_t) {});
}

// A kernel's work units: whole RNG groups (one workgroup draws the group's planes back to back; the seeding is paid once
// per group), or -- `split`, chosen by the launcher when there are too few groups to fill the chip -- single planes, the
// workgroup fast-forwarding the group's streams to its plane.  Same values either way.
struct GroupWalk {
    int64_t grp;
    int first, count;
    __device__ __forceinline__ GroupWalk(int64_t unit, int group, int split)
        : grp(split ? unit / group : unit), first(split ? (int)(unit % group) : 0), count(split ? 1 : group) {}
};

// ---- unit complex normal z = rho e^{i theta}, E|z|^2 = 1, from raw generator bits (about 30 instruction slots) --------------
// radius: 23 random bits become the mantissa of a float f in [1, 2) in ONE v_alignbit; u = 2 - f is uniform on (0, 1] and
//   rho^2 = -ln u (the 1/sqrt(2) of "(a + ib) / sqrt 2" folded into the radius), so rho <= sqrt(23 ln 2) = 3.99 (5.65 sigma
//   per component).
// angle: 16 random bits -> f in [1, 2) the same way; v_sin / v_cos take revolutions and are periodic, so they are fed f
//   directly.  One 32-bit draw serves two elements; 65536 directions x a 23-bit radius is far below fp32 output resolution
//   after the 8192-term FFT sums.
__device__ __forceinline__ float unit_mantissa(uint32_t hi_bits_in_msb) {  // bits 31..9 -> [1, 2)
    return __uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, hi_bits_in_msb, 9));
}
__device__ __forceinline__ float neg_ln_u(uint32_t r) {  // -ln(u), u = 2 - f in (0, 1]
    return -0.6931471805599453f * __builtin_amdgcn_logf(2.0f - unit_mantissa(r));
}
__device__ __forceinline__ c32 unit_complex_normal(uint32_t r, uint32_t t16) {
    const float rho = __builtin_amdgcn_sqrtf(neg_ln_u(r));
    const float f = __uint_as_float(0x3f800000u | (t16 << 7));
    return make_float2(rho * __builtin_amdgcn_cosf(f), rho * __builtin_amdgcn_sinf(f));
}
// one drawn spectrum element times the filter value (the replay path multiplies the dumped element by f the same way)
__device__ __forceinline__ c32 drawn_elem(uint32_t r, uint32_t t16, float f) {
    const c32 z = unit_complex_normal(r, t16);
    return make_float2(z.x * f, z.y * f);
}

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
            T0[tid] = drawn_elem(r0, t & 0xFFFFu, filter[tid * Wh]);
            TM[tid] = drawn_elem(rm, t >> 16, filter[tid * Wh + M]);
            if (edge_seq) {  // whole waves take this branch (H is a multiple of 64): publish "this wave's edge rows are in LDS"
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if ((tid & 63) == 0) __hip_atomic_fetch_add(edge_seq, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        },
        [&](int, int p, uint32_t ra, uint32_t rb, uint32_t t) {
            const int pn = min(p + NT, PAIRS - 1);
            const float na = filter[fpos(pn)], nb = filter[fpos(pn) + (H / 2) * Wh];
            c32* const a = A + (p >> LM) * S + 1 + (p & (M - 1));
            a[0] = drawn_elem(ra, t & 0xFFFFu, fa);
            a[(H / 2) * S] = drawn_elem(rb, t >> 16, fb);
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

// SRC: 0 = spectrum `z` supplied (replay), 1 = spectrum drawn on device, 2 = `z` is a REAL H x W plane: forward r2c FFT in
// LDS, x filter, then the same inverse (spectral filter: out = irfft2(rfft2(x) * filter), py/nodes/powernoise.py:356-366)
template <int H, int W, int SRC, bool STATS, bool NORM>
__global__ void __launch_bounds__((plane_threads<H, W>()), (H * W >= 32768 ? 2 : SONAR_FFT_WAVES)) power_irfft2_kernel(const float* __restrict__ z,
                                                                       const float* __restrict__ filter, float* out,
                                                                       int64_t planes, uint64_t seed, uint64_t stream_id,
                                                                       int64_t plane_offset, int group, int split, double* partials,
                                                                       NormArgs na) {
    using C = PlaneCfg<H, W>;
    constexpr int NT = plane_threads<H, W>();
    constexpr int M = C::M, S = C::S;
    constexpr int RN1 = C::RN1, RN2 = C::RN2;
    // FAST shapes (W = 128, H = 64 / 128, 8 waves): every wave owns ONE residue n2 in both twiddled passes, so all
    // twiddles are wave-uniform AND loop-invariant -> loaded once into scalar registers before the plane loop
    // (no s_load / lgkmcnt(0) stall inside the passes); the column split is H = (H/8) x 8 instead of 8 x (H/8).
    constexpr bool FAST = (W == 128) && (H == 128 || H == 64) && (NT == 512) && !SONAR_FFT_TW_LDS;
    constexpr int CN1 = FAST ? H / 8 : C::CN1, CN2 = FAST ? 8 : C::CN2;
    __shared__ c32 A[C::kLdsComplex];
    c32* const T0 = A + H * S;      // raw column kx = 0
    c32* const TM = T0 + H;         // raw column kx = M
    c32* const TW = TM + H;         // e^{2 pi i j / 256}
    __shared__ double red[2 * NT / 64];
    __shared__ NormDecision shd;
    __shared__ int edge_seq;  // FAST generate path: edge-column waves drawn so far (two per plane), see the fill
    const int tid = threadIdx.x;
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
    auto tw = [&](int idx, int n, bool uni) -> c32 {
        if (uni) return c_tw256[(__builtin_amdgcn_readfirstlane(idx) * (256 / n)) & 255];
        return TW[(idx * (256 / n)) & 255];
    };
#endif
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    c32 ctw[CN1], gtw[RN1], ptw[RN1];
    auto load_twiddles = [&](int w) {
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
    for (int64_t unit = blockIdx.x; unit < (split ? planes : planes / group); unit += gridDim.x) {
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
#pragma unroll 2
        for (int item = tid; item < RN1 * H; item += NT) {
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
                const int item = tid + it * NT;
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
                const int item = tid + it * NT;
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
        for (int item = tid; item < (M / 2 + 1) * H; item += NT) {
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
                const c32 t = cmulc(o, tw(k, W, H % 64 == 0));
                row[k] = make_float2(e.x + t.x, e.y + t.y);
                row[M - k] = make_float2(e.x - t.x, -(e.y - t.y));
            }
        }
        __syncthreads();
        // columns, pass b': DFT over k2 -> n2 for fixed k1 (rows CN2 k1 + .), twiddle
        for (int item = tid; item < CN1 * M; item += NT) {
            const int c = item % M, k1 = item / M;
            c32 u[CN2];
#pragma unroll
            for (int k2 = 0; k2 < CN2; ++k2) u[k2] = A[(CN2 * k1 + k2) * S + c];
            fdft<CN2>(u);
#pragma unroll
            for (int n2 = 1; n2 < CN2; ++n2) u[n2] = cmulc(u[n2], tw(n2 * k1, H, M % 64 == 0));
#pragma unroll
            for (int n2 = 0; n2 < CN2; ++n2) A[(CN2 * k1 + n2) * S + c] = u[n2];
        }
        __syncthreads();
        // columns, pass a': DFT over k1 -> n1: Z[ky = CN2 n1 + n2][c], in place
        for (int item = tid; item < CN2 * M; item += NT) {
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
        for (int ky = tid; ky < H; ky += NT) {
            const int kn = (H - ky) & (H - 1);
            const c32 p = A[ky * S], pn = A[kn * S];
            const float f0 = SRC == 3 ? 1.0f : filter[ky * Wh], fm = SRC == 3 ? 1.0f : filter[ky * Wh + M];
            T0[ky] = make_float2(0.5f * (p.x + pn.x) * f0, 0.5f * (p.y - pn.y) * f0);
            TM[ky] = make_float2(0.5f * (p.y + pn.y) * fm, -0.5f * (p.x - pn.x) * fm);
        }
        if constexpr (SRC != 3) {
        // unrolled: the filter values are global loads (L2 hits) -- eight in flight instead of a wait per element
#pragma unroll 8
        for (int j = tid; j < H * M; j += NT) {
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
            for (int j = tid; j < H * Wh; j += NT) {
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
#pragma unroll
                for (int n1 = 0; n1 < CN1; ++n1) v[n1] = A[(CN2 * n1 + n2) * S + c];
                idft<CN1>(v);
#pragma unroll
                for (int k1 = 1; k1 < CN1; ++k1) v[k1] = cmul(v[k1], ctw[k1]);
#pragma unroll
                for (int k1 = 0; k1 < CN1; ++k1) A[(CN2 * k1 + n2) * S + c] = v[k1];
            }
            SONAR_STAMP(4);
            __syncthreads();
            SONAR_STAMP(5);
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
        } else {
        // ---------------------------------------------------------------- columns, pass a
        // Column 0 is built on the fly from the raw kx = 0 / kx = M columns:
        //   Q[ky] = sym(Z0)[ky] + i sym(ZM)[ky],  sym(Z)[ky] = (Z[ky] + conj Z[-ky]) / 2
SONAR_UNROLL_ITEMS
        for (int item = tid; item < CN2 * M; item += NT) {
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
        for (int item = tid; item < CN1 * M; item += NT) {
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
                const int item = tid + it * NT;
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
                const int item = tid + it * NT;
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
        float ps = 0.0f, pq = 0.0f;  // per-plane fp32 partials (<= 64 values per thread), folded into fp64 below
SONAR_UNROLL_ITEMS
        for (int item = tid; item < RN1 * H; item += NT) {
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
    if constexpr (STATS) write_partial<NT>(s, q, partials, red);
}

// Statistics of the output WITHOUT computing it (Parseval, ortho-normalised transform):
//   sum   x  = sqrt(H W) * Re(Zf[0][0])
//   sum x^2  = sum_ky ( |sym(Zf[:,0])[ky]|^2 + |sym(Zf[:,M])[ky]|^2 ) + 2 sum_ky sum_{0<kx<M} |Zf[ky][kx]|^2
// (only the Hermitian-symmetric part of the kx = 0 and kx = M columns survives the c2r stage).
template <int H, int W>
__global__ void __launch_bounds__((plane_threads<H, W>())) power_stats_kernel(const float* __restrict__ filter, int64_t planes, uint64_t seed,
                                                                   uint64_t stream_id, int64_t plane_offset, int group, int split,
                                                                   double* partials) {
    constexpr int NT = plane_threads<H, W>(), M = W / 2, Wh = M + 1;
    __shared__ c32 EDGE[2][2][H];  // [plane parity][kx = 0 | kx = M][ky]: double-buffered -> one barrier per plane
    __shared__ double red[2 * NT / 64];
    const int tid = threadIdx.x;
    double s = 0.0, q = 0.0;
    int par = 0;
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
    constexpr float kNegLn2 = -0.6931471805599453f;
    float wgt[2 * ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int p = tid + it * NT, ky = p >> LM, kx = 1 + (p & (M - 1));
        const bool live = p < PAIRS && kx < M;
        const float fa = live ? filter[ky * Wh + kx] : 0.0f, fb = live ? filter[(ky + H / 2) * Wh + kx] : 0.0f;
        wgt[it] = kNegLn2 * (fa * fa);
        wgt[it + ITER] = kNegLn2 * (fb * fb);
    }
    const float f0 = tid < H ? filter[tid * Wh] : 0.0f, fm = tid < H ? filter[tid * Wh + M] : 0.0f;
    for (int64_t unit = blockIdx.x; unit < (split ? planes : planes / group); unit += gridDim.x) {
        const GroupWalk gw(unit, group, split);
        SpectrumRng rng = spectrum_rng<H, false>(seed, stream_id, plane_offset / group + gw.grp, tid);
        for (int i = 0; i < gw.first; ++i) skip_plane<H, W, false>(rng, tid);
        // The planes of a group meet the same weight at the same slot, so sum_planes w log2 u = w log2(prod_planes u): ONE logarithm per
        // slot and group instead of one per plane (a group has at most 4 planes: the product of four u in [2^-23, 1] stays a normal float,
        // and its rounding error, 3 x 2^-24 relative, is below the logarithm's own).
        float prod[2 * ITER];
#pragma unroll
        for (int it = 0; it < 2 * ITER; ++it) prod[it] = 1.0f;
        int in_prod = 0;
        auto flush = [&]() {
            float acc = 0.0f;
#pragma unroll
            for (int it = 0; it < 2 * ITER; ++it) {
                acc = __builtin_fmaf(wgt[it], __builtin_amdgcn_logf(prod[it]), acc);
                prod[it] = 1.0f;
            }
            q += 2.0 * (double)acc;
            in_prod = 0;
        };
        for (int gp = gw.first; gp < gw.first + gw.count; ++gp) {
            draw_plane<H, W, false>(
                rng, tid,
                [&](uint32_t r0, uint32_t rm, uint32_t t) {
                    EDGE[par][0][tid] = drawn_elem(r0, t & 0xFFFFu, f0);
                    EDGE[par][1][tid] = drawn_elem(rm, t >> 16, fm);
                },
                [&](int it, int, uint32_t ra, uint32_t rb, uint32_t) {
                    prod[it] *= 2.0f - unit_mantissa(ra);
                    prod[it + ITER] *= 2.0f - unit_mantissa(rb);
                });
            if (++in_prod == 4) flush();
            __syncthreads();           // this plane's edge columns are complete; the other buffer's readers finished last iteration
            edge_terms(par);           // overlaps with the next plane's draw (which writes the other buffer)
            par ^= 1;
        }
        if (in_prod) flush();
    }
    write_partial<NT>(s, q, partials, red);
}

// the spectrum draw_plane yields for (seed, stream_id, plane_offset, group), unit filter: zout[planes][H][W/2+1] complex64
template <int H, int W>
__global__ void __launch_bounds__((plane_threads<H, W>())) power_spectrum_kernel(float* zout, int64_t planes, uint64_t seed, uint64_t stream_id,
                                                                      int64_t plane_offset, int group, int split) {
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
                    zp[tid * Wh] = unit_complex_normal(r0, t & 0xFFFFu);
                    zp[tid * Wh + M] = unit_complex_normal(rm, t >> 16);
                },
                [&](int, int p, uint32_t ra, uint32_t rb, uint32_t t) {
                    const int ky = p >> draw_shift<W>(), kx = 1 + (p & (M - 1));
                    if (kx < M) {
                        zp[ky * Wh + kx] = unit_complex_normal(ra, t & 0xFFFFu);
                        zp[(ky + H / 2) * Wh + kx] = unit_complex_normal(rb, t >> 16);
                    }
                });
        }
    }
}

template <int H, int W>
static int power_grid(int64_t planes) {
    using C = PlaneCfg<H, W>;
    static_assert(C::kLdsBytes + 256 <= 160 * 1024, "plane does not fit in LDS");
    // blocks/CU by LDS; persistent grid of resident blocks (<= kNPart so each owns a partial slot)
    // 16 waves per CU at the kernel's 128-VGPR budget
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(1024 / plane_threads<H, W>(), (160 * 1024) / (C::kLdsBytes + 256)));
    return (int)std::min<int64_t>(std::min<int64_t>(planes, (int64_t)256 * per_cu), kNPart);
}

// what: 0 = irfft2 (z given or drawn; optional statistics), 1 = normalised generate (stats pass + final pass),
//       2 = dump the drawn spectrum into `out`, 3 = spectral filter of the real planes `z`, 4 = forward rfft2 of the real planes `z` into `out`
template <int H, int W>
static int launch_power(int what, const float* z, const float* filter, float* out, int64_t planes, uint64_t seed,
                        uint64_t stream_id, int64_t plane_offset, int group, double* partials, NormArgs na, hipStream_t st) {
    // too few RNG groups to fill the chip: one workgroup per plane (it fast-forwards the group's streams), same values
    const int split = group > 1 && planes / group < 2 * 256 ? 1 : 0;
    const int64_t ngroups = split ? planes : planes / group;  // work units
    const int g = power_grid<H, W>(ngroups);
    const dim3 blk(plane_threads<H, W>());
#define SONAR_PW(G, ST, NM, PART) \
    hipLaunchKernelGGL((power_irfft2_kernel<H, W, G, ST, NM>), dim3(g), blk, 0, st, z, filter, out, planes, seed, stream_id, plane_offset, group, split, PART, na)
    if (what == 4) {
        SONAR_PW(3, false, false, nullptr);
    } else if (what == 3) {
        if (partials) SONAR_PW(2, true, false, partials); else SONAR_PW(2, false, false, partials);
    } else if (what == 2) {
        hipLaunchKernelGGL((power_spectrum_kernel<H, W>), dim3(std::min<int64_t>(ngroups, 2048)), blk, 0, st, out, planes, seed, stream_id, plane_offset, group, split);
    } else if (what == 1) {
        hipLaunchKernelGGL((power_stats_kernel<H, W>), dim3(std::min<int64_t>(ngroups, kNPart)), blk, 0, st, filter, planes, seed, stream_id,
                           plane_offset, group, split, partials);
        SONAR_PW(1, false, true, nullptr);
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

#include "power_any.h"

using namespace sonar;

static int power_dispatch(int what, const float* z, const float* filter, float* out, int64_t planes, int64_t H, int64_t W,
                          uint64_t seed, uint64_t stream_id, int64_t plane_offset, int group, double* partials, NormArgs na,
                          hipStream_t st) {
    const bool gen = what == 1 || what == 2 || (what == 0 && z == nullptr);
    if (!gen) group = 1;
    SONAR_REQUIRE(group >= 1 && planes % group == 0 && plane_offset % group == 0, SONAR_ERR_ARG,
                  "sonar_power_*: planes (%lld) and plane_offset (%lld) must be multiples of the RNG group (%d)", (long long)planes,
                  (long long)plane_offset, group);
#define SONAR_CASE(HH, WW) \
    if (H == HH && W == WW) return launch_power<HH, WW>(what, z, filter, out, planes, seed, stream_id, plane_offset, group, partials, na, st)
    SONAR_CASE(128, 128);
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
#undef SONAR_CASE
    if (what == 4) {
        set_error("sonar_rfft2_f32: power-of-two planes from 16 x 16 to 256 x 128 only (got %lld x %lld)", (long long)H, (long long)W);
        return SONAR_ERR_UNSUPPORTED;
    }
    if (any_plane_ok(H, W)) return launch_power_any(what, z, filter, out, planes, H, W, seed, stream_id, plane_offset, group, partials, na, st);
    set_error("sonar_power_*: unsupported plane %lld x %lld (even sizes whose half-spectrum fits in LDS: H <= 512, W <= 1024, about 19k complex values)",
              (long long)H, (long long)W);
    return SONAR_ERR_UNSUPPORTED;
}

#ifdef SONAR_PW_TRACE
extern "C" int sonar_debug_pw_trace(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(sonar::g_pw_trace), sizeof(sonar::g_pw_trace));
}
#endif

extern "C" int sonar_power_plane_kind(int64_t H, int64_t W) {
    static const int fast[][2] = {{128, 128}, {64, 64}, {32, 32}, {16, 16}, {256, 128}, {128, 256}, {128, 64}, {64, 128}, {64, 32}, {32, 64},
                                  {256, 64}, {64, 256}};
    for (const auto& f : fast)
        if (H == f[0] && W == f[1]) return 1;
    return any_plane_ok(H, W) ? 2 : 0;
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
