// Shared device/host helpers for libsonar_hip.so (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/sonar_hip.h"

namespace sonar {

constexpr int kWave = 64;
constexpr int kBlock = 256;           // 4 waves, one per SIMD
constexpr int kMaxGrid = 256 * 8;     // 256 CUs x 8 resident 256-thread blocks (guide G11)
constexpr int kNPart = SONAR_NPART;   // (sum,sumsq) partial pairs written by stats producers

// ---- error plumbing -------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);
// hipFuncAttributeMaxDynamicSharedMemorySize for `kern`, set ONCE per (kernel, device) of the process (runtime.hip): the attribute belongs
// to the device's copy of the function, so a flag per instantiation alone leaves a second GPU of the process at the 64 KB default.
void lds_attr(const void* kern, int bytes);

#define SONAR_REQUIRE(cond, code, ...)   \
    do {                                 \
        if (!(cond)) {                   \
            sonar::set_error(__VA_ARGS__); \
            return (code);               \
        }                                \
    } while (0)

static inline int grid_for(int64_t work_items, int per_block) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > kMaxGrid) g = kMaxGrid;
    return (int)g;
}

// ---- Philox4x32-10 --------------------------------------------------------------------------
struct Philox4 {
    uint32_t v[4];
};

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                 uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 multiply per product (v_mad_u64_u32) instead of separate mul_hi / mul_lo
        const uint64_t p0 = (uint64_t)M0 * (uint64_t)c0;
        const uint64_t p1 = (uint64_t)M1 * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += W0; k1 += W1;
    }
    return Philox4{{c0, c1, c2, c3}};
}

// counter = (group index (64 bit), stream id (64 bit)); key = seed
__device__ __forceinline__ Philox4 philox_group(uint64_t seed, uint64_t stream_id, uint64_t group) {
    return philox4x32_10((uint32_t)group, (uint32_t)(group >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32),
                         (uint32_t)seed, (uint32_t)(seed >> 32));
}

// 24-bit uniform in [0,1)
__device__ __forceinline__ float u01(uint32_t r) { return (float)(r >> 8) * 0x1p-24f; }
// 24-bit uniform in (0,1): safe for log
__device__ __forceinline__ float u01_open(uint32_t r) { return (float)(r >> 8) * 0x1p-24f + 0x1p-25f; }

// Box-Muller on hardware transcendental units: v_log_f32 is log2, v_sin/v_cos take revolutions.
__device__ __forceinline__ void box_muller(uint32_t ra, uint32_t rb, float& z0, float& z1) {
    const float u1 = u01_open(ra);
    const float u2 = u01(rb);
    // r = sqrt(-2 ln u1) = sqrt(-2 ln2 * log2 u1)
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    z0 = r * __builtin_amdgcn_cosf(u2);
    z1 = r * __builtin_amdgcn_sinf(u2);
}

__device__ __forceinline__ void philox_normal4(uint64_t seed, uint64_t stream_id, uint64_t group, float (&z)[4]) {
    const Philox4 p = philox_group(seed, stream_id, group);
    box_muller(p.v[0], p.v[1], z[0], z[1]);
    box_muller(p.v[2], p.v[3], z[2], z[3]);
}

// ---- tile-keyed random streams ---------------------------------------------------------------------
// 32-bit integer multiplies and the 20 of a Philox4x32-10 block are used only to SEED short bursts of a cheap generator.  A stream
// is keyed by (seed, stream id, tile, lane): every draw is still a pure function of the global position, so batches can be sharded
// over GPUs without changing values.
//
// The burst generator is a multiply-with-carry: MWC64X (D. B. Thomas), 64-bit state (x, c), x' : c' = A x + c, output x ^ c; period
// (A 2^32 - 2) / 2 ~ 2^63, passes TestU01's BigCrush.  On gfx950 the whole state update is ONE v_mad_u64_u32 (32 x 32 + 64 -> 64;
// measured 1.8 ns per wave-instruction and SIMD, the cost of a shift) and a word costs 2.8 ns with the exclusive-or, against 8.8 ns
// for the xoshiro128+ step rounds 1-4 used (seven operations, two of them half-rate shifts: scratch/ubench/valu_rate.hip) -- the draw
// of a spectrum value was one third generator.  Two registers per stream instead of four.  Streams are random points of the one
// cycle (2^63 against at most 2^12 words per stream).  Round 5; the change moved every generate-mode value (replay mode is untouched).
struct Mwc {
    uint32_t x, c;
    static constexpr uint32_t A = 4294883355u;
    __device__ __forceinline__ uint32_t next() {
        const uint32_t r = x ^ c;
        const uint64_t t = (uint64_t)A * (uint64_t)x + (uint64_t)c;
        x = (uint32_t)t;
        c = (uint32_t)(t >> 32);
        return r;
    }
    __device__ __forceinline__ uint32_t next_high() { return next(); }  // (callers that keep only a word's high bits: every bit is good here)
    // (x, c) from two random words: c in [1, 2^31) -- below A and never the all-zero state
    static __device__ __forceinline__ Mwc seeded(uint32_t a, uint32_t b) { return Mwc{a, (b >> 1) | 1u}; }
    __device__ __forceinline__ void normal4(float (&z)[4]) {
        const uint32_t a = next(), b = next(), c2 = next(), d = next();
        box_muller(a, b, z[0], z[1]);
        box_muller(c2, d, z[2], z[3]);
    }
    __device__ __forceinline__ void words4_high(uint32_t (&r)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = next();
    }
    __device__ __forceinline__ void uniform4(float (&u)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) u[k] = u01(next());
    }
};
using TileRng = Mwc;

__device__ __forceinline__ TileRng rng_stream(uint64_t seed, uint64_t stream_id, uint64_t tile, uint32_t lane) {
    const Philox4 p = philox4x32_10((uint32_t)tile, (uint32_t)(tile >> 32), (uint32_t)stream_id,
                                    (uint32_t)((stream_id >> 32) << 16) ^ lane, (uint32_t)seed, (uint32_t)(seed >> 32));
    return Mwc::seeded(p.v[0] ^ p.v[2], p.v[1] ^ p.v[3]);
}

// One-instruction conversions of generator words (power_core.h has the derivation): 23 bits into the mantissa of a float in [1, 2)
// (u = 2 - m is a uniform in (0, 1]); 16 bits as a fraction of a revolution inside a float in [128, 256) -- v_sin / v_cos take
// revolutions, are periodic and accept |x| <= 256 -- from the low or the high half of one word.
__device__ __forceinline__ float unit_mantissa(uint32_t hi_bits_in_msb) {  // bits 31..9 -> [1, 2)
    return __uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, hi_bits_in_msb, 9));
}
__device__ __forceinline__ float angle_lo(uint32_t t) { return __uint_as_float(__builtin_amdgcn_bitop3_b32(t, 0x007FFFFFu, 0x43000000u, 0xEA)); }  // (t & mask) | 128.0f
__device__ __forceinline__ float angle_hi(uint32_t t) { return __uint_as_float(__builtin_amdgcn_alignbit(0x4300u, t, 16)); }

// Flat buffers are drawn in tiles of kTileIters x 64 lanes x 4 elements: global element e belongs to
// tile e / kTileElems, lane (e % 256) / 4, burst step (e % kTileElems) / 256, slot e % 4.
constexpr int kTileIters = 16;
constexpr int kTileElems = kTileIters * 256;

// Every 64-byte line of a kernel's argument block requested at its first instruction.  The compiler loads arguments where it first needs
// them, a batch and a wait at a time; with ~560 bytes of arguments (the levels' tables) that was four or five scalar-cache misses one after
// the other before the pyramid plane kernel's first table entry could be written: 2.2-2.4 k cycles, 1 us, at the top of every workgroup
// (trace build, round 5: 1.35 k with this line); the pipelined power kernel, two lines of arguments, 40.6 -> 39.9 us per call.  Every kernel
// of the library starts with it: nothing for one line of arguments, one early batch of scalar loads otherwise.
template <size_t BYTES>
__device__ __forceinline__ void kernarg_touch() {
#ifdef SONAR_NO_KERNARG_TOUCH  // (profiling builds: the A/B of this line)
    return;
#endif
    if constexpr (BYTES <= 64) return;  // one line: the kernel's own first load is the touch
    const uint32_t __attribute__((address_space(4)))* ka = (const uint32_t __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t any = 0;
#pragma unroll
    for (size_t k = 0; k < BYTES; k += 64) any |= ka[k / 4];
    asm volatile("" ::"s"(any));
}
// ... for a kernel's parameter list (passed as it is declared): the block's size from the parameters' types, in declaration order
template <typename... T>
constexpr size_t kernarg_bytes() {
    size_t off = 0;
    ((off = (off + alignof(T) - 1) / alignof(T) * alignof(T) + sizeof(T)), ...);
    return off;
}
static_assert(kernarg_bytes<float*, int64_t>() == 16 && kernarg_bytes<int, double*>() == 16 && kernarg_bytes<char, int64_t, int>() == 20,
              "the argument block's layout: every parameter at its natural alignment, in declaration order");
template <typename... T>
__device__ __forceinline__ void kernarg_touch_for(const T&...) {
    kernarg_touch<kernarg_bytes<T...>()>();
}

// ---- reductions -----------------------------------------------------------------------------
// Wave-wide sums on the DPP data path (no LDS crossbar traffic, unlike __shfl / ds_bpermute): butterfly inside each quad
// (quad_perm), rotate-and-add inside each row of 16 lanes (row_ror:4, row_ror:8), then the row totals travel down the wave with
// row_bcast:15 (rows 1, 3) and row_bcast:31 (rows 2, 3); lane 63 holds the total, returned to every lane through v_readlane.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_move(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <typename T>
__device__ __forceinline__ T wave_sum_dpp(T v) {
    v += dpp_move<0xB1>(v);         // quad_perm:[1,0,3,2]
    v += dpp_move<0x4E>(v);         // quad_perm:[2,3,0,1]
    v += dpp_move<0x124>(v);        // row_ror:4
    v += dpp_move<0x128>(v);        // row_ror:8  -> every lane holds its row's sum
    v += dpp_move<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v += dpp_move<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 = wave total
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
    v = wave_sum_dpp(v);
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float wave_sum(float v) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum_dpp(v)), 63));
}

// Block-wide (sum, sumsq) in fp64; result valid in thread 0.  `red` = 2*kBlock/64 doubles of LDS.
template <int BLOCK>
__device__ __forceinline__ void block_sum2(double& s, double& q, double* red) {
    constexpr int NW = BLOCK / 64;
    s = wave_sum(s);
    q = wave_sum(q);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) {
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
        s = ss;
        q = qq;
    }
}

// Every stats-producing kernel owns partial slot blockIdx.x (grid <= kNPart); block 0 zeroes the
// slots no block owns, so consumers can always reduce all kNPart pairs in a fixed order.
// `bid` of `nb`: the block's index among the blocks that own a slot (a kernel whose grid also holds other work passes its own)
template <int BLOCK>
__device__ __forceinline__ void write_partial_at(double s, double q, double* partials, double* red, int bid, int nb) {
    block_sum2<BLOCK>(s, q, red);
    if (threadIdx.x == 0) {
        partials[2 * bid + 0] = s;
        partials[2 * bid + 1] = q;
    }
    if (bid == 0)
        for (int j = nb + threadIdx.x; j < kNPart; j += BLOCK) {
            partials[2 * j] = 0.0;
            partials[2 * j + 1] = 0.0;
        }
}
template <int BLOCK>
__device__ __forceinline__ void write_partial(double s, double q, double* partials, double* red) {
    write_partial_at<BLOCK>(s, q, partials, red, (int)blockIdx.x, (int)gridDim.x);
}

// ---- normalisation decision (py/utils.py:100-105), identical in every block ---------------------
struct NormDecision {
    float mean, stdv;
    int do_sub, do_div;
};

// fp64 reciprocal and reciprocal square root from the fp32 hardware seeds and Newton steps in fp64 (error 2^-23 -> 2^-46 -> 2^-92: the
// last bit of a double at most).  The compiler's IEEE double division and sqrt are chains of ~30 dependent instructions each, half of them
// quarter-rate; the normalisation decision below ran five of them in ONE thread with the whole workgroup waiting at a barrier: 1.6 us at
// the top of every normalising kernel (pipelined power kernel, trace build, round 5) -- as long as a 33 MB fill.
__device__ __forceinline__ double rcp_f64(double x) {
    double r = (double)__builtin_amdgcn_rcpf((float)x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double rsqrt_f64(double x) {  // x > 0, in fp32 range
    double r = (double)__builtin_amdgcn_rsqf((float)x);
    r = __builtin_fma(r * __builtin_fma(-x * r, r, 1.0), 0.5, r);
    r = __builtin_fma(r * __builtin_fma(-x * r, r, 1.0), 0.5, r);
    return r;
}

// the decision from the totals (sum, sumsq) of n_total values
__device__ __forceinline__ NormDecision decision_from_totals(double s, double q, int64_t n_total, float thr_sd) {
    const double nt = (double)n_total;
    const double mean = s * rcp_f64(nt);
    // unbiased (py/utils.py:100: noise.std()); n_total == 1 -> NaN like torch (0 x inf)
    const double var = (q - s * mean) * rcp_f64(nt - 1.0);
    double sd;
    if (var > 1e-30 && var < 1e30) sd = var * rsqrt_f64(var);             // every sensible tensor
    else sd = sqrt(var > 0.0 || !(var == var) ? var : 0.0);                // degenerate: zero, denormal, huge or NaN variance
    NormDecision d;
    d.mean = (float)mean;
    d.stdv = (float)sd;
    const double thr = (double)thr_sd * rsqrt_f64(nt);
    d.do_sub = fabs((double)d.mean) > thr;
    d.do_div = fabs(1.0 - (double)d.stdv) > thr;
    return d;
}

// the decision from this thread's share (s, q) of the (sum, sumsq) partials; every thread of the block takes part
template <int BLOCK>
__device__ __forceinline__ NormDecision decide_from_sums(double s, double q, int64_t n_total, float thr_sd, double* red, NormDecision* sh) {
    block_sum2<BLOCK>(s, q, red);
    if (threadIdx.x == 0) *sh = decision_from_totals(s, q, n_total, thr_sd);
    __syncthreads();
    return *sh;
}

template <int BLOCK>
__device__ __forceinline__ NormDecision decide_norm(const double* __restrict__ partials, int64_t npart, int64_t n_total,
                                                    float thr_sd, double* red, NormDecision* sh) {
    double s = 0.0, q = 0.0;
    for (int64_t i = threadIdx.x; i < npart; i += BLOCK) {
        s += partials[2 * i];
        q += partials[2 * i + 1];
    }
    return decide_from_sums<BLOCK>(s, q, n_total, thr_sd, red, sh);
}

// arguments of the kernels that normalise inside the generating pass
struct NormArgs {
    const double* partials;
    int64_t n_total;
    float factor, thr_sd;
};

__device__ __forceinline__ float apply_norm(float v, const NormDecision& d, float factor, bool do_mul) {
    if (d.do_sub) v = v - d.mean;
    if (d.do_div) v = v / d.stdv;
    if (do_mul) v = v * factor;
    return v;
}

// Fused generate+normalise kernels (device draws, no bit-parity reference): one reciprocal per block instead of a
// division per element; differs from apply_norm by at most one ulp.
struct NormFast {
    float mean, inv_std, factor;
    int do_sub, do_scale;
    __device__ __forceinline__ NormFast(const NormDecision& d, float f)
        : mean(d.mean), inv_std(d.do_div ? 1.0f / d.stdv : 1.0f), factor(f), do_sub(d.do_sub), do_scale(d.do_div || f != 1.0f) {}
    __device__ __forceinline__ float operator()(float v) const {
        if (do_sub) v = v - mean;
        if (do_scale) v = v * inv_std * factor;
        return v;
    }
};

// x / d as x * (1/d) when d is a power of two (bit-identical), true division otherwise
struct Divider {
    float d, inv;
    int exact;
    __host__ __device__ explicit Divider(float div) : d(div), inv(1.0f / div) {
        int e;
        exact = (frexpf(div, &e) == 0.5f) || (frexpf(div, &e) == -0.5f);
    }
    __device__ __forceinline__ float operator()(float x) const { return exact ? x * inv : x / d; }
    // u01(r) / d + term for a generator word: with a power-of-two divisor both scalings are exact, so the sum is ONE fused multiply-add of
    // the 24-bit integer -- the same bits as the rounded steps, three instructions instead of five
    __device__ __forceinline__ float from_word(uint32_t r, float term) const {
        return exact ? __builtin_fmaf((float)(r >> 8), inv * 0x1p-24f, term) : u01(r) / d + term;
    }
};

// Divider::from_word with the `exact` decision taken once, by the caller (a launch has ONE divisor): inside a tile loop the run-time flag was
// a scalar branch, or two, in front of every value -- 64 per tile and lane in the Perlin kernels (round 5).  Same expressions, same bits.
struct WordScale {     // power-of-two divisor
    float k;           // inv * 2^-24
    __device__ __forceinline__ float from_word(uint32_t r, float term) const { return __builtin_fmaf((float)(r >> 8), k, term); }
};
struct WordDivide {    // any other divisor
    float d;
    __device__ __forceinline__ float from_word(uint32_t r, float term) const { return u01(r) / d + term; }
};
// f(word-to-value converter): the tile loop instantiated for the divisor's kind
template <typename F>
__device__ __forceinline__ void with_divider(const Divider& div, F&& f) {
    if (div.exact) f(WordScale{div.inv * 0x1p-24f});
    else f(WordDivide{div.d});
}

// ---- blend modes (py/utils.py:17-21) -----------------------------------------------------------
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

// 16-byte store of a result nobody in this launch reads again.  `nt`: with the non-temporal hint the lines leave the L2 as they are
// written instead of sitting dirty until the end-of-kernel write-back -- a launch-bound kernel that leaves 16.8 MB dirty pays 2-3 us for
// that flush after its last workgroup has finished (batch 64: pyramid 26.4 -> 24.5 us, the cfg3 chain 31.1 -> 28.6 us per call); a
// bandwidth-bound launch is better off with the write-back cache (batch 512: Perlin 45.5 -> 47.8 us with the hint), hence by size:
// nt_stores_host(n) for tensors up to 8 Mi elements (32 MiB), in the pyramid plane kernel and scale_noise_kernel.  Not in the fills and the
// Perlin kernels (batch 64: +0.4-0.5 us with the hint), and not where a wave's stores do not cover whole cache lines: the power kernels' row
// pass writes 64-byte runs per four lanes and loses with it (batch 64: 14.9 -> 16.0 us).
constexpr int64_t kNtMaxElems = 8 << 20;
static inline bool nt_stores_host(int64_t n) { return n <= kNtMaxElems; }
typedef float sonar_v4f __attribute__((ext_vector_type(4)));
template <bool NT>  // a template parameter of the kernel: a run-time branch per store cost the batch-512 launches ~1 %
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
    if constexpr (NT) __builtin_nontemporal_store(sonar_v4f{a, b, c, d}, reinterpret_cast<sonar_v4f*>(p));
    else *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}

// torch.lerp as the CPU reference's vectorised kernel evaluates it (ATen/native/cpu/Lerp.h
// lerp_vec): coeff = |w| < 0.5 ? w : w - 1, base = |w| < 0.5 ? a : b, result = fma(coeff, b - a, base).
template <typename T>
__device__ __forceinline__ T blend(int mode, T a, T b, T t) {
    if (mode == SONAR_BLEND_LERP) {
        const T diff = b - a;
        const bool small = (t < T(0.5)) && (t > T(-0.5));
        return fma_t(small ? t : t - T(1), diff, small ? a : b);
    }
    if (mode == SONAR_BLEND_INJECT) return b * t + a;
    return a - b * t;
}

}  // namespace sonar

// power_fft.hip (power_any.h): the LDS line transforms as passes through a complex workspace, for the planes beyond LDS; false = not
// taken (odd width, misaligned buffer, line too long): dft_direct.hip then runs its direct sums
bool sonar_lines_rows_r2c(const float* x, float* y, int64_t rows, int64_t W, hipStream_t st);
bool sonar_lines_cols(const float* in, const float* filter, float* out, int64_t planes, int64_t H, int64_t K, int inverse, hipStream_t st);
bool sonar_lines_rows_c2r(const float* y, float* out, int64_t rows, int64_t W, float scale, double* partials, hipStream_t st);
bool sonar_lines_rows_c2r_norm(const float* y, float* out, int64_t rows, int64_t W, float scale, double* partials, const sonar::NormArgs* norm,
                               hipStream_t st);
