// General-size planes for the power-noise path: any even H x W whose half-spectrum H x (W/2 + 1) (complex64) fits in LDS.
// The fast kernels in power_fft.hip cover the power-of-two latents (SDXL 1024^2 -> 128 x 128 ...); everything else
// (832 x 1216 px -> 104 x 152, 768^2 -> 96 x 96 ...) lands here.  Same pipeline, same semantics
// (torch.fft.irfft2(z * filter, s=(H, W), norm="ortho"), py/nodes/powernoise.py:338-366), written for generality:
//   * a length-N line DFT is the two-factor Cooley-Tukey split N = N1 x N2 (the host picks the divisor pair with the smallest
//     N1 + N2; a prime N degenerates to the plain O(N^2) sum) with table twiddles e^{2 pi i j / N} built in LDS per launch;
//     each pass gathers its outputs in registers, then barrier, then writes -> in place, one plane buffer;
//   * rows use the half-length complex trick (W real values = W/2 complex) exactly like the fast kernel;
//   * the kx = 0 and kx = W/2 columns are ordinary columns; c2r takes their real parts after the column transform.
// Templates and inline device functions only (power_core.h's helpers): included by power_any.h (the run-time-size kernels, launchers
// and the pass-per-launch line kernels, one translation unit) and by power_buckets_*.hip (the same kernel with compile-time factor
// pairs for the SDXL buckets).
#pragma once
#include "power_core.h"

namespace sonar {

constexpr int kAnyThreads = 1024;        // one workgroup per CU (16 waves) when only one plane buffer fits; two of kAnySlots threads otherwise
constexpr int kAnySlots = kFftThreads;   // drawing thread slots: streams are keyed by (group, slot) like the fast path
constexpr int kAnyPer = 4;               // outputs per thread and batch held in registers across a pass's barrier
constexpr size_t kAnyLdsLimit = 160 * 1024 - 2048;

// ---- register codelets of any length up to kAnyCodelet ---------------------------------------------------------------------------
// idft_any<N>: in-place inverse (sign +) DFT, natural order in and out.  Powers of two are the fixed-size kernels' codelets; an odd
// prime P is the symmetric form X[k], X[P - k] = (v0 + sum_j cos(2 pi j k / P) (v[j] + v[P - j])) +- i sum_j sin(2 pi j k / P) (v[j] -
// v[P - j]) -- (P - 1)^2 / 2 packed FMAs; a composite N = A x B is Cooley-Tukey in registers (B transforms of length A, constant
// twiddles, A transforms of length B).  Every coefficient is a compile-time constant (static_for hands the loop indices to the
// lambdas as types), so a codelet is straight-line packed arithmetic on register pairs.
constexpr int kAnyCodelet = 16;
#ifndef SONAR_ANY_STORE16
#define SONAR_ANY_STORE16 1
#endif
#ifndef SONAR_ANY_PRIMES  // codelets for 17 and 19 too (136 = 8 x 17 and 152 = 8 x 19 are SDXL sides): 104 x 152 216 -> 152 us, the other sizes +3 %
#define SONAR_ANY_PRIMES 1
#endif

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

constexpr double ct_series(double x, bool cosine) {  // |x| <= pi
    double term = cosine ? 1.0 : x, sum = term;
    for (int k = 1; k < 20; ++k) {
        const double a = cosine ? 2.0 * k - 1.0 : 2.0 * k, b = a + 1.0;
        term *= -x * x / (a * b);
        sum += term;
    }
    return sum;
}
// cos / sin of 2 pi m / n, exact on the axes
constexpr double ct_cos2pi(int m, int n) {
    m = ((m % n) + n) % n;
    if (4 * m % n == 0) return 4 * m / n == 0 ? 1.0 : 4 * m / n == 2 ? -1.0 : 0.0;
    double x = 6.283185307179586476925286766559 * m / n;
    if (x > 3.14159265358979323846) x -= 6.283185307179586476925286766559;
    return ct_series(x, true);
}
constexpr double ct_sin2pi(int m, int n) {
    m = ((m % n) + n) % n;
    if (4 * m % n == 0) return 4 * m / n == 1 ? 1.0 : 4 * m / n == 3 ? -1.0 : 0.0;
    double x = 6.283185307179586476925286766559 * m / n;
    if (x > 3.14159265358979323846) x -= 6.283185307179586476925286766559;
    return ct_series(x, false);
}
constexpr int ct_first_factor(int n) {  // the first-pass length of a composite codelet: 4 when it divides, else the smallest prime
    if (n % 4 == 0) return 4;
    for (int a = 2; a * a <= n; ++a)
        if (n % a == 0) return a;
    return n;
}
constexpr bool ct_pow2(int n) { return (n & (n - 1)) == 0; }

template <int N>
__device__ __forceinline__ void idft_any(c32 (&v)[N]) {
    if constexpr (ct_pow2(N)) {
        idft<N>(v);
    } else if constexpr (ct_first_factor(N) == N) {
        constexpr int h = (N - 1) / 2;
        c32 a[h + 1], b[h + 1];
        const c32 v0 = v[0];
        c32 sum = v0;
        static_for<1, h + 1>([&](auto jc) {
            constexpr int j = jc;
            a[j] = cadd(v[j], v[N - j]);
            b[j] = csub(v[j], v[N - j]);
            sum = cadd(sum, a[j]);
        });
        static_for<1, h + 1>([&](auto kc) {
            constexpr int k = kc;
            v2f cs = vv(v0), sn = {0.0f, 0.0f};
            static_for<1, h + 1>([&](auto jc) {
                constexpr int j = jc;
                constexpr float c = (float)ct_cos2pi(j * k, N), t = (float)ct_sin2pi(j * k, N);
                cs = __builtin_elementwise_fma(vv(a[j]), v2f{c, c}, cs);
                sn = __builtin_elementwise_fma(vv(b[j]), v2f{t, t}, sn);
            });
            v[k] = cadd_i(cc(cs), cc(sn));
            v[N - k] = csub_i(cc(cs), cc(sn));
        });
        v[0] = sum;
    } else {
        constexpr int A = ct_first_factor(N), B = N / A;
        c32 t[N];
        static_for<0, B>([&](auto n2c) {
            constexpr int n2 = n2c;
            c32 u[A];
            static_for<0, A>([&](auto n1c) { constexpr int n1 = n1c; u[n1] = v[n1 * B + n2]; });
            idft_any<A>(u);
            static_for<0, A>([&](auto k1c) {
                constexpr int k1 = k1c;
                if constexpr (k1 * n2 == 0) {
                    t[k1 * B + n2] = u[k1];
                } else {
                    constexpr float c = (float)ct_cos2pi(k1 * n2, N), sgn = (float)ct_sin2pi(k1 * n2, N);
                    t[k1 * B + n2] = cmul(u[k1], make_float2(c, sgn));
                }
            });
        });
        static_for<0, A>([&](auto k1c) {
            constexpr int k1 = k1c;
            c32 u[B];
            static_for<0, B>([&](auto n2c) { constexpr int n2 = n2c; u[n2] = t[k1 * B + n2]; });
            idft_any<B>(u);
            static_for<0, B>([&](auto k2c) { constexpr int k2 = k2c; v[k1 + A * k2] = u[k2]; });
        });
    }
}
// An odd prime P with its outputs handed out as they are made -- emit(k, X[k]) for k = 0, then the pairs (k, P - k) -- instead of
// collected in v: a pass that stores each output at once (radix_pass0: a family's outputs go back to its own slots) holds the (P - 1)
// folded inputs and one output pair, not 2 P values -- the 19-point pass of the 152-row buckets spilled 13-22 registers otherwise.
// FWD: the forward transform (conjugate twiddles: the sine sums change sign).
template <int P, bool FWD, typename Emit>
__device__ __forceinline__ void dft_prime_stream(const c32 (&v)[P], Emit&& emit) {
    static_assert(P % 2 == 1 && ct_first_factor(P) == P, "odd prime");
    constexpr int h = (P - 1) / 2;
    c32 a[h + 1], b[h + 1];
    const c32 v0 = v[0];
    c32 sum = v0;
    static_for<1, h + 1>([&](auto jc) {
        constexpr int j = jc;
        a[j] = cadd(v[j], v[P - j]);
        b[j] = csub(v[j], v[P - j]);
        sum = cadd(sum, a[j]);
    });
    emit(std::integral_constant<int, 0>{}, sum);
    static_for<1, h + 1>([&](auto kc) {
        constexpr int k = kc;
        v2f cs = vv(v0), sn = {0.0f, 0.0f};
        static_for<1, h + 1>([&](auto jc) {
            constexpr int j = jc;
            constexpr float c = (float)ct_cos2pi(j * k, P), t = (float)ct_sin2pi(j * k, P);
            cs = __builtin_elementwise_fma(vv(a[j]), v2f{c, c}, cs);
            sn = __builtin_elementwise_fma(vv(b[j]), v2f{t, t}, sn);
        });
        const c32 plus = cadd_i(cc(cs), cc(sn)), minus = csub_i(cc(cs), cc(sn));
        emit(std::integral_constant<int, k>{}, FWD ? minus : plus);
        emit(std::integral_constant<int, P - k>{}, FWD ? plus : minus);
    });
}
constexpr bool ct_stream_prime(int n) { return n >= 11 && n % 2 == 1 && ct_first_factor(n) == n; }

// forward (sign -) through the inverse codelet, as fdft
template <int N>
__device__ __forceinline__ void fdft_any(c32 (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = make_float2(v[i].y, v[i].x);
    idft_any<N>(v);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = make_float2(v[i].y, v[i].x);
}

struct AnyPlan {
    int H, W, M, S;        // M = W / 2, S = M + 1 = row stride in complex values (odd: M is even for every W = 4 q)
    int hn1, hn2;          // H = hn1 * hn2
    int mn1, mn2;          // M = mn1 * mn2
    int c2r_fuse;          // the rows' first pass forms the c2r pre-twiddle as it loads (c2r_pass0): pays when its batches of whole rows are few and full
};

// N = n1 * n2: n1 is the first-pass length.  Both factors <= kAnyCodelet: two register-codelet passes (the most balanced pair, the
// larger factor first: pass 0 needs no batching); n <= kAnyCodelet: one pass.  Otherwise (a prime factor above 16, or N > 256): a
// first pass of length n1 <= kAnyCodelet runs as a codelet (cost ~ 2 terms per value), any other pair costs (n1 + n2) / 2 terms per value as
// direct sums (conjugate output pairs share their products); pick the cheapest.
// Which lengths a kernel's run-time switch holds as register codelets (`set`): every codelet of the switch is inlined at each call
// site, and the register allocator provides for the largest of them -- with all seventeen (2 .. 16, 17, 19) every instantiation of the
// plane kernel spilled 32-74 vector registers (round 4), and a kernel with a scratch segment stalls its process' queue for 50-90 ms
// when the runtime sizes the scratch.  kSetSmall (2 .. 12: the plane kernel, 118 registers at most) and kSetLines
// (+ 16: the pass-per-launch line kernels and the column-block kernel) are spill-free; the lengths 13 - 19 exist as COMPILE-TIME factors
// only (power_buckets_*.hip), where a kernel holds exactly its four codelets.
constexpr int kSetSmall = 0, kSetLines = 1, kSetAll = 2;
constexpr bool radix_in_set(int n, int set) {
    return n >= 2 && (n <= 12 || (set >= kSetLines && n == 16) || (set >= kSetAll && (n <= kAnyCodelet || (SONAR_ANY_PRIMES && (n == 17 || n == 19)))));
}
static inline bool codelet_len(int n, int set) { return n == 1 || radix_in_set(n, set); }
static inline void best_split(int n, int& n1, int& n2, int set) {
    if (codelet_len(n, set)) {
        n1 = n;
        n2 = 1;
        return;
    }
    int best_pair = 0;
    for (int a = 2; a <= 19; ++a)
        if (n % a == 0 && codelet_len(a, set) && codelet_len(n / a, set) && n / a <= a && (best_pair == 0 || a < best_pair)) best_pair = a;
    if (best_pair) {
        n1 = best_pair;
        n2 = n / best_pair;
        return;
    }
    n1 = 1;
    n2 = n;
    float best = 0.5f * (1 + n);
    for (int a = 1; a * a <= n; ++a)
        if (n % a == 0 && 0.5f * (a + n / a) < best) {
            best = 0.5f * (a + n / a);
            n1 = a;
            n2 = n / a;
        }
    for (int r = 2; r <= 19; ++r)
        if (n % r == 0 && codelet_len(r, set) && 2.0f + 0.5f * (n / r) <= best) {
            best = 2.0f + 0.5f * (n / r);
            n1 = r;
            n2 = n / r;
        }
}

// 0 = no, 1 = yes
static inline int any_plane_ok(int64_t H, int64_t W) {
    if (H < 2 || W < 2 || (H & 1) || (W & 1) || H > kAnySlots || W > 2 * kAnySlots) return 0;
    const size_t elems = (size_t)H * (W / 2 + 1);
    return elems * sizeof(c32) + (size_t)(H + W) * sizeof(c32) <= kAnyLdsLimit ? 1 : 0;
}

template <bool NEED_T>
__device__ __forceinline__ SpectrumRng spectrum_rng_dyn(uint64_t seed, uint64_t stream_id, int64_t ggroup, int tid, int H) {
    return spectrum_seed<NEED_T>(seed, stream_id, ggroup, tid, tid < H);  // one Philox block read at three depths (power_fft.hip)
}

// draw_plane with run-time sizes (slots tid < kAnySlots): pair p -> ky = p / M, kx = 1 + p % M (handed to `pair`), partner H/2 rows below;
// the kx = M slot of a row is drawn and discarded, as in the fast path
template <bool NEED_T, typename Edge, typename Pair>
__device__ __forceinline__ void draw_plane_dyn(SpectrumRng& g, int tid, int H, int M, Edge&& edge, Pair&& pair) {
    if (tid < H) {
        const uint32_t r0 = g.E.next();
        const uint32_t rm = g.E.next();
        const uint32_t t = g.E.next();
        edge(r0, rm, t);
    }
    // (ky, kx) of pair p walk along with it: one division per plane instead of one per pair
    const int pairs = (H / 2) * M, dky = kAnySlots / M, dkx = kAnySlots - dky * M;
    int ky = tid / M, kx = 1 + tid - ky * M;
    for (int p = tid; p < pairs; p += kAnySlots) {
        const uint32_t ra = g.R.next();
        const uint32_t rb = g.R.next();
        const uint32_t t = NEED_T ? g.T.next() : 0u;
        pair(ky, kx, ra, rb, t);
        ky += dky;
        kx += dkx;
        if (kx > M) {
            kx -= M;
            ++ky;
        }
    }
}

// One pass of the two-factor line DFT over `lines` lines of N = N1 * N2 complex values (element stride es, line stride ls),
// twiddles tw[j] = e^{2 pi i j / TN} with N = TN / ts (FWD: conjugated).
//   PASS 0: out[k1 N2 + n2] = w_N^{n2 k1} sum_{n1} in[n1 N2 + n2] w_N1^{n1 k1}      (K = N1 terms, G = N2 sums per index)
//   PASS 1: out[k1 + N1 k2] = sum_{n2} in[k1 N2 + n2] w_N2^{n2 k2}                  (K = N2 terms, G = N1)
// A work item is the output PAIR (k, K - k) of one sum family: their twiddles are conjugates, so the four real products
// v.x w.x, v.y w.y, v.x w.y, v.y w.x -- four FMAs per term -- serve both (k = 0 and k = K/2 are single outputs).
// In place: lines are taken in batches of whole lines (a line's outputs depend on that line only); a batch's items -- kAnyPer
// per thread -- are gathered in registers, barrier, written, barrier.  Threads take consecutive LINES (conflict-free: odd row
// stride).
template <int NT, int PASS, bool FWD>
__device__ __forceinline__ void line_dft_pass(c32* A, const c32* __restrict__ tw, int TN, int ts, int N1, int N2, int lines, int es, int ls,
                                              int tid) {
    const int K = PASS == 0 ? N1 : N2, G = PASS == 0 ? N2 : N1, HK = K / 2 + 1;
    const int per_line = G * HK;
    const int per_batch = max(1, (NT * kAnyPer) / per_line);
    const int unit = (PASS == 0 ? N2 : N1) * ts;  // twiddle step of k = 1: w_K = w_N^{N / K}
    const float rG = 1.0f / (float)G;
#pragma unroll 1
    for (int l0 = 0; l0 < lines; l0 += per_batch) {
        const int nl = min(per_batch, lines - l0), total = nl * per_line;
        const float rnl = 1.0f / (float)nl;
        c32 lo[kAnyPer], hi[kAnyPer];
        int where[kAnyPer];  // line | k << 10 | g << 20 of the item (sizes are <= 1023), -1 = none: decoded once, used on both sides of the barrier
#pragma unroll
        for (int j = 0; j < kAnyPer; ++j) {
            const int idx = tid + j * NT;
            v2f pa = {0.0f, 0.0f}, pb = {0.0f, 0.0f};  // (sum v.x w.x, sum v.x w.y) and (sum v.y w.y, sum v.y w.x): two packed FMAs per term
            int k = 0, g = 0;
            where[j] = -1;
            if (idx < total) {
                // exact small-integer divisions through float reciprocals (idx < 4096 * 4, quotients far below 2^23)
                int it = (int)(((float)idx + 0.5f) * rnl);
                it -= it * nl > idx;
                it += (it + 1) * nl <= idx;
                const int line = l0 + idx - it * nl;
                k = (int)(((float)it + 0.5f) * rG);
                k -= k * G > it;
                k += (k + 1) * G <= it;
                g = it - k * G;
                where[j] = line | (k << 10) | (g << 20);
                const int step = unit * k;  // < TN
                const int first = PASS == 0 ? g : g * N2, sstep = (PASS == 0 ? N2 : 1) * es;
                const c32* src = A + line * ls + first * es;
                int t = 0;
#pragma unroll 2
                for (int n = 0; n < K; ++n) {
                    const v2f v = vv(src[0]), w = vv(tw[t]);
                    pa = __builtin_elementwise_fma(v.xx, w, pa);
                    pb = __builtin_elementwise_fma(v.yy, w.yx, pb);
                    src += sstep;
                    t += step;
                    if (t >= TN) t -= TN;
                }
            }
            // sum v w = (p1 - p2, p3 + p4) belongs to k for the inverse (w) and to K - k for the forward (conj w); sum v conj(w) the other
            const float p1 = pa.x, p3 = pa.y, p2 = pb.x, p4 = pb.y;
            c32 a = make_float2(p1 - p2, p3 + p4), b = make_float2(p1 + p2, p4 - p3);
            if (FWD) {
                const c32 sw = a;
                a = b;
                b = sw;
            }
            if (PASS == 0 && idx < total) {
                const int kc = k == 0 ? 0 : K - k;
                c32 wa = tw[g * k * ts], wb = tw[g * kc * ts];
                if (FWD) {
                    wa.y = -wa.y;
                    wb.y = -wb.y;
                }
                a = make_float2(a.x * wa.x - a.y * wa.y, a.x * wa.y + a.y * wa.x);
                b = make_float2(b.x * wb.x - b.y * wb.y, b.x * wb.y + b.y * wb.x);
            }
            lo[j] = a;
            hi[j] = b;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kAnyPer; ++j) {
            if (where[j] >= 0) {
                const int line = where[j] & 1023, k = (where[j] >> 10) & 1023, g = where[j] >> 20, kc = k == 0 ? 0 : K - k;
                c32* base = A + line * ls;
                const int da = PASS == 0 ? k * N2 + g : g + N1 * k, db = PASS == 0 ? kc * N2 + g : g + N1 * kc;
                base[da * es] = lo[j];
                if (kc != k) base[db * es] = hi[j];
            }
        }
        __syncthreads();
    }
}

// PASS 0 for N1 = R <= kAnyCodelet: one thread owns a whole sum family (line, n2) -- R loads, the length-R register codelet,
// R - 1 twiddles, R stores to the same slots: no batching, no barrier until the end of the pass.
template <int NT, int R, bool FWD>
__device__ __forceinline__ void radix_pass0(c32* A, const c32* __restrict__ tw, int ts, int N2, int lines, int es, int ls, int tid) {
    const int total = lines * N2;
    const float rl = 1.0f / (float)lines;
    for (int idx = tid; idx < total; idx += NT) {
        int g = (int)(((float)idx + 0.5f) * rl);
        g -= g * lines > idx;
        g += (g + 1) * lines <= idx;
        const int line = idx - g * lines;
        c32* base = A + line * ls + g * es;
        const int stride = N2 * es;
        c32 v[R];
#pragma unroll
        for (int n = 0; n < R; ++n) v[n] = base[n * stride];
        if constexpr (ct_stream_prime(R)) {
            dft_prime_stream<R, FWD>(v, [&](auto kc, c32 x) {
                constexpr int k = kc;
                if constexpr (k > 0) {
                    c32 w = tw[g * k * ts];
                    if (FWD) w.y = -w.y;
                    x = make_float2(x.x * w.x - x.y * w.y, x.x * w.y + x.y * w.x);
                }
                base[k * stride] = x;
            });
        } else {
        if (FWD) fdft_any<R>(v); else idft_any<R>(v);
#pragma unroll
        for (int k = 1; k < R; ++k) {
            c32 w = tw[g * k * ts];
            if (FWD) w.y = -w.y;
            v[k] = make_float2(v[k].x * w.x - v[k].y * w.y, v[k].x * w.y + v[k].y * w.x);
        }
#pragma unroll
        for (int k = 0; k < R; ++k) base[k * stride] = v[k];
        }
    }
    __syncthreads();
}

// PASS 1 for N2 = R <= kAnyCodelet: one thread owns the family (line, k1) -- R loads from in[k1 R + n2], the codelet, R stores to
// out[k1 + N1 k2].  Outputs land on other families' inputs of the SAME line, so lines are taken in batches of whole lines (all loads,
// barrier, all stores, barrier); lanes take consecutive lines (odd row stride: conflict-free).
template <int NT, int R, bool FWD>
__device__ __forceinline__ void codelet_pass1(c32* A, int N1, int lines, int es, int ls, int tid) {
    const int per_batch = max(1, NT / N1);
#pragma unroll 1
    for (int l0 = 0; l0 < lines; l0 += per_batch) {
        const int nl = min(per_batch, lines - l0), total = nl * N1;
        c32 v[R];
        c32* base = nullptr;
        if (tid < total) {
            int k1 = (int)(((float)tid + 0.5f) / (float)nl);
            k1 -= k1 * nl > tid;
            k1 += (k1 + 1) * nl <= tid;
            base = A + (l0 + tid - k1 * nl) * ls + k1 * es;
            const c32* src = base + k1 * (R - 1) * es;  // element k1 R of the line
#pragma unroll
            for (int n = 0; n < R; ++n) v[n] = src[n * es];
            if (FWD) fdft_any<R>(v); else idft_any<R>(v);
        }
        __syncthreads();
        if (tid < total) {
            const int stride = N1 * es;
#pragma unroll
            for (int k = 0; k < R; ++k) base[k * stride] = v[k];
        }
        __syncthreads();
    }
}

#define SONAR_ANY_RADICES(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(19)
// not inlined: four call sites per kernel, fifteen codelets per pass
#ifndef SONAR_ANY_INLINE
#define SONAR_ANY_INLINE 1
#endif
#if SONAR_ANY_INLINE
#define SONAR_ANY_LINKAGE __forceinline__
#else
#define SONAR_ANY_LINKAGE __noinline__
#endif
// CR1 / CR2 > 0: the factor is known at compile time (the SDXL buckets' kernels, power_buckets_*.hip) -- the pass IS that codelet.  The
// run-time switch over all seventeen codelets, force-inlined at four call sites, made the register allocator provide for the largest
// of them everywhere: every instantiation spilled 32-74 vector registers at its 128 (round 4), and a kernel with a scratch segment costs
// the process a 50-90 ms queue stall when the runtime sizes the scratch.
// SET: the codelets of the run-time switch (radix_in_set); a length outside it runs as direct sums (line_dft_pass).
template <int NT, bool FWD, int CR2 = 0, int SET = kSetSmall>
__device__ SONAR_ANY_LINKAGE void line_pass1(c32* A, const c32* tw, int TN, int ts, int N1, int N2, int lines, int es, int ls, int tid) {
    if constexpr (CR2 == 1) {
        return;
    } else if constexpr (CR2 > 1) {
        codelet_pass1<NT, CR2, FWD>(A, N1, lines, es, ls, tid);
        return;
    }
    switch (N2 == 1 ? 1 : radix_in_set(N2, SET) ? N2 : 0) {  // uniform
        case 1: break;  // pass 0 was the whole transform
#define SONAR_ANY_CASE(R) case R: if constexpr (radix_in_set(R, SET)) codelet_pass1<NT, R, FWD>(A, N1, lines, es, ls, tid); break;
        SONAR_ANY_RADICES(SONAR_ANY_CASE)
#undef SONAR_ANY_CASE
        default: line_dft_pass<NT, 1, FWD>(A, tw, TN, ts, N1, N2, lines, es, ls, tid);
    }
}
template <int NT, bool FWD, int CR1 = 0, int CR2 = 0, int SET = kSetSmall>
__device__ SONAR_ANY_LINKAGE void line_dft(c32* A, const c32* tw, int TN, int ts, int N1, int N2, int lines, int es, int ls, int tid) {
    if constexpr (CR1 > 0) {
        radix_pass0<NT, CR1, FWD>(A, tw, ts, N2, lines, es, ls, tid);
    } else {
        switch (radix_in_set(N1, SET) ? N1 : 0) {  // uniform
#define SONAR_ANY_CASE(R) case R: if constexpr (radix_in_set(R, SET)) radix_pass0<NT, R, FWD>(A, tw, ts, N2, lines, es, ls, tid); break;
            SONAR_ANY_RADICES(SONAR_ANY_CASE)
#undef SONAR_ANY_CASE
            default: line_dft_pass<NT, 0, FWD>(A, tw, TN, ts, N1, N2, lines, es, ls, tid);
        }
    }
    line_pass1<NT, FWD, CR2, SET>(A, tw, TN, ts, N1, N2, lines, es, ls, tid);
}

// The c2r pre-twiddle G[k] = (X[k] + conj X[M-k]) + i (X[k] - conj X[M-k]) w^k (X[0], X[M] contribute their real parts) as a pass over
// the plane of its own: every row's (k, M - k) pairs read, combined and written back.
template <int NT>
__device__ __forceinline__ void c2r_pretwiddle(c32* A, const c32* __restrict__ twW, int H, int M, int S, int tid) {
    const int Q = M / 2 + 1, qdr = NT / Q, qdk = NT - qdr * Q;  // (row, k) of item j walk along with it
    int qr = tid / Q, qk = tid - qr * Q;
    for (int j = tid; j < H * Q; j += NT) {
        const int r = qr, k = qk;
        qr += qdr;
        qk += qdk;
        if (qk >= Q) {
            qk -= Q;
            ++qr;
        }
        c32* row = A + r * S;
        if (k == 0) {
            const float x0 = row[0].x, xm = row[M].x;
            row[0] = make_float2(x0 + xm, x0 - xm);
        } else {
            const int kk = M - k;
            const c32 xa = row[k], xb = row[kk];
            {
                const c32 e = make_float2(xa.x + xb.x, xa.y - xb.y), d = make_float2(xa.x - xb.x, xa.y + xb.y);
                const c32 w = twW[k];
                const c32 o = make_float2(d.x * w.x - d.y * w.y, d.x * w.y + d.y * w.x);
                row[k] = make_float2(e.x - o.y, e.y + o.x);
            }
            if (kk != k) {
                const c32 e = make_float2(xb.x + xa.x, xb.y - xa.y), d = make_float2(xb.x - xa.x, xb.y + xa.y);
                const c32 w = twW[kk];
                const c32 o = make_float2(d.x * w.x - d.y * w.y, d.x * w.y + d.y * w.x);
                row[kk] = make_float2(e.x - o.y, e.y + o.x);
            }
        }
    }
    __syncthreads();
}

// ... or folded into the rows' first pass when that is a codelet: the thread that owns family (row, n2) forms G[n1 N2 + n2] from
// X[k] and X[M - k] as it loads (one more LDS read and the w^k table read per value; the pass of its own costs both reads, both
// writes and the item indexing).  The mirrored values belong to another family of the SAME row, so rows are taken in batches of
// whole rows: all loads, barrier, all stores, barrier.
template <int NT, int R>
__device__ __forceinline__ void c2r_pass0(c32* A, const c32* __restrict__ twW, int N2, int H, int M, int S, int tid) {
    // rows per batch: as many as the threads take, spread evenly over the batches (96 rows of 6 families: 48 + 48, not 85 + 11)
    const int most = max(1, NT / N2), batches = (H + most - 1) / most, per_batch = (H + batches - 1) / batches;
#pragma unroll 1
    for (int l0 = 0; l0 < H; l0 += per_batch) {
        const int nl = min(per_batch, H - l0), total = nl * N2;
        c32 v[R];
        c32* row = nullptr;
        int n2 = 0;
        if (tid < total) {
            n2 = (int)(((float)tid + 0.5f) / (float)nl);
            n2 -= n2 * nl > tid;
            n2 += (n2 + 1) * nl <= tid;
            row = A + (l0 + tid - n2 * nl) * S;
#pragma unroll
            for (int n1 = 0; n1 < R; ++n1) {
                const int k = n1 * N2 + n2;
                if (n1 == 0 && n2 == 0) {  // k = 0
                    const float x0 = row[0].x, xm = row[M].x;
                    v[n1] = make_float2(x0 + xm, x0 - xm);
                } else {
                    const c32 xa = row[k], xb = row[M - k], w = twW[k];
                    const c32 e = make_float2(xa.x + xb.x, xa.y - xb.y), d = make_float2(xa.x - xb.x, xa.y + xb.y);
                    const c32 o = make_float2(d.x * w.x - d.y * w.y, d.x * w.y + d.y * w.x);
                    v[n1] = make_float2(e.x - o.y, e.y + o.x);
                }
            }
            idft_any<R>(v);
#pragma unroll
            for (int k1 = 1; k1 < R; ++k1) {
                const c32 w = twW[2 * n2 * k1];  // e^{2 pi i n2 k1 / M}
                v[k1] = make_float2(v[k1].x * w.x - v[k1].y * w.y, v[k1].x * w.y + v[k1].y * w.x);
            }
        }
        __syncthreads();
        if (tid < total) {
#pragma unroll
            for (int k1 = 0; k1 < R; ++k1) row[k1 * N2 + n2] = v[k1];
        }
        __syncthreads();
    }
}

// rows of the inverse: pre-twiddle + length-M complex inverse DFT (value m of a row is then (x[2m], x[2m+1]))
#ifndef SONAR_ANY_FUSE_C2R
#define SONAR_ANY_FUSE_C2R 1
#endif
template <int NT, int CR1 = 0, int CR2 = 0, int SET = kSetSmall>
__device__ SONAR_ANY_LINKAGE void c2r_rows(c32* A, const c32* twW, int W, int N1, int N2, int H, int M, int S, int fuse, int tid) {
    if (!SONAR_ANY_FUSE_C2R || !fuse) {  // uniform
        c2r_pretwiddle<NT>(A, twW, H, M, S, tid);
        line_dft<NT, false, CR1, CR2, SET>(A, twW, W, 2, N1, N2, H, 1, S, tid);
        return;
    }
    if constexpr (CR1 > 0) {
        c2r_pass0<NT, CR1>(A, twW, N2, H, M, S, tid);
    } else {
        switch (radix_in_set(N1, SET) ? N1 : 0) {  // uniform
#define SONAR_ANY_CASE(R) case R: if constexpr (radix_in_set(R, SET)) c2r_pass0<NT, R>(A, twW, N2, H, M, S, tid); break;
            SONAR_ANY_RADICES(SONAR_ANY_CASE)
#undef SONAR_ANY_CASE
            default:
                c2r_pretwiddle<NT>(A, twW, H, M, S, tid);
                line_dft_pass<NT, 0, false>(A, twW, W, 2, N1, N2, H, 1, S, tid);
        }
    }
    line_pass1<NT, false, CR2, SET>(A, twW, W, 2, N1, N2, H, 1, S, tid);
}

// Parseval statistics of the drawn, filtered spectrum (see power_stats_kernel), run-time sizes; kAnySlots threads.  Workgroup `bid` of
// `nb`: the statistics kernel's grid, or the trailing workgroups of a generate launch that computes the NEXT call's statistics beside
// this call's planes (power_irfft2_any_kernel's StatsAhead, as in the fixed-size kernels).  EDGE: [parity][column 0 | column M][ky].
__device__ __forceinline__ void power_stats_any_body(const float* __restrict__ filter, int64_t planes, const AnyPlan& pl, uint64_t seed,
                                                     uint64_t stream_id, int64_t plane_offset, int group, int split, double* partials, int64_t bid,
                                                     int64_t nb, c32* EDGE, double* red) {
    const int H = pl.H, M = pl.M, S = pl.S;
    const int tid = threadIdx.x;
    double s = 0.0, q = 0.0;
    int par = 0;
    for (int64_t unit = bid; unit < (split ? planes : planes / group); unit += nb) {
        const GroupWalk gw(unit, group, split);
        SpectrumRng rng = spectrum_rng_dyn<false>(seed, stream_id, plane_offset / group + gw.grp, tid, H);
        for (int i = 0; i < gw.first; ++i)
            draw_plane_dyn<false>(rng, tid, H, M, [](uint32_t, uint32_t, uint32_t) {}, [](int, int, uint32_t, uint32_t, uint32_t) {});
        for (int gp = gw.first; gp < gw.first + gw.count; ++gp) {
            float acc = 0.0f;
            c32* const e0 = EDGE + (par * 2 + 0) * H;
            c32* const em = EDGE + (par * 2 + 1) * H;
            draw_plane_dyn<false>(
                rng, tid, H, M,
                [&](uint32_t r0, uint32_t rm, uint32_t t) {
                    e0[tid] = drawn_elem(r0, angle_lo(t), filter[tid * S]);
                    em[tid] = drawn_elem(rm, angle_hi(t), filter[tid * S + M]);
                },
                [&](int ky, int kx, uint32_t ra, uint32_t rb, uint32_t) {
                    if (kx < M) {
                        const float fa = filter[ky * S + kx], fb = filter[(ky + H / 2) * S + kx];
                        acc = __builtin_fmaf(fa * fa, neg_ln_u(ra), acc);
                        acc = __builtin_fmaf(fb * fb, neg_ln_u(rb), acc);
                    }
                });
            q += 2.0 * (double)acc;
            __syncthreads();
            float edge = 0.0f;
            for (int ky = tid; ky < H; ky += kFftThreads) {
                const int kn = ky == 0 ? 0 : H - ky;
                const c32 a = e0[ky], an = e0[kn], b = em[ky], bn = em[kn];
                const float ar = 0.5f * (a.x + an.x), ai = 0.5f * (a.y - an.y), br = 0.5f * (b.x + bn.x), bi = 0.5f * (b.y - bn.y);
                edge += (ar * ar + ai * ai) + (br * br + bi * bi);
                if (ky == 0) s += (double)(sqrtf((float)H * (float)pl.W) * ar);
            }
            q += (double)edge;
            par ^= 1;
        }
    }
    write_partial_at<kFftThreads>(s, q, partials, red, (int)bid, (int)nb);
}

// SRC as in power_irfft2_kernel: 0 = spectrum supplied, 1 = drawn on device, 2 = real plane in (forward, x filter, inverse)
// HN1 x HN2 = H and MN1 x MN2 = W / 2: the factor pairs when the plane size is known at compile time (0: taken from `pl` at run time)
// SET: the codelets of the run-time switches (HN1 = 0): kSetSmall, or kSetAll for the plane sizes that need a factor of 13 .. 19 and are no
// bucket (that instantiation spills 5-48 registers; direct sums instead cost those sizes 25-60 % more time than the spills do)
#ifdef SONAR_ANY_TRACE  // profiling builds (scratch/any_trace.py): thread 0's cycle stamps of a workgroup's planes
static __device__ unsigned long long g_any_trace[512 * 8 * 8];
#define SONAR_ANY_STAMP(slot) do { if (threadIdx.x == 0 && blockIdx.x < 512 && any_pidx < 8) g_any_trace[(blockIdx.x * 8 + any_pidx) * 8 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define SONAR_ANY_STAMP(slot) do { } while (0)
#endif
template <int NT, int SRC, bool STATS, bool NORM, int HN1 = 0, int HN2 = 0, int MN1 = 0, int MN2 = 0, int SET = kSetSmall>
__global__ void __launch_bounds__(NT, 4) power_irfft2_any_kernel(const float* __restrict__ z, const float* __restrict__ filter,
                                                                       float* out, int64_t planes, AnyPlan pl, uint64_t seed,
                                                                       uint64_t stream_id, int64_t plane_offset, int group, int split,
                                                                       double* partials, NormArgs na, StatsAhead sa) {
    kernarg_touch_for(z, filter, out, planes, pl, seed, stream_id, plane_offset, group, split, partials, na, sa);
    extern __shared__ __align__(16) unsigned char any_lds[];
    __shared__ double red[2 * NT / 64];
    __shared__ NormDecision shd;
    // a bucket's plane size is its factor pairs' product: every stride, trip count and division of the passes folds at compile time
    const int H = HN1 > 0 ? HN1 * HN2 : pl.H, M = MN1 > 0 ? MN1 * MN2 : pl.M, W = 2 * M, S = M + 1, NC = H * S;
    if constexpr (HN1 > 0) {
        pl.H = H, pl.W = W, pl.M = M, pl.S = S;
        pl.hn1 = HN1, pl.hn2 = HN2, pl.mn1 = MN1, pl.mn2 = MN2;
    }
    c32* const A = reinterpret_cast<c32*>(any_lds);
    int64_t bid = blockIdx.x, nblk = gridDim.x;
    if constexpr (NT == kFftThreads && SRC == 1 && NORM && !STATS) {  // the launch-bound batch sizes: the next call's statistics in this launch
        if (sa.partials) {
            if ((int)blockIdx.x >= sa.main_blocks) {
                power_stats_any_body(filter, planes, pl, seed, sa.stream_id, plane_offset, group, split, sa.partials, bid - sa.main_blocks,
                                     nblk - sa.main_blocks, A /* 4 H values: the launcher checks they fit the plane buffer */, red);
                return;
            }
            nblk = sa.main_blocks;
        }
    }
    c32* const twH = A + NC;   // e^{2 pi i j / H}
    c32* const twW = twH + H;  // e^{2 pi i j / W}
    const int tid = threadIdx.x;
    for (int j = tid; j < H + W; j += NT) {
        const int n = j < H ? H : W, i = j < H ? j : j - H;
        double sn, cs;
        sincospi(2.0 * (double)i / (double)n, &sn, &cs);
        twH[j] = make_float2((float)cs, (float)sn);
    }
    float scale = SRC == 2 ? 1.0f / ((float)H * (float)W) : 1.0f / sqrtf((float)H * (float)W);
    float nm = scale, nc = 0.0f;
    if constexpr (NORM) {
        const NormDecision dec = decide_norm<NT>(na.partials, kNPart, na.n_total, na.thr_sd, red, &shd);
        const float g = (dec.do_div ? 1.0f / dec.stdv : 1.0f) * na.factor;
        nm = scale * g;
        nc = dec.do_sub ? dec.mean * g : 0.0f;
    }
    double s = 0.0, q = 0.0;
    [[maybe_unused]] int any_pidx = 0;  // (trace builds)
    for (int64_t unit = bid; unit < group_units(planes, group, split); unit += nblk) {  // (split may be the mixed form: power_core.h)
        const GroupWalk gw(unit, group, split);
        SpectrumRng rng;
        if constexpr (SRC == 1) {
            if (tid < kAnySlots) {
                rng = spectrum_rng_dyn<true>(seed, stream_id, plane_offset / group + gw.grp, tid, H);
                for (int i = 0; i < gw.first; ++i)
                    draw_plane_dyn<true>(rng, tid, H, M, [](uint32_t, uint32_t, uint32_t) {}, [](int, int, uint32_t, uint32_t, uint32_t) {});
            }
        }
        for (int gp = gw.first; gp < gw.first + gw.count; ++gp) {
            const int64_t plane = gw.grp * group + gp;
            __syncthreads();  // previous plane fully consumed (and the twiddle tables visible)
            SONAR_ANY_STAMP(0);
            // With the sizes known at compile time every LDS address of a plane is loop-invariant and the optimiser hoists them ALL out of
            // the plane loop -- and spills 2-31 registers to hold them.  An opaque copy of the thread index per plane keeps the address
            // arithmetic (now a few constant multiplies) inside the loop, as in the fixed-size kernels (power_fft.hip).
            int ptid = tid;
            if constexpr (HN1 > 0) asm volatile("" : "+v"(ptid));
            if constexpr (SRC == 1) {
                if (tid < kAnySlots)
                    draw_plane_dyn<true>(
                        rng, tid, H, M,
                        [&](uint32_t r0, uint32_t rm, uint32_t t) {
                            A[tid * S] = drawn_elem(r0, angle_lo(t), filter[tid * S]);
                            A[tid * S + M] = drawn_elem(rm, angle_hi(t), filter[tid * S + M]);
                        },
                        [&](int ky, int kx, uint32_t ra, uint32_t rb, uint32_t t) {
                            if (kx < M) {
                                const int a = ky * S + kx, b = a + (H / 2) * S;
                                A[a] = drawn_elem(ra, angle_lo(t), filter[a]);
                                A[b] = drawn_elem(rb, angle_hi(t), filter[b]);
                            }
                        });
            } else if constexpr (SRC == 0) {
                const c32* zp = reinterpret_cast<const c32*>(z) + plane * NC;
                for (int j = tid; j < NC; j += NT) {
                    const c32 v = zp[j];
                    const float f = filter[j];
                    A[j] = make_float2(v.x * f, v.y * f);
                }
            } else {
                // ---- forward r2c: rows as W/2 complex values, forward DFT, split into the half-spectrum, forward columns, x filter
                const float* xin = z + plane * (int64_t)H * W;
                if (SONAR_ANY_STORE16 && (M & 1) == 0 && (reinterpret_cast<uintptr_t>(z) & 15u) == 0) {  // uniform: 16-byte loads, two values per item
                    const int Mh = M >> 1, hdr = NT / Mh, hdm = NT - hdr * Mh;
                    int hr = tid / Mh, hm = tid - hr * Mh;
                    for (int j = tid; j < H * Mh; j += NT) {
                        const int r = hr, m = 2 * hm;
                        hr += hdr;
                        hm += hdm;
                        if (hm >= Mh) {
                            hm -= Mh;
                            ++hr;
                        }
                        const float4 q4 = *reinterpret_cast<const float4*>(xin + (int64_t)r * W + 2 * m);
                        A[r * S + m] = make_float2(q4.x, q4.y);
                        A[r * S + m + 1] = make_float2(q4.z, q4.w);
                    }
                } else {
                    const int ldr = NT / M, ldm = NT - ldr * M;
                    int lr = tid / M, lm = tid - lr * M;
                    for (int j = tid; j < H * M; j += NT) {
                        const int r = lr, m = lm;
                        lr += ldr;
                        lm += ldm;
                        if (lm >= M) {
                            lm -= M;
                            ++lr;
                        }
                        A[r * S + m] = *reinterpret_cast<const float2*>(xin + (int64_t)r * W + 2 * m);
                    }
                }
                __syncthreads();
                line_dft<NT, true, MN1, MN2, SET>(A, twW, W, 2, pl.mn1, pl.mn2, H, 1, S, ptid);
                // X[k] = E + w^k O, X[M-k] = conj(E - w^k O), E = (C[k] + conj C[M-k]) / 2, O = (C[k] - conj C[M-k]) / 2i, w = e^{-2 pi i / W}
                const int FQ = M / 2 + 1, fdr = NT / FQ, fdk = NT - fdr * FQ;
                int fr = tid / FQ, fk = tid - fr * FQ;
                for (int j = tid; j < H * FQ; j += NT) {
                    const int r = fr, k = fk;
                    fr += fdr;
                    fk += fdk;
                    if (fk >= FQ) {
                        fk -= FQ;
                        ++fr;
                    }
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
                        const c32 w = twW[k];  // conj -> e^{-2 pi i k / W}
                        const c32 t = make_float2(o.x * w.x + o.y * w.y, o.y * w.x - o.x * w.y);
                        row[k] = make_float2(e.x + t.x, e.y + t.y);
                        if (kk != k) row[kk] = make_float2(e.x - t.x, -(e.y - t.y));
                    }
                }
                __syncthreads();
                line_dft<NT, true, HN1, HN2, SET>(A, twH, H, 1, pl.hn1, pl.hn2, S, S, 1, ptid);
                for (int j = tid; j < NC; j += NT) {
                    const float f = filter[j];
                    c32 v = A[j];
                    v.x *= f;
                    v.y *= f;
                    A[j] = v;
                }
            }
            SONAR_ANY_STAMP(1);
            __syncthreads();
            SONAR_ANY_STAMP(2);
            // ---- inverse columns: every one of the W/2 + 1 columns, length H
            line_dft<NT, false, HN1, HN2, SET>(A, twH, H, 1, pl.hn1, pl.hn2, S, S, 1, ptid);
            SONAR_ANY_STAMP(3);
            // ---- rows: c2r pre-twiddle + length-M complex inverse DFT; value m of a row is (x[2m], x[2m+1])
            c2r_rows<NT, MN1, MN2, SET>(A, twW, W, pl.mn1, pl.mn2, H, M, S, pl.c2r_fuse, ptid);
            SONAR_ANY_STAMP(4);
            float* const oplane = out + plane * (int64_t)H * W;
            float ps = 0.0f, pq = 0.0f;
            if (SONAR_ANY_STORE16 && (M & 1) == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0) {  // uniform
                // two adjacent values of a row per item: 16-byte stores (the store phase is bound by its memory instructions)
                const int Mh = M >> 1, hdr = NT / Mh, hdm = NT - hdr * Mh;
                int hr = tid / Mh, hm = tid - hr * Mh;
                for (int j = tid; j < H * Mh; j += NT) {
                    const int r = hr, m = 2 * hm;
                    hr += hdr;
                    hm += hdm;
                    if (hm >= Mh) {
                        hm -= Mh;
                        ++hr;
                    }
                    const c32 g0 = A[r * S + m], g1 = A[r * S + m + 1];
                    float4 o;
                    if constexpr (NORM) {
                        o = make_float4(__builtin_fmaf(g0.x, nm, -nc), __builtin_fmaf(g0.y, nm, -nc), __builtin_fmaf(g1.x, nm, -nc), __builtin_fmaf(g1.y, nm, -nc));
                    } else {
                        o = make_float4(g0.x * scale, g0.y * scale, g1.x * scale, g1.y * scale);
                    }
                    *reinterpret_cast<float4*>(oplane + (int64_t)r * W + 2 * m) = o;
                    if constexpr (STATS) {
                        ps += (o.x + o.y) + (o.z + o.w);
                        pq = __builtin_fmaf(o.x, o.x, __builtin_fmaf(o.y, o.y, __builtin_fmaf(o.z, o.z, __builtin_fmaf(o.w, o.w, pq))));
                    }
                }
            } else {
            const int sdr = NT / M, sdm = NT - sdr * M;
            int sr = tid / M, sm = tid - sr * M;
            for (int j = tid; j < H * M; j += NT) {
                const int r = sr, m = sm;
                sr += sdr;
                sm += sdm;
                if (sm >= M) {
                    sm -= M;
                    ++sr;
                }
                const c32 g = A[r * S + m];
                float a, b;
                if constexpr (NORM) {
                    a = __builtin_fmaf(g.x, nm, -nc);
                    b = __builtin_fmaf(g.y, nm, -nc);
                } else {
                    a = g.x * scale;
                    b = g.y * scale;
                }
                *reinterpret_cast<float2*>(oplane + (int64_t)r * W + 2 * m) = make_float2(a, b);
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
            SONAR_ANY_STAMP(5);
            ++any_pidx;
        }
    }
    if constexpr (STATS) write_partial<NT>(s, q, partials, red);
}

// the look-ahead of the general-size kernels (launch_power_any): two workgroups per CU, every workgroup resident at once
static inline bool any_ahead_ok(int64_t planes, int64_t H, int64_t W, int group) {
    if (!any_plane_ok(H, W) || W < 6 || group < 1 || planes < 1 || planes % group) return false;
    const size_t lds = ((size_t)H * (W / 2 + 1) + H + W) * sizeof(c32);
    if (2 * (lds + 1024) > 160 * 1024) return false;
    const int split = group > 1 && planes / group < 512 ? 1 : 0;
    return (split ? planes : planes / group) <= 256;
}

// does a plane size (no bucket) need a codelet outside kSetSmall?  Then its kernel is the kSetAll instantiation (power_any_all.hip).
static inline bool any_needs_all(int64_t H, int64_t W) {
    int a, b, c2, d;
    best_split((int)H, a, b, kSetAll);
    best_split((int)W / 2, c2, d, kSetAll);
    auto big = [](int n) { return n > 1 && radix_in_set(n, kSetAll) && !radix_in_set(n, kSetSmall); };
    return big(a) || big(b) || big(c2) || big(d);
}
// the kernels of one call (what: as launch_power), with the plane kernel's factor pairs as template arguments (0: run time)
constexpr int kNotABucket = -1000;
__global__ void __launch_bounds__(kFftThreads) power_stats_any_kernel(const float* __restrict__ filter, int64_t planes, AnyPlan pl, uint64_t seed,
                                                                      uint64_t stream_id, int64_t plane_offset, int group, int split, double* partials);
__global__ void __launch_bounds__(kFftThreads) power_spectrum_any_kernel(float* zout, int64_t planes, AnyPlan pl, uint64_t seed, uint64_t stream_id,
                                                                         int64_t plane_offset, int group, int split);
template <int HN1, int HN2, int MN1, int MN2, int SET = kSetSmall>
static int launch_power_any_t(int what, const float* z, const float* filter, float* out, int64_t planes, int64_t H, int64_t W, uint64_t seed,
                              uint64_t stream_id, int64_t plane_offset, int group, double* partials, NormArgs na, hipStream_t st, Ahead ah) {
    AnyPlan pl;
    pl.H = (int)H;
    pl.W = (int)W;
    pl.M = (int)W / 2;
    pl.S = pl.M + 1;
    if (HN1 > 0) {  // a bucket: its factor pairs are the kernel's (they are what best_split picks among ALL codelet lengths)
        if (HN1 * HN2 != pl.H || MN1 * MN2 != pl.M) return kNotABucket;
        pl.hn1 = HN1, pl.hn2 = HN2, pl.mn1 = MN1, pl.mn2 = MN2;
    } else {
        best_split(pl.H, pl.hn1, pl.hn2, SET);
        best_split(pl.M, pl.mn1, pl.mn2, SET);
    }
    const size_t lds = ((size_t)H * pl.S + H + W) * sizeof(c32);
    {
        // measured (scratch/size_sweep.py, fused against the pre-twiddle pass of its own): 104 x 152 (416 families, one batch) 155 ->
        // 143 us, 144 x 112 (2 x 511) 142 -> 139; 96 x 96 (2 x 288) 137 -> 140, 192 x 192 and 160 x 160 (three batches) +6 %
        const int nt = 2 * (lds + 1024) <= 160 * 1024 ? kAnySlots : kAnyThreads;
        const int families = pl.H * pl.mn2, batches = (pl.H + std::max(1, nt / pl.mn2) - 1) / std::max(1, nt / pl.mn2);
        pl.c2r_fuse = (HN1 > 0 || codelet_len(pl.mn1, SET)) && pl.mn1 > 1 && batches <= 2 && 5 * families >= 4 * batches * nt ? 1 : 0;
    }
    const size_t lds_stats = (size_t)4 * H * sizeof(c32);
    const int split = group > 1 && planes / group < 512 ? 1 : 0;
    const int64_t units = split ? planes : planes / group;
    // resident workgroups: 16 waves per CU at the codelets' 128-register budget -- two 512-thread workgroups when two plane buffers fit
    // (two planes in flight per CU: one's barriers under the other's passes), else one of 1024 threads
    const int per_cu = 2 * (lds + 1024) <= 160 * 1024 ? 2 : 1;
    // The plane kernel's units (generate mode): whole groups for the launch's full rounds of workgroups, single planes for what is left
    // when that is cheaper than a last round of whole groups on a few workgroups (a single-plane unit costs ~1.3 planes: it seeds and
    // skips).  530 latents of 104 x 152 on 512 workgroups: 2 x 4 plane-times -> 4 + 1.3 (`scratch/any_trace.py`: 14.5 us per plane).
    int split_main = split;
    if (group > 1 && !split && what <= 1 && z == nullptr && !ah.next && planes / group < (int64_t)1 << 29) {
        const int64_t slots = std::min<int64_t>(256 * per_cu, kNPart), G = planes / group, rounds = G / slots, tail = G % slots;
        if (tail > 0) {
            const double whole = (double)group, single = 1.3 * (double)((tail * group + slots - 1) / slots);
            if (single < whole) split_main = 2 + (int)(rounds * slots);
        }
    }
    const int64_t units_main = group_units(planes, group, split_main);
    const int g = (int)std::min<int64_t>(std::min<int64_t>(units_main, 256 * per_cu), kNPart);
#define SONAR_PA_NT(NT, G, ST, NM, PART)                                                                                                   \
    do {                                                                                                                                   \
        auto kern = power_irfft2_any_kernel<NT, G, ST, NM, HN1, HN2, MN1, MN2, SET>;                                                                                \
        lds_attr(reinterpret_cast<const void*>(kern), (int)kAnyLdsLimit);                                                                 \
        hipLaunchKernelGGL(kern, dim3(g), dim3(NT), lds, st, z, filter, out, planes, pl, seed, stream_id, plane_offset, group, split_main, \
                           PART, na, StatsAhead());                                                                                        \
    } while (0)
#define SONAR_PA(G, ST, NM, PART)                                                                                                          \
    do {                                                                                                                                   \
        if (per_cu == 2) SONAR_PA_NT(kAnySlots, G, ST, NM, PART);                                                                          \
        else SONAR_PA_NT(kAnyThreads, G, ST, NM, PART);                                                                                    \
    } while (0)
    if (what == 3) {
        if (partials) SONAR_PA(2, true, false, partials); else SONAR_PA(2, false, false, partials);
    } else if (what == 2) {
        hipLaunchKernelGGL(power_spectrum_any_kernel, dim3((int)std::min<int64_t>(units, 2048)), dim3(kFftThreads), 0, st, out, planes, pl, seed,
                           stream_id, plane_offset, group, split);
    } else if (what == 1) {
        if (!ah.have_stats)
            hipLaunchKernelGGL(power_stats_any_kernel, dim3((int)std::min<int64_t>(units, kNPart)), dim3(kFftThreads), lds_stats, st, filter, planes,
                               pl, seed, stream_id, plane_offset, group, split, partials);
        if (ah.next) {  // any_ahead_ok: the next call's statistics as extra workgroups of this launch
            StatsAhead sa;
            sa.partials = ah.next;
            sa.stream_id = ah.next_stream;
            sa.main_blocks = g;
            auto kern = power_irfft2_any_kernel<kAnySlots, 1, false, true, HN1, HN2, MN1, MN2, SET>;
            lds_attr(reinterpret_cast<const void*>(kern), (int)kAnyLdsLimit);
            hipLaunchKernelGGL(kern, dim3(g + (int)std::min<int64_t>(units, kNPart)), dim3(kAnySlots), lds, st, z, filter, out, planes, pl, seed, stream_id,
                               plane_offset, group, split, (double*)nullptr, na, sa);
        } else {
            SONAR_PA(1, false, true, nullptr);
        }
    } else if (z == nullptr) {
        if (partials) SONAR_PA(1, true, false, partials); else SONAR_PA(1, false, false, partials);
    } else {
        if (partials) SONAR_PA(0, true, false, partials); else SONAR_PA(0, false, false, partials);
    }
#undef SONAR_PA
#undef SONAR_PA_NT
    return check_launch("sonar_power_* (general-size plane)");
}

}  // namespace sonar
