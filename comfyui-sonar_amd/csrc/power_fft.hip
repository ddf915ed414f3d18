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

__device__ __forceinline__ c32 cadd(c32 a, c32 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ c32 csub(c32 a, c32 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ c32 cmul(c32 a, c32 b) {
    return make_float2(__builtin_fmaf(a.x, b.x, -(a.y * b.y)), __builtin_fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ c32 cmul_i(c32 a) { return make_float2(-a.y, a.x); }  // a * (+i)

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
    const c32 t2 = cadd(v[1], v[3]), t3 = cmul_i(csub(v[1], v[3]));
    v[0] = cadd(t0, t2);
    v[2] = csub(t0, t2);
    v[1] = cadd(t1, t3);
    v[3] = csub(t1, t3);
}

template <>
__device__ __forceinline__ void idft<8>(c32 (&v)[8]) {
    constexpr float r = 0.70710678118654752f;
    c32 e[4] = {v[0], v[2], v[4], v[6]};
    c32 o[4] = {v[1], v[3], v[5], v[7]};
    idft<4>(e);
    idft<4>(o);
    const c32 t0 = o[0];
    const c32 t1 = make_float2(r * (o[1].x - o[1].y), r * (o[1].x + o[1].y));   // * e^{i pi/4}
    const c32 t2 = cmul_i(o[2]);                                                // * i
    const c32 t3 = make_float2(-r * (o[3].x + o[3].y), r * (o[3].x - o[3].y));  // * e^{3 i pi/4}
    v[0] = cadd(e[0], t0); v[4] = csub(e[0], t0);
    v[1] = cadd(e[1], t1); v[5] = csub(e[1], t1);
    v[2] = cadd(e[2], t2); v[6] = csub(e[2], t2);
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
    t[2] = make_float2(r * (o[2].x - o[2].y), r * (o[2].x + o[2].y));
    t[3] = cmul(o[3], make_float2(s1, c1));
    t[4] = cmul_i(o[4]);
    t[5] = cmul(o[5], make_float2(-s1, c1));
    t[6] = make_float2(-r * (o[6].x + o[6].y), r * (o[6].x - o[6].y));
    t[7] = cmul(o[7], make_float2(-c1, s1));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v[i] = cadd(e[i], t[i]);
        v[i + 8] = csub(e[i], t[i]);
    }
}

// e^{+2 pi i idx / N} for N | 256
template <int N>
__device__ __forceinline__ c32 twiddle(int idx) {
    return c_tw256[(idx * (256 / N)) & 255];
}

constexpr int split_n1(int n) { return n >= 256 ? 16 : n >= 64 ? 8 : n == 32 ? 4 : n == 16 ? 2 : 1; }

template <int H, int W>
struct PlaneCfg {
    static constexpr int M = W / 2;       // complex length of the c2r stage
    static constexpr int Wh = M + 1;      // half-spectrum width
    static constexpr int S = M + 1;       // LDS row stride (complex): odd -> rows hit distinct banks
    static constexpr int CN1 = split_n1(H), CN2 = H / CN1;
    static constexpr int RN1 = split_n1(M), RN2 = M / RN1;
    static constexpr int kLdsComplex = H * S + H;  // plane + side column
    static constexpr size_t kLdsBytes = (size_t)kLdsComplex * sizeof(c32);
};

template <int H, int W, bool GEN, bool STATS>
__global__ void __launch_bounds__(kBlock) power_irfft2_kernel(const float* __restrict__ z,
                                                               const float* __restrict__ filter, float* out,
                                                               int64_t planes, uint64_t seed, uint64_t stream_id,
                                                               int64_t cplx_offset, double* partials) {
    using C = PlaneCfg<H, W>;
    constexpr int M = C::M, Wh = C::Wh, S = C::S;
    constexpr int CN1 = C::CN1, CN2 = C::CN2, RN1 = C::RN1, RN2 = C::RN2;
    constexpr int NC = H * Wh;  // complex per plane (even, H is even)
    __shared__ c32 A[C::kLdsComplex];
    c32* const T = A + H * S;
    __shared__ double red[2 * kBlock / 64];
    const int tid = threadIdx.x;
    const float scale = 1.0f / sqrtf((float)H * (float)W);  // norm="ortho"
    double s = 0.0, q = 0.0;

    for (int64_t plane = blockIdx.x; plane < planes; plane += gridDim.x) {
        __syncthreads();  // previous plane's LDS reads are done
        // ---------------------------------------------------------------- fill: z * filter
        for (int j = tid; j < NC / 2; j += kBlock) {
            const int c0 = 2 * j;
            c32 z0, z1;
            if constexpr (GEN) {
                float n[4];
                philox_normal4(seed, stream_id, (uint64_t)((cplx_offset + plane * NC + c0) >> 1), n);
                constexpr float rs = 0.70710678118654752f;  // complex normal: (a + ib) * sqrt(1/2)
                z0 = make_float2(n[0] * rs, n[1] * rs);
                z1 = make_float2(n[2] * rs, n[3] * rs);
            } else {
                const float4 t = *reinterpret_cast<const float4*>(z + (plane * NC + c0) * 2);
                z0 = make_float2(t.x, t.y);
                z1 = make_float2(t.z, t.w);
            }
            const float2 f = *reinterpret_cast<const float2*>(filter + c0);
            z0.x *= f.x; z0.y *= f.x;
            z1.x *= f.y; z1.y *= f.y;
            const int ky0 = c0 / Wh, kx0 = c0 - ky0 * Wh;
            const int c1 = c0 + 1;
            const int ky1 = c1 / Wh, kx1 = c1 - ky1 * Wh;
            if (kx0 < M) A[ky0 * S + kx0] = z0; else T[ky0] = z0;
            if (kx1 < M) A[ky1 * S + kx1] = z1; else T[ky1] = z1;
        }
        __syncthreads();
        // ---------------------------------------------------------------- fix-up of columns 0 / M
        {
            c32 qv[(H + kBlock - 1) / kBlock];
#pragma unroll
            for (int i = 0; i < (H + kBlock - 1) / kBlock; ++i) {
                const int ky = tid + i * kBlock;
                if (ky < H) {
                    const int kn = (H - ky) & (H - 1);
                    const c32 a = A[ky * S], an = A[kn * S], b = T[ky], bn = T[kn];
                    const c32 z0s = make_float2(0.5f * (a.x + an.x), 0.5f * (a.y - an.y));
                    const c32 zms = make_float2(0.5f * (b.x + bn.x), 0.5f * (b.y - bn.y));
                    qv[i] = make_float2(z0s.x - zms.y, z0s.y + zms.x);
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < (H + kBlock - 1) / kBlock; ++i) {
                const int ky = tid + i * kBlock;
                if (ky < H) A[ky * S] = qv[i];
            }
        }
        __syncthreads();
        // ---------------------------------------------------------------- columns, pass a
        for (int item = tid; item < CN2 * M; item += kBlock) {
            const int c = item % M;
            int n2 = item / M;
            if constexpr (M % 64 == 0) n2 = __builtin_amdgcn_readfirstlane(n2);
            c32 v[CN1];
#pragma unroll
            for (int n1 = 0; n1 < CN1; ++n1) v[n1] = A[(CN2 * n1 + n2) * S + c];
            idft<CN1>(v);
#pragma unroll
            for (int k1 = 1; k1 < CN1; ++k1) v[k1] = cmul(v[k1], twiddle<H>(n2 * k1));
#pragma unroll
            for (int k1 = 0; k1 < CN1; ++k1) A[(CN2 * k1 + n2) * S + c] = v[k1];
        }
        __syncthreads();
        // ---------------------------------------------------------------- columns, pass b
        // LDS row r = CN2*k1 + k2 now holds spatial row y = k1 + CN1*k2
        for (int item = tid; item < CN1 * M; item += kBlock) {
            const int c = item % M;
            const int k1 = item / M;
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
            constexpr int ITEMS = (RN2 * H + kBlock - 1) / kBlock;
            c32 g[ITEMS][RN1];
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int item = tid + it * kBlock;
                if (item < RN2 * H) {
                    const int r = item % H;
                    int n2 = item / H;
                    if constexpr (H % 64 == 0) n2 = __builtin_amdgcn_readfirstlane(n2);
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
                        const c32 o = cmul(d, twiddle<W>(k));
                        g[it][n1] = make_float2(e.x - o.y, e.y + o.x);
                    }
                    idft<RN1>(g[it]);
#pragma unroll
                    for (int k1 = 1; k1 < RN1; ++k1) g[it][k1] = cmul(g[it][k1], twiddle<M>(n2 * k1));
                }
            }
            __syncthreads();  // every mirrored read is done before anyone overwrites
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int item = tid + it * kBlock;
                if (item < RN2 * H) {
                    const int r = item % H;
                    const int n2 = item / H;
#pragma unroll
                    for (int k1 = 0; k1 < RN1; ++k1) A[r * S + RN2 * k1 + n2] = g[it][k1];
                }
            }
        }
        __syncthreads();
        // ---------------------------------------------------------------- rows, pass b -> global
        float* const oplane = out + plane * (int64_t)H * W;
        for (int item = tid; item < RN1 * H; item += kBlock) {
            const int k1 = item % RN1;
            const int r = item / RN1;
            const int y = (r / CN2) + CN1 * (r % CN2);
            c32 u[RN2];
#pragma unroll
            for (int n2 = 0; n2 < RN2; ++n2) u[n2] = A[r * S + RN2 * k1 + n2];
            idft<RN2>(u);
            float* orow = oplane + (int64_t)y * W;
#pragma unroll
            for (int k2 = 0; k2 < RN2; ++k2) {
                const float a = u[k2].x * scale, b = u[k2].y * scale;
                *reinterpret_cast<float2*>(orow + 2 * (k1 + RN1 * k2)) = make_float2(a, b);
                if constexpr (STATS) {
                    const double da = a, db = b;
                    s += da; q += da * da;
                    s += db; q += db * db;
                }
            }
        }
    }
    if constexpr (STATS) write_partial<kBlock>(s, q, partials, red);
}

template <int H, int W>
static int launch_power(const float* z, const float* filter, float* out, int64_t planes, uint64_t seed,
                        uint64_t stream_id, int64_t cplx_offset, double* partials, hipStream_t st) {
    using C = PlaneCfg<H, W>;
    static_assert(C::kLdsBytes + 64 <= 160 * 1024, "plane does not fit in LDS");
    // blocks/CU by LDS; persistent grid of resident blocks (<= kNPart so each owns a partial slot)
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / (C::kLdsBytes + 64)));
    const int g = (int)std::min<int64_t>(std::min<int64_t>(planes, (int64_t)256 * per_cu), kNPart);
#define SONAR_PW(G, ST) \
    hipLaunchKernelGGL((power_irfft2_kernel<H, W, G, ST>), dim3(g), dim3(kBlock), 0, st, z, filter, out, planes, seed, stream_id, cplx_offset, partials)
    if (z == nullptr) {
        if (partials) SONAR_PW(true, true); else SONAR_PW(true, false);
    } else {
        if (partials) SONAR_PW(false, true); else SONAR_PW(false, false);
    }
#undef SONAR_PW
    return check_launch("sonar_power_irfft2_f32");
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

using namespace sonar;

extern "C" int sonar_power_irfft2_f32(const float* z, const float* filter, float* out, int64_t planes, int64_t H,
                                      int64_t W, uint64_t seed, uint64_t stream_id, int64_t cplx_offset,
                                      double* partials, void* stream) {
    SONAR_REQUIRE(filter && out && planes >= 0 && H > 0 && W > 0 && cplx_offset >= 0, SONAR_ERR_ARG,
                  "sonar_power_irfft2_f32: bad argument");
    SONAR_REQUIRE((cplx_offset & 1) == 0, SONAR_ERR_ARG, "sonar_power_irfft2_f32: cplx_offset must be even");
    SONAR_REQUIRE((reinterpret_cast<uintptr_t>(out) & 7u) == 0 && (reinterpret_cast<uintptr_t>(filter) & 7u) == 0 &&
                      (z == nullptr || (reinterpret_cast<uintptr_t>(z) & 15u) == 0),
                  SONAR_ERR_ARG, "sonar_power_irfft2_f32: misaligned buffer");
    if (planes == 0) return SONAR_OK;
    hipStream_t st = (hipStream_t)stream;
#define SONAR_CASE(HH, WW) \
    if (H == HH && W == WW) return launch_power<HH, WW>(z, filter, out, planes, seed, stream_id, cplx_offset, partials, st)
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
    set_error("sonar_power_irfft2_f32: unsupported plane %lld x %lld (powers of two, 16..256, LDS-resident)",
              (long long)H, (long long)W);
    return SONAR_ERR_UNSUPPORTED;
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
