// Shared device/host helpers for libsonar_hip.so (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
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
        const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
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
    const float r = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
    z0 = r * __builtin_amdgcn_cosf(u2);
    z1 = r * __builtin_amdgcn_sinf(u2);
}

__device__ __forceinline__ void philox_normal4(uint64_t seed, uint64_t stream_id, uint64_t group, float (&z)[4]) {
    const Philox4 p = philox_group(seed, stream_id, group);
    box_muller(p.v[0], p.v[1], z[0], z[1]);
    box_muller(p.v[2], p.v[3], z[2], z[3]);
}

// ---- reductions -----------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
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
template <int BLOCK>
__device__ __forceinline__ void write_partial(double s, double q, double* partials, double* red) {
    block_sum2<BLOCK>(s, q, red);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x + 0] = s;
        partials[2 * blockIdx.x + 1] = q;
    }
    if (blockIdx.x == 0)
        for (int j = gridDim.x + threadIdx.x; j < kNPart; j += BLOCK) {
            partials[2 * j] = 0.0;
            partials[2 * j + 1] = 0.0;
        }
}

// ---- blend modes (py/utils.py:17-21) -----------------------------------------------------------
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

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
