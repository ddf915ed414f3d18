// Generated power-law noise on planes whose half-spectrum does not fit in LDS (256 x 256 = a 2048 px latent, 512 x 512, 384 x 512 ...):
// the spectrum is drawn on device in BLOCKS OF COLUMNS, a block is filtered and column-transformed in LDS and written once to a complex
// workspace; the row pass (lines_c2r_kernel, power_any.h) reads it once and writes the tensor once, normalised when the Parseval
// statistics of the draw (power_block_stats_kernel: no transform, no angles outside the two edge columns) were computed first:
//     statistics (no HBM traffic) -> draw + filter + columns (write W) -> rows c2r + normalise (read W, write out)      ~ 3 x the tensor
// against white noise -> rfft2 -> x filter -> irfft2 -> scale (round 3's route for these planes: ~ 10 x the tensor).
// (py/nodes/powernoise.py:338-366: the reference multiplies the rfft2 of white noise by the filter; the rfft2 of white noise IS a
// complex-normal half-spectrum, drawn directly here as for the LDS-resident planes.)
//
// Stream definition (seed compatibility): the S = W/2 + 1 columns of a plane split into nblk = ceil(S / 32) draw blocks of
// bw = ceil(S / nblk) columns (the last one may be narrower); draw block d of RNG group g owns the 512 slots d * 512 + tid of that
// group (spectrum_seed: Philox counter = (group, stream id, slot)); slot `tid` draws the pairs p = tid, tid + 512, ... of its block,
// pair p = element (ky = p / ncd, column c0 + p % ncd) and its partner H/2 rows below: two radius words from R, one angle word
// from T (low / high half).  Every column is an ordinary column (no E streams): the row pass takes the real parts of columns 0 and M
// after the column transform, like the general-size LDS kernels.  Included by power_fft.hip after power_any.h.
#pragma once

namespace sonar {

constexpr int kBlockSlots = 512;  // = kLinesThreads: line_dft<512, ...> is shared with the workspace passes
constexpr int kBlockColsMax = 32;
static_assert(kBlockSlots == kLinesThreads, "the block kernels run the line passes' transforms");

struct BlockPlan {
    int H, W, M, S;  // M = W / 2, S = M + 1 columns
    int nblk, bw;    // draw blocks and their width
    int hn1, hn2;    // H = hn1 * hn2 (best_split)
};

// 0 = no, 1 = yes: even H <= 512 (a block of 32 columns fits LDS), even W in [4, 2048], not an LDS-resident plane
static inline int block_plane_ok(int64_t H, int64_t W) {
    if (H < 2 || W < 4 || (H & 1) || (W & 1) || H > 512 || W > kLinesMax) return 0;
    return 1;
}

static inline BlockPlan block_plan(int64_t H, int64_t W) {
    BlockPlan pl;
    pl.H = (int)H;
    pl.W = (int)W;
    pl.M = (int)W / 2;
    pl.S = pl.M + 1;
    pl.nblk = (pl.S + kBlockColsMax - 1) / kBlockColsMax;
    pl.bw = (pl.S + pl.nblk - 1) / pl.nblk;
    best_split(pl.H, pl.hn1, pl.hn2);
    return pl;
}

template <bool NEED_T, typename Pair>
__device__ __forceinline__ void draw_block(SpectrumRng& g, int tid, int H, int ncd, Pair&& pair) {
    const int pairs = (H / 2) * ncd, dky = kBlockSlots / ncd, dc = kBlockSlots - dky * ncd;
    int ky = tid / ncd, c = tid - ky * ncd;
    for (int p = tid; p < pairs; p += kBlockSlots) {
        const uint32_t ra = g.R.next_high();
        const uint32_t rb = g.R.next_high();
        const uint32_t t = NEED_T ? g.T.next() : 0u;
        pair(ky, c, ra, rb, t);
        ky += dky;
        c += dc;
        if (c >= ncd) {
            c -= ncd;
            ++ky;
        }
    }
}

// MODE 0: ws[plane][ky][c0 .. c0 + ncd) = inverse column DFT of (drawn x filter); MODE 1: the drawn spectrum itself (unit filter, no transform)
template <int MODE>
__global__ void __launch_bounds__(kBlockSlots, 4) power_block_cols_kernel(const float* __restrict__ filter, c32* __restrict__ ws, int64_t planes,
                                                                          BlockPlan pl, uint64_t seed, uint64_t stream_id, int64_t plane_offset,
                                                                          int group, int split) {
    extern __shared__ __align__(16) unsigned char any_lds[];
    const int H = pl.H, S = pl.S, Sb = pl.bw | 1, tid = threadIdx.x;  // odd LDS row stride
    c32* const A = reinterpret_cast<c32*>(any_lds);
    c32* const tw = A + (size_t)H * Sb;
    if constexpr (MODE == 0) lines_table(tw, H, tid);
    const int64_t units = (split ? planes : planes / group) * pl.nblk;
    for (int64_t u = blockIdx.x; u < units; u += gridDim.x) {
        const int64_t unit = u / pl.nblk;
        const int d = (int)(u - unit * pl.nblk), c0 = d * pl.bw, ncd = min(pl.bw, S - c0);
        const GroupWalk gw(unit, group, split);
        SpectrumRng rng = spectrum_seed<true>(seed, stream_id, plane_offset / group + gw.grp, d * kBlockSlots + tid, false);
        for (int i = 0; i < gw.first; ++i) draw_block<true>(rng, tid, H, ncd, [](int, int, uint32_t, uint32_t, uint32_t) {});
        for (int gp = gw.first; gp < gw.first + gw.count; ++gp) {
            c32* const dst = ws + (gw.grp * group + gp) * (int64_t)H * S + c0;
            if constexpr (MODE == 1) {
                draw_block<true>(rng, tid, H, ncd, [&](int ky, int c, uint32_t ra, uint32_t rb, uint32_t t) {
                    dst[(int64_t)ky * S + c] = unit_complex_normal(ra, t & 0xFFFFu);
                    dst[(int64_t)(ky + H / 2) * S + c] = unit_complex_normal(rb, t >> 16);
                });
            } else {
                __syncthreads();  // the previous plane's block is stored (and the table visible)
                draw_block<true>(rng, tid, H, ncd, [&](int ky, int c, uint32_t ra, uint32_t rb, uint32_t t) {
                    const float* f = filter + (int64_t)ky * S + c0 + c;
                    A[ky * Sb + c] = drawn_elem(ra, t & 0xFFFFu, f[0]);
                    A[(ky + H / 2) * Sb + c] = drawn_elem(rb, t >> 16, f[(int64_t)(H / 2) * S]);
                });
                __syncthreads();
                line_dft<kBlockSlots, false>(A, tw, H, 1, pl.hn1, pl.hn2, ncd, Sb, 1, tid);
                for (LinesWalk lw(tid, ncd); lw.j < H * ncd; lw.next(ncd)) dst[(int64_t)lw.r * S + lw.c] = A[lw.r * Sb + lw.c];
            }
        }
    }
}

// Parseval statistics of the drawn, filtered spectrum (power_stats_kernel): sum x = sqrt(H W) Re Zf[0][0]; sum x^2 = the Hermitian parts of
// columns 0 and M + twice the interior columns' |Zf|^2 = f^2 (-ln u): the radius words alone.  A unit = (RNG group | plane, draw block).
template <bool NEED_T>
__device__ __forceinline__ void block_stats_unit(const float* __restrict__ filter, const BlockPlan& pl, uint64_t seed, uint64_t stream_id,
                                                 int64_t ggroup, const GroupWalk& gw, int d, c32* EDGE, double& s, double& q, int tid) {
    const int H = pl.H, S = pl.S, M = pl.M, c0 = d * pl.bw, ncd = min(pl.bw, S - c0);
    const int e0 = c0 == 0 ? 0 : -1, em = (M >= c0 && M < c0 + ncd) ? M - c0 : -1;  // the edge columns of this block, if any
    c32* const E0 = EDGE;
    c32* const EM = EDGE + H;
    SpectrumRng rng = spectrum_seed<NEED_T>(seed, stream_id, ggroup, d * kBlockSlots + tid, false);
    for (int i = 0; i < gw.first; ++i) draw_block<NEED_T>(rng, tid, H, ncd, [](int, int, uint32_t, uint32_t, uint32_t) {});
    for (int gp = gw.first; gp < gw.first + gw.count; ++gp) {
        float acc = 0.0f;
        draw_block<NEED_T>(rng, tid, H, ncd, [&](int ky, int c, uint32_t ra, uint32_t rb, uint32_t t) {
            const float* f = filter + (int64_t)ky * S + c0 + c;
            const float fa = f[0], fb = f[(int64_t)(H / 2) * S];
            if (NEED_T && (c == e0 || c == em)) {
                c32* const E = c == e0 ? E0 : EM;
                E[ky] = drawn_elem(ra, t & 0xFFFFu, fa);
                E[ky + H / 2] = drawn_elem(rb, t >> 16, fb);
            } else {
                acc = __builtin_fmaf(fa * fa, neg_ln_u(ra), acc);
                acc = __builtin_fmaf(fb * fb, neg_ln_u(rb), acc);
            }
        });
        q += 2.0 * (double)acc;
        if constexpr (NEED_T) {
            __syncthreads();
            float edge = 0.0f;
            for (int ky = tid; ky < H; ky += kBlockSlots) {
                const int kn = ky == 0 ? 0 : H - ky;
                if (e0 >= 0) {
                    const c32 a = E0[ky], an = E0[kn];
                    const float ar = 0.5f * (a.x + an.x), ai = 0.5f * (a.y - an.y);
                    edge += ar * ar + ai * ai;
                    if (ky == 0) s += (double)(sqrtf((float)H * (float)pl.W) * ar);
                }
                if (em >= 0) {
                    const c32 b = EM[ky], bn = EM[kn];
                    const float br = 0.5f * (b.x + bn.x), bi = 0.5f * (b.y - bn.y);
                    edge += br * br + bi * bi;
                }
            }
            q += (double)edge;
            __syncthreads();  // read before the next plane's columns overwrite them
        }
    }
}

__global__ void __launch_bounds__(kBlockSlots) power_block_stats_kernel(const float* __restrict__ filter, int64_t planes, BlockPlan pl, uint64_t seed,
                                                                        uint64_t stream_id, int64_t plane_offset, int group, int split,
                                                                        double* partials) {
    extern __shared__ __align__(16) unsigned char any_lds[];
    __shared__ double red[2 * kBlockSlots / 64];
    c32* const EDGE = reinterpret_cast<c32*>(any_lds);  // [column 0 | column M][ky]
    const int tid = threadIdx.x;
    double s = 0.0, q = 0.0;
    const int64_t units = (split ? planes : planes / group) * pl.nblk;
    for (int64_t u = blockIdx.x; u < units; u += gridDim.x) {
        const int64_t unit = u / pl.nblk;
        const int d = (int)(u - unit * pl.nblk);
        const GroupWalk gw(unit, group, split);
        const bool edge = d == 0 || d == pl.nblk - 1;  // uniform: column 0 in the first block, column M in the last
        if (edge) block_stats_unit<true>(filter, pl, seed, stream_id, plane_offset / group + gw.grp, gw, d, EDGE, s, q, tid);
        else block_stats_unit<false>(filter, pl, seed, stream_id, plane_offset / group + gw.grp, gw, d, EDGE, s, q, tid);
    }
    write_partial<kBlockSlots>(s, q, partials, red);
}

// mode 0: out = irfft2(drawn x filter, norm = "ortho") (+ statistics of out when `partials`); 1: the same, normalised (`partials` is the
// statistics workspace); 2: the drawn spectrum itself into `ws`
static int launch_power_block(int mode, const float* filter, float* ws, float* out, int64_t planes, int64_t H, int64_t W, uint64_t seed,
                              uint64_t stream_id, int64_t plane_offset, int group, double* partials, NormArgs na, hipStream_t st) {
    const BlockPlan pl = block_plan(H, W);
    const int Sb = pl.bw | 1;
    const size_t lds = ((size_t)pl.H * Sb + pl.H) * sizeof(c32);
    const int split = group > 1 && (planes / group) * pl.nblk < 2 * 256 ? 1 : 0;
    const int64_t units = (split ? planes : planes / group) * pl.nblk;
    const int per_cu = 2 * (lds + 1024) <= 160 * 1024 ? 2 : 1;
    c32* const wsc = reinterpret_cast<c32*>(ws);
    if (mode == 2) {
        hipLaunchKernelGGL(power_block_cols_kernel<1>, dim3((int)std::min<int64_t>(units, 2048)), dim3(kBlockSlots), 0, st, filter, wsc, planes, pl, seed,
                           stream_id, plane_offset, group, split);
        return check_launch("sonar_power_block_f32");
    }
    if (mode == 1)
        hipLaunchKernelGGL(power_block_stats_kernel, dim3((int)std::min<int64_t>(units, kNPart)), dim3(kBlockSlots), (size_t)2 * pl.H * sizeof(c32), st,
                           filter, planes, pl, seed, stream_id, plane_offset, group, split, partials);
    lines_lds_attr(power_block_cols_kernel<0>);
    hipLaunchKernelGGL(power_block_cols_kernel<0>, dim3((int)std::min<int64_t>(units, 256 * per_cu)), dim3(kBlockSlots), lds, st, filter, wsc, planes, pl,
                       seed, stream_id, plane_offset, group, split);
    const float scale = 1.0f / sqrtf((float)H * (float)W);
    if (!sonar_lines_rows_c2r_norm(ws, out, planes * H, W, scale, mode == 1 ? nullptr : partials, mode == 1 ? &na : nullptr, st)) {
        set_error("sonar_power_block_f32: no row pass for width %lld", (long long)W);
        return SONAR_ERR_UNSUPPORTED;
    }
    return check_launch("sonar_power_block_f32");
}

}  // namespace sonar
