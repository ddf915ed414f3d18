// Generated power-law noise on planes whose half-spectrum does not fit in LDS (256 x 256 = a 2048 px latent, 512 x 512, 384 x 512 ...):
// the spectrum is drawn on device in BLOCKS OF COLUMNS, a block is filtered and column-transformed in LDS and written once to a complex
// workspace; the row pass (lines_c2r_kernel, power_any.h) reads it once and writes the tensor once -- normalised, when the column kernel
// also summed the Parseval statistics of the filtered spectrum it had in hand:
//     draw + filter + statistics + columns (write W) -> rows c2r + normalise (read W, write out)                        ~ 3 x the tensor
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
    best_split(pl.H, pl.hn1, pl.hn2, kSetLines);
    return pl;
}

// Slot `tid` walks the pairs p = tid + it * 512 of a block: (ky, c) advance by constants.
struct BlockWalk {
    int ky, c, dky, dc, ncd;
    __device__ __forceinline__ BlockWalk(int tid, int ncd_) : ky(tid / ncd_), c(tid - (tid / ncd_) * ncd_), dky(kBlockSlots / ncd_), dc(kBlockSlots - (kBlockSlots / ncd_) * ncd_), ncd(ncd_) {}
    __device__ __forceinline__ void next() {
        ky += dky;
        c += dc;
        if (c >= ncd) {
            c -= ncd;
            ++ky;
        }
    }
};

// pre(it, ky, c) -> whatever `pair` wants to know about the pair beyond the random words (its filter values): asked for one iteration
// ahead, so a load behind it has the previous pair's arithmetic to arrive in (a slot meets the same pairs in every plane of a unit)
template <bool NEED_T, typename Pre, typename Pair>
__device__ __forceinline__ void draw_block(SpectrumRng& g, int tid, int H, int ncd, Pre&& pre, Pair&& pair) {
    const int pairs = (H / 2) * ncd;
    if (tid >= pairs) return;
    BlockWalk w(tid, ncd);
    auto cur = pre(0, w.ky, w.c);
    int it = 0;
    for (int p = tid; p < pairs; p += kBlockSlots, ++it) {
        const int ky = w.ky, c = w.c;
        w.next();
        auto nxt = cur;
        if (p + kBlockSlots < pairs) nxt = pre(it + 1, w.ky, w.c);
        const uint32_t ra = g.R.next();
        const uint32_t rb = g.R.next();
        const uint32_t t = NEED_T ? g.T.next() : 0u;
        pair(ky, c, ra, rb, t, cur);
        cur = nxt;
    }
}
template <bool NEED_T>
__device__ __forceinline__ void skip_block(SpectrumRng& g, int tid, int H, int ncd) {
    draw_block<NEED_T>(g, tid, H, ncd, [](int, int, int) { return 0; }, [](int, int, uint32_t, uint32_t, uint32_t, int) {});
}

// MODE 0: ws[plane][ky][c0 .. c0 + ncd) = inverse column DFT of (drawn x filter); MODE 1: the drawn spectrum itself (unit filter, no transform).
// STATS: the statistics of the tensor the row pass will make of the workspace, by Parseval on the filtered spectrum while it is in hand
// (ortho-normalised transforms): sum x = sqrt(H W) Re Zf[0][0]; sum x^2 = twice the interior columns' |Zf|^2 + the Hermitian parts of
// columns 0 and M (what the row pass keeps of them) -- two FMAs per drawn value, no pass over anything.
// CR1, CR2 > 0: the column length's factor pair at compile time (256 rows = 16 x 16)
template <int MODE, bool STATS = false, int CR1 = 0, int CR2 = 0>
__global__ void __launch_bounds__(kBlockSlots, 4) power_block_cols_kernel(const float* __restrict__ filter, c32* __restrict__ ws, int64_t planes,
                                                                          BlockPlan pl, uint64_t seed, uint64_t stream_id, int64_t plane_offset,
                                                                          int group, int split, double* partials) {
    kernarg_touch_for(filter, ws, planes, pl, seed, stream_id, plane_offset, group, split, partials);
    extern __shared__ __align__(16) unsigned char any_lds[];
    __shared__ double red[2 * kBlockSlots / 64];
    const int H = pl.H, S = pl.S, Sb = pl.bw | 1, tid = threadIdx.x;  // odd LDS row stride
    c32* const A = reinterpret_cast<c32*>(any_lds);
    c32* const tw = A + (size_t)H * Sb;
    if constexpr (MODE == 0) lines_table(tw, H, tid);
    [[maybe_unused]] double s = 0.0, q = 0.0;
    const int64_t units = (split ? planes : planes / group) * pl.nblk;
    for (int64_t u = blockIdx.x; u < units; u += gridDim.x) {
        const int64_t unit = u / pl.nblk;
        const int d = (int)(u - unit * pl.nblk), c0 = d * pl.bw, ncd = min(pl.bw, S - c0);
        const GroupWalk gw(unit, group, split);
        SpectrumRng rng = spectrum_seed<true>(seed, stream_id, plane_offset / group + gw.grp, d * kBlockSlots + tid, false);
        for (int i = 0; i < gw.first; ++i) skip_block<true>(rng, tid, H, ncd);
        for (int gp = gw.first; gp < gw.first + gw.count; ++gp) {
            c32* const dst = ws + (gw.grp * group + gp) * (int64_t)H * S + c0;
            if constexpr (MODE == 1) {
                draw_block<true>(rng, tid, H, ncd, [](int, int, int) { return 0; }, [&](int ky, int c, uint32_t ra, uint32_t rb, uint32_t t, int) {
                    dst[(int64_t)ky * S + c] = unit_complex_normal(ra, angle_lo(t));
                    dst[(int64_t)(ky + H / 2) * S + c] = unit_complex_normal(rb, angle_hi(t));
                });
            } else {
                __syncthreads();  // the previous plane's block is stored (and the table visible)
                const int e0 = c0 == 0 ? 0 : -1, em = (pl.M >= c0 && pl.M < c0 + ncd) ? pl.M - c0 : -1;  // the edge columns of this block, if any
                [[maybe_unused]] float acc = 0.0f;
                draw_block<true>(
                    rng, tid, H, ncd,
                    [&](int, int ky, int c) {
                        const float* f = filter + (int64_t)ky * S + c0 + c;
                        return make_float2(f[0], f[(int64_t)(H / 2) * S]);
                    },
                    [&](int ky, int c, uint32_t ra, uint32_t rb, uint32_t t, float2 f) {
                        const c32 za = drawn_elem(ra, angle_lo(t), f.x), zb = drawn_elem(rb, angle_hi(t), f.y);
                        A[ky * Sb + c] = za;
                        A[(ky + H / 2) * Sb + c] = zb;
                        if constexpr (STATS) {
                            if (c != e0 && c != em) acc = __builtin_fmaf(za.x, za.x, __builtin_fmaf(za.y, za.y, __builtin_fmaf(zb.x, zb.x, __builtin_fmaf(zb.y, zb.y, acc))));
                        }
                    });
                __syncthreads();
                if constexpr (STATS) {
                    q += 2.0 * (double)acc;
                    if (e0 >= 0 || em >= 0) {  // uniform
                        float edge = 0.0f;
                        for (int ky = tid; ky < H; ky += kBlockSlots) {
                            const int kn = ky == 0 ? 0 : H - ky;
                            if (e0 >= 0) {
                                const c32 a = A[ky * Sb + e0], an = A[kn * Sb + e0];
                                const float ar = 0.5f * (a.x + an.x), ai = 0.5f * (a.y - an.y);
                                edge += ar * ar + ai * ai;
                                if (ky == 0) s += (double)(sqrtf((float)H * (float)pl.W) * ar);
                            }
                            if (em >= 0) {
                                const c32 b = A[ky * Sb + em], bn = A[kn * Sb + em];
                                const float br = 0.5f * (b.x + bn.x), bi = 0.5f * (b.y - bn.y);
                                edge += br * br + bi * bi;
                            }
                        }
                        q += (double)edge;
                        __syncthreads();  // the transform's first pass writes in place
                    }
                }
                line_dft<kBlockSlots, false, CR1, CR2, kSetLines>(A, tw, H, 1, pl.hn1, pl.hn2, ncd, Sb, 1, tid);
                for (LinesWalk lw(tid, ncd); lw.j < H * ncd; lw.next(ncd)) dst[(int64_t)lw.r * S + lw.c] = A[lw.r * Sb + lw.c];
            }
        }
    }
    if constexpr (STATS) write_partial<kBlockSlots>(s, q, partials, red);
}

// mode 0: out = irfft2(drawn x filter, norm = "ortho") (+ statistics of out when `partials`); 1: the same, normalised (`partials` is the
// statistics workspace); 2: the drawn spectrum itself into `ws`
static int launch_power_block(int mode, const float* filter, float* ws, float* out, int64_t planes, int64_t H, int64_t W, uint64_t seed,
                              uint64_t stream_id, int64_t plane_offset, int group, double* partials, NormArgs na, hipStream_t st) {
    const BlockPlan pl = block_plan(H, W);
    const int Sb = pl.bw | 1;
    const size_t lds = ((size_t)pl.H * Sb + pl.H) * sizeof(c32);
    const int per_cu = 2 * (lds + 1024) <= 160 * 1024 ? 2 : 1;
    // Units of whole RNG groups (the seeding and nothing else is shared by a group's planes) or of single planes that fast-forward the
    // group's streams -- a work decomposition, not part of the stream definition.  Time in rounds of the resident workgroups (a last
    // round that leaves every CU one workgroup instead of two runs in about 0.55 of a round), a single-plane unit costs about 1.3 planes'
    // worth in groups of four (seeding and skipping).  Measured on 256 x 256: 128 latents 76 us as groups (640 units on 512 slots), 89 us
    // as planes; 32 latents 64 against 55; 32 latents of 512 x 512 (one workgroup per CU) 455 against 334.
    const int64_t slots = 256 * per_cu, u_group = (planes / group) * pl.nblk, u_plane = planes * pl.nblk;
    auto rounds = [&](int64_t u) {
        const int64_t tail = u % slots;
        return (double)(u / slots) + (tail == 0 ? 0.0 : (per_cu == 2 && tail <= slots / 2) ? 0.55 : 1.0);
    };
    const double t_group = rounds(u_group) * group, t_plane = rounds(u_plane) * (1.0 + 0.1 * (group - 1));
    const int split = group > 1 && t_plane < t_group ? 1 : 0;
    const int64_t units = split ? u_plane : u_group;
    c32* const wsc = reinterpret_cast<c32*>(ws);
    const dim3 blk(kBlockSlots), grid((int)std::min<int64_t>(units, slots));
    if (mode == 2) {
        hipLaunchKernelGGL(power_block_cols_kernel<1>, dim3((int)std::min<int64_t>(units, 2048)), blk, 0, st, filter, wsc, planes, pl, seed, stream_id,
                           plane_offset, group, split, (double*)nullptr);
        return check_launch("sonar_power_block_f32");
    }
    if (mode == 1 && pl.hn1 == 16 && pl.hn2 == 16) {
        lines_lds_attr(power_block_cols_kernel<0, true, 16, 16>);
        hipLaunchKernelGGL((power_block_cols_kernel<0, true, 16, 16>), grid, blk, lds, st, filter, wsc, planes, pl, seed, stream_id, plane_offset, group,
                           split, partials);
    } else if (mode == 1) {  // the statistics ride in the column kernel
        lines_lds_attr(power_block_cols_kernel<0, true>);
        hipLaunchKernelGGL((power_block_cols_kernel<0, true>), grid, blk, lds, st, filter, wsc, planes, pl, seed, stream_id, plane_offset, group, split,
                           partials);
    } else {
        lines_lds_attr(power_block_cols_kernel<0>);
        hipLaunchKernelGGL(power_block_cols_kernel<0>, grid, blk, lds, st, filter, wsc, planes, pl, seed, stream_id, plane_offset, group, split,
                           (double*)nullptr);
    }
    const float scale = 1.0f / sqrtf((float)H * (float)W);
    if (!sonar_lines_rows_c2r_norm(ws, out, planes * H, W, scale, mode == 1 ? nullptr : partials, mode == 1 ? &na : nullptr, st)) {
        set_error("sonar_power_block_f32: no row pass for width %lld", (long long)W);
        return SONAR_ERR_UNSUPPORTED;
    }
    return check_launch("sonar_power_block_f32");
}

}  // namespace sonar
