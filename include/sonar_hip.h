/*
 * sonar_hip.h — C ABI of libsonar_hip.so (MI355X / gfx950 only).
 *
 * This is the drop-in boundary for the procedural-noise + momentum-step + wavelet-split hot
 * path of blepping/ComfyUI-sonar (last row).  The reference is 100 % Python over
 * torch ops, so it has no FFI of its own; every entry point below replaces one torch-op
 * sequence of the reference, cited as `file:line` relative to the reference root.
 *
 * Conventions (all entry points):
 *   - returns 0 on success, SONAR_ERR_* (<0) on failure; sonar_last_error() gives the text
 *   - every buffer is a caller-owned DEVICE pointer (PyTorch-ROCm tensor.data_ptr()),
 *     contiguous NCHW, fp32 unless the name says f64; sizes are int64_t; scalars by value
 *   - asynchronous on `stream` (a hipStream_t passed as void*); no device sync, no allocation
 *   - workspace, when needed, is passed in by the caller (sonar_*_ws_bytes gives the size)
 *   - "partials" = array of `npart` (sum, sum-of-squares) fp64 pairs produced by a producing
 *     kernel and consumed by sonar_scale_noise_f32 (which reduces them in a fixed order on
 *     device, so normalisation needs no host sync and no atomics)
 */
#ifndef SONAR_HIP_H
#define SONAR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SONAR_OK 0
#define SONAR_ERR_ARG (-1)         /* null pointer / negative size / bad enum */
#define SONAR_ERR_UNSUPPORTED (-2) /* shape the kernel does not handle */
#define SONAR_ERR_HIP (-3)         /* HIP runtime error, see sonar_last_error() */

/* blend modes: py/utils.py:17-21 (BLENDING_MODES) */
#define SONAR_BLEND_LERP 0       /* torch.lerp(a, b, t) */
#define SONAR_BLEND_INJECT 1     /* a + b*t */
#define SONAR_BLEND_SUBTRACT_B 2 /* a - b*t */

/* momentum modes: py/sonar.py:40-43 */
#define SONAR_MODE_CLASSIC 0
#define SONAR_MODE_NEW 1
#define SONAR_MODE_DENOISED 2

/* history init applied inside the step when no history exists yet: py/sonar.py:169-206 */
#define SONAR_INIT_NONE 0        /* ZERO (history stays unset) or history already present */
#define SONAR_INIT_SAMPLE 1      /* history = x (denoised in DENOISED mode) */
#define SONAR_INIT_SAMPLE_NORM 2 /* history = x / sigma (denoised / sigma in DENOISED mode) */

/* number of (sum,sumsq) partial pairs every stats-producing kernel writes */
#define SONAR_NPART 1024

int sonar_abi_version(void);
const char* sonar_last_error(void);
/* Version of the GENERATE-mode value streams (device draws, cpu = False): the number changes whenever a seed gives other values than the
 * build before did -- saved seeds and workflows that use device draws are reproducible only within one version.  Replay mode (the
 * reference's host-generator draws handed in, cpu = True: the parity mode) is not versioned, its values are the reference's.
 *   3: round 3 (xoshiro128 bursts seeded by Philox4x32-10 at three depths)
 *   5: round 5 (MWC64X bursts; power-law spectra: filter weight under the radius' square root, 23 radius / 16 angle bits)
 *   6: round 6 (Brownian noise only: node bursts seeded by hashing a per-sub-tile Philox state with the node id, the power-law draw's
 *      conversions; every other generator's values are version 5's) */
int sonar_noise_stream_version(void);

/* ---------------------------------------------------------------- normalisation (row N) */
/* py/utils.py:100 — whole-tensor mean/std inputs: writes SONAR_NPART fp64 (sum,sumsq) pairs */
int sonar_stats_f32(const float* x, int64_t n, double* partials, void* stream);
/* reduce partials -> out3 = {sum, sumsq, n} (device fp64[3]); used for the optional cross-rank
 * all-reduce and by tests */
int sonar_stats_finalize(const double* partials, int64_t npart, int64_t n, double* out3, void* stream);
/* py/utils.py:93-106 — scale_noise: normalized=0 -> x*=factor; normalized=1 -> data-dependent
 * (x-mean)/std with thresholds threshold_std_devs/sqrt(n_total) evaluated ON DEVICE, then *factor.
 * partials/npart: from any producing kernel; n_total: element count the stats cover (may exceed
 * n when the stats were all-reduced over ranks).  In place. */
int sonar_scale_noise_f32(float* x, int64_t n, float factor, int normalized, float threshold_std_devs,
                          const double* partials, int64_t npart, int64_t n_total, void* stream);
/* sonar_scale_noise_f32(normalized = 1) that also writes the (sum, sumsq) of the RESULT into out_partials (1024 pairs; derived from the
 * input statistics and the decision, no extra pass), so a wrapper that normalises the same tensor again (chain -> scheduled -> chain,
 * py/noise.py:194,676,1405) needs no statistics sweep; when the decision is "leave as is" the kernel returns without touching x. */
int sonar_scale_noise_stats_f32(float* x, int64_t n, float factor, float threshold_std_devs, const double* partials,
                                int64_t npart, int64_t n_total, double* out_partials, void* stream);
/* py/utils.py:96-99 — normalize_dims variant: rows = groups, each of `inner` contiguous elements:
 * y = x/std_row; y -= mean_row(y); y *= factor.  (dims must be trailing & contiguous) */
int sonar_scale_noise_rows_f32(float* x, int64_t rows, int64_t inner, float factor, void* stream);

/* ---------------------------------------------------------------- elementwise (rows U, C, S, L) */
/* out = blend(a, b, t) — py/utils.py:17-21; out may alias a or b */
int sonar_blend_f32(int mode, const float* a, const float* b, float t, float* out, int64_t n, void* stream);
/* out = blend(a, b, t[i % tn]) with per-element/broadcast weights (BlendedNoise mask, py/noise.py:1364-1379) */
int sonar_blend_tensor_f32(int mode, const float* a, const float* b, const float* t, int64_t tn, float* out,
                           int64_t n, void* stream);
/* y = y*ymul + x*xmul  (chain accumulate py/noise.py:192; ancestral noise add py/sonar.py:565) */
int sonar_axpby_f32(float* y, float ymul, const float* x, float xmul, int64_t n, void* stream);
/* the same, and the (sum, sumsq) partials of the result (1024 fp64 pairs): the last accumulation of a chain feeds its scale_noise */
int sonar_axpby_stats_f32(float* y, float ymul, const float* x, float xmul, int64_t n, double* partials, void* stream);
/* ModulatedNoise spectral_signum (py/noise.py:938-1015): fftn over the modulation dims -> log amplitude -> per-sample quantiles of its
 * absolute value (sonar_abs_quantile_rows_f32 on `full`) -> soft clamp of the bins outside the 5 % / 95 % quantiles -> inverse.
 *  sonar_rfft2_f32                 z_out[planes][H][W/2+1] (complex64) = rfft2(x) unscaled, LDS-resident (power-of-two planes)
 *  sonar_cdft_mid_f32              DFT along the middle axis of complex z[outer][C][inner] (the channel axis), out of place; real_in:
 *                                  float input; real_out: real part only, float output; inverse without the 1/C
 *  sonar_spectral_logamp_f32       la = log(sqrt(re^2 + im^2)); full[planes][H][W] = |la| over the whole spectrum (dropped rfft2 columns
 *                                  from their Hermitian partners; C = channels when a channel DFT was applied, else 1)
 *  sonar_spectral_signum_mask_f32  z *= gain * (mult_low * mult_high)^intensity, q[nq][3] = (low, high, max) quantile rows: row 0 when
 *                                  nq == 1, the bin's CHANNEL index when nq == C (how the reference's expand() lines the vector up);
 *                                  channel_sym: z is a HALF spectrum under a channel DFT -> the mask is averaged with its Hermitian
 *                                  partner's (row (C - c) % C), which is what the real part of the reference's complex inverse keeps */
int sonar_rfft2_f32(const float* x, float* z_out, int64_t planes, int64_t H, int64_t W, void* stream);
int sonar_cdft_mid_f32(const float* z_in, float* z_out, int64_t outer, int64_t C, int64_t inner, int inverse, int real_in,
                       int real_out, void* stream);
int sonar_spectral_logamp_f32(const float* z, float* la, float* full, int64_t planes, int64_t C, int64_t H, int64_t W, int64_t Wz,
                              void* stream);
int sonar_spectral_signum_mask_f32(float* z, const float* la, const float* q, int64_t nq, int64_t planes, int64_t C,
                                   int64_t plane_elems, float intensity, float gain, int channel_sym, void* stream);
/* CompositeNoise py/noise.py:524-531: out = dst*(1-mask) + src*mask; mask is [mask_n] broadcast over n/mask_n */
int sonar_mask_mix_f32(const float* dst, const float* src, const float* mask, int64_t mask_n, float* out,
                       int64_t n, void* stream);
/* x = (x - sub)*mul + add in place (UniformNoiseGenerator, py/noise_generation.py:508-514) */
int sonar_affine_f32(float* x, float sub, float mul, float add, int64_t n, void* stream);
/* out = a*s (op 0) | a/s (op 1, true division) | (a-b)/s (op 2: k-diffusion to_d, py/sonar.py:300) */
int sonar_scalar_op_f32(int op, const float* a, const float* b, float s, float* out, int64_t n, void* stream);
/* per-row mean and unbiased std over `inner` contiguous elements (guidance_shift / prepare_ref_latent,
 * py/sonar.py:336-341,372-377) */
int sonar_rowstats_f32(const float* x, int64_t rows, int64_t inner, float* mean, float* stdv, void* stream);
/* op 0: out = (x - a[row]) / b[row]   op 1: out = x * b[row] + a[row]   (same call sites) */
int sonar_row_affine_f32(int op, const float* x, int64_t rows, int64_t inner, const float* a, const float* b,
                         float* out, void* stream);
/* StudentTNoiseGenerator.generate, py/noise_generation.py:667-677, in three steps:
 *  sonar_studentt_f32            x (a standard-normal draw) -> loc + scale * x * rsqrt(max(gamma/0.5, tiny) / df), gamma = the
 *                                torch._standard_gamma(df/2) draw of torch.distributions.StudentT.rsample (Chi2 = Gamma(df/2, 1/2)); in place
 *  sonar_abs_quantile_rows_f32   out[row] = torch.quantile(|x[row]|, q) with linear interpolation; the caller passes the fp32 rank
 *                                q*(inner-1) split into rank_lo + rank_frac (radix select on bit patterns, one workgroup per row)
 *  sonar_clamp_signpow_rows_f32  x = copysign(|clamp(x, -lim, lim)|^p, x), lim = limit[row] * mul; in place */
/* acc = (first ? 0 : acc) + mul*z*z -- the chi-square of an integer df as a sum of squared normals (on-device StudentT draws) */
int sonar_sq_acc_f32(float* acc, const float* z, float mul, int first, int64_t n, void* stream);
int sonar_studentt_f32(float* x, const float* gamma, float loc, float scale, float df, int64_t n, void* stream);
int sonar_abs_quantile_rows_f32(const float* x, int64_t rows, int64_t inner, int64_t rank_lo, float rank_frac, float* out,
                                void* stream);
int sonar_clamp_signpow_rows_f32(float* x, int64_t rows, int64_t inner, const float* limit, float mul, float p, void* stream);
/* RippleFilteredNoise, py/noise.py:1197-1200: x[i] *= table[(i / inner) % len] (a sin / cos gain profile along one dimension, or along
 * the flattened trailing dimensions with inner = 1); follow_sign: the result takes the sign of 1 - table[..] (torch.copysign). In place. */
int sonar_mul_table_f32(float* x, const float* table, int64_t n, int64_t inner, int64_t len, int follow_sign, void* stream);
/* LaplacianNoiseGenerator.generate, py/noise_generation.py:796-802: x = x/div_fac + Laplace(loc, scale), the variate built from a
 * uniform u in (eps-1, 1) the way torch.distributions.Laplace.rsample does: loc - scale*sign(u)*log1p(-max(|u|, tiny)).  In place. */
int sonar_laplace_add_f32(float* x, const float* u, float div_fac, float loc, float scale, int64_t n, void* stream);
/* ModulatedNoise (py/noise.py:784-866), three steps that never read a scalar back to the host:
 *  sonar_std_mid_f32   unbiased std over the middle axis of x[outer][mid][inner] -> stdv[outer][inner]
 *                      (torch.std(dim=-3, keepdim=True), :795-799; the other two modulation_dims use sonar_rowstats_f32)
 *  sonar_bcast_gain_f32  v = x*k*(1/(std*|strength|+1) + 1) (:800-803), std broadcast by `bcast`: 0 = stdv[outer],
 *                      1 = stdv[outer][mid], 2 = stdv[outer][inner]; `out` and/or `partials` may be NULL; each partial slot pair
 *                      holds (sum x^2, sum v^2) of its block (1024 pairs, same layout as the statistics kernels)
 *  sonar_ratio_mix_f32   out = a*(a_mul*rho) + x*x_mul, rho = sqrt(num_mul * sum num[2i] / sum den[2i+1]): the L2-norm ratio
 *                      "noise_norm / scaled_noise_norm" (:805-810, :860-866) taken from partial slots */
int sonar_std_mid_f32(const float* x, int64_t outer, int64_t mid, int64_t inner, float* stdv, void* stream);
int sonar_bcast_gain_f32(const float* x, const float* stdv, int64_t outer, int64_t mid, int64_t inner, int bcast,
                         float abs_strength, float k, float* out, double* partials, void* stream);
int sonar_ratio_mix_f32(const float* a, float a_mul, const float* x, float x_mul, const double* num_partials,
                        double num_mul, const double* den_partials, float* out, int64_t n, void* stream);
/* PowerLawNoiseGenerator, py/noise_generation.py:775-779: x = (use_sign ? sign(x) : x) * |x|^alpha, in place */
int sonar_powerlaw_f32(float* x, float alpha, int use_sign, int64_t n, void* stream);
/* x viewed as [outer, mid, inner]: amax over `mid` of x (use_abs 0) or |x| (use_abs 1) -> peak[outer, inner]
 * (torch.amax over one non-trailing dim, py/noise_generation.py:780-785) */
int sonar_amax_mid_f32(const float* x, int64_t outer, int64_t mid, int64_t inner, int use_abs, float* peak, void* stream);
/* x[o, m, i] /= d[o, i], in place, true division (same call site) */
int sonar_div_mid_f32(float* x, int64_t outer, int64_t mid, int64_t inner, const float* d, void* stream);
/* py/utils.py:452-470 normalize_to_scale: per row min/max rescale to [lo,hi] */
int sonar_minmax_rows_f32(const float* x, int64_t rows, int64_t inner, float* out_min, float* out_max, void* stream);
/* the rest of normalize_to_scale (py/utils.py:462-469): out = clamp(((x - lo[r]) / ((hi[r] - lo[r]) + eps)) * (target_max - target_min)
 * + target_min, target_min, target_max), one (lo, hi) per row, every step rounded separately as the reference's tensor ops are; the
 * targets arrive as doubles (Python floats in the reference): their difference is rounded to fp32 once */
int sonar_minmax_rescale_f32(const float* x, int64_t rows, int64_t inner, const float* lo, const float* hi, float eps,
                             double target_min, double target_max, float* out, void* stream);

/* A pending global normalisation of a noise tensor (py/utils.py:100-105): the decision scale_noise(normalized=True) would take,
 * computed ON THE DEVICE from the tensor's (sum, sumsq) partials and left in device memory, so that the kernel that consumes the
 * noise (the sampler-step kernels below take it as `noise_norm`, nullable) applies `((v - mean) / std) * factor` -- only the parts
 * the thresholds ask for, in scale_noise's own order: same bits -- while it reads the noise, instead of a separate read + write of
 * the tensor.  sonar_apply_norm_f32 materialises it in place for any other consumer. */
typedef struct sonar_noise_norm {
    float mean, stdv, inv_std /* correctly rounded 1 / stdv */, factor;
    int32_t do_sub, do_div;
} sonar_noise_norm;
int sonar_norm_decision_f32(const double* partials, int64_t npart, int64_t n_total, float factor, float threshold_std_devs,
                            sonar_noise_norm* out /* device */, void* stream);
int sonar_apply_norm_f32(float* x, int64_t n, const sonar_noise_norm* norm /* device */, void* stream);

/* ---------------------------------------------------------------- momentum step (rows M, M2) */
typedef struct sonar_momentum_cfg {
    float momentum;      /* SonarConfig.momentum             py/sonar.py:47 */
    float hist_ratio;    /* history_ratios[0] = momentum_hist py/sonar.py:208-219 */
    float hist_scale;    /* history_ratios[1] */
    float md_scale;      /* history_ratios[2] = direction */
    int32_t mode;        /* SONAR_MODE_* */
    int32_t momentum_blend; /* SONAR_BLEND_* for momentum_mix  py/sonar.py:256-260 */
    int32_t history_blend;  /* SONAR_BLEND_* for update_hist   py/sonar.py:231-236 */
    int32_t use_momentum;   /* check_step(step)                py/sonar.py:221-225 */
    int32_t update_hist;    /* momentum_hist != 1 && check_step(step, is_history=True) */
    int32_t init_kind;      /* SONAR_INIT_* (applies only when h_in == NULL) */
    int32_t h_in_fresh;     /* 1: h_in was created by RAND init in this very step -> not used by the
                               denoised-mix of this step (py/sonar.py:273-283 reads hd before init) */
    int32_t reserved;
} sonar_momentum_cfg;

/* One Sonar Euler step, fused (py/sonar.py:262-320): x_out = momentum_d*(sigma_down-sigma) + x and
 * the twice-updated history.  h_in may be NULL (no history yet); h_out must be valid whenever the
 * step can create/update history; *h_out_present (host int, may be NULL) tells whether h_out holds
 * a history afterwards.  noise (nullable) adds noise*noise_scale (ancestral, py/sonar.py:563-566).
 * x_out may alias x; h_out may alias h_in. */
int sonar_momentum_euler_f32(const float* x, const float* denoised, const float* h_in, float* x_out,
                             float* h_out, const float* noise, float noise_scale, float sigma, float dt,
                             const sonar_momentum_cfg* cfg, int64_t n, int* h_out_present, const sonar_noise_norm* noise_norm,
                             void* stream);

/* DPM-Solver++(SDE) half steps, py/sonar.py:649-735.  Stage 1:
 *   md1 = MD(den, sigma); m_d = D(expm1_a * md1); x2 = ratio_a*x - m_d + noise*noise_scale
 * Stage 2 (den2 = model(x2, sigma_s)):
 *   md2 = MD(den2, sigma_s); dd = (1-fac)*md1 + fac*md2; m_d = D(expm1_b * dd);
 *   x_out = ratio_b*x - m_d (+ noise*noise_scale when noise != NULL)
 * `adj_is_one` reproduces the `adjusted_momentum == 1` early-out of get_momentum_d while the
 * blend weight stays cfg->momentum (py/sonar.py:298-303). */
int sonar_dpmpp_stage1_f32(const float* x, const float* denoised, const float* h_in, float* x2_out, float* md1_out,
                           float* h_out, const float* noise, float noise_scale, float sigma, float expm1_a,
                           float ratio_a, int adj_is_one, const sonar_momentum_cfg* cfg, int64_t n,
                           int* h_out_present, const sonar_noise_norm* noise_norm, void* stream);
int sonar_dpmpp_stage2_f32(const float* x, const float* denoised2, const float* md1, const float* h_in,
                           float* x_out, float* dd_out, float* h_out, const float* noise, float noise_scale,
                           float sigma_s, float expm1_b, float ratio_b, float fac, int adj_is_one,
                           const sonar_momentum_cfg* cfg, int64_t n, int* h_out_present, const sonar_noise_norm* noise_norm,
                           void* stream);

/* ---------------------------------------------------------------- base generators (rows G1, G2) */
/* On-device counter RNG.  Philox4x32-10 (key = seed) seeds one xoshiro128++ burst per (stream id, tile, lane);
 * a flat buffer is drawn in tiles of 4096 elements, so the value of element e depends only on
 * (seed, stream_id, elem_offset + e) -> independent of how a batch is sharded over GPUs. */
int sonar_philox_normal_f32(float* out, int64_t n, uint64_t seed, uint64_t stream_id, int64_t elem_offset,
                            double* partials /*nullable*/, void* stream);
/* out = (u - sub)*mul + add, u~U[0,1)  (py/noise_generation.py:508-514) */
int sonar_philox_uniform_f32(float* out, int64_t n, uint64_t seed, uint64_t stream_id, int64_t elem_offset,
                             float sub, float mul, float add, double* partials /*nullable*/, void* stream);
/* The same draws followed by scale_noise(factor, normalized = 1) with the tensor written ONCE: a statistics pass re-draws the
 * values without storing them, the final pass re-draws, normalises and stores (uniform = 0: N(0,1); 1: (U[0,1) - sub)*mul + add).
 * N(0,1) with factor 1 passes both thresholds 98.7 % of the time: that case stores the draws with their statistics in one pass and
 * lets sonar_scale_noise_f32 decide on the device (it returns without touching the tensor when there is nothing to do).
 * partials: 1024 fp64 pairs of workspace. */
int sonar_philox_noise_f32(int uniform, float* out, int64_t n, uint64_t seed, uint64_t stream_id, int64_t elem_offset,
                           float sub, float mul, float add, float factor, float threshold_std_devs, double* partials,
                           void* stream);
/* The same call in a sampler's steady state, for prepared plans (round 6): ONE launch runs this call's final pass (statistics in
 * `partials`: left there by the previous call's launch when `have_stats`, else computed first by the ordinary statistics pass) and the
 * statistics pass of the call that will draw with `next_stream_id`, into `partials_next` (another 1024 fp64 pairs) -- the final pass is
 * store-bound, the statistics pass pure vector-ALU work, one wave does both for its tiles.  Output bits and `partials_next` are those of
 * sonar_philox_noise_f32 (N(0,1) with factor 1 included: with the decision known before the first store the raw draws go out as they
 * are 98.7 % of the time, and otherwise with scale_noise's own subtract / divide sequence -- no second launch either way).
 * sonar_philox_noise_ahead_ok(): whether the form pays for the shape -- n > 0, and for N(0,1) with factor 1 (whose ordinary route draws
 * every value once) only up to 8 Mi elements, where the call is launch-bound; the entry point itself takes every shape. */
int sonar_philox_noise_ahead_ok(int uniform, int64_t n, float factor);
int sonar_philox_noise_ahead_f32(int uniform, float* out, int64_t n, uint64_t seed, uint64_t stream_id, int64_t elem_offset,
                                 float sub, float mul, float add, float factor, float threshold_std_devs, double* partials,
                                 int have_stats, uint64_t next_stream_id, double* partials_next, void* stream);

/* Brownian-interval noise (the reference wraps ComfyUI's BrownianTreeNoiseSampler -> torchsde, un-vendored:
 * py/noise_generation.py:262-286, py/nodes/powernoise.py:383-393).  out[e] = sum_k coefs[k] * z(node_ids[k], e) with
 * z a counter-based N(0,1) keyed by (seed, node id (48 bits), global element index elem_offset + e).  node_ids / coefs are
 * HOST arrays (<= 96 entries): the expansion of the queried increment / point over the node normals, kept by the host in fp64
 * (each queried time is a Brownian bridge between the nearest times known before it).  latent_elems =
 * elements per latent (0 if unknown): when it is a multiple of 4096 and there is one seed, z(node, .) is the tile-keyed burst
 * stream of the Gaussian fill with stream id = node (one Philox seeding per 64 values); otherwise one Philox4x32-10 call per
 * 4 values.  latent_seeds (device, nullable): one seed per latent (the sampler's batched-seed mode), replacing `seed`. */
int sonar_brownian_f32(float* out, int64_t n, int64_t elem_offset, const uint64_t* node_ids, const float* coefs, int nnodes,
                       uint64_t seed, const uint64_t* latent_seeds, int64_t latent_elems, void* stream);
/* One path point instead of an increment: W = sum_k coefs[k] z(node_ids[k], e) with the coefficients of W(t) itself;
 * w_out = W (nullable) and out = scale * (W - prev) (out, prev nullable; prev = a W(t') kept from an earlier call). */
int sonar_brownian_point_f32(float* out, float* w_out, const float* prev, float scale, int64_t n, int64_t elem_offset,
                             const uint64_t* node_ids, const float* coefs, int nnodes, uint64_t seed,
                             const uint64_t* latent_seeds, int64_t latent_elems, void* stream);
/* The bridge form of a path point: W = fa * base_a + fb * base_b + sum_k coefs[k] z(node_ids[k], e), base_a / base_b (nullable) =
 * the cached W tensors of the two times the new one was bridged between -- ONE fresh normal per element instead of the point's
 * whole expansion.  Also accumulates an expansion longer than 96 terms in chunks (base_a = the partial sum, fa = 1). */
int sonar_brownian_bridge_f32(float* out, float* w_out, const float* prev, float scale, const float* base_a, float fa,
                              const float* base_b, float fb, int64_t n, int64_t elem_offset, const uint64_t* node_ids,
                              const float* coefs, int nnodes, uint64_t seed, const uint64_t* latent_seeds, int64_t latent_elems,
                              double* partials /* nullable: (sum, sumsq) of `out`, SONAR_NPART pairs */, void* stream);

/* Accumulating forms of the generators (noise chains: result = sum_i item_i * factor_i, py/noise.py:188-194).  Instead of writing
 * a fresh tensor that sonar_axpby_f32 then folds into the running sum (read 8N + write 4N more), the generator reads the sum and
 * writes it back: y <- y * y_mul + x * x_mul with x the values the plain form would have stored (products rounded separately and
 * skipped for a multiplier of exactly 1: bit-identical to plain form + sonar_axpby_f32).  partials (nullable, SONAR_NPART pairs)
 * receives (sum, sumsq) of the new y, which is what a following scale_noise needs. */
typedef struct sonar_accumulate {
    float* y;         /* running sum, updated in place */
    float y_mul;
    float x_mul;
    double* partials; /* nullable */
} sonar_accumulate;
int sonar_philox_normal_acc_f32(const sonar_accumulate* acc, int64_t n, uint64_t seed, uint64_t stream_id, int64_t elem_offset,
                                void* stream);
int sonar_perlin_generate_acc_f32(const sonar_accumulate* acc, const float* terms, int64_t B, int64_t chw, int64_t iters, float div_fac,
                                  uint64_t seed, uint64_t stream_id, int64_t elem_offset, void* stream);
int sonar_brownian_bridge_acc_f32(const sonar_accumulate* acc, float* w_out, const float* prev, float scale, const float* base_a, float fa,
                                  const float* base_b, float fb, int64_t n, int64_t elem_offset, const uint64_t* node_ids,
                                  const float* coefs, int nnodes, uint64_t seed, const uint64_t* latent_seeds, int64_t latent_elems,
                                  void* stream);
/* Two chain items in ONE pass over the running sum (py/noise.py:188-194 with a Gaussian or Perlin item followed by a Brownian one):
 * `pre` describes the fold the PREVIOUS item would have made with its own entry point (kind NORMAL: sonar_philox_normal_acc_f32 with
 * that seed / stream_id; kind PERLIN: sonar_perlin_generate_acc_f32 with iters == 1, that summed lattice, chw == latent_elems); the
 * Brownian kernel evaluates it per element, applies y1 = y * pre.y_mul + x * pre.x_mul and then its own fold on y1 -- the same
 * operations as the two launches, so the same bits, with one read + write of the sum instead of two.  Requires the tile route of the
 * Brownian kernel (one seed, latents of whole 4096-element tiles, 16-byte aligned tensors; the two items share elem_offset):
 * SONAR_ERR_UNSUPPORTED otherwise, and the caller applies `pre` with its own entry point first. */
#define SONAR_PREFIX_NORMAL 1
#define SONAR_PREFIX_PERLIN 2
typedef struct sonar_fold_prefix {
    int32_t kind;
    float y_mul;      /* y1 = y * y_mul + x * x_mul */
    float x_mul;
    float div_fac;    /* PERLIN */
    uint64_t seed;
    uint64_t stream_id;
    const float* terms; /* PERLIN: the summed lattice [chw] */
    int64_t chw;        /* PERLIN */
    int32_t fresh;      /* 1: the item is the chain's FIRST -- the sum holds nothing yet, y1 = x (y is not read; x_mul must be 1 and the
                           hosting fold's y_mul carries the item's factor, as sonar_axpby_f32 would) */
} sonar_fold_prefix;
int sonar_brownian_bridge_chain_f32(const sonar_accumulate* acc, const sonar_fold_prefix* pre, float* w_out, const float* prev, float scale,
                                    const float* base_a, float fa, const float* base_b, float fb, int64_t n, int64_t elem_offset,
                                    const uint64_t* node_ids, const float* coefs, int nnodes, uint64_t seed, int64_t latent_elems,
                                    void* stream);
/* sonar_philox_normal_acc_f32 / sonar_perlin_generate_acc_f32 (iters == 1: `terms` is the summed lattice) with `pre` riding along.
 * Whole 4-element groups only (n, elem_offset, chw multiples of 4; 16-byte aligned tensors): SONAR_ERR_UNSUPPORTED otherwise. */
int sonar_philox_normal_chain_f32(const sonar_accumulate* acc, const sonar_fold_prefix* pre, int64_t n, uint64_t seed, uint64_t stream_id,
                                  int64_t elem_offset, void* stream);
int sonar_perlin_generate_chain_f32(const sonar_accumulate* acc, const sonar_fold_prefix* pre, const float* terms, int64_t B, int64_t chw,
                                    float div_fac, uint64_t seed, uint64_t stream_id, int64_t elem_offset, void* stream);
/* Pyramid noise (sonar_pyramid_generate_f32's values, levels drawn in the kernel or passed in) folded into the running sum, `pre`
 * (nullable) as above: the plane kernel's generator shares the tile keying of the Gaussian / Perlin generators, so any whole-plane
 * shape it can run can host them.  SONAR_ERR_UNSUPPORTED when the plane kernel cannot run the shape (nothing was done). */
int sonar_pyramid_generate_acc_f32(const sonar_accumulate* acc, const sonar_fold_prefix* pre, int64_t planes, int64_t H, int64_t W,
                                   int64_t nlevels, const float* const* level_ptrs, const int64_t* level_h, const int64_t* level_w,
                                   const float* level_weight, int mode, uint64_t seed, uint64_t stream_id, int64_t elem_offset,
                                   void* stream);
/* sonar_pyramid_generate_acc_f32 hosting a Perlin item (`pre`, kind PERLIN) for a chain that is called step after step with stream ids it
 * can foresee (the prepared plans): extra leading workgroups of the same launch compute the summed lattice [lattice_channels][H][W] of a
 * LATER call's Perlin item into `lattice_out` (as sonar_perlin_lattice_f32 with `seed`, `lattice_stream_id` would) -- independent of the
 * rest of the launch, and one launch less on that later call's critical path. */
int sonar_pyramid_generate_acc_ahead_f32(const sonar_accumulate* acc, const sonar_fold_prefix* pre, int64_t planes, int64_t H, int64_t W,
                                         int64_t nlevels, const float* const* level_ptrs, const int64_t* level_h, const int64_t* level_w,
                                         const float* level_weight, int mode, uint64_t seed, uint64_t stream_id, int64_t elem_offset,
                                         float* lattice_out, int64_t lattice_iters, int64_t lattice_channels, int blend_mode,
                                         uint64_t lattice_stream_id, void* stream);

/* ---------------------------------------------------------------- Perlin (row P) */
/* py/noise_generation.py:465-476,388-405 at the only position generate() uses (grid == output,
 * block 1x1, pos (0.5,0.5)): angles[iters][C][H+1][W+1] -> terms[iters][C][H][W] evaluating the
 * reference's blend tree.  blend_mode: SONAR_BLEND_* */
int sonar_perlin_terms_f32(const float* angles, float* terms, int64_t iters, int64_t C, int64_t H, int64_t W,
                           int blend_mode, void* stream);
/* generate mode: the same terms with the lattice angles drawn where they are used (counter-based, keyed by seed / stream_id /
 * lattice point / iteration; shared by every latent and every rank), summed over the iterations: terms_sum[C][H][W] */
int sonar_perlin_lattice_f32(float* terms_sum, int64_t iters, int64_t C, int64_t H, int64_t W, int blend_mode, uint64_t seed,
                             uint64_t stream_id, void* stream);
/* replay mode, py/noise_generation.py:478-493: out = base/div_fac (+ terms[i] broadcast over B, in order) */
int sonar_perlin_apply_f32(const float* base, const float* terms, float* out, int64_t B, int64_t chw,
                           int64_t iters, float div_fac, double* partials /*nullable*/, void* stream);
/* generate mode: base u~U[0,1) from Philox(seed, stream_id, elem_offset+e) fused with the above */
int sonar_perlin_generate_f32(const float* terms, float* out, int64_t B, int64_t chw, int64_t iters, float div_fac,
                              uint64_t seed, uint64_t stream_id, int64_t elem_offset, double* partials /*nullable*/,
                              void* stream);

/* generate mode + scale_noise(factor, normalized=True) without a second sweep over the tensor (SURVEY.md §8d
 * "stats-before-write"): pass 1 re-draws the values and only reduces (sum, sumsq) -> partials; pass 2 re-draws,
 * applies the on-device normalisation decision and writes the final tensor once. */
int sonar_perlin_noise_f32(const float* terms, float* out, int64_t B, int64_t chw, int64_t iters, float div_fac,
                           uint64_t seed, uint64_t stream_id, int64_t elem_offset, float factor, float threshold_std_devs,
                           double* partials /*workspace, SONAR_NPART pairs*/, void* stream);
/* sonar_perlin_lattice_f32 + sonar_perlin_noise_f32 for a sampler that is called step after step with stream ids it can foresee (the
 * prepared plans, below): ONE launch per call in the steady state.  The launch runs this call's final pass (`terms`: its summed lattice,
 * `partials` + have_stats: its statistics, both left by the previous launches; have_stats == 0: the statistics pass is launched first),
 * then the statistics pass of the NEXT call (tile stream `next_stream_id` against `terms_next`, nullable) into `partials_next`, and in
 * extra blocks the lattice of a LATER call (`lattice_out`, nullable; `lattice_iters`, C, H, W, `blend_mode`, `lattice_stream_id` as
 * sonar_perlin_lattice_f32 takes them) -- nothing inside the launch depends on anything else inside it.  The output bits are those of the
 * two entry points it replaces.  Only where sonar_perlin_noise_ahead_ok() says 1 (whole 4096-element tiles per latent, at most 4096
 * tiles: the launch-bound sizes); SONAR_ERR_UNSUPPORTED otherwise. */
int sonar_perlin_noise_ahead_ok(int64_t B, int64_t chw, int64_t elem_offset);
int sonar_perlin_noise_ahead_f32(const float* terms, float* out, int64_t B, int64_t chw, float div_fac, uint64_t seed, uint64_t stream_id,
                                 int64_t elem_offset, float factor, float threshold_std_devs, double* partials, int have_stats,
                                 uint64_t next_stream_id, const float* terms_next, double* partials_next, float* lattice_out,
                                 int64_t lattice_iters, int64_t C, int64_t H, int64_t W, int blend_mode, uint64_t lattice_stream_id,
                                 void* stream);

/* ---------------------------------------------------------------- Pyramid (row Y) */
/* dst[B*C][H][W] += bilinear_upsample(src[B*C][h][w]) * scale   (F.interpolate(mode="bilinear",
 * align_corners=False), py/utils.py:58-67 <- py/noise_generation.py:629-646).  mode: 0 bilinear,
 * 1 nearest-exact, 2 area (adaptive average, used when downscaling), 3 nearest (legacy floor(dst * in / out)), 4 bicubic
 * (A = -0.75, clamped taps), 5 bicubic with align_corners=True (py/noise.py:583-588, GuidedNoise's reference latent),
 * 6 bilinear with align_corners=True (py/nodes/powernoise.py:133, the filter's gain curve) */
int sonar_resample_acc_f32(float* dst, const float* src, int64_t planes, int64_t H, int64_t W, int64_t h, int64_t w,
                           float scale, int mode, int accumulate, double* partials /*nullable*/, void* stream);
/* generate mode: out = N(0,1) * base_scale + sum_l upsample(levels[l]) * weights[l].  level_ptrs[l]: device pointer to a
 * [planes][h_l][w_l] grid, or NULL = drawn on device: a NULL level of the latent's own size is folded into the base draw
 * (sum of two independent normals: base_scale = sqrt(1 + w^2)); a smaller NULL level is drawn by the plane kernel (stream
 * id stream_id + 2 + l, keyed by global plane) -- SONAR_ERR_UNSUPPORTED if that kernel cannot run this shape. */
int sonar_pyramid_generate_f32(float* out, int64_t planes, int64_t H, int64_t W, int64_t nlevels,
                               const float* const* level_ptrs /*host array of device ptrs*/, const int64_t* level_h,
                               const int64_t* level_w, const float* level_weight, int mode, uint64_t seed,
                               uint64_t stream_id, int64_t elem_offset, double* partials /*nullable*/, void* stream);

/* same, with the normaliser folded in (see sonar_perlin_noise_f32) */
int sonar_pyramid_noise_f32(float* out, int64_t planes, int64_t H, int64_t W, int64_t nlevels,
                            const float* const* level_ptrs, const int64_t* level_h, const int64_t* level_w,
                            const float* level_weight, int mode, uint64_t seed, uint64_t stream_id, int64_t elem_offset,
                            float factor, float threshold_std_devs, double* partials, void* stream);
/* The same call (every level drawn in the kernel, bilinear: the generate-mode call of PyramidNoiseGenerator) in a sampler's steady state,
 * for prepared plans (round 6): ONE launch.  Half of its workgroups run this call's planes and store them normalised -- with the
 * statistics in `partials` (left there by the previous call's launch when `have_stats`; else a launch of this call's planes without
 * stores computes them first) and scale_noise's own operation sequence: the output bits are sonar_pyramid_noise_f32's -- the other half
 * run the planes of the call that will draw with `next_stream_id` and the `next_*` level table without storing anything and leave its
 * (sum, sumsq) partials in `partials_next`, the ones its own generating launch would leave.  SONAR_ERR_UNSUPPORTED (nothing launched) when
 * a level table is beyond the plane kernel's stretched-rows form. */
int sonar_pyramid_noise_ahead_f32(float* out, int64_t planes, int64_t H, int64_t W, int64_t nlevels, const int64_t* level_h,
                                  const int64_t* level_w, const float* level_weight, int mode, uint64_t seed, uint64_t stream_id,
                                  int64_t elem_offset, float factor, float threshold_std_devs, double* partials, int have_stats,
                                  uint64_t next_stream_id, int64_t next_nlevels, const int64_t* next_level_h, const int64_t* next_level_w,
                                  const float* next_level_weight, double* partials_next, void* stream);

/* Levels that are drawn only to be shrunk, with on-device draws: PyramidOld (py/noise_generation.py:567-606: noise = sum_i discount^i *
 * F.interpolate(normal(std = 0.5^i) at (2^(i+1) H) x (2^(i+1) W), size = (H, W), mode)) and HighresPyramid (:517-564: levels of up to 15 x
 * the latent's sides).  The levels are never materialised: level l's value at (global plane, ys, xs) is a counter-based normal keyed by
 * its global element index (Philox4x32-10 of group e / 4, Box-Muller, slot e % 4; stream stream_id + l) times level_sd[l], and the kernel
 * draws exactly the taps the interpolation reads -- mode 1 nearest-exact / 3 nearest one, 0 bilinear 2 x 2, 4 bicubic 4 x 4 per level and
 * output, by the index and weight rules of sonar_resample_acc_f32 (same ids):
 *     out[p][y][x] (+)= sum_l level_weight[l] * interpolate(level_l)[p][y][x]        (accumulate != 0: added to out)
 * Mode 2 (area) only for levels of whole multiples of the output size: the mean of a block of independent normals is itself a normal of
 * std level_sd / sqrt(block size), independent from block to block, and is drawn as such (keyed by the output element; the same joint
 * distribution as drawing the level and pooling it); any other ratio: SONAR_ERR_UNSUPPORTED (the windows overlap: draw the level with
 * sonar_level_normal_f32 and resample it).  At most 16 levels.  sonar_level_normal_f32 writes one whole level [planes][h][w] from the same
 * keys: resampling those levels with sonar_resample_acc_f32 gives the sampled kernel's values (the tests do). */
int sonar_levels_sampled_f32(float* out, int64_t planes, int64_t H, int64_t W, int nlevels, const int64_t* level_h, const int64_t* level_w,
                             const float* level_weight, const float* level_sd, int mode, uint64_t seed, uint64_t stream_id,
                             int64_t plane_offset, int accumulate, void* stream);
int sonar_level_normal_f32(float* level, int64_t planes, int64_t h, int64_t w, float sd, uint64_t seed, uint64_t stream_id,
                           int64_t plane_offset, void* stream);

/* ---------------------------------------------------------------- power-law rFFT noise (row PW) */
/* Which kernel family serves an H x W plane of the power-noise path: 1 = the fixed-size LDS FFT kernels (powers of two, 16..256),
 * 2 = the general-size kernels (any even H <= 512, even W <= 1024 with H*(W/2+1) + H + W <= 20224 complex values: two-factor table-twiddle DFTs in LDS;
 * e.g. 104 x 152 for 832 x 1216 px), 4 = half-spectrum beyond LDS (even H <= 512, even W <= 2048: 256 x 256 for 2048 px): GENERATED noise runs
 * in column blocks through a complex workspace (sonar_power_block_f32); a supplied spectrum and the spectral filter take the
 * sonar_dft_* passes as for kind 0, 0 = unsupported by the LDS kernels (the sonar_power_* / sonar_spectral_filter_f32 calls return -2 for
 * kinds 0 and 4). */
int sonar_power_plane_kind(int64_t H, int64_t W);
/* Kind-4 planes, spectrum drawn on device (py/nodes/powernoise.py:338-366 -- the reference filters the rfft2 of white noise, which IS
 * a complex-normal half-spectrum: drawn directly, no forward transform):
 *   mode 0: out = irfft2(drawn * filter, norm="ortho"); statistics of out into `partials` when given
 *   mode 1: the same, normalised like sonar_power_noise_f32 (the column pass sums the Parseval statistics of the filtered spectrum it holds
 *           into `partials`, the row pass writes normalised values): two launches, the tensor written once, the workspace written and read
 *           once (3 x the tensor of HBM traffic)
 *   mode 2: the drawn spectrum itself into `ws` as [planes][H][W/2+1] complex64 (filter, out unused): what modes 0 / 1 transform
 *   ws      [planes][H][W/2+1] complex64 scratch, sonar_power_block_ws_bytes(planes, H, W) bytes (-1: not a kind-4 plane)
 * Streams are keyed by (rng_group of global planes, block of <= 32 spectrum columns, thread slot): shards of a batch agree; the values
 * differ from the white-noise route round 3 used for these planes. */
int64_t sonar_power_block_ws_bytes(int64_t planes, int64_t H, int64_t W);
int sonar_power_block_f32(const float* filter, float* ws, float* out, int64_t planes, int64_t H, int64_t W, uint64_t seed, uint64_t stream_id,
                          int64_t plane_offset, int rng_group, int mode, float factor, float threshold_std_devs, double* partials /*mode 1: required*/,
                          void* stream);
/* py/nodes/powernoise.py:366-377: out = irfft2(z * filter, s=(H,W), norm="ortho").
 *   z      [planes][H][W/2+1] complex64 (interleaved re,im) or NULL -> drawn on device: complex normal
 *          (a+ib)*sqrt(1/2); plane p of this call is global plane plane_offset + p of the logical batch.
 *   rng_group  device draws are keyed per group of `rng_group` consecutive global planes (the Philox seeding of the
 *          xoshiro bursts is paid once per group); planes and plane_offset must be multiples of it.  It is part of
 *          the stream definition: every shard of a batch must pass the same value (the host uses 4 when C % 4 == 0,
 *          else 1).  Ignored when z is supplied.
 *   filter [H][W/2+1] fp32 (broadcast over planes)
 * Supported: H, W powers of two in 16..256 with the half-spectrum LDS-resident (else SONAR_ERR_UNSUPPORTED). */
int sonar_power_irfft2_f32(const float* z, const float* filter, float* out, int64_t planes, int64_t H, int64_t W,
                           uint64_t seed, uint64_t stream_id, int64_t plane_offset, int rng_group,
                           double* partials /*nullable*/, void* stream);
/* draw + filter + irfft2 + scale_noise(factor, normalized=True) writing the tensor ONCE: the output statistics are
 * obtained from the spectrum by Parseval's identity in a first RNG-only pass (no FFT, no stores; only the radius
 * uniforms of the interior columns and the two edge columns are drawn there). */
int sonar_power_noise_f32(const float* filter, float* out, int64_t planes, int64_t H, int64_t W, uint64_t seed,
                          uint64_t stream_id, int64_t plane_offset, int rng_group, float factor, float threshold_std_devs,
                          double* partials /*workspace*/, void* stream);

/* sonar_power_noise_f32 with a look-ahead for samplers that call the same generator step after step (consecutive stream ids).
 * have_stats != 0: `partials` already holds THIS call's statistics (what an earlier call left in its partials_next for exactly this
 * filter / shape / seed / stream_id / plane_offset / rng_group): the statistics launch is skipped.  partials_next (may be NULL):
 * receives the statistics of the same call with stream id next_stream_id, computed by the final pass's otherwise idle waves -- pass it
 * as `partials` with have_stats = 1 if the next call turns out to be that one, ignore it otherwise.  Output bits are those of
 * sonar_power_noise_f32 either way.  Only where sonar_power_noise_ahead_ok() says 1: the pipelined 128 x 128 path with at most 512 work
 * units (idle waves of the final pass), and -- as extra workgroups of the final pass's launch -- every fixed-size or general-size
 * LDS-resident plane with at most 256 work units (the launch-bound batch sizes: one launch per call); SONAR_ERR_UNSUPPORTED elsewhere.  Same call site: py/nodes/powernoise.py:338-408 followed by py/utils.py:85-106. */
int sonar_power_noise_ahead_ok(int64_t planes, int64_t H, int64_t W, int rng_group);
/* process-wide switch between the two generate kernels for 128 x 128 planes: 1 (default) the pipelined kernel where it applies (more than
 * 256 planes), 0 the phase-serial kernel everywhere; < 0 only asks.  Same stream definition, the same bits either way
 * (tests/test_gpu_round3.py); returns the previous setting.  For A/B timing and that test -- not a per-call option. */
int sonar_power_pipeline(int enable);
int sonar_power_noise_ahead_f32(const float* filter, float* out, int64_t planes, int64_t H, int64_t W, uint64_t seed,
                                uint64_t stream_id, int64_t plane_offset, int rng_group, float factor, float threshold_std_devs,
                                double* partials, int have_stats, uint64_t next_stream_id, double* partials_next, void* stream);
/* the spectrum the two entry points above draw for (seed, stream_id, plane_offset, rng_group):
 * z_out[planes][H][W/2+1] complex64 (tests / replaying a device draw) */
int sonar_power_spectrum_f32(float* z_out, int64_t planes, int64_t H, int64_t W, uint64_t seed, uint64_t stream_id,
                             int64_t plane_offset, int rng_group, void* stream);
/* spectral filter of real planes: out = irfft2(rfft2(x, norm="ortho") * filter, s=(H,W), norm="ortho") with
 * filter[H][W/2+1] real; forward and inverse FFT both LDS-resident, x read once, out written once (x != out).
 * PowerFilterNoiseItem / time_brownian path (py/nodes/powernoise.py:356-366,471-522) and, with a symmetrised
 * filter, Re(ifft2(fft2(x) * F)) of OneF / GreenTest (py/noise_generation.py:680-759).  partials: optional (sum, sumsq). */
int sonar_spectral_filter_f32(const float* x, const float* filter, float* out, int64_t planes, int64_t H, int64_t W,
                              double* partials, void* stream);
/* x *= mul / std, std = unbiased standard deviation from the (sum, sumsq) partials of n_total elements
 * (GreenTestNoiseGenerator, py/noise_generation.py:702: noise *= scale / noise.std()); no host sync */
int sonar_std_scale_f32(float* x, int64_t n, float mul, const double* partials, int64_t npart, int64_t n_total, void* stream);
/* py/nodes/powernoise.py:96-101 ChannelMixer.apply: out[b][i] = sum_j mixer[i][j] * in[b][j] over planes of hw */
int sonar_channel_mix_f32(const float* in, const float* mixer, float* out, int64_t B, int64_t C, int64_t hw,
                          double* partials /*nullable*/, void* stream);

/* ---------------------------------------------------------------- 2-D DWT / IDWT (rows W, WC, WF) */
/* pytorch_wavelets DWTForward/DWTInverse semantics == pywt.dwt2/idwt2 per level
 * (py/wavelet_functions.py:56-105).  One level per call; the host loops levels.
 *   mode: 0 zero, 1 symmetric, 2 reflect, 3 periodization, 4 periodic(ppd), 5 constant(replicate)
 *   dec_lo/dec_hi (rec_lo/rec_hi): HOST arrays of `flen` (<= 64) taps in pywt order
 *   forward : x[planes][H][W] -> ll[planes][h][w], hi[planes][3][h][w]  (orientations LH,HL,HH == pywt cH,cV,cD;
 *             h = sonar_dwt_out_len(H), w = sonar_dwt_out_len(W))
 *   inverse : ll[planes][ll_h][ll_w] (only its leading h x w block is used: pytorch_wavelets drops the extra
 *             row/column of a coarser ll), hi[planes][3][h][w] -> out[planes][Ho][Wo], Ho/Wo <= full size
 *             (2h - flen + 2, or 2h for periodization)
 *   ws      : caller-provided workspace of sonar_dwt2_ws_bytes(...) bytes (row/column pass intermediate)
 */
int64_t sonar_dwt_out_len(int64_t n, int64_t flen, int mode);
int64_t sonar_dwt2_ws_bytes(int64_t planes, int64_t H, int64_t W, int flen, int mode, int elem_size, int inverse);
int sonar_dwt2_fwd_f32(const float* x, float* ll, float* hi, int64_t planes, int64_t H, int64_t W,
                       const double* dec_lo, const double* dec_hi, int flen, int mode, void* ws, void* stream);
int sonar_dwt2_fwd_f64(const double* x, double* ll, double* hi, int64_t planes, int64_t H, int64_t W,
                       const double* dec_lo, const double* dec_hi, int flen, int mode, void* ws, void* stream);
int sonar_dwt2_inv_f32(const float* ll, int64_t ll_h, int64_t ll_w, const float* hi, float* out, int64_t planes,
                       int64_t h, int64_t w, int64_t Ho, int64_t Wo, const double* rec_lo, const double* rec_hi,
                       int flen, int mode, void* ws, void* stream);
int sonar_dwt2_inv_f64(const double* ll, int64_t ll_h, int64_t ll_w, const double* hi, double* out, int64_t planes,
                       int64_t h, int64_t w, int64_t Ho, int64_t Wo, const double* rec_lo, const double* rec_hi,
                       int flen, int mode, void* ws, void* stream);
/* 1-D DWT / IDWT of flattened latents, one level per call (replaces pytorch_wavelets DWT1DForward / DWT1DInverse behind
 * Wavelet(use_1d_dwt=True), py/wavelet_functions.py:56-57; the callers flatten [B,C,H,W] to [B,C,H*W],
 * py/wavelet_cfg.py:713-715, py/noise_generation.py:1982-1986).  x[rows][L] -> lo[rows][n], hi[rows][n] with
 * n = sonar_dwt_out_len(L, flen, mode); the inverse reads the leading n samples of each lo row (row pitch lo_len >= n:
 * a coarser approximation may be one sample longer than its band) and writes out[rows][Lo], Lo <= 2n - flen + 2
 * (2n in periodization mode).  Same taps / mode conventions as the 2-D entry points; no workspace. */
int sonar_dwt1_fwd_f32(const float* x, float* lo, float* hi, int64_t rows, int64_t L, const double* dec_lo,
                       const double* dec_hi, int flen, int mode, void* stream);
int sonar_dwt1_fwd_f64(const double* x, double* lo, double* hi, int64_t rows, int64_t L, const double* dec_lo,
                       const double* dec_hi, int flen, int mode, void* stream);
int sonar_dwt1_inv_f32(const float* lo, int64_t lo_len, const float* hi, float* out, int64_t rows, int64_t n, int64_t Lo,
                       const double* rec_lo, const double* rec_hi, int flen, int mode, void* stream);
int sonar_dwt1_inv_f64(const double* lo, int64_t lo_len, const double* hi, double* out, int64_t rows, int64_t n, int64_t Lo,
                       const double* rec_lo, const double* rec_hi, int flen, int mode, void* stream);
/* WaveletCFG band arithmetic, py/wavelet_cfg.py:750-791, for one band tensor of n elements whose element i
 * belongs to orientation group g = (i / group_size) % groups (groups = 3 for yh[B,C,3,h,w], 6 for the dual-tree transform's [B,C,6,h,w,2] with group_size = 2 h w, 1 for yl):
 *   c = cond*s_cond[g]; u = uncond*s_uncond[g]; d = (c-u)*s_diff[g]; out = blend(u, d, strength)*s_final[g]
 * s_* are HOST arrays of `groups` (<= 8) doubles; multiplications by exactly 1 are skipped like the reference. */
int sonar_wcfg_band_f32(const float* cond, const float* uncond, float* out, int64_t n, int64_t group_size,
                        int64_t groups, const double* s_cond, const double* s_uncond, const double* s_diff,
                        const double* s_final, int blend_mode, double strength, void* stream);
int sonar_wcfg_band_f64(const double* cond, const double* uncond, double* out, int64_t n, int64_t group_size,
                        int64_t groups, const double* s_cond, const double* s_uncond, const double* s_diff,
                        const double* s_final, int blend_mode, double strength, void* stream);
/* The same arithmetic on a band of the 1-D transform ([B, C, l] rows of row_len coefficients, use_1d_dwt).  The reference's
 * wavelet_scaling (py/wavelet_functions.py:208-215) indexes axis 2 of every band: the orientation axis of a 2-D band, but the
 * COEFFICIENT axis of a 1-D band, whose scale table has one entry -- so only the first coefficient of each row is scaled.
 * s_* are HOST pointers to ONE double each (NULL = 1): applied where i % row_len == 0; every other element uses unit scales. */
int sonar_wcfg_band_head_f32(const float* cond, const float* uncond, float* out, int64_t n, int64_t row_len,
                             const double* s_cond, const double* s_uncond, const double* s_diff, const double* s_final,
                             int blend_mode, double strength, void* stream);
int sonar_wcfg_band_head_f64(const double* cond, const double* uncond, double* out, int64_t n, int64_t row_len,
                             const double* s_cond, const double* s_uncond, const double* s_diff, const double* s_final,
                             int blend_mode, double strength, void* stream);
/* The whole transform-domain step of WaveletCFG in `2 * levels` launches (py/wavelet_cfg.py:750-791,729-748), LDS-staged:
 * each level's analysis handles cond and uncond together (fp32 inputs cast in registers), applies
 * blend(u*s_u, (c*s_c - u*s_u)*s_d, strength)*s_f per band before storing; each level's synthesis is one launch; the last
 * one writes out = x - result (subtract_from_x) or result, cropped to H x W, as fp32.  _f32 / _f64 = arithmetic type
 * (high_precision_mode).  yl_scales[4] = {cond, uncond, diff, final}; yh_scales[levels][4][3] = the same per level
 * (finest first) and orientation (cH, cV, cD).  ws: sonar_wcfg_fused_ws_bytes(...) bytes; returns
 * SONAR_ERR_UNSUPPORTED (nothing launched) when a level does not fit the LDS tile.
 * perfect_reconstruction != 0: the caller vouches that (dec, mode_fwd) / (rec, mode_inv) reconstruct exactly (one wavelet both
 * ways).  Rules that then only scale the difference bands (every cond / uncond / final scale 1, any per-level, per-orientation
 * diff scales) take the single-tensor route: IDWT(blend(U, D (C - U), t)) = ku u + kt IDWT(D DWT(c - u)) -- c - u is formed on
 * load, ONE transform runs instead of two, and the last synthesis adds ku * uncond. */
int64_t sonar_wcfg_fused_ws_bytes(int64_t planes, int64_t H, int64_t W, int levels, int dec_len, int mode_fwd, int rec_len,
                                  int mode_inv, int elem_size);
int sonar_wcfg_fused_f32(const float* cond, const float* uncond, const float* x, float* out, int64_t planes, int64_t H, int64_t W,
                         int levels, const double* dec_lo, const double* dec_hi, int dec_len, int mode_fwd, const double* rec_lo,
                         const double* rec_hi, int rec_len, int mode_inv, const double* yl_scales, const double* yh_scales,
                         int blend_mode, double strength, int subtract_from_x, int perfect_reconstruction, void* ws, int64_t ws_bytes,
                         void* stream);
/* sonar_wcfg_fused_f64 keeps level 1's three detail bands -- 3/4 of that level's coefficients, written once and read once -- in fp32 in
 * its workspace (arithmetic on both sides and every other band stay fp64; the result is an fp32 tensor): 110 MB less HBM traffic per 256 SDXL
 * latents.  sonar_wcfg_hi_storage(0) stores them in fp64 (1: fp32, the default; -1: query); returns the previous setting.  Process-wide. */
int sonar_wcfg_hi_storage(int fp32);
int sonar_wcfg_fused_f64(const float* cond, const float* uncond, const float* x, float* out, int64_t planes, int64_t H, int64_t W,
                         int levels, const double* dec_lo, const double* dec_hi, int dec_len, int mode_fwd, const double* rec_lo,
                         const double* rec_hi, int rec_len, int mode_inv, const double* yl_scales, const double* yh_scales,
                         int blend_mode, double strength, int subtract_from_x, int perfect_reconstruction, void* ws, int64_t ws_bytes,
                         void* stream);
/* ---------------------------------------------------------------- DTCWT (row 8f-4)
 * The dual-tree complex wavelet transform the reference reaches through pytorch_wavelets' DTCWTForward / DTCWTInverse
 * (py/wavelet_functions.py:56-73; Kingsbury's dtwavexfm2 / dtwaveifm2).  Each of its stages (odd-length filters with symmetric
 * extension, decimating / interpolating dual-tree filters) is a sparse linear map along one axis with `taps` terms per output row:
 *   out[o][j][i] (+)= sum_k coef[j][k] * x[o][idx[j][k]][i],  o < outer, j < n_out, i < inner
 * idx [n_out][taps] (int32) / coef [n_out][taps]: DEVICE tables built once per length by the host (py/dtcwt.py).  x != out. */
int sonar_axis_taps_f32(const float* x, float* out, int64_t outer, int64_t n_in, int64_t n_out, int64_t inner, const int* idx,
                        const float* coef, int taps, int accumulate, void* stream);
int sonar_axis_taps_f64(const double* x, double* out, int64_t outer, int64_t n_in, int64_t n_out, int64_t inner, const int* idx,
                        const double* coef, int taps, int accumulate, void* stream);
/* q2c / c2q: three real planes lh, hh, hl [planes][2h][2w] <-> bands [planes][6][h][w][2] (orientations 15, 45, 75, 105, 135, 165
 * degrees, last axis re / im): quad (a b / c d) <-> ((a - d) + i (b + c)) / sqrt 2 and ((a + d) + i (b - c)) / sqrt 2; lh feeds
 * orientations (0, 5), hh (1, 4), hl (2, 3). */
int sonar_dtcwt_q2c_f32(const float* lh, const float* hh, const float* hl, float* bands, int64_t planes, int64_t h, int64_t w, void* stream);
int sonar_dtcwt_c2q_f32(const float* bands, float* lh, float* hh, float* hl, int64_t planes, int64_t h, int64_t w, void* stream);
int sonar_dtcwt_q2c_f64(const double* lh, const double* hh, const double* hl, double* bands, int64_t planes, int64_t h, int64_t w, void* stream);
int sonar_dtcwt_c2q_f64(const double* bands, double* lh, double* hh, double* hl, int64_t planes, int64_t h, int64_t w, void* stream);
/* normalize_to_scale_adv (py/utils.py:473-510; NormalizeToScaleNoise's advanced mode, py/noise.py:1262-1286): per row of `inner` elements the
 * negative values are rescaled between their own extremes to [min_neg, max_neg] (max_neg >= 0: the row's largest negative value) and the
 * positive ones to [min_pos, max_pos] (min_pos < 0: the row's smallest positive value); zeros stay, a sign whose range is degenerate
 * (:482-483) is copied.  stats_ws: 4 floats per row, 16-byte aligned.  Two launches. */
int sonar_signed_rescale_f32(const float* x, int64_t rows, int64_t inner, double min_neg, double max_neg, double min_pos, double max_pos, float eps,
                             float* stats_ws, float* out, void* stream);
/* Direct real 2-D DFT passes for planes the LDS-resident FFT kernels do not take (odd heights / widths -- 1080-line video gives
 * 135-row latents -- or planes beyond the LDS budget): torch.fft.rfft2 / irfft2 (py/nodes/powernoise.py:338-408,
 * py/noise_generation.py:680-759, py/nodes/freeu_extreme.py:10-29) as rows r2c -> columns (optionally x a real filter [H][K] on the
 * way in; inverse: e^{+}) -> rows c2r (x scale; the imaginary parts of the DC / Nyquist columns are ignored, as irfft does), through a
 * caller-owned complex64 workspace [planes][H][W/2+1].  Lines of at most 2048; unscaled except for `scale`.  Rows and columns run the
 * LDS line transforms of the general-size kernels (register codelets, a factor above 19 as direct sums; even widths with the
 * half-length trick, odd widths as full-length complex lines); an odd width that is a prime above 19, or buffers that are not 8-byte
 * aligned, run direct sums, O(N) per output.  sonar_dft_cols_f32: inverse = 0 forward, 1 inverse (the
 * filter multiplies the INPUT), 2 forward, x filter, inverse in one pass (the spectral filter's middle; SONAR_ERR_UNSUPPORTED when the
 * columns cannot go through LDS: run 0 then 1).  partials (nullable): (sum, sumsq) of the real output. */
int sonar_dft_rows_r2c_f32(const float* x, float* y, int64_t rows, int64_t W, void* stream);
int sonar_dft_cols_f32(const float* in, const float* filter /*nullable*/, float* out, int64_t planes, int64_t H, int64_t K, int inverse,
                       void* stream);
int sonar_dft_rows_c2r_f32(const float* y, float* out, int64_t rows, int64_t W, float scale, double* partials /*nullable*/, void* stream);
/* max over a non-empty device vector with torch.max's NaN rule, returned to the host: WaveletCFG's `sigma.max().item()`
 * (py/wavelet_cfg.py:795-796) as one launch that writes into pinned host memory + one wait.  `begin` only launches; `end` BLOCKS
 * until the value has landed (as `.item()` does) -- the host prepares everything that does not depend on the value in between.
 * One request per host thread and device at a time (`begin` twice without `end`: SONAR_ERR_ARG).  `sonar_max_to_host_f32` = both. */
int sonar_max_to_host_begin_f32(const float* x, int64_t n, void* stream);
int sonar_max_to_host_end_f32(float* result, void* stream);
int sonar_max_to_host_f32(const float* x, int64_t n, float* result, void* stream);
/* WaveletCFG for difference-only rules with ONE detail scale per level (py/wavelet_cfg.py:750-791 with `cond` / `uncond` / `final`
 * absent; the node's placeholder rule, BASELINE cfg4).  By linearity and perfect reconstruction
 *   IDWT(blend(DWT u, D (DWT c - DWT u), t)) = ku u + kt (g[0] v + Up_1(g[1] LL_1 + Up_2(... g[J] LL_J))),  v = c - u,
 * LL_j = low-pass analysis chain of v, Up_j = level-j synthesis with zero details, g[0] = d_1, g[j] = d_{j+1} - d_j, g[J] = l - d_J
 * (d_j: the detail scale of level j, l: the approximation scale; (ku, kt) = (1, t) inject, (1 - t, t) lerp, (1, -t) subtract_b).
 * One launch, a workgroup per plane, the LL pyramid stays in LDS: out = x - result (subtract_from_x) or result; fp32 tensors,
 * _f32 / _f64 = arithmetic type.  dec_lo / rec_lo: the `flen` (even, <= 20) low-pass taps of ONE wavelet (perfect-reconstruction
 * pair); g: HOST array of levels + 1 doubles.  SONAR_ERR_UNSUPPORTED when the pyramid does not fit in LDS
 * (sonar_wcfg_lowpass_lds_bytes < 0): callers use sonar_wcfg_fused_* then. */
int64_t sonar_wcfg_lowpass_lds_bytes(int64_t H, int64_t W, int levels, int flen, int mode_fwd, int mode_inv, int elem_size);
int sonar_wcfg_lowpass_f32(const float* cond, const float* uncond, const float* x, float* out, int64_t planes, int64_t H,
                           int64_t W, int levels, const double* dec_lo, const double* rec_lo, int flen, int mode_fwd,
                           int mode_inv, const double* g, double ku, double kt, int subtract_from_x, void* stream);
int sonar_wcfg_lowpass_f64(const float* cond, const float* uncond, const float* x, float* out, int64_t planes, int64_t H,
                           int64_t W, int levels, const double* dec_lo, const double* rec_lo, int flen, int mode_fwd,
                           int mode_inv, const double* g, double ku, double kt, int subtract_from_x, void* stream);
/* WaveletCFG with ANY per-level, per-orientation band scales in one launch, the coefficients resident in LDS (py/wavelet_cfg.py:750-791,
 * py/wavelet_functions.py:193-238).  Phi_D(v) = IDWT(D DWT(v)) with D = (yl_scale for the approximation, yh_scales[levels][3] = (cH, cV,
 * cD) per level, finest first) is linear in v, and with a perfect-reconstruction pair a level needs its approximation and ONE detail band
 * on chip (csrc/dwt_bands.h).  v = a - b (b nullable: v = a);  result = ku * b + kt * Phi_D(v);  out = x - (float)result
 * (subtract_from_x) or (float)result; x may alias out.  Difference-only rules: a = cond, b = uncond, D = the difference scales, (ku, kt)
 * as in sonar_wcfg_lowpass_*.  Any other rule with a linear blend is Phi_A(cond) + Phi_B(uncond): two launches, the second with x = out of
 * the first.  fp32 tensors, _f32 / _f64 = arithmetic type; dec / rec: the four filters (flen taps each, even, <= 20) of ONE wavelet.
 * SONAR_ERR_UNSUPPORTED when the coefficients do not fit in LDS (sonar_wcfg_bands_lds_bytes < 0) or the extension pair does not
 * reconstruct: callers use sonar_wcfg_fused_* then. */
int64_t sonar_wcfg_bands_lds_bytes(int64_t H, int64_t W, int levels, int flen, int mode_fwd, int mode_inv, int elem_size, int per_orientation);
int sonar_wcfg_bands_f32(const float* a, const float* b, const float* x, float* out, int64_t planes, int64_t H, int64_t W, int levels,
                         const double* dec_lo, const double* dec_hi, const double* rec_lo, const double* rec_hi, int flen, int mode_fwd,
                         int mode_inv, const double* yh_scales, double yl_scale, double ku, double kt, int subtract_from_x, void* stream);
int sonar_wcfg_bands_f64(const float* a, const float* b, const float* x, float* out, int64_t planes, int64_t H, int64_t W, int levels,
                         const double* dec_lo, const double* dec_hi, const double* rec_lo, const double* rec_hi, int flen, int mode_fwd,
                         int mode_inv, const double* yh_scales, double yl_scale, double ku, double kt, int subtract_from_x, void* stream);
/* process_output, py/wavelet_cfg.py:729-748: out = x - (float)crop(result)  (subtract_from_x = 1, target DENOISED)
 * or out = (float)crop(result); result is [planes][Hr][Wr] (f64 or f32), x/out are [planes][H][W] fp32 */
int sonar_wcfg_output_f32(const float* x, const void* result, int result_is_f64, float* out, int64_t planes,
                          int64_t H, int64_t W, int64_t Hr, int64_t Wr, int subtract_from_x, void* stream);
/* fp32 -> fp64 conversion of a contiguous buffer (get_context cast, py/wavelet_cfg.py:707,764-765) */
int sonar_cast_f32_f64(const float* in, double* out, int64_t n, void* stream);

/* ---------------------------------------------------------------- prepared call plans (host floor of rows C, G1, P, Y, PW) */
/* The reference composes a sampler step from Python closures (py/noise.py:137-257, py/noise_generation.py:134-258); at the batch sizes
 * a ComfyUI run uses (1-4, cfg3's 64) the interpreter work per step outweighs the kernels.  A plan is the step resolved once: the entry
 * points of this header it issues, in order, with their arguments.  sonar_plan_run patches the values that change from call to call into
 * the recorded arguments -- tensor addresses (slots), the RNG seed and stream ids, the pyramid's level table -- and calls the same entry
 * points: same launches, same bits, one foreign call per step.  Host-side objects; not thread safe per plan. */
typedef struct sonar_plan sonar_plan;
#define SONAR_PATCH_SLOT 0   /* slots[index] + addend: a device address (or scalar) handed to sonar_plan_run */
#define SONAR_PATCH_STREAM 1 /* stream_base + addend: an RNG stream id */
#define SONAR_PATCH_SEED 2   /* the RNG seed */
#define SONAR_PATCH_BLOB 3   /* address of byte `addend` of the record's blob (array and struct arguments) */
#define SONAR_PATCH_LEVELS 4 /* pyramid level table (sonar_plan_levels at blob offset `index`) recomputed from (seed, stream_base + addend);
                                the level count goes to argument `target` */
typedef struct sonar_plan_patch {
    int32_t source; /* SONAR_PATCH_* */
    int32_t target; /* >= 0: argument index; < 0: byte offset -(target + 1) into the record's blob */
    int32_t index;  /* SLOT: slot number; LEVELS: blob offset of the rule */
    int32_t width;  /* bytes written at a blob target: 4 or 8 */
    int64_t addend;
} sonar_plan_patch;
typedef struct sonar_plan_levels {
    int64_t H, W;
    double discount;
    int32_t iterations;
    int32_t reserved;
    int64_t h_offset, w_offset, weight_offset; /* blob offsets of the int64 / int64 / float tables (`iterations` entries each) */
} sonar_plan_levels;
/* index of a replayable entry point (one that only launches on its last argument, the stream) or -1; its argument count */
int sonar_plan_fn_id(const char* name);
int sonar_plan_fn_nargs(int fn_id);
sonar_plan* sonar_plan_create(int nslots);
void sonar_plan_destroy(sonar_plan* plan);
int sonar_plan_length(const sonar_plan* plan);
/* append one call: `args` = one 64-bit word per argument (integers sign-extended, float / double bit patterns, addresses), `blob` = bytes
 * the arguments point into (copied), `patches` = what sonar_plan_run rewrites before the call (the last argument, the stream, always is) */
int sonar_plan_add(sonar_plan* plan, int fn_id, const uint64_t* args, int nargs, const void* blob, int64_t blob_bytes,
                   const sonar_plan_patch* patches, int npatches);
/* issue every record on `stream`; stops at the first entry point that fails and returns its code (`failed_record`, nullable: its index,
 * -1 when all ran; sonar_last_error() is that entry point's message) */
int sonar_plan_run(sonar_plan* plan, const uint64_t* slots, int nslots, uint64_t seed, uint64_t stream_base, void* stream,
                   int* failed_record);
/* the level sizes and weights of a device-mode pyramid draw (py/noise_generation.py:609-649: r = rand * 2 + 2 per level from a host
 * splitmix64 sequence keyed by (seed, stream_id)); returns the level count.  Pure host arithmetic. */
int sonar_pyramid_levels(int64_t H, int64_t W, int iterations, double discount, uint64_t seed, uint64_t stream_id, int64_t* level_h,
                         int64_t* level_w, float* weight);

#ifdef __cplusplus
}
#endif
#endif /* SONAR_HIP_H */
