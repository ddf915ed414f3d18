// Prepared call plans: a noise sampler's step resolved ONCE into an array of launch records and replayed with one foreign call.
//
// The reference builds a sampler step out of Python closures (py/noise.py:137-257: chain -> items -> generators); the host side here
// keeps that object model, and at batch 1-64 its ~200 interpreter-level calls per step cost more than the kernels they launch.  A plan is
// what such a step issues when nothing but the RNG position and the output tensors changes between calls: the entry points of this
// library, in order, with their arguments.  sonar_plan_run patches the per-call values (tensor addresses, seed, stream ids, the pyramid's
// level table) into the recorded arguments and calls the SAME entry points -- same launches, same bits, one ctypes crossing.
#include <math.h>
#include <string.h>

#include <memory>
#include <type_traits>
#include <utility>
#include <vector>

#include "common.h"

namespace sonar {
namespace {

// ---- an entry point called from a row of 64-bit argument words --------------------------------------------------------------------
template <typename T>
inline T word_to(uint64_t w) {
    if constexpr (std::is_pointer_v<T>) {
        return reinterpret_cast<T>(static_cast<uintptr_t>(w));
    } else if constexpr (std::is_same_v<T, float>) {
        const uint32_t lo = (uint32_t)w;
        float f;
        memcpy(&f, &lo, sizeof f);
        return f;
    } else if constexpr (std::is_same_v<T, double>) {
        double d;
        memcpy(&d, &w, sizeof d);
        return d;
    } else {
        static_assert(std::is_integral_v<T>, "entry points take pointers, integers, float and double");
        return static_cast<T>(w);
    }
}

using Thunk = int (*)(const uint64_t*);

template <auto Fn>
struct Entry;
template <typename... A, int (*Fn)(A...)>
struct Entry<Fn> {
    static constexpr int kArgs = (int)sizeof...(A);
    template <size_t... I>
    static int call(const uint64_t* w, std::index_sequence<I...>) {
        return Fn(word_to<A>(w[I])...);
    }
    static int thunk(const uint64_t* w) { return call(w, std::index_sequence_for<A...>{}); }
};

struct Replayable {
    const char* name;
    Thunk thunk;
    int nargs;
};

// Every entry point that only LAUNCHES on its stream argument (the last one): no host-visible result, no host synchronisation.
#define SONAR_REPLAYABLE(fn) Replayable{#fn, &Entry<&fn>::thunk, Entry<&fn>::kArgs}
const Replayable kReplayable[] = {
    SONAR_REPLAYABLE(sonar_stats_f32),
    SONAR_REPLAYABLE(sonar_scale_noise_f32),
    SONAR_REPLAYABLE(sonar_scale_noise_stats_f32),
    SONAR_REPLAYABLE(sonar_axpby_f32),
    SONAR_REPLAYABLE(sonar_axpby_stats_f32),
    SONAR_REPLAYABLE(sonar_affine_f32),
    SONAR_REPLAYABLE(sonar_norm_decision_f32),
    SONAR_REPLAYABLE(sonar_apply_norm_f32),
    SONAR_REPLAYABLE(sonar_philox_normal_f32),
    SONAR_REPLAYABLE(sonar_philox_uniform_f32),
    SONAR_REPLAYABLE(sonar_philox_noise_f32),
    SONAR_REPLAYABLE(sonar_philox_noise_ahead_f32),
    SONAR_REPLAYABLE(sonar_philox_normal_acc_f32),
    SONAR_REPLAYABLE(sonar_philox_normal_chain_f32),
    SONAR_REPLAYABLE(sonar_perlin_lattice_f32),
    SONAR_REPLAYABLE(sonar_perlin_generate_f32),
    SONAR_REPLAYABLE(sonar_perlin_generate_acc_f32),
    SONAR_REPLAYABLE(sonar_perlin_generate_chain_f32),
    SONAR_REPLAYABLE(sonar_perlin_noise_f32),
    SONAR_REPLAYABLE(sonar_perlin_noise_ahead_f32),
    SONAR_REPLAYABLE(sonar_pyramid_generate_f32),
    SONAR_REPLAYABLE(sonar_pyramid_generate_acc_f32),
    SONAR_REPLAYABLE(sonar_pyramid_generate_acc_ahead_f32),
    SONAR_REPLAYABLE(sonar_pyramid_noise_f32),
    SONAR_REPLAYABLE(sonar_pyramid_noise_ahead_f32),
    SONAR_REPLAYABLE(sonar_power_noise_f32),
    SONAR_REPLAYABLE(sonar_power_block_f32),
    SONAR_REPLAYABLE(sonar_power_noise_ahead_f32),
    SONAR_REPLAYABLE(sonar_power_irfft2_f32),
    SONAR_REPLAYABLE(sonar_spectral_filter_f32),
    SONAR_REPLAYABLE(sonar_channel_mix_f32),
    SONAR_REPLAYABLE(sonar_std_scale_f32),
    SONAR_REPLAYABLE(sonar_powerlaw_f32),
    SONAR_REPLAYABLE(sonar_mul_table_f32),
    SONAR_REPLAYABLE(sonar_stats_finalize),
    SONAR_REPLAYABLE(sonar_scale_noise_rows_f32),
    SONAR_REPLAYABLE(sonar_blend_f32),
    SONAR_REPLAYABLE(sonar_blend_tensor_f32),
    SONAR_REPLAYABLE(sonar_scalar_op_f32),
    SONAR_REPLAYABLE(sonar_rowstats_f32),
    SONAR_REPLAYABLE(sonar_row_affine_f32),
    SONAR_REPLAYABLE(sonar_amax_mid_f32),
    SONAR_REPLAYABLE(sonar_div_mid_f32),
    SONAR_REPLAYABLE(sonar_mask_mix_f32),
    SONAR_REPLAYABLE(sonar_minmax_rows_f32),
    SONAR_REPLAYABLE(sonar_minmax_rescale_f32),
    SONAR_REPLAYABLE(sonar_levels_sampled_f32),
    SONAR_REPLAYABLE(sonar_level_normal_f32),
    SONAR_REPLAYABLE(sonar_resample_acc_f32),
    SONAR_REPLAYABLE(sonar_power_spectrum_f32),
    SONAR_REPLAYABLE(sonar_rfft2_f32),
    SONAR_REPLAYABLE(sonar_cdft_mid_f32),
    SONAR_REPLAYABLE(sonar_spectral_logamp_f32),
    SONAR_REPLAYABLE(sonar_spectral_signum_mask_f32),
    SONAR_REPLAYABLE(sonar_std_mid_f32),
    SONAR_REPLAYABLE(sonar_bcast_gain_f32),
    SONAR_REPLAYABLE(sonar_ratio_mix_f32),
    SONAR_REPLAYABLE(sonar_sq_acc_f32),
    SONAR_REPLAYABLE(sonar_studentt_f32),
    SONAR_REPLAYABLE(sonar_abs_quantile_rows_f32),
    SONAR_REPLAYABLE(sonar_clamp_signpow_rows_f32),
    SONAR_REPLAYABLE(sonar_laplace_add_f32),
    SONAR_REPLAYABLE(sonar_dft_rows_r2c_f32),
    SONAR_REPLAYABLE(sonar_dft_cols_f32),
    SONAR_REPLAYABLE(sonar_dft_rows_c2r_f32),
    SONAR_REPLAYABLE(sonar_perlin_terms_f32),
    SONAR_REPLAYABLE(sonar_perlin_apply_f32),
};
#undef SONAR_REPLAYABLE
constexpr int kReplayableCount = (int)(sizeof(kReplayable) / sizeof(kReplayable[0]));
constexpr int kMaxArgs = 32;

struct Record {
    int fn;
    int nargs;
    uint64_t args[kMaxArgs];
    std::vector<uint8_t> blob;  // arrays and structs the arguments point at
    std::vector<sonar_plan_patch> patches;
};

// a patched value goes into the CALL's copy of the argument words / blob: the plan's records are read-only while it runs
inline void put(uint64_t* args, uint8_t* blob, int32_t target, int32_t width, uint64_t value) {
    if (target >= 0) {
        args[target] = value;
    } else {
        memcpy(blob + (size_t)(-(target + 1)), &value, (size_t)width);  // little endian: the low `width` bytes
    }
}

}  // namespace
}  // namespace sonar

struct sonar_plan {
    std::vector<sonar::Record> records;
    int nslots;
};

using namespace sonar;

// PyramidNoiseGenerator._plan / _level_ratios (py/noise_generation.py:609-649: `r = rand(1) * 2 + 2` per level, sizes shrink by r^i
// cumulatively, weight discount^i, stop at a 1-pixel side): the host-side scalar sequence of a device-mode draw, keyed by the call's
// (seed, stream) -- splitmix64, the same doubles in the same order as comfyui-sonar_amd/py/noise_generation.py computes them.
extern "C" int sonar_pyramid_levels(int64_t H, int64_t W, int iterations, double discount, uint64_t seed, uint64_t stream_id,
                                    int64_t* level_h, int64_t* level_w, float* weight) {
    SONAR_REQUIRE(H > 0 && W > 0 && iterations >= 0 && level_h && level_w && weight, SONAR_ERR_ARG, "sonar_pyramid_levels: bad argument");
    uint64_t state = seed * 0x9E3779B97F4A7C15ull + stream_id;
    int64_t cw = W, ch = H;
    int n = 0;
    for (int i = 0; i < iterations; ++i) {
        state += 0x9E3779B97F4A7C15ull;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        const double r = (double)(z >> 11) * (2.0 / 9007199254740992.0) + 2.0;
        const double shrink = pow(r, (double)i);
        cw = std::max<int64_t>(1, (int64_t)((double)cw / shrink));
        ch = std::max<int64_t>(1, (int64_t)((double)ch / shrink));
        level_h[n] = ch;
        level_w[n] = cw;
        weight[n] = (float)pow(discount, (double)i);
        ++n;
        if (cw == 1 || ch == 1) break;
    }
    return n;
}

extern "C" int sonar_plan_fn_id(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < kReplayableCount; ++i)
        if (strcmp(kReplayable[i].name, name) == 0) return i;
    return -1;
}

extern "C" int sonar_plan_fn_nargs(int fn_id) { return fn_id >= 0 && fn_id < kReplayableCount ? kReplayable[fn_id].nargs : -1; }

extern "C" sonar_plan* sonar_plan_create(int nslots) {
    if (nslots < 0) return nullptr;
    sonar_plan* p = new (std::nothrow) sonar_plan();
    if (p) p->nslots = nslots;
    return p;
}

extern "C" void sonar_plan_destroy(sonar_plan* plan) { delete plan; }

extern "C" int sonar_plan_length(const sonar_plan* plan) { return plan ? (int)plan->records.size() : -1; }

extern "C" int sonar_plan_add(sonar_plan* plan, int fn_id, const uint64_t* args, int nargs, const void* blob, int64_t blob_bytes,
                              const sonar_plan_patch* patches, int npatches) {
    SONAR_REQUIRE(plan && fn_id >= 0 && fn_id < kReplayableCount, SONAR_ERR_ARG, "sonar_plan_add: unknown plan or entry point");
    const Replayable& fn = kReplayable[fn_id];
    SONAR_REQUIRE(nargs == fn.nargs && nargs <= kMaxArgs && args, SONAR_ERR_ARG, "sonar_plan_add: %s takes %d arguments, got %d", fn.name,
                  fn.nargs, nargs);
    SONAR_REQUIRE(blob_bytes >= 0 && (blob_bytes == 0 || blob) && npatches >= 0 && (npatches == 0 || patches), SONAR_ERR_ARG,
                  "sonar_plan_add: bad blob / patch list");
    Record r;
    r.fn = fn_id;
    r.nargs = nargs;
    memcpy(r.args, args, sizeof(uint64_t) * (size_t)nargs);
    r.blob.assign((const uint8_t*)blob, (const uint8_t*)blob + blob_bytes);
    for (int i = 0; i < npatches; ++i) {
        const sonar_plan_patch& pt = patches[i];
        const bool arg_target = pt.target >= 0;
        SONAR_REQUIRE(arg_target ? pt.target < nargs - 1 : (int64_t)(-(pt.target + 1)) + pt.width <= blob_bytes && (pt.width == 4 || pt.width == 8),
                      SONAR_ERR_ARG, "sonar_plan_add: patch %d of %s writes outside its record", i, fn.name);
        switch (pt.source) {
            case SONAR_PATCH_SLOT:
                SONAR_REQUIRE(pt.index >= 0 && pt.index < plan->nslots, SONAR_ERR_ARG, "sonar_plan_add: patch %d names slot %d of %d", i,
                              pt.index, plan->nslots);
                break;
            case SONAR_PATCH_STREAM:
            case SONAR_PATCH_SEED:
                break;
            case SONAR_PATCH_BLOB:
                SONAR_REQUIRE(pt.addend >= 0 && pt.addend <= blob_bytes, SONAR_ERR_ARG, "sonar_plan_add: patch %d points outside the blob", i);
                break;
            case SONAR_PATCH_LEVELS: {
                // index: blob offset of a sonar_plan_levels rule; its table offsets must lie inside the blob
                SONAR_REQUIRE(pt.index >= 0 && (int64_t)pt.index + (int64_t)sizeof(sonar_plan_levels) <= blob_bytes && arg_target, SONAR_ERR_ARG,
                              "sonar_plan_add: patch %d: level rule outside the blob", i);
                sonar_plan_levels lv;
                memcpy(&lv, r.blob.data() + pt.index, sizeof lv);
                const int64_t n = lv.iterations;
                // every bound without an addition of caller-supplied 64-bit values (a huge offset must not wrap past the check); the tables
                // are read and written as int64 / float: their offsets must be aligned for that
                auto inside = [&](int64_t off, int64_t elem) { return off >= 0 && off % elem == 0 && n <= (blob_bytes - off) / elem && off <= blob_bytes; };
                SONAR_REQUIRE(n >= 0 && n <= 64 && lv.H > 0 && lv.W > 0 && inside(lv.h_offset, 8) && inside(lv.w_offset, 8) && inside(lv.weight_offset, 4),
                              SONAR_ERR_ARG, "sonar_plan_add: patch %d: level rule out of range (H, W > 0, 0 <= iterations <= 64, aligned tables inside the blob)", i);
                break;
            }
            default:
                SONAR_REQUIRE(false, SONAR_ERR_ARG, "sonar_plan_add: patch %d has an unknown source %d", i, pt.source);
        }
        r.patches.push_back(pt);
    }
    plan->records.push_back(std::move(r));
    return SONAR_OK;
}

// Re-entrant: the per-call values are patched into a copy of each record's argument words (and of its blob, when a patch or an argument
// points into it) on this call's stack -- two threads may replay the SAME plan at once on their own streams (round 5; round 4 patched the
// shared records in place).
extern "C" int sonar_plan_run(sonar_plan* plan, const uint64_t* slots, int nslots, uint64_t seed, uint64_t stream_base, void* stream,
                              int* failed_record) {
    SONAR_REQUIRE(plan && nslots == plan->nslots && (nslots == 0 || slots), SONAR_ERR_ARG, "sonar_plan_run: bad plan / slot table");
    if (failed_record) *failed_record = -1;
    int idx = 0;
    constexpr size_t kStackBlob = 1024;
    alignas(16) uint8_t stack_blob[kStackBlob];
    std::vector<uint8_t> heap_blob;
    for (const Record& r : plan->records) {
        uint64_t args[kMaxArgs];
        memcpy(args, r.args, sizeof(uint64_t) * (size_t)r.nargs);
        uint8_t* blob = nullptr;
        if (!r.blob.empty()) {
            if (r.blob.size() <= kStackBlob) {
                blob = stack_blob;
            } else {
                heap_blob.resize(r.blob.size());
                blob = heap_blob.data();
            }
            memcpy(blob, r.blob.data(), r.blob.size());
        }
        for (const sonar_plan_patch& pt : r.patches) {
            switch (pt.source) {
                case SONAR_PATCH_SLOT: put(args, blob, pt.target, pt.width, slots[pt.index] + (uint64_t)pt.addend); break;
                case SONAR_PATCH_STREAM: put(args, blob, pt.target, pt.width, stream_base + (uint64_t)pt.addend); break;
                case SONAR_PATCH_SEED: put(args, blob, pt.target, pt.width, seed); break;
                case SONAR_PATCH_BLOB: put(args, blob, pt.target, pt.width, (uint64_t)(uintptr_t)(blob + pt.addend)); break;
                case SONAR_PATCH_LEVELS: {
                    sonar_plan_levels lv;
                    memcpy(&lv, blob + pt.index, sizeof lv);
                    const int n = sonar_pyramid_levels(lv.H, lv.W, lv.iterations, lv.discount, seed, stream_base + (uint64_t)pt.addend,
                                                       reinterpret_cast<int64_t*>(blob + lv.h_offset), reinterpret_cast<int64_t*>(blob + lv.w_offset),
                                                       reinterpret_cast<float*>(blob + lv.weight_offset));
                    if (n < 0) {
                        if (failed_record) *failed_record = idx;
                        return n;
                    }
                    args[pt.target] = (uint64_t)(int64_t)n;
                    break;
                }
            }
        }
        args[r.nargs - 1] = (uint64_t)(uintptr_t)stream;
        const int rc = kReplayable[r.fn].thunk(args);
        if (rc != SONAR_OK) {
            if (failed_record) *failed_record = idx;
            return rc;
        }
        ++idx;
    }
    return SONAR_OK;
}
