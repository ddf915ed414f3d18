#!/usr/bin/env python3
"""profiles/rNN_traffic.json from tools/rocpd_traffic.py's per-kernel table (scratch/prof_r04.sh writes it as traffic_raw.json):
    python tools/make_traffic_json.py gpurun_out/r03/traffic_raw.json > profiles/r03_traffic.json
Kernels are picked by name fragments; batch-512 and batch-64 launches of one kernel (same name) are split in proportion to the batch."""
import json
import sys

N = 4 * 128 * 128 * 4  # bytes of one fp32 SDXL latent


def pick(raw, *frags, exclude=()):
    hits = {k: v for k, v in raw.items() if all(f in k for f in frags) and not any(e in k for e in exclude)}
    if len(hits) != 1:
        raise SystemExit(f"{frags}: {len(hits)} kernels match: {sorted(hits)}")
    return next(iter(hits.values()))


def total(*rows):
    return int(sum(r["hbm_bytes_per_launch"] for r in rows))


def main(path):
    raw = json.load(open(path))
    out = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes over scratch/prof_workload.py (scratch/prof_r04.sh, "
                   "tools/rocpd_traffic.py, tools/make_traffic_json.py); raw counters are KiB per dispatch.  hbm_bytes_per_launch = 2 x FETCH_SIZE + "
                   "WRITE_SIZE: the gfx950 correction of MI355X_MICROARCH.md, confirmed in the same run by the calibration kernels below "
                   "(one 134 217 728-byte tensor written / read / read + written)."}
    out["calibration"] = {k: raw[k] for k in raw if k.startswith(("stream_fill_kernel<", "stats_kernel<", "scale_noise_kernel<"))}
    scale = pick(raw, "scale_noise_kernel<4, false>")  # the calibration launch at 512 latents (the 32 MiB-and-below form is <4, true>)
    fin = pick(raw, "power_pipe_kernel<128, 128, false, true>")
    out["power_noise_b512"] = {"kernels": {"power_pipe_kernel<128,128,NORM> (final pass + the next call's statistics in its idle waves)": fin},
                               "hbm_bytes_per_launch": total(fin), "algorithmic_bytes_4N": 512 * N, "contract_bytes_12N": 3 * 512 * N,
                               "note": "a sampler's steady state: one launch per call; the first call of a sampler also runs power_stats_kernel "
                                       "(no stores: a few KiB)"}
    stats = [v for k, v in raw.items() if "power_stats_kernel" in k]
    if stats:
        out["power_noise_b512"]["kernels"]["power_stats_kernel<128,128> (first call only)"] = stats[0]
    cols, rows = pick(raw, "power_block_cols_kernel<0, true"), pick(raw, "lines_c2r_kernel<false, true")
    out["power_noise_256x256_b128"] = {"kernels": {"power_block_cols_kernel<0, STATS> (draw + filter + statistics + columns -> workspace)": cols,
                                                   "lines_c2r_kernel<NORM> (workspace -> rows, normalised)": rows},
                                       "hbm_bytes_per_launch": total(cols, rows), "algorithmic_bytes_4N": 128 * 4 * 256 * 256 * 4,
                                       "workspace_bytes": 128 * 4 * 256 * 129 * 8,
                                       "note": "128 latents of 4 x 256 x 256: the complex workspace is written and read once, the tensor written once (12.2N)"}
    sf = pick(raw, "spectral_filter128_kernel<128, 128,")  # (rounds 1-4: power_irfft2_kernel<128, 128, 2, ...>)
    out["spectral_filter_b512"] = dict(sf, algorithmic_bytes_8N=2 * 512 * N)
    # Perlin: lattice + statistics pass + final pass, run equally often at batch 512 and 64 (same kernel names)
    per = [pick(raw, "perlin_lattice_kernel"), pick(raw, "perlin_generate_kernel<1,"), pick(raw, "perlin_generate_kernel<2,")]
    both = 2 * total(*per)
    out["perlin_b512"] = {"hbm_bytes_per_launch": int(both * 512 / 576), "algorithmic_bytes_4N": 512 * N, "contract_bytes_12N": 3 * 512 * N,
                          "note": "lattice + statistics pass (no stores) + final pass; the workload ran batch 512 and batch 64 equally often, bytes split in proportion to the batch"}
    out["perlin_b64"] = {"hbm_bytes_per_launch": int(both * 64 / 576), "algorithmic_bytes_4N": 64 * N, "contract_bytes_12N": 3 * 64 * N}
    # round 4: the plane kernel is a different instantiation at batch 512 and batch 64 (non-temporal stores up to 32 MiB): no split by batch
    pyr512 = pick(raw, "pyramid_plane_kernel<true, true, 0, false, false>")["hbm_bytes_per_launch"]
    pyr64 = pick(raw, "pyramid_plane_kernel<true, true, 0, true, false>")["hbm_bytes_per_launch"]
    out["pyramid_b512"] = {"hbm_bytes_per_launch": int(pyr512) + 2 * 512 * N,
                           "kernels": {"pyramid_plane_kernel": int(pyr512), "scale_noise_kernel (in place, read + write)": 2 * 512 * N},
                           "algorithmic_bytes_12N": 3 * 512 * N,
                           "note": "generate pass with statistics (one write) + in-place scale_noise (calibrated read + write above)"}
    out["pyramid_b64"] = {"hbm_bytes_per_launch": int(pyr64) + 2 * 64 * N, "algorithmic_bytes_12N": 3 * 64 * N,
                          "note": "both launches store with the non-temporal hint at this size"}
    mom = pick(raw, "EulerOp")
    out["momentum_euler_b512"] = dict(mom, algorithmic_bytes_20N=5 * 512 * N, note="x, denoised, history in; x', history' out")
    for tag, T in (("fp64", "double"), ("fp32", "float")):
        low = pick(raw, f"wcfg_lowpass_kernel<{T},")
        out[f"wcfg_lowpass_{tag}_b256"] = dict(low, algorithmic_bytes_16N=4 * 256 * N, ratio_to_16N=round(low["hbm_bytes_per_launch"] / (4 * 256 * N), 2),
                                               note="cond and uncond are read twice (analysis, then the output phase); the second read mostly comes out of the "
                                                    "256 MB memory-side cache: removing it (profiling build) saves 8-10 us of the kernel's time")
        if T == "float":
            # round 5: fp32 arithmetic runs the single-launch kernel by default (every band resident in LDS): ONE launch for difference-only rules
            # (the AHEAD instantiation: LDS leaves it two workgroups per CU), TWO for rules that scale cond / uncond / final as well
            one = pick(raw, "wcfg_bands_kernel<float, float,", ", true>")
            two = pick(raw, "wcfg_bands_kernel<float, float,", ", false>")
            out["wcfg_bands_difference_fp32_b256"] = {"kernels": {"wcfg_bands_kernel<float, float, AHEAD> (one launch)": one},
                                                      "hbm_bytes_per_launch": total(one), "algorithmic_bytes_16N": 4 * 256 * N,
                                                      "ratio_to_16N": round(total(one) / (4 * 256 * N), 2),
                                                      "note": "cond and uncond are read twice (level 1's analysis, then the output stage), like the low-pass kernel's"}
            out["wcfg_bands_pair_fp32_b256"] = {"kernels": {"wcfg_bands_kernel<float, float> x 2 (B . DWT(uncond) into out, then A . DWT(cond))": dict(two, launches=2)},
                                                "hbm_bytes_per_launch": 2 * total(two), "algorithmic_bytes_16N": 4 * 256 * N,
                                                "ratio_to_16N": round(2 * total(two) / (4 * 256 * N), 2)}
            continue
        # fp64 arithmetic: level 1 by the tile kernels, the deeper levels in wcfg_bands_kernel<T, T> (coefficients resident in LDS) -- once for
        # difference-only rules, twice (A . DWT(cond) + B . DWT(uncond)) for rules that scale cond / uncond / final as well
        deep = pick(raw, f"wcfg_bands_kernel<{T}, {T},")
        for route, mode, ndeep in (("bands_difference", "2", 1), ("bands_pair", "1", 2)):
            ks = {"dwt2_tile_kernel (level 1 analysis)": pick(raw, f"dwt2_tile_kernel<{T}, float, {mode},"),
                  f"wcfg_bands_kernel<{T}, {T}> (levels 2..5 in LDS) x {ndeep}": dict(deep, launches=ndeep),
                  "idwt2_tile_kernel (level 1 synthesis + output)": pick(raw, f"idwt2_tile_kernel<{T}, {mode},")}
            tot = int(sum(v["hbm_bytes_per_launch"] * v.get("launches", 1) for v in ks.values()))
            out[f"wcfg_{route}_{tag}_b256"] = {"kernels": ks, "hbm_bytes_per_launch": tot, "algorithmic_bytes_16N": 4 * 256 * N,
                                               "ratio_to_16N": round(tot / (4 * 256 * N), 2)}
        one = pick(raw, f"wcfg_bands_kernel<{T}, float,")
        out[f"wcfg_single_launch_bands_{tag}_b256"] = dict(one, algorithmic_bytes_16N=4 * 256 * N, ratio_to_16N=round(one["hbm_bytes_per_launch"] / (4 * 256 * N), 2),
                                                           note="sonar_wcfg_bands_*: every coefficient band resident in LDS, one launch per difference-only rule "
                                                                "(two for cond / uncond rules; bytes per launch here); fp64's default stays the tile route -- see DESIGN.md 7")
    # round 6: the look-ahead forms of the plans (one launch per call in a sampler's steady state) and the Brownian tree call; optional rows
    # (a table made from an older workload has none of them)
    def maybe(*frags, exclude=()):
        hits = {k: v for k, v in raw.items() if all(f in k for f in frags) and not any(e in k for e in exclude)}
        return next(iter(hits.values())) if len(hits) == 1 else None

    fa = maybe("stream_fill_ahead_kernel")
    if fa:
        out["uniform_ahead_b512"] = dict(fa, algorithmic_bytes_4N=512 * N, note="normalised uniform fill inside a plan: this call's final pass + the next call's statistics pass (no stores) in one launch")
    pa = maybe("perlin_ahead_kernel")
    if pa:
        out["perlin_ahead_b512"] = dict(pa, algorithmic_bytes_4N=512 * N, note="normalised Perlin call inside a plan (fused form): final pass + next call's statistics in the same waves + a later call's lattice")
    pya = maybe("pyramid_plane_kernel<false, true, 0, false, true>")
    if pya:
        out["pyramid_ahead_b512"] = dict(pya, algorithmic_bytes_4N=512 * N, note="normalised pyramid call inside a plan: this call's planes stored normalised + the next call's planes for their statistics (no stores); the two-launch form moves 12N; the average includes the store-less launches a call without statistics left for it runs first")
    pya64 = maybe("pyramid_plane_kernel<false, true, 0, true, true>")
    if pya64:
        out["pyramid_ahead_b64"] = dict(pya64, algorithmic_bytes_4N=64 * N)
    bt = maybe("brownian_burst_kernel<0, 512>")
    if bt:
        out["brownian_tree_cfg5_shard"] = dict(bt, tensor_bytes=128 * 16 * 128 * 128 * 4, note="tree mode, two-stage evaluation: reads the two coarse-grid tensors and the kept previous point, writes W(t) and the increment; ~20 node bursts per element (compute-bound)")
    br = pick(raw, "brownian_burst_kernel<0, 256>")
    out["brownian_bridge_cfg5_shard"] = dict(br, tensor_bytes=128 * 16 * 128 * 128 * 4,
                                             note="128 x 16 x 128 x 128: reads the kept neighbour tensor(s), writes W(t) and the increment (or reads and "
                                                  "writes the chain's running sum): 3-4 tensors per call")
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
