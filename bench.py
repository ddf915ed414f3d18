#!/usr/bin/env python3
"""bench.py — headline benchmark of the Sonar hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): noise-latents/sec on SDXL 4x128x128 latents.  One "step" = one call of the normalised power-law (pink,
alpha = 1) rFFT noise sampler for a batch of 512 latents per GPU (cfg2 of BASELINE.json at the north_star's batch), through the
reference's plugin API (PowerNoiseItem.make_noise_sampler -> ns(sigma, sigma_next)), generate mode (cpu=False: spectrum drawn
in-kernel, nothing read from HBM but the 33 KB filter).  A step's launches: the pipelined final pass (draw, filter, LDS-resident C2R
FFT, normalise, ONE write), which also computes the Parseval statistics of the call the sampler expects next in its idle waves; the
statistics launch of a call only runs when that expectation failed (the first call of a sampler).  Every step therefore does one
statistics computation and one final pass, as before -- the statistics of step k are computed during step k - 1.  `--no-lookahead`
times the two-launch form (statistics launch + final pass per call); `extra.power_noise_two_launch_us` reports it either way.  N > 1: one process per GPU, every rank generates its own 512-latent
shard of one logical N*512 batch (weak scaling, no data-path collective; shard-invariant counters) — the only collectives are the
timing barrier / max, and, outside `value`, the optional final gather (RCCL all-gather vs direct peer copies).

Before the W warm-up steps the contract asks for, `--prewarm` (default 1000) untimed steps bring the GPU to its sustained clocks:
the step time falls from ~67 us in the first milliseconds of a process to 60.5 us after ~30 ms of work and stays there
(scratch/warm_curve.py); `config.prewarm_steps` records it.  `--prewarm 0` measures the cold figure.

Prints ONE JSON line (rank 0).  Keys beyond the contract:
  roofline      the dominant launch pair (statistics pass + final pass of one C-ABI call), HIP-event timed on the launch stream
                inside the timed region.  `achieved` / `frac` are at the bytes the kernels REALLY move (4N per latent: one write;
                `traffic` = the rocprofv3 PMC figure from profiles/r02_traffic.json); `frac_contract_12N` is the same time priced
                at SURVEY.md 8d's 12N (the reference-structured write + read + write).  `bound` = what the counters and the
                per-pass timings in profiles/ show limits the pair (vector-ALU issue + LDS latency), `peak` stays the HBM peak the
                fraction is taken against.  `kernels` = the other rows of the path, each event-timed here, with its own bytes.
  cpu_baseline  oracle/sonar_oracle.py (PyTorch-CPU restatement pinned to the reference) on the host cores: all cores and one
                thread, bounded samples, CPU model string (N = 1 only).
  extra         other configurations of BASELINE.json (cfg3 / cfg4 / cfg5) event-timed outside the timed region.
"""
from __future__ import annotations

import argparse
import datetime
import importlib
import json
import math
import os
import socket
import statistics
import subprocess
import sys
import threading
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BATCH = 512
C, H, W = 4, 128, 128
N_LATENT = C * H * W


def power_item(pn, factor=1.0, channels="1,1,1,1,1,1"):
    return pn.PowerNoiseItem(factor, time_brownian=False, alpha=1.0, max_freq=0.7071, min_freq=0.0, stretch=1.0, rotate=0.0, pnorm=2.0,
                             mix=1.0, common_mode=0.0, channel_correlation=channels)


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(target_s: float = 8.0):
    """Oracle (PyTorch-CPU restatement of the reference, `port`) on the host cores: the same workload at a bounded batch, with every
    host core, with 16 threads and with one thread; `value` is the best of the three (torch's CPU FFT / elementwise kernels at this
    size lose more to thread hand-offs than they gain: all cores is the slowest)."""
    from oracle import sonar_oracle as orc

    shape = (64, C, H, W)
    filt = orc.power_filter_normalize(orc.power_filter_build(shape, alpha=1.0, max_freq=0.7071), shape)

    def run(seconds):
        torch.manual_seed(0)
        orc.power_noise(orc.draw_power(shape), filt, shape, None, 1.0, True)  # warm
        t0 = time.perf_counter()
        reps = 0
        while True:
            orc.power_noise(orc.draw_power(shape), filt, shape, None, 1.0, True)
            reps += 1
            if time.perf_counter() - t0 >= seconds or reps >= 400:
                break
        return reps, time.perf_counter() - t0

    cores = torch.get_num_threads()
    rows = {}
    for label, threads, share in (("all_cores", cores, 0.5), ("threads_16", min(16, cores), 0.25), ("one_thread", 1, 0.25)):
        torch.set_num_threads(threads)
        reps, dt = run(target_s * share)
        rows[label] = {"value": reps * shape[0] / dt, "threads": threads, "calls": reps, "seconds": round(dt, 1)}
    torch.set_num_threads(cores)
    best = max(rows.values(), key=lambda r: r["value"])
    return {"value": best["value"], "unit": "latents/s", "cores": best["threads"], "kind": "port", "cpu_model": cpu_model(), **rows,
            "sample": " + ".join(f"{r['calls']} calls ({k}, {r['seconds']} s)" for k, r in rows.items())
                      + f" of normalised power-law noise at batch {shape[0]} (SDXL 4x128x128), oracle/sonar_oracle.py"}


def event_us(fn, steps=20, warmup=5):
    """Average GPU time of fn() in microseconds: HIP events on torch's current stream (the stream every sonar_* call launches on)."""
    for _ in range(warmup):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


BURSTS = {}  # row tag -> {"host_us": {min, median, max}, "gpu_us": {...}, "bursts": n} of the launch-bound rows (extra.bursts)


def host_and_event_us(fn, steps=200, warmup=100, burst=25, tag=None):
    """(host issue time per call, GPU span per call) in microseconds: what a launch-bound step costs on either side.  Issued in bursts of
    `burst` calls with a synchronisation between them: a host that issues a step in 10 us runs hundreds of steps ahead of a GPU that needs
    30, and once the runtime's command queue is full every further launch blocks in the driver for far longer than either figure (seen
    as 200+ us per call on both clocks).  A sampler never queues more than a few noise calls ahead.
    The figures are the MEDIAN over the bursts (round 5): one burst in a few hundred meets a 50-90 ms freeze of the whole process that is
    not the kernels' (the container's CPU quota throttling torch's CPU thread pool after a sampler is built), and a mean
    over eight bursts then reads 400 us per call.  `BURSTS[tag]` keeps min / median / max of both clocks."""
    for _ in range(warmup):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    hosts, gpus = [], []
    done = 0
    while done < steps:
        n = min(burst, steps - done)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(n):
            fn()
        hosts.append((time.perf_counter() - t0) / n * 1e6)
        e1.record()
        torch.cuda.synchronize()
        gpus.append(e0.elapsed_time(e1) / n * 1e3)
        done += n
    if tag is not None:
        BURSTS[tag] = {"bursts": len(hosts), "calls_per_burst": burst,
                       "host_us": {"min": min(hosts), "median": statistics.median(hosts), "mean": statistics.fmean(hosts), "max": max(hosts)},
                       "gpu_us": {"min": min(gpus), "median": statistics.median(gpus), "mean": statistics.fmean(gpus), "max": max(gpus)}}
    return statistics.median(hosts), statistics.median(gpus)


def traffic_table() -> dict:
    """The newest tracked PMC traffic table that parses (an annotation: a missing or damaged file must not cost the run its line)."""
    for r in (5, 4, 3):
        path = os.path.join(ROOT, "profiles", f"r0{r}_traffic.json")
        try:
            with open(path) as fh:
                table = json.load(fh)
            if isinstance(table, dict) and table:
                return table
        except (OSError, ValueError):
            continue
    return {}


VALU_PMC_FILE = "profiles/r05_pmc_issue_a.txt"


def valu_active_per_launch():
    """SQ_ACTIVE_INST_VALU of the headline kernel per dispatch (rocprofv3 --pmc pass of the round's profile set, VALU_PMC_FILE):
    vector-ALU busy cycles summed over the chip's 1024 SIMDs; None when the profile file is absent.  A DERIVED figure in the bench
    line: the counter comes from that tracked file, not from this run."""
    path = os.path.join(ROOT, VALU_PMC_FILE)
    try:
        inside = False
        with open(path) as fh:
            for line in fh:
                if "power_pipe_kernel" in line:
                    inside = True
                elif inside and "SQ_ACTIVE_INST_VALU" in line:
                    return float(line.split()[-1])
                elif inside and line.startswith("void "):
                    inside = False
    except (OSError, ValueError, IndexError):
        pass
    return None


def kernel_entry(name, us, bytes_per_launch, traffic, note=None):
    gbps = bytes_per_launch / (us * 1e-6) / 1e9
    e = {"kernel": name, "avg_launch_us": us, "bytes_per_launch": bytes_per_launch, "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
         "frac": gbps / HBM_PEAK_GBPS, "traffic": traffic}
    if note:
        e["note"] = note
    return e


def secondary_rows(device, hl, pn, ng, nz, x, sig):
    """Other rows of the path and the other BASELINE configurations, event-timed (N = 1 only, outside `value`)."""
    tr = traffic_table()
    kernels, extra = [], {}
    # achievable HBM bandwidth on this box (SURVEY 8d asks for it beside the 8 TB/s spec): device-to-device copy of 1 GiB
    big_a = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=device)
    big_b = torch.empty_like(big_a)
    extra["hbm_copy_GBps_read_plus_write"] = 2 * big_a.numel() * 4 / (event_us(lambda: big_b.copy_(big_a), 10, 3) * 1e-6) / 1e9
    del big_a, big_b
    # cfg3: Perlin and pyramid, normalised, generate mode, at the north_star batch (512) and at the configured batch (64)
    x64 = torch.zeros((64, C, H, W), device=device)
    for name in ("perlin", "pyramid"):
        for tag, xb in (("b512", x), ("b64", x64)):
            ns = nz.get_noise_sampler(name, xb, 0.03, 14.6, seed=None, cpu=False, normalized=True)
            us = event_us(lambda: ns(*sig))
            b = xb.shape[0]
            extra[f"{name}_{tag}_latents_per_s"] = b / (us * 1e-6)
            real = (4 if name == "perlin" else 12) * N_LATENT * b  # Perlin is written once; pyramid generates, then scales in place
            kernels.append(kernel_entry(f"{name} normalised generate, batch {b}", us, real, tr.get(f"{name}_{tag}", {}).get("hbm_bytes_per_launch"),
                                        "4N real (12N contract)" if name == "perlin" else "12N real = contract"))
    # rows G1 / G2 normalised at the north_star batch: the fill look-ahead of round 6 (a plan runs the next call's statistics pass in the
    # storing waves; N(0,1) with factor 1 keeps its one-pass route at this size)
    for name in ("uniform", "gaussian"):
        ns = nz.get_noise_sampler(name, x, 0.03, 14.6, seed=None, cpu=False, normalized=True)
        us = event_us(lambda: ns(*sig))
        extra[f"{name}_b512_latents_per_s"] = BATCH / (us * 1e-6)
        kernels.append(kernel_entry(f"{name} normalised generate, batch {BATCH}", us, 4 * N_LATENT * BATCH, None, "4N real (12N contract): one write"))
    # cfg3 as a chain: Perlin + pyramid items of one CustomNoiseChain, normalised (the Perlin item is evaluated inside the pyramid
    # kernel: lattice + ONE generating launch + the normalisation pass)
    for tag, xb in (("b512", x), ("b64", x64)):
        chain3 = nz.CustomNoiseChain()
        chain3.add(nz.CustomNoiseItem(0.5, noise_type="perlin"))
        chain3.add(nz.CustomNoiseItem(0.5, noise_type="pyramid"))
        ns3 = chain3.make_noise_sampler(xb, 0.03, 14.6, seed=None, cpu=False, normalized=True)
        extra[f"cfg3_chain_{tag}_latents_per_s"] = xb.shape[0] / (event_us(lambda: ns3(*sig), 30, 10) * 1e-6)
    # the configurations as BASELINE.json writes them and as a ComfyUI run calls them: cfg2 at 1 and 4 latents, cfg3's chain at 4 --
    # launch-bound steps, so the host's issue time per call stands beside the GPU's (prepared plans: one foreign call per step)
    for tag, bsz in (("b1", 1), ("b4", 4)):
        xs_ = torch.zeros((bsz, C, H, W), device=device)
        item = power_item(pn)
        ns_small = item.make_noise_sampler(xs_, None, None, seed=None, cpu=False, normalized=True)
        host, gpu = host_and_event_us(lambda: ns_small(*sig), tag=f"power_noise_{tag}")
        extra[f"power_noise_{tag}_us"] = gpu
        extra[f"power_noise_{tag}_host_us_per_call"] = host
    # the same single-latent call on latents off the 128 x 128 path: an SDXL portrait bucket (general-size kernels; the next call's statistics
    # ride in the launch) and a 2048 px latent (column blocks through a workspace: two launches)
    for tag, (hh, ww) in (("104x152", (104, 152)), ("112x144", (112, 144)), ("96x168", (96, 168)), ("256x256", (256, 256))):
        xs_ = torch.zeros((1, C, hh, ww), device=device)
        ns_off = power_item(pn).make_noise_sampler(xs_, None, None, seed=None, cpu=False, normalized=True)
        host, gpu = host_and_event_us(lambda: ns_off(*sig), tag=f"power_noise_{tag}_b1")
        extra[f"power_noise_{tag}_b1_us"] = gpu
        extra[f"power_noise_{tag}_b1_host_us_per_call"] = host
    for tag, xb in (("b4", torch.zeros((4, C, H, W), device=device)), ("b64", x64)):
        chain3 = nz.CustomNoiseChain()
        chain3.add(nz.CustomNoiseItem(0.5, noise_type="perlin"))
        chain3.add(nz.CustomNoiseItem(0.5, noise_type="pyramid"))
        ns3 = chain3.make_noise_sampler(xb, 0.03, 14.6, seed=None, cpu=False, normalized=True)
        host, gpu = host_and_event_us(lambda: ns3(*sig), tag=f"cfg3_chain_{tag}")
        extra[f"cfg3_chain_{tag}_us"] = gpu
        extra[f"cfg3_chain_{tag}_host_us_per_call"] = host
    for name in ("perlin", "pyramid"):
        ns1 = nz.get_noise_sampler(name, x64, 0.03, 14.6, seed=None, cpu=False, normalized=True)
        host, gpu = host_and_event_us(lambda: ns1(*sig), tag=f"{name}_b64")
        extra[f"{name}_b64_us"] = gpu
        extra[f"{name}_b64_host_us_per_call"] = host
    # momentum step (row M): 3 reads + 2 writes
    sonar = importlib.import_module("comfyui_sonar_amd.py.sonar")
    sb = sonar.SonarBase(sonar.SonarBase.get_config(None, {}))
    den, xs = torch.randn_like(x), torch.randn_like(x)
    sb.momentum_step(0, xs, den, torch.tensor(10.0), torch.tensor(8.0))
    us = event_us(lambda: sb.momentum_step(1, xs, den, torch.tensor(8.0), torch.tensor(6.0)))
    extra["momentum_euler_latent_steps_per_s"] = BATCH / (us * 1e-6)
    kernels.append(kernel_entry("ew_kernel<EulerOp> fused momentum step, batch 512", us, 20 * N_LATENT * BATCH,
                                tr.get("momentum_euler_b512", {}).get("hbm_bytes_per_launch")))
    # spectral filter (PowerFilterNoiseItem / ffilter): one read, one write
    filt = torch.rand(H, W // 2 + 1, device=device) + 0.5
    us = event_us(lambda: hl.spectral_filter(xs, filt))
    extra["spectral_filter_latents_per_s"] = BATCH / (us * 1e-6)
    kernels.append(kernel_entry("spectral_filter128_kernel (rfft2 x filter, irfft2 in LDS), batch 512", us, 8 * N_LATENT * BATCH,
                                tr.get("spectral_filter_b512", {}).get("hbm_bytes_per_launch")))
    # the same call on planes off the 128 x 128 path, 33.5 M values each: SD 1.5 latents (the fixed-size kernels' general passes), an
    # SDXL portrait bucket (general-size kernels: codelets 13 x 8 and 19 x 4), 2048 px (beyond LDS: drawn and column-transformed in blocks of
    # columns into a complex workspace, rows out of it -- sonar_power_block_f32)
    for tag, (hh, ww, nb) in {"64x64": (64, 64, 2048), "104x152": (104, 152, 530), "112x144": (112, 144, 520), "96x168": (96, 168, 520),
                              "256x256": (256, 256, 128)}.items():
        try:
            fz = torch.rand(hh, ww // 2 + 1, device=device) + 0.5
            shp, ctr = (nb, C, hh, ww), [0]

            def sized_call():
                ctr[0] += 1
                return hl.power_noise(fz, shp, seed=11, stream_id=ctr[0], plane_offset=0, factor=1.0)

            us = event_us(sized_call, 20, 5)
            extra[f"power_noise_{tag}_us"] = us
            # the same call through the sampler API (a prepared plan issues its launches with one foreign call: no host gap between them)
            xs_sz = torch.zeros(shp, device=device)
            ns_sz = power_item(pn).make_noise_sampler(xs_sz, None, None, seed=None, cpu=False, normalized=True)
            extra[f"power_noise_{tag}_sampler_us"] = event_us(lambda: ns_sz(*sig), 20, 8)
            del xs_sz, ns_sz
            kernels.append(kernel_entry(f"power noise, normalised generate, {nb} latents of {C} x {hh} x {ww}", us, 4 * nb * C * hh * ww,
                                        tr.get(f"power_noise_{tag}_b{nb}", {}).get("hbm_bytes_per_launch"),
                                        "two launches (statistics + final pass; 256 x 256: statistics + columns into a workspace + rows, 3 x the tensor of traffic); bytes = the tensor written once"
                                        + ("; SDXL bucket: the general-size kernel with compile-time factor pairs (power_buckets_*.hip)" if (hh, ww) in ((104, 152), (112, 144), (96, 168)) else "")))
        except Exception as exc:  # secondary figure only
            extra[f"power_noise_{tag}_error"] = repr(exc)[:200]
    # brownian (cfg5's third source).  Default since round 5: the virtual Brownian tree (W(t) a function of (seed, t) alone, up to 25 normals
    # per element and evaluation); the rows below keep measuring the path of bridges (SONAR_BROWNIAN_TREE=0: one new path point per call,
    # bridged between the kept tensors of its neighbours -- rounds 2-4's default), the tree's rows follow cfg5's
    ng = importlib.import_module("comfyui_sonar_amd.py.noise_generation")
    tree_default = ng.BROWNIAN_TREE_DEPTH
    ng.BROWNIAN_TREE_DEPTH = 0
    ns_b = nz.get_noise_sampler("brownian", x64, 0.03, 14.6, seed=7, cpu=False, normalized=False)
    sched = torch.linspace(14.6, 0.03, 41).tolist()
    pos = [0]

    def brownian_step():
        i = pos[0] % 40
        pos[0] += 1
        return ns_b(torch.tensor(sched[i]), torch.tensor(sched[i + 1]))

    extra["brownian_b64_latents_per_s"] = 64 / (event_us(brownian_step, 30, 5) * 1e-6)
    # the same on cfg5's shard (128 Flux latents = one 134 MB tensor): a DPM++ SDE run's (t, s), (t, t') queries -> 3 or 4 tensors cross HBM
    try:
        xs5 = torch.zeros((128, 16, H, W), device=device)
        ns_s = nz.get_noise_sampler("brownian", xs5, 0.03, 14.6, seed=7, cpu=False, normalized=False)
        sch = torch.linspace(14.6, 0.5, 21).tolist()
        calls = [pair for k in range(20) for pair in ((sch[k], math.sqrt(sch[k] * sch[k + 1])), (sch[k], sch[k + 1]))]
        at = [0]

        def shard_step():
            a, b = calls[at[0] % len(calls)]
            at[0] += 1
            return ns_s(torch.tensor(a), torch.tensor(b))

        us = event_us(shard_step, 30, 4)
        extra["brownian_cfg5_shard_us_per_call"] = us
        kernels.append(kernel_entry("brownian_burst_kernel, bridge route (SONAR_BROWNIAN_TREE=0), cfg5 shard (128 x 16 x 128 x 128)", us, int(3.5 * xs5.numel() * 4),
                                    tr.get("brownian_bridge_cfg5_shard", {}).get("hbm_bytes_per_launch"),
                                    "reads the kept neighbour tensor(s), writes W(t) and the increment: 3 or 4 tensors per call, 3.5 on average"))
        del xs5, ns_s
    except Exception as exc:  # secondary figure only
        extra["brownian_cfg5_shard_error"] = repr(exc)[:200]
    finally:
        ng.BROWNIAN_TREE_DEPTH = tree_default
    # cfg4: WaveletCFG db4 / level 5 / symmetric, fp32 I/O, 256 latents (cond, uncond, x -> out: 16N bytes per latent)
    wc = importlib.import_module("comfyui_sonar_amd.py.wavelet_cfg")
    b4 = 256
    ms = types.SimpleNamespace(sigma_min=torch.tensor(0.03), sigma_max=torch.tensor(14.6), timestep=lambda sg: (
        999 * (1 - (sg.log() - math.log(0.03)) / (math.log(14.6) - math.log(0.03)))).clamp(0, 999))
    cond, uncond, xin = (torch.randn(b4, C, H, W, device=device) for _ in range(3))
    wargs = {"cond_denoised": cond, "uncond_denoised": uncond, "cond": xin - cond, "uncond": xin - uncond, "input": xin, "cond_scale": 7.0,
             "sigma": torch.full((b4,), 7.0, device=device), "model": types.SimpleNamespace(model_sampling=ms),
             "model_options": {"transformer_options": {"sample_sigmas": torch.cat([torch.linspace(14.6, 0.03, 20), torch.zeros(1)])}}}
    for tag, hp in (("fp64", True), ("fp32", False)):  # the node's placeholder rule: db4, level 5, symmetric, difference scales 5 / 3
        cfg_fn = wc.WaveletCFG(existing_cfg=None, rules=wc.WCFGRules.build(difference=dict(yl_scale=5.0, yh_scales=3.0), high_precision_mode=hp))
        try:
            us = event_us(lambda: cfg_fn(wargs), 10, 3)
            extra[f"wavelet_cfg_{tag}_end_to_end_us"] = us
            extra[f"wavelet_cfg_{tag}_latents_per_s"] = b4 / (us * 1e-6)
            w = cfg_fn.rules[0].make_wavelet()
            kus = event_us(lambda: hl.wcfg_lowpass(cond, uncond, xin, levels=5, dec_lo=w.dec_lo, rec_lo=w.rec_lo, mode="symmetric", inv_mode="symmetric",
                                                   g=[3.0, 0.0, 0.0, 0.0, 0.0, 2.0], ku=1.0, kt=1.0, subtract_from_x=True, high_precision=hp), 10, 3)
            extra[f"wavelet_cfg_{tag}_kernel_us"] = kus
            kernels.append(kernel_entry(f"wcfg_lowpass_kernel<{'double' if hp else 'float'}, 8> WaveletCFG placeholder rule, batch 256", kus,
                                        16 * N_LATENT * b4, tr.get(f"wcfg_lowpass_{tag}_b256", {}).get("hbm_bytes_per_launch"),
                                        "end to end adds the reference's sigma.max().item() sync and the host rule logic"))
        except Exception as exc:  # secondary figure only; the headline must still print
            extra[f"wavelet_cfg_{tag}_error"] = repr(exc)[:200]
    # the rules the low-pass kernel does not take: per-orientation difference scales (one tensor, cond - uncond) and cond / uncond scales
    # beside the difference (both tensors transformed): level 1 by the tile kernels (its bands cross HBM), levels 2-5 by the LDS-resident
    # band kernel, launched before the sigma value is known (round 4)
    band_rules = {"bands_difference": dict(difference=dict(yl_scale=5.0, yh_scales=[[3.0, 2.5, 2.0]] * 5)),
                  "bands_pair": dict(cond=dict(yl_scale=1.1, yh_scales=1.0), uncond=dict(yl_scale=1.0, yh_scales=0.9),
                                     difference=dict(yl_scale=5.0, yh_scales=3.0))}
    for rtag, params in band_rules.items():
        for tag, hp in (("fp64", True), ("fp32", False)):
            try:
                cfg_fn = wc.WaveletCFG(existing_cfg=None, rules=wc.WCFGRules.build(**params, high_precision_mode=hp))
                us = event_us(lambda: cfg_fn(wargs), 10, 3)
                extra[f"wavelet_cfg_{rtag}_{tag}_end_to_end_us"] = us
                single = tag == "fp32"  # WaveletCFG.single_launch_bands = None: by precision (py/wavelet_cfg.py)
                route = ("wcfg_bands_kernel, every band resident in LDS (one launch; two for the cond / uncond rule)" if single else
                         "dwt2_tile + wcfg_bands (levels 2-5 resident in LDS) + idwt2_tile kernels")
                kernels.append(kernel_entry(f"{route}, WaveletCFG {rtag.replace('_', ' ')} rule, {tag}, batch 256 (end to end, the default route)",
                                            us, 16 * N_LATENT * b4, tr.get(f"wcfg_{rtag}_{tag}_b256", {}).get("hbm_bytes_per_launch")))
                # both routes forced, whichever is the default: every band resident in LDS (sonar_wcfg_bands_*: least traffic, one or two
                # workgroups per CU) and level 1 through the tile kernels
                for forced, key in ((True, "single_launch_us"), (False, "tile_route_us")):
                    wc.WaveletCFG.single_launch_bands = forced
                    try:
                        extra[f"wavelet_cfg_{rtag}_{tag}_{key}"] = event_us(lambda: cfg_fn(wargs), 10, 3)
                    finally:
                        wc.WaveletCFG.single_launch_bands = None
            except Exception as exc:  # secondary figure only
                extra[f"wavelet_cfg_{rtag}_{tag}_error"] = repr(exc)[:200]
    # cfg5: one rank's shard (128 Flux latents), scheduled power + Perlin + Brownian chain, SonarDPMPPSDE with momentum, per step
    try:
        extra["cfg5_step_ms"] = cfg5_shard_step_ms(device, hl, pn, nz, sonar)
        extra["cfg5_latent_steps_per_s"] = 128 / (extra["cfg5_step_ms"] * 1e-3)
    except Exception as exc:
        extra["cfg5_error"] = repr(exc)[:200]
    # the same step on the path of bridges (SONAR_BROWNIAN_TREE=0, rounds 2-4's default: one normal per element and Brownian call, but a
    # function of the query history), and the tree's noise call alone on the shard
    ng = importlib.import_module("comfyui_sonar_amd.py.noise_generation")
    depth0 = ng.BROWNIAN_TREE_DEPTH
    extra["brownian_tree_depth"] = depth0
    try:
        ng.BROWNIAN_TREE_DEPTH = 0
        extra["cfg5_step_brownian_bridges_ms"] = cfg5_shard_step_ms(device, hl, pn, nz, sonar)
        ng.BROWNIAN_TREE_DEPTH = depth0 or 24
        xs5 = torch.zeros((128, 16, H, W), device=device)
        ns_t = nz.get_noise_sampler("brownian", xs5, 0.03, 14.6, seed=7, cpu=False, normalized=False)
        sch = torch.linspace(14.6, 0.5, 21).tolist()
        calls = [pair for k in range(20) for pair in ((sch[k], math.sqrt(sch[k] * sch[k + 1])), (sch[k], sch[k + 1]))]
        at = [0]

        def tree_step():
            a, b = calls[at[0] % len(calls)]
            at[0] += 1
            return ns_t(torch.tensor(a), torch.tensor(b))

        extra["brownian_tree_cfg5_shard_us_per_call"] = event_us(tree_step, 20, 4)
        del xs5, ns_t
    except Exception as exc:  # secondary figure only
        extra["cfg5_brownian_error"] = repr(exc)[:200]
    finally:
        ng.BROWNIAN_TREE_DEPTH = depth0
    return kernels, extra


def cfg5_shard_step_ms(device, hl, pn, nz, sonar):
    """cfg5 on one rank's shard (128 Flux latents): scheduled power + Perlin + Brownian chain, SonarDPMPPSDE with momentum; ms per sampler step
    (5 steps per run: 10 noise calls, 10 stand-in model evaluations)."""
    xf = torch.randn(128, 16, H, W, device=device) * 10.0
    inner = nz.CustomNoiseChain()
    inner.add(power_item(pn, 0.5, "1"))
    inner.add(nz.CustomNoiseItem(0.3, noise_type="perlin"))
    inner.add(nz.CustomNoiseItem(0.2, noise_type="brownian"))
    fallback = nz.CustomNoiseChain()
    fallback.add(nz.CustomNoiseItem(1.0, noise_type="gaussian"))
    chain = nz.CustomNoiseChain()
    chain.add(nz.ScheduledNoise(1.0, noise=inner, start_sigma=20.0, end_sigma=0.0, normalize=None, fallback_noise=fallback))
    sigmas = torch.cat([torch.linspace(14.6, 0.5, 11), torch.zeros(1)])
    ns5 = chain.make_noise_sampler(xf, 0.5, 14.6, seed=3, cpu=False, normalized=True)

    def run5():
        return sonar.SonarDPMPPSDE.sampler(lambda t, sigma, **_k: hl.mul_scalar(t, 0.5), xf, sigmas[:6], {"seed": 3}, None, True, None, dict(momentum=0.95),
                                           1.0, 1.0, ns5)

    return event_us(run5, 3, 1) / 5 / 1e3


def cfg3_chain_us(device, nz, batch, sig):
    """(host issue time, GPU span) per call of cfg3's chain -- Perlin + pyramid items of one CustomNoiseChain, normalised -- at `batch` latents."""
    xb = torch.zeros((batch, C, H, W), device=device)
    chain3 = nz.CustomNoiseChain()
    chain3.add(nz.CustomNoiseItem(0.5, noise_type="perlin"))
    chain3.add(nz.CustomNoiseItem(0.5, noise_type="pyramid"))
    ns3 = chain3.make_noise_sampler(xb, 0.03, 14.6, seed=None, cpu=False, normalized=True)
    return host_and_event_us(lambda: ns3(*sig))


def gpus_on_this_node() -> int:
    """GPUs of this node counted WITHOUT a HIP / HSA call (the launcher role must not initialise the GPU: it starts other programs): KFD's
    topology lists every compute agent, GPUs are the nodes with SIMDs; the visibility variables narrow the count like the runtime would."""
    root = "/sys/class/kfd/kfd/topology/nodes"
    count = 0
    try:
        for node in os.listdir(root):
            try:
                with open(os.path.join(root, node, "properties")) as fh:
                    props = dict(line.split(None, 1) for line in fh if line.strip())
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                count += 1
    except OSError:
        return 0
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        val = os.environ.get(var)
        if val is not None:
            count = min(count, len([v for v in val.split(",") if v.strip()]))
    return count


def device_identity(index: int) -> dict:
    """What tells two ranks' GPUs apart in the result line: PCI address, UUID, architecture."""
    props = torch.cuda.get_device_properties(index)
    ident = {"device_name": props.name, "gcn_arch": getattr(props, "gcnArchName", None)}
    if all(hasattr(props, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        ident["pci_bus_id"] = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
    if getattr(props, "uuid", None) is not None:
        ident["uuid"] = str(props.uuid)
    return ident


def _free_port() -> int:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n: int) -> None:
    """`python bench.py --gpus N` without a launcher around it: start N fresh rank processes (torch.distributed.run, one per GPU)
    and relay rank 0's JSON line and the launcher's exit code.  This process never touches the GPU -- no HIP / HSA call, no library load;
    the GPUs are counted from KFD's sysfs topology -- so starting other programs from it is safe on this pool."""
    have = gpus_on_this_node()
    if have < n and os.environ.get("SONAR_BENCH_BACKEND", "nccl") == "nccl":
        print(f"[bench] --gpus {n} needs {n} GPUs on this node, {have} visible: not measuring fewer GPUs under that label",
              file=sys.stderr, flush=True)
        sys.exit(2)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in res.stdout.splitlines():
        if ln.startswith('{"metric"'):
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    if res.returncode == 0 and line is None:
        print("[bench] the ranks exited cleanly but printed no result line", file=sys.stderr, flush=True)
        sys.exit(3)
    sys.exit(res.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--prewarm", type=int, default=1000,
                    help="untimed steps before the W warm-up steps: the GPU reaches its sustained clocks after ~30 ms of work "
                         "(scratch/warm_curve.py: 67 us per step cold, 60.5 us from step ~400 on)")
    ap.add_argument("--no-lookahead", action="store_true",
                    help="every call launches its own statistics pass (the sampler is given no PowerLookahead)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    args = ap.parse_args()

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)  # does not return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("SONAR_BENCH_BACKEND", "nccl")  # "gloo": self-test of the N > 1 path with ranks sharing a GPU
    share_gpu = backend == "gloo" and torch.cuda.device_count() < world
    device_index = 0 if share_gpu else local_rank
    distributed = world > 1 or bool(os.environ.get("SONAR_BENCH_FORCE_DIST"))  # the env knob runs the RCCL path with one rank (self-test)
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(device_index)
        # RCCL prints a version banner on stdout when the communicator is created: keep stdout for the one JSON line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            # a collective that hangs ends the process after two minutes instead of holding the node
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", device_index), timeout=datetime.timedelta(seconds=120))
            else:
                dist.init_process_group(backend, timeout=datetime.timedelta(seconds=120))
            dist.barrier()
        finally:
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    # the number of GPUs reported is the number of ranks the process group really has, never the flag
    n_gpus = dist.get_world_size() if distributed else 1
    if args.gpus != n_gpus:
        if rank == 0:
            print(f"[bench] --gpus {args.gpus} but the process group has {n_gpus} rank(s): refusing to report a mislabelled figure",
                  file=sys.stderr, flush=True)
        if distributed:
            dist.destroy_process_group()
        sys.exit(2)
    device = torch.device("cuda", device_index)

    import sonar_pkg

    pkg = sonar_pkg.load()
    hl = pkg.hip_lib
    hl.load()
    pn = importlib.import_module("comfyui_sonar_amd.py.nodes.powernoise")
    ng = importlib.import_module("comfyui_sonar_amd.py.noise_generation")
    nz = importlib.import_module("comfyui_sonar_amd.py.noise")

    torch.manual_seed(0)
    x = torch.zeros((BATCH, C, H, W), device=device)
    sig = (torch.tensor(14.6), torch.tensor(10.0))
    out = None
    with ng.shard_offset(rank * BATCH):  # this rank's slice of the logical N*512 batch
        def make_sampler(lookahead):
            item = power_item(pn)
            item.stats_lookahead = lookahead
            return item.make_noise_sampler(x, None, None, seed=None, cpu=False, normalized=True)

        ns = make_sampler(not args.no_lookahead)

        def step():
            return ns(*sig)

        # HIP events on the launch stream (torch's current stream is the stream every sonar_* call launches on) bracket the
        # timed region: average launch-pair duration = event span / steps.  (Per-call event pairs perturb the pipeline:
        # they add ~10 us of idle time per step.)  The events are created and recorded once before the region: the first
        # record of an event allocates it.
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        ev1.record()
        # the cold figure (what `--prewarm 0` measures): W + K steps from an idle GPU, before the clock ramp below
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(args.steps):
            step()
        ev1.record()
        torch.cuda.synchronize()
        cold_us = ev0.elapsed_time(ev1) / args.steps * 1e3
        for _ in range(max(args.prewarm, 0)):  # clock ramp, outside the W + K steps of the contract
            step()
        torch.cuda.synchronize()
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(args.steps):
            step()
        ev1.record()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        span_ms = ev0.elapsed_time(ev1)
        pair_us_local = span_ms / args.steps * 1e3
        ranks = None
        if distributed:
            dist.barrier()
            coll_dev = device if backend == "nccl" else "cpu"
            t = torch.tensor([elapsed, pair_us_local], device=coll_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the slowest rank's wall clock and launch-pair time
            elapsed, pair_us_max = t[0].item(), t[1].item()
            ranks = [None] * n_gpus
            dist.all_gather_object(ranks, {"rank": rank, "device": f"cuda:{device_index}", **device_identity(device_index),
                                           "shard_start": rank * BATCH, "shard_count": BATCH, "pair_us": pair_us_local})
        else:
            pair_us_max = pair_us_local

        if rank == 0:
            value = n_gpus * BATCH * args.steps / elapsed
            pair_us = pair_us_max  # N > 1: the slowest rank's launch pair; every rank runs the same pair on its own shard
            peak = HBM_PEAK_GBPS * n_gpus  # the job's roofline: N x 8 TB/s
            # one call = the final pass (draw, filter, LDS-resident C2R FFT, normalise, ONE write) + one statistics computation (re-draw
            # of the radius words, Parseval, no stores: in the final pass's idle waves, or its own launch): 4N bytes per latent really
            # cross HBM (profiles/r05_traffic.json)
            real_bytes = 4 * N_LATENT * BATCH * n_gpus  # all ranks' launches together
            contract_bytes = 12 * N_LATENT * BATCH * n_gpus
            achieved = real_bytes / (pair_us * 1e-6) / 1e9
            tr = traffic_table()
            step_s = elapsed / args.steps
            out = {
                "metric": "noise-latents/sec (SDXL 4x128x128)", "value": value, "unit": "latents/s", "n_gpus": n_gpus,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_s * 1e3, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "cfg2: power-law (pink, alpha=1) rFFT noise, normalised, SDXL 4x128x128, batch 512 per GPU, "
                                       "generate mode (in-kernel Philox-seeded multiply-with-carry streams)", "batch_per_gpu": BATCH, "global_batch": BATCH * n_gpus, "prewarm_steps": max(args.prewarm, 0),
                           "stats_lookahead": not args.no_lookahead,
                           "parallelism": f"batch-shard x{n_gpus}"},
                "roofline": {"bound": "valu", "limiter": "vector-ALU issue + LDS / barrier latency of two 8-wave teams per CU (phase timeline and counters: "
                                                         "profiles/r05_power_kernel.md); HBM moves 4N per latent and would allow ~21 us per launch",
                             "kernel": "power_pipe_kernel<128,128,NORM> with the next call's statistics in its idle waves (one C-ABI call, "
                                       "sonar_power_noise_ahead_f32); --no-lookahead: power_stats_kernel<128,128> + power_pipe_kernel",
                             "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
                             "frac_valu": None if valu_active_per_launch() is None else 4.0 * valu_active_per_launch() / 1024 / (pair_us * 2400.0),
                             "frac_valu_derived": {"derived": True, "counter": "SQ_ACTIVE_INST_VALU per dispatch", "source": VALU_PMC_FILE,
                                                   "assumed_clock_ghz": 2.4, "measured_in_this_run": False},
                             "frac_valu_note": "what `bound` names: vector-ALU busy cycles per SIMD (4 x SQ_ACTIVE_INST_VALU -- the counter ticks "
                                               "once per 4-cycle issue -- / 1024 SIMDs) over this run's launch time at an assumed 2.4 GHz engine "
                                               "clock; the counter is read from the tracked profile file, not measured by this run",
                             "traffic": tr.get("power_noise_b512", {}).get("hbm_bytes_per_launch"), "bytes_per_launch": real_bytes,
                             "avg_launch_us": pair_us, "achieved_contract_12N": contract_bytes / (pair_us * 1e-6) / 1e9,
                             "frac_contract_12N": contract_bytes / (pair_us * 1e-6) / 1e9 / peak,
                             "note": "achieved / frac: the 4N bytes the launch pair really moves (one write of the tensor; statistics come from "
                                     "the spectrum by Parseval); *_contract_12N: the same time at SURVEY 8d's 12N for the reference-structured "
                                     "write + read + write"},
                "path": {"step_GBps_real_4N": real_bytes / step_s / 1e9, "step_GBps_at_12N": contract_bytes / step_s / 1e9,
                         "host_us_per_step_beyond_gpu": max(0.0, step_s * 1e6 - pair_us)},
            }
    if rank == 0 and n_gpus == 1 and not args.no_extra:
        kernels, extra = secondary_rows(device, hl, pn, ng, nz, x, sig)
        ns_two = make_sampler(False)
        extra["power_noise_cold_us"] = cold_us
        extra["power_noise_two_launch_us"] = event_us(lambda: ns_two(*sig), 100, 300)
        ns_one = make_sampler(True)
        extra["power_noise_lookahead_us"] = event_us(lambda: ns_one(*sig), 100, 300)
        copy_gbps = extra.get("hbm_copy_GBps_read_plus_write")
        if copy_gbps:
            for k in kernels:
                k["frac_of_copy"] = k["achieved"] / copy_gbps  # against what a device-to-device copy reaches on this box (read + write)
            out["roofline"]["frac_of_copy"] = out["roofline"]["achieved"] / copy_gbps
        extra["bursts"] = BURSTS
        out["roofline"]["kernels"] = kernels
        out["extra"] = extra
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if distributed and not args.no_extra:
        # N > 1: the two configurations BASELINE.json defines beside the headline, on EVERY rank's own shard -- cfg5 is an 8-GPU job of 128
        # Flux latents per rank, cfg3's chain runs at its configured 64 -- with the slowest and fastest rank in the line
        sonar_mod = importlib.import_module("comfyui_sonar_amd.py.sonar")
        mine = {"rank": rank}
        with ng.shard_offset(rank * 128):
            try:
                mine["cfg5_step_ms"] = cfg5_shard_step_ms(device, hl, pn, nz, sonar_mod)
                # the same step on the path of bridges: the difference is what the seed-determined Brownian tree costs on this rank
                depth0 = ng.BROWNIAN_TREE_DEPTH
                try:
                    ng.BROWNIAN_TREE_DEPTH = 0
                    mine["cfg5_step_brownian_bridges_ms"] = cfg5_shard_step_ms(device, hl, pn, nz, sonar_mod)
                finally:
                    ng.BROWNIAN_TREE_DEPTH = depth0
                mine["brownian_tree_depth"] = depth0
            except Exception as exc:  # secondary figure only
                mine["cfg5_error"] = repr(exc)[:200]
        with ng.shard_offset(rank * 64):
            try:
                mine["cfg3_chain_b64_host_us"], mine["cfg3_chain_b64_us"] = cfg3_chain_us(device, nz, 64, sig)
            except Exception as exc:
                mine["cfg3_error"] = repr(exc)[:200]
        per_rank = [None] * n_gpus
        dist.all_gather_object(per_rank, mine)
        if rank == 0:
            def spread(key):
                vals = [r[key] for r in per_rank if key in r]
                return {"max": max(vals), "min": min(vals), "ranks": len(vals)} if vals else None

            out["extra"] = {"per_rank": per_rank, "cfg5_shard_step_ms": spread("cfg5_step_ms"),
                            "cfg5_shard_step_brownian_bridges_ms": spread("cfg5_step_brownian_bridges_ms"),
                            "cfg3_chain_b64_us": spread("cfg3_chain_b64_us")}
            ms = out["extra"]["cfg5_shard_step_ms"]
            if ms:  # the job's rate: every rank steps its 128-latent shard, the slowest sets the pace
                out["extra"]["cfg5_latent_steps_per_s"] = 128 * n_gpus / (ms["max"] * 1e-3)
            us = out["extra"]["cfg3_chain_b64_us"]
            if us:
                out["extra"]["cfg3_chain_b64_latents_per_s"] = 64 * n_gpus / (us["max"] * 1e-6)
    if rank == 0 and distributed:
        rccl = None
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001 -- a build without the binding
            pass
        out["collectives"] = {"backend": dist.get_backend(), "rccl_version": rccl, "world_size": dist.get_world_size(),
                              "hsa_enable_ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    if rank == 0 and ranks is not None:
        out["ranks"] = ranks
    if distributed:
        # optional final gather, outside `value`: RCCL all-gather vs direct peer-to-peer copies over xGMI, reported
        # under extra.gather.  Every rank learns whether all ranks succeeded before the next collective.  The result line waits for
        # it, so a watchdog on rank 0 prints the line without the gather figures if a collective hangs (the process group's own
        # two-minute timeout then ends the ranks).
        printed = threading.Event()
        lock = threading.Lock()

        def emit(gather_result):
            with lock:
                if printed.is_set():
                    return
                out.setdefault("extra", {})["gather"] = gather_result
                print(json.dumps(out), flush=True)
                printed.set()

        watchdog = None
        if rank == 0:
            watchdog = threading.Timer(90.0, emit, args=({"error": "no answer from the gather collectives within 90 s"},))
            watchdog.daemon = True
            watchdog.start()
        par = importlib.import_module("comfyui_sonar_amd.parallel")
        coll_dev = device if backend == "nccl" else "cpu"
        with ng.shard_offset(rank * BATCH):
            shard = ns(*sig)
        gather = {"backend": dist.get_backend(), "bytes_per_shard": shard.numel() * 4}
        for tag, direct in (("rccl_all_gather", False), ("direct_peer_copies", True)):
            ok, secs = 1.0, 0.0
            try:
                full = par.gather_batch(shard, BATCH * world, direct=direct)
                torch.cuda.synchronize()
                if tuple(full.shape) != (BATCH * world, C, H, W) or not torch.equal(full[rank * BATCH:(rank + 1) * BATCH], shard):
                    raise RuntimeError("gathered batch does not hold this rank's shard at its place")
                del full
                g0 = time.perf_counter()
                for _ in range(5):
                    par.gather_batch(shard, BATCH * world, direct=direct)
                torch.cuda.synchronize()
                secs = (time.perf_counter() - g0) / 5
            except Exception as exc:
                ok = 0.0
                gather[tag + "_error"] = repr(exc)[:200]
            gt = torch.tensor([secs, -ok], device=coll_dev, dtype=torch.float64)
            dist.all_reduce(gt, op=dist.ReduceOp.MAX)  # slowest rank; -ok is 0 if any rank failed
            if gt[1].item() == -1.0:
                gather[tag + "_ms"] = gt[0].item() * 1e3
                gather[tag + "_GBps_per_rank_in"] = shard.numel() * 4 * (world - 1) / max(gt[0].item(), 1e-12) / 1e9
            else:
                gather.setdefault(tag + "_error", "failed on another rank")
                break
        if rank == 0:
            watchdog.cancel()
            emit(gather)
        dist.barrier()
        dist.destroy_process_group()
    elif rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
