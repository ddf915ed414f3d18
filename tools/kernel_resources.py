#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of the library, from the compiler's own metadata (no GPU needed):

    python tools/kernel_resources.py [source.hip ...] > profiles/r03_kernel_resources.txt

Each csrc/*.hip is compiled for gfx950 to assembly (device side only, the product's flags) and the amdhsa.kernels notes are read.
The summary lists, per source, the number of kernels, those that use scratch memory (register spills or stack) and those above 128
vector registers (fewer than four waves per SIMD); the headline kernels are listed in full."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "comfyui-sonar_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "--cuda-device-only", "-Wno-unused-function", "-S"]
HEADLINE = ("power_pipe_kernel", "power_irfft2_kernelILi128ELi128", "power_irfft2_kernelILi64ELi64", "power_irfft2_any_kernel", "power_stats_kernel",
            "lines_", "power_block", "levels_sampled_kernel", "wcfg_lowpass_kernel", "pyramid_plane_kernel", "perlin_generate_kernel")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return out.splitlines()
    except Exception:
        return names


def kernels_of(asm_path):
    text = open(asm_path).read()
    rows = []
    for block in text.split("  - .agpr_count:")[1:]:
        def field(key, default="0"):
            m = re.search(r"\." + key + r":\s+(\S+)", block)
            return m.group(1) if m else default
        rows.append({"name": field("name", "?"), "vgpr": int(field("vgpr_count")), "sgpr": int(field("sgpr_count")),
                     "lds": int(field("group_segment_fixed_size")), "scratch": int(field("private_segment_fixed_size")),
                     "spill": int(field("vgpr_spill_count")), "sspill": int(field("sgpr_spill_count"))})
    return rows


def main():
    print("kernel resources from the compiler's metadata (hipcc " + " ".join(FLAGS) + ")")
    with tempfile.TemporaryDirectory() as tmp:
        only = sys.argv[1:]
        for src in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") and (not only or f in only)):
            asm = os.path.join(tmp, src + ".s")
            res = subprocess.run(["hipcc", *FLAGS, src, "-o", asm], cwd=CSRC, capture_output=True, text=True)
            if res.returncode != 0:
                print(f"{src}: compile failed\n{res.stderr[-400:]}")
                continue
            rows = kernels_of(asm)
            names = demangle([r["name"] for r in rows])
            for r, n in zip(rows, names):
                r["pretty"] = re.sub(r"\(.*", "", n)
            scratch = [r for r in rows if r["scratch"] > 0]
            big = [r for r in rows if r["vgpr"] > 128]
            print(f"\n== {src}: {len(rows)} kernels, {len(scratch)} use scratch memory, {len(big)} above 128 vector registers")
            for r in scratch:
                print(f"  scratch {r['scratch']:4d} B (spilled vector registers {r['spill']:3d}, scalar {r['sspill']:3d}; vgpr {r['vgpr']:3d})  {r['pretty']}")
            for r in big:
                print(f"  vgpr {r['vgpr']:3d} (lds {r['lds']:6d} B)  {r['pretty']}")
            for r in rows:
                if any(h in r["name"] for h in HEADLINE) and r not in scratch and r not in big:
                    print(f"  vgpr {r['vgpr']:3d} sgpr {r['sgpr']:3d} lds {r['lds']:6d} B scratch 0  {r['pretty']}")


if __name__ == "__main__":
    sys.exit(main())
