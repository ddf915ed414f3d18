#!/usr/bin/env python3
"""Kernel resources of the BUILT library, read from its code objects' metadata in seconds (no recompilation):

    python tools/so_kernels.py [libsonar_hip.so] [--scratch]      (--scratch: only the kernels with a private segment)

The .hip_fatbin section of the shared object holds one clang offload bundle per translation unit; each is unbundled for gfx950 and its
amdhsa.kernels notes are read with llvm-readelf.  `kernels(path)` returns [{name, vgpr, sgpr, lds, scratch, spill_v, spill_s}]."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _section(path, name=".hip_fatbin"):
    out = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-S", "-W", path], capture_output=True, text=True, check=True).stdout
    for line in out.splitlines():
        m = re.search(r"\]\s+" + re.escape(name) + r"\s+\S+\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", line)
        if m:
            off, size = int(m.group(2), 16), int(m.group(3), 16)
            with open(path, "rb") as fh:
                fh.seek(off)
                return fh.read(size)
    raise RuntimeError(f"{path}: no {name} section")


def kernels(path=None):
    path = path or os.path.join(ROOT, "comfyui-sonar_amd", "libsonar_hip.so")
    blob = _section(path)
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for i, s in enumerate(starts):
            e = starts[i + 1] if i + 1 < len(starts) else len(blob)
            bundle = os.path.join(tmp, f"b{i}.bundle")
            with open(bundle, "wb") as fh:
                fh.write(blob[s:e])
            co = os.path.join(tmp, f"b{i}.co")
            res = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                  f"--input={bundle}", f"--output={co}"], capture_output=True, text=True)
            if res.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
            for block in notes.split("  - .agpr_count:")[1:]:
                def field(key, default="0"):
                    m = re.search(r"\." + key + r":\s+(\S+)", block)
                    return m.group(1) if m else default
                rows.append({"name": field("name", "?"), "vgpr": int(field("vgpr_count")), "sgpr": int(field("sgpr_count")),
                             "lds": int(field("group_segment_fixed_size")), "scratch": int(field("private_segment_fixed_size")),
                             "spill_v": int(field("vgpr_spill_count")), "spill_s": int(field("sgpr_spill_count"))})
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    for r, n in zip(rows, names):
        r["pretty"] = re.sub(r"\(.*", "", n)
    return rows


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rows = kernels(args[0] if args else None)
    only = "--scratch" in sys.argv
    shown = [r for r in rows if r["scratch"] > 0] if only else rows
    for r in sorted(shown, key=lambda r: r["pretty"]):
        print(f"vgpr {r['vgpr']:3d} sgpr {r['sgpr']:3d} lds {r['lds']:6d} scratch {r['scratch']:4d} (spilled vector {r['spill_v']:3d}, scalar {r['spill_s']:3d})  {r['pretty']}")
    print(f"{len(rows)} kernels, {sum(1 for r in rows if r['scratch'] > 0)} with a scratch segment, {sum(1 for r in rows if r['vgpr'] > 128)} above 128 vector registers")


if __name__ == "__main__":
    main()
