"""Builds libsonar_hip.so (gfx950) from csrc/*.hip with hipcc.  In-tree, no JIT cache.

Called by ``__graft_entry__.build()``; can also be run directly: ``python _build.py [--force]``.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(HERE, "build")
LIB_PATH = os.path.join(HERE, "libsonar_hip.so")
ARCH = "gfx950"
# -ffp-contract=off: kernels restate the reference's op order; FMAs appear only where written.
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]
CXXFLAGS += os.environ.get("SONAR_EXTRA_CXXFLAGS", "").split()  # tuning experiments only (part of the build stamp)


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libsonar_hip.so cannot be built")


def _sources() -> list[str]:
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest() -> str:
    h = hashlib.sha256(" ".join(CXXFLAGS).encode())
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))]
    files.append(os.path.join(HERE, "..", "include", "sonar_hip.h"))
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _deps_digest(obj: str, src: str) -> str | None:
    """Hash of the flags, the source and every header the last compile of ``obj`` read (its -MD file); None: never compiled."""
    dep = obj[:-2] + ".d"
    if not (os.path.exists(obj) and os.path.exists(dep)):
        return None
    text = open(dep).read().replace("\\\n", " ")
    files = [f for f in text.split(":", 1)[1].split() if f.startswith(("/root", HERE, CSRC)) or not f.startswith("/")]
    h = hashlib.sha256((" ".join(CXXFLAGS) + src).encode())
    for f in sorted(set(os.path.normpath(os.path.join(CSRC, f)) for f in files)):
        if f.startswith("/opt/"):
            continue  # the toolchain's own headers
        try:
            with open(f, "rb") as fh:
                h.update(fh.read())
        except OSError:
            return None
    return h.hexdigest()


def build_library(force: bool = False, verbose: bool = False) -> str:
    """One object per .hip source; an object is recompiled when the flags, its source or a header it included changed (per-object
    stamps from the compiler's dependency files), the library relinked when any object was."""
    stamp = os.path.join(OBJDIR, "stamp")
    digest = _digest()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(stamp) and open(stamp).read() == digest:
        return LIB_PATH
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src: str) -> str:
        obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
        ostamp = obj[:-2] + ".stamp"
        if not force and os.path.exists(ostamp) and open(ostamp).read() == _deps_digest(obj, src):
            return obj
        cmd = [hipcc, *CXXFLAGS, "-MD", "-MF", obj[:-2] + ".d", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)
        with open(ostamp, "w") as fh:
            fh.write(_deps_digest(obj, src) or "")
        return obj

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 2)) as pool:
        objs = list(pool.map(compile_one, _sources()))
    # No rpath on purpose: the HIP runtime is the one PyTorch-ROCm already loaded (same soname).
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB_PATH, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as fh:
        fh.write(digest)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
