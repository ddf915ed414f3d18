#!/usr/bin/env python3
"""Host-side sanitizer run of the C ABI (no GPU needed; GPU AddressSanitizer is not available on this pool):

    python tools/host_sanitize.py            # builds csrc/*.hip with -fsanitize=address,undefined (host code only) + a generated driver

The driver is generated from include/sonar_hip.h: every entry point is called (a) with every argument zero / NULL and (b) with sizes
of 4 and 1 but NULL buffers -- the argument-validation paths, which must return an error code (or 0 / -1 for the pure size queries)
before anything touches the HIP runtime.  The sanitizers watch the host code those paths run (table copies, plan arithmetic, error
formatting).  Output: the driver's log, also written to profiles/r06_host_sanitizer.txt."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "comfyui-sonar_amd", "csrc")
HEADER = os.path.join(ROOT, "include", "sonar_hip.h")
FLAGS = ["-O1", "-g", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-fsanitize=address,undefined", "-fno-gpu-sanitize",
         "-fno-sanitize-recover=undefined", "-Wno-unused-function"]
SIZES = r"\b(n|planes|rows|B|outer|H|W|h|w|L|n_in|n_out|inner|chw|levels|taps|iters|npart|n_total|groups|group_size|row_len|mid|C|nq|plane_elems)\b"


def prototypes():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for m in re.finditer(r"\b(int64_t|int|const char\*)\s+(sonar_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        yield ret, name, params


def zero_arg(param: str, size: int) -> str:
    if "*" in param:
        return "nullptr"
    if re.search(r"\b(float|double)\b", param):
        return "1.0" if size else "0.0"
    if size and re.search(SIZES, param.split()[-1]):
        return str(size)
    return "0"


def driver_source() -> str:
    lines = ['#include <cstdio>', '#include <cstdint>', f'#include "{HEADER}"', "int main() {", "    int calls = 0, bad = 0;"]
    for ret, name, params in prototypes():
        for size in (0, 4, 1):
            call = f"{name}({', '.join(zero_arg(p, size) for p in params)})"
            if ret == "const char*":
                lines.append(f"    (void){call}; ++calls;")
            else:
                query = re.search(r"version|kind|bytes|_len$|_pipeline$|_hi_storage$|_fn_nargs$|_fn_id$|_plan_length$|_pyramid_levels$|_ahead_ok$", name) is not None  # pure queries / switches answer with a number
                check = f'if (rc > 0) {{ std::printf("{name}: rc %lld\\n", rc); ++bad; }}' if ret == "int" and not query else ""
                lines.append(f"    {{ long long rc = (long long){call}; ++calls; (void)rc; {check} }}")
    lines += ['    std::printf("%d calls through the C ABI with NULL / degenerate arguments, %d unexpected return codes; last error text: %s\\n", calls, bad, sonar_last_error());',
              "    return bad != 0;", "}"]
    return "\n".join(lines) + "\n"


def main():
    out_dir = tempfile.mkdtemp(prefix="sonar_san_")
    drv = os.path.join(out_dir, "driver.cpp")
    with open(drv, "w") as fh:
        fh.write(driver_source())
    objs, procs = [], []
    for src in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
        obj = os.path.join(out_dir, src[:-4] + ".o")
        objs.append(obj)
        procs.append(subprocess.Popen(["hipcc", *FLAGS, "-c", src, "-o", obj], cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        log = p.communicate()[0].decode()
        if p.returncode:
            sys.exit(log)
    exe = os.path.join(out_dir, "driver")
    subprocess.run(["hipcc", *FLAGS, "-x", "hip", drv, "-x", "none", *objs, "-o", exe], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([exe], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    log = run.stdout.decode()
    report = (f"host sanitizer run (tools/host_sanitize.py): hipcc {' '.join(FLAGS[6:9])}, {len(objs)} sources + generated driver, "
              f"exit code {run.returncode}\n" + log)
    print(report)
    with open(os.path.join(ROOT, "profiles", "r06_host_sanitizer.txt"), "w") as fh:
        fh.write(report)
    sys.exit(run.returncode)


if __name__ == "__main__":
    main()
