#!/usr/bin/env python3
"""Scalar loads inside loops, per kernel, from the compiler's assembly (hipcc -S --cuda-device-only): a loop that loads a kernel argument
or a table entry through the scalar cache and waits for it every iteration runs at that latency (the pyramid plane kernel's level loop
did: round 5).  With --lanes: v_readlane / v_writelane inside loops instead -- scalar values spilled to vector-register lanes and fetched
back in front of every use (the fp64 band kernel's filter taps did: round 5).  With --branches: scalar conditional branches inside
loops -- a run-time flag tested in front of every value (the Perlin kernels' divisor kind was: round 5).
Usage: python tools/asm_loop_loads.py [--lanes | --branches] file.s [name-filter]"""
import re, subprocess, sys

argv = [a for a in sys.argv[1:] if a not in ("--lanes", "--branches")]
LANES = "--lanes" in sys.argv
BRANCHES = "--branches" in sys.argv
text = open(argv[0]).read().splitlines()
flt = argv[1] if len(argv) > 1 else ""
PATTERN = r"\bv_(readlane|writelane)_b32" if LANES else r"\bs_cbranch_(vcc|scc)" if BRANCHES else r"\bs_(load|buffer_load)_"
WHAT = "lane moves" if LANES else "scalar branches" if BRANCHES else "scalar loads"
kernel, inloop, depth, found = None, False, 0, {}
for line in text:
    m = re.match(r"^(_Z\w+):", line)
    if m:
        kernel, inloop = m.group(1), False
        continue
    if kernel is None:
        continue
    if re.match(r"^\.LBB\d+_\d+:", line) or re.match(r"^; %bb\.\d+:", line):
        inloop = "Loop" in line
        d = re.search(r"Depth=(\d+)", line)
        depth = int(d.group(1)) if d else (1 if inloop else 0)
        continue
    if "Loop Header" in line or "in Loop" in line:
        inloop = True
        d = re.search(r"Depth=(\d+)", line)
        if d:
            depth = max(depth, int(d.group(1)))
        continue
    if inloop and re.search(PATTERN, line):
        found.setdefault(kernel, []).append((depth, line.strip()))
    if "s_endpgm" in line:
        inloop = False
names = list(found)
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines() if names else []
for n, d in zip(names, dem):
    if flt and flt not in d:
        continue
    rows = found[n]
    print(f"{len(rows):4d} {WHAT} in loops (deepest {max(r[0] for r in rows)})  {d[:150]}")
