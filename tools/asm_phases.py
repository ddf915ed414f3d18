#!/usr/bin/env python3
"""Static instruction counts of one kernel between its workgroup barriers, from the compiler's assembly (no GPU needed):

    python tools/asm_phases.py file.s <mangled-name-substring>

Prints, for every stretch of straight-line code between s_barrier instructions / labels, the number of vector-ALU, transcendental,
LDS, global-memory and scalar instructions -- the per-phase wave-instruction table of the pipelined power kernel in profiles/ is made
from it (the stretches of the steady-state loop are the ones with 8 + 8 v_sin / v_cos)."""
import re
import sys


def kernel_text(path, key):
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(key) + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start:end + 1]


def main():
    path, key = sys.argv[1], sys.argv[2]
    body = kernel_text(path, key)
    seg = dict(valu=0, trans=0, lds=0, vmem=0, salu=0, wait=0, xor=0, pk=0)
    first = 0
    print(f"{'lines':>13} {'valu':>5} {'trans':>5} {'pk':>4} {'xor/bitop':>9} {'lds':>4} {'vmem':>4} {'salu':>5} {'waitcnt':>7}  ends with")
    def flush(i, why):
        nonlocal seg, first
        if seg["valu"] + seg["lds"] + seg["vmem"] > 0:
            print(f"{first:6d}-{i:6d} {seg['valu']:5d} {seg['trans']:5d} {seg['pk']:4d} {seg['xor']:9d} {seg['lds']:4d} {seg['vmem']:4d} {seg['salu']:5d} {seg['wait']:7d}  {why}")
        seg = dict.fromkeys(seg, 0)
        first = i + 1
    for i, l in enumerate(body):
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            if re.match(r"^\.LBB\S+:", t):
                flush(i, t.split(":")[0])
            continue
        op = t.split()[0]
        if op == "s_barrier":
            flush(i, "s_barrier")
        elif op.startswith("v_"):
            seg["valu"] += 1
            if re.match(r"v_(sin|cos|log|exp|sqrt|rsq|rcp)_", op):
                seg["trans"] += 1
            if op.startswith("v_pk_"):
                seg["pk"] += 1
            if op.startswith(("v_xor", "v_bitop3")):
                seg["xor"] += 1
        elif op.startswith("ds_"):
            seg["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            seg["vmem"] += 1
        elif op == "s_waitcnt":
            seg["wait"] += 1
        elif op.startswith("s_"):
            seg["salu"] += 1
    flush(len(body), "end")


if __name__ == "__main__":
    main()
