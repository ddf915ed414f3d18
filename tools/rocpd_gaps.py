#!/usr/bin/env python3
"""Per-kernel durations AND the idle gap in front of each dispatch from a rocprofv3 (rocpd sqlite) kernel trace: launch-bound sequences
spend their time between kernels.  Usage: python tools/rocpd_gaps.py trace.db [min_calls]"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name).replace("void ", "").replace("sonar::", "")
    return name if len(name) < 90 else name[:87] + "..."


def main(path, min_calls=20):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
    acc = defaultdict(lambda: [0, 0.0, 0.0, 0, 0])
    prev_end = None
    for name, st, en, gx, wx in rows:
        a = acc[(short(name), gx, wx)]
        a[0] += 1
        a[1] += en - st
        if prev_end is not None and st - prev_end < 50_000:  # gaps above 50 us are host pauses, not launch latency
            a[2] += max(0, st - prev_end)
            a[3] += 1
        prev_end = en
    print("| kernel | grid | wg | calls | avg us | avg gap before, us |")
    print("|---|---:|---:|---:|---:|---:|")
    for (name, gx, wx), (n, dur, gap, ng, _) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        if n >= min_calls:
            print(f"| `{name}` | {gx} | {wx} | {n} | {dur / n / 1e3:.2f} | {gap / max(ng, 1) / 1e3:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20)
