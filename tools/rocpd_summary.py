#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / avg / min / max duration,
registers and LDS.  Usage: python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/x.md"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)  # drop the argument list
    name = name.replace("void ", "")
    return name if len(name) < 110 else name[:107] + "..."


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), avg(duration), min(duration), max(duration), sum(duration), max(vgpr_count), max(sgpr_count),"
        " max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc"
    ).fetchall()
    total = sum(r[5] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary ({path.split('/')[-1]})\n")
    print("| kernel | calls | avg us | min us | max us | total ms | % | VGPR | SGPR | LDS B | grid | wg |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for name, calls, avg, mn, mx, tot, vg, sg, lds, gx, wx in rows[:25]:
        print(f"| `{short(name)}` | {calls} | {avg/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {tot/1e6:.3f} | {100*tot/total:.1f} | {vg} | {sg} | {lds} | {gx} | {wx} |")


if __name__ == "__main__":
    main(sys.argv[1])
