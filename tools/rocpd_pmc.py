#!/usr/bin/env python3
"""Per-kernel PMC averages from a rocprofv3 (rocpd sqlite) counter-collection run.
Usage: python tools/rocpd_pmc.py results.db [kernel-substring]"""
import sqlite3
import sys
from collections import defaultdict


def main(path, needle=""):
    db = sqlite3.connect(path)
    cols = [d[0] for d in db.execute("select * from counters_collection limit 1").description]
    rows = db.execute("select * from counters_collection").fetchall()
    ki, ci, vi = cols.index("kernel_name") if "kernel_name" in cols else cols.index("name"), cols.index("counter_name"), cols.index("value")
    di = cols.index("dispatch_id")
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(set)
    for r in rows:
        if needle in r[ki]:
            acc[r[ki]][r[ci]] += r[vi]
            cnt[r[ki]].add(r[di])
    for k, d in acc.items():
        n = max(len(cnt[k]), 1)
        print(k[:100], f"({n} dispatches)")
        for c, v in sorted(d.items()):
            print(f"    {c:28s} {v / n:16.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
