#!/usr/bin/env python3
"""HBM traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately, rocpd sqlite):
    python tools/rocpd_traffic.py fetch.db write.db > table.json
Per kernel name: dispatches, average FETCH_SIZE / WRITE_SIZE (KiB per dispatch, raw) and the corrected bytes per launch
(2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md, confirmed by the calibration kernels of the same run)."""
import json
import re
import sqlite3
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name).replace("void ", "").replace("sonar::", "")
    return name


def averages(path, counter):
    db = sqlite3.connect(path)
    cols = [d[0] for d in db.execute("select * from counters_collection limit 1").description]
    ki = cols.index("kernel_name") if "kernel_name" in cols else cols.index("name")
    ci, vi, di = cols.index("counter_name"), cols.index("value"), cols.index("dispatch_id")
    acc, cnt = defaultdict(float), defaultdict(set)
    for r in db.execute("select * from counters_collection"):
        if r[ci] == counter:
            acc[short(r[ki])] += r[vi]
            cnt[short(r[ki])].add(r[di])
    return {k: (acc[k] / max(len(cnt[k]), 1), len(cnt[k])) for k in acc}


def main(fetch_db, write_db):
    f, w = averages(fetch_db, "FETCH_SIZE"), averages(write_db, "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w)):
        fk, n = f.get(k, (0.0, 0))
        wk, n2 = w.get(k, (0.0, 0))
        out[k] = {"dispatches": max(n, n2), "FETCH_SIZE_KiB": round(fk, 1), "WRITE_SIZE_KiB": round(wk, 1),
                  "hbm_bytes_per_launch": int(round((2 * fk + wk) * 1024))}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
