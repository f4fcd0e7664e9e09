#!/usr/bin/env python3
"""tools/rocpd_traffic.py <fetch.db> <write.db> -> JSON {kernel: {fetch_kb, write_kb, launches}} with the
rocprofv3 FETCH_SIZE / WRITE_SIZE counters (KB) averaged per launch (steady state: the first launch of
every kernel, which sees the bootstrap frame, is skipped when there are more than two)."""
import json
import sqlite3
import sys


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, value, start from counters_collection where counter_name = ? "
                     "order by start", (counter,)).fetchall()
    agg = {}
    for name, v, _ in rows:
        short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("kvfe::", "")
        agg.setdefault(short, []).append(v)
    out = {}
    for k, vals in agg.items():
        if len(vals) > 2:
            vals = vals[1:]
        out[k] = (sum(vals) / len(vals), len(vals))
    return out


def main():
    f = per_kernel(sys.argv[1], "FETCH_SIZE")
    w = per_kernel(sys.argv[2], "WRITE_SIZE")
    res = {k: {"fetch_kb": round(f[k][0], 1), "write_kb": round(w.get(k, (0, 0))[0], 1), "launches": f[k][1]}
           for k in f if not k.startswith("__amd")}
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
