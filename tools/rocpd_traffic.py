#!/usr/bin/env python3
"""tools/rocpd_traffic.py --leg NAME <fetch.db> <write.db> [--leg NAME2 <fetch.db> <write.db> ...]
-> JSON {leg: {kernel: {fetch_kb, write_kb, launches, shapes}}} from rocprofv3 PMC passes (FETCH_SIZE and
WRITE_SIZE, KB, collected in separate passes of the same command).

Every pass runs ONE leg of bench.py alone (`--legs none`, `--config c5 --legs none`, `--legs dense` ...), so
every launch of a kernel belongs to that leg.  Launches are grouped by grid shape; within a shape the counter is
averaged per launch (steady state: the first launch of a shape, which sees the bootstrap frame, is skipped when
there are more than two); `fetch_kb` / `write_kb` add the shapes up = KB per step for kernels launched once per
shape and step (the pyramid is one launch per level; rectify / min-eig have one shape)."""
import json
import sqlite3
import sys


def short_name(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("kvfe::", "")


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, grid_size_x, grid_size_y, grid_size_z, value, start from "
                     "counters_collection where counter_name = ? order by start", (counter,)).fetchall()
    agg = {}
    for name, gx, gy, gz, v, _ in rows:
        agg.setdefault(short_name(name), {}).setdefault(f"{gx}x{gy}x{gz}", []).append(v)
    out = {}
    for k, shapes in agg.items():
        o = {}
        for sh, vals in shapes.items():
            if len(vals) > 2:
                vals = vals[1:]
            o[sh] = (sum(vals) / len(vals), len(vals))
        out[k] = o
    return out


def leg(fetch_db, write_db):
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    res = {}
    for k, shapes in f.items():
        if k.startswith("__amd"):
            continue
        ws = w.get(k, {})
        res[k] = {"fetch_kb": round(sum(v[0] for v in shapes.values()), 1),
                  "write_kb": round(sum(v[0] for v in ws.values()), 1),
                  "launches": sum(v[1] for v in shapes.values()),
                  "shapes": {sh: {"fetch_kb": round(v[0], 1), "write_kb": round(ws.get(sh, (0, 0))[0], 1),
                                  "launches": v[1]} for sh, v in shapes.items()}}
    return res


def main():
    a = sys.argv[1:]
    out = {}
    while a:
        assert a[0] == "--leg" and len(a) >= 4, __doc__
        out[a[1]] = leg(a[2], a[3])
        a = a[4:]
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
