#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.x, rocpd sqlite) kernel trace into the same table that
`rocprofv3 --stats` prints: per-kernel calls, total / average / min / max duration, percentage.
Usage: tools/rocpd_stats.py <results.db> [--skip-first N] > profiles/<name>.md"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, lds_size, vgpr_count, "
                     "sgpr_count, scratch_size from kernels order by start").fetchall()
    agg = {}
    for name, s, e, gx, gy, gz, wx, lds, vg, sg, scr in rows:
        short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("kvfe::", "")
        a = agg.setdefault(short, dict(n=0, tot=0, mn=1e18, mx=0, grid=(gx, gy, gz), wg=wx, lds=lds, vgpr=vg,
                                       sgpr=sg, scratch=scr))
        d = e - s
        a["n"] += 1
        a["tot"] += d
        a["mn"] = min(a["mn"], d)
        a["mx"] = max(a["mx"], d)
    total = sum(a["tot"] for a in agg.values()) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % | grid (threads) | wg | LDS B | VGPR | SGPR | scratch |")
    print("|---|---:|---:|---:|---:|---:|---:|---|---:|---:|---:|---:|---:|")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["tot"]):
        print(f"| {k} | {a['n']} | {a['tot']/1e6:.3f} | {a['tot']/a['n']/1e3:.2f} | {a['mn']/1e3:.2f} | "
              f"{a['mx']/1e3:.2f} | {100*a['tot']/total:.1f} | {a['grid']} | {a['wg']} | {a['lds']} | {a['vgpr']} | "
              f"{a['sgpr']} | {a['scratch']} |")


if __name__ == "__main__":
    main()
