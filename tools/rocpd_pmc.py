#!/usr/bin/env python3
"""Per-kernel PMC summary of rocprofv3 rocpd databases (one or more passes):
tools/rocpd_pmc.py <results.db>...  ->  markdown table, counters averaged per launch."""
import sqlite3
import sys


def main():
    agg = {}
    names = []
    for db in [a for a in sys.argv[1:] if not a.startswith("--")]:
        c = sqlite3.connect(db)
        q = ("select kernel_name, counter_name, count(*), sum(value), sum(duration) from counters_collection "
             "group by 1, 2")
        for k, cn, n, v, d in c.execute(q):
            short = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("kvfe::", "")
            a = agg.setdefault(short, {})
            a[cn] = v / n
            a["_n"] = n
            a.setdefault("_dur_us", d / n / 1e3)
            if cn not in names:
                names.append(cn)
    if "--json" in sys.argv:   # {kernel: {counter: mean per launch}} for bench.py (profiles/pmc_valu_latest.json)
        import json
        print(json.dumps({k: {n: a[n] for n in names if n in a} for k, a in agg.items() if not k.startswith("__amd")},
                         indent=1, sort_keys=True))
        return
    print("| kernel | launches | avg us | " + " | ".join(names) + " |")
    print("|---|---:|---:|" + "---:|" * len(names))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1].get("_dur_us", 0) * kv[1]["_n"]):
        if k.startswith("__amd"):
            continue
        print(f"| {k} | {a['_n']} | {a['_dur_us']:.1f} | " +
              " | ".join(f"{a.get(n, float('nan')):.4g}" for n in names) + " |")


if __name__ == "__main__":
    main()
