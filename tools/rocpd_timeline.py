#!/usr/bin/env python3
"""Print the kernel timeline (start offset, duration, queue, name) of a window of a rocprofv3
rocpd database: tools/rocpd_timeline.py <results.db> [first_kernel_index] [count]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
    rows = rows[first:first + count]
    t0 = rows[0][1]
    for name, s, e, q, st in rows:
        short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("kvfe::", "")
        print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f} us  q{q} s{st}  {short}")


if __name__ == "__main__":
    main()
