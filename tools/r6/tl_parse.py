#!/usr/bin/env python3
"""tools/r6/tl_parse.py <stderr of a run with KVFE_PROF_TIMELINE=N>: the stages of each sampled step by begin time"""
import re
import sys
txt = open(sys.argv[1]).read()
for line in txt.splitlines():
    m = re.match(r'KVFE_PROF_TIMELINE sample (\d+):(.*)', line)
    if not m:
        continue
    print("sample", m.group(1))
    items = re.findall(r'(\w+) (-?\d+)-(-?\d+)', m.group(2))
    for n, a, b in sorted(items, key=lambda t: int(t[1])):
        print(f"   {int(a):7d} -> {int(b):7d} (+{int(b) - int(a):5d})  {n}")
