#!/usr/bin/env python3
"""Static instruction counts of the gfx950 ISA of one kernel source, per kernel and basic block, by category.

Why: the dense kernels and LK are bound by the issue of ALL their instructions -- one instruction of any category
per SIMD and 4-cycle slot (profiles/r2_v5_lk_analysis.md) -- so instructions per pixel / point, every category, is the
figure to drive down, and it can be read off the ISA without a GPU.

  python tools/isa_count.py kimera_vio_amd/csrc/k_track.hip lk_kernel_sysILi24      # one kernel, block by block
  python tools/isa_count.py kimera_vio_amd/csrc/k_detect.hip                        # every kernel, totals only
  python tools/isa_count.py kimera_vio_amd/csrc/k_track.hip lk_kernel_sysILi24 -D KVFE_LK_PROF=1

Blocks are listed with their loop depth (from the compiler's "in Loop: ... Depth=N" annotations); multiply by the trip
counts you know (the tool cannot).  Categories: valu, salu (incl. s_waitcnt / s_nop, listed separately as `wait`),
lds (ds_*), vmem (global / buffer / flat / scratch), smem (s_load / s_buffer_load), branch (s_cbranch / s_branch).
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
         "-I" + os.path.join(ROOT, "include"), "-x", "hip", "--cuda-device-only", "-S"]


def category(op):
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep")):
        return "wait"
    if op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime", "s_dcache")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    return "other"


CATS = ("valu", "salu", "lds", "vmem", "smem", "branch", "wait", "other")


def parse(asm):
    kernels, cur, block = {}, None, None
    for line in asm.splitlines():
        s = line.strip()
        m = re.match(r"^(_Z\w+|\w+):\s*; @", line)
        if m and not s.startswith("."):
            cur = {"name": m.group(1), "blocks": [], "meta": {}}
            kernels[m.group(1)] = cur
            block = {"label": "entry", "depth": 0, "counts": dict.fromkeys(CATS, 0)}
            cur["blocks"].append(block)
            continue
        if cur is None:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):(.*)$", s)
        if m:
            depth = 0
            d = re.search(r"Depth=(\d+)", m.group(2))
            if d:
                depth = int(d.group(1))
            block = {"label": m.group(1), "depth": depth, "counts": dict.fromkeys(CATS, 0)}
            cur["blocks"].append(block)
            continue
        d = re.search(r";\s+=>\s*This (?:Inner )?Loop Header: Depth=(\d+)", s)
        if d and block is not None:
            block["depth"] = int(d.group(1))
        if s.startswith(".amdhsa_"):
            parts = s.split()
            if len(parts) == 2 and parts[0] in (".amdhsa_next_free_vgpr", ".amdhsa_next_free_sgpr",
                                                ".amdhsa_group_segment_fixed_size",
                                                ".amdhsa_private_segment_fixed_size"):
                name = getattr(parse, "_desc", None)
                if name and name in kernels:
                    kernels[name]["meta"][parts[0][8:]] = parts[1]
        m = re.match(r"^\.amdhsa_kernel\s+(\S+)", s)
        if m:
            parse._desc = m.group(1)
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            if s.startswith("s_endpgm"):
                pass
            continue
        op = s.split()[0]
        if block is not None and re.match(r"^[a-z_0-9]+$", op):
            block["counts"][category(op)] += 1
            if op == "s_endpgm":
                block = None
    return kernels


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("source")
    ap.add_argument("kernel", nargs="?", default=None, help="substring of the mangled kernel name: block-by-block listing")
    ap.add_argument("-D", action="append", default=[], help="extra -D definitions")
    ap.add_argument("--min", type=int, default=8, help="hide blocks with fewer instructions than this")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + ["-D" + d for d in a.D] + [os.path.abspath(a.source), "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(a.source)))
        if r.returncode != 0:
            sys.exit(r.stderr)
        asm = open(out).read()
    kernels = parse(asm)
    for name, k in kernels.items():
        if a.kernel and a.kernel not in name:
            continue
        tot = dict.fromkeys(CATS, 0)
        for b in k["blocks"]:
            for c in CATS:
                tot[c] += b["counts"][c]
        n = sum(tot.values())
        if n == 0:
            continue
        meta = k["meta"]
        print(f"{name}\n  static: {n} instructions  " + "  ".join(f"{c} {tot[c]}" for c in CATS if tot[c]) +
              f"\n  vgpr {meta.get('next_free_vgpr', '?')}  sgpr {meta.get('next_free_sgpr', '?')}  "
              f"lds {meta.get('group_segment_fixed_size', '?')} B  scratch {meta.get('private_segment_fixed_size', '?')} B")
        if a.kernel:
            print("  block            depth  total  " + "  ".join(f"{c:>6}" for c in CATS))
            for b in k["blocks"]:
                t = sum(b["counts"].values())
                if t < a.min:
                    continue
                print(f"  {b['label']:<16} {b['depth']:>5}  {t:>5}  " + "  ".join(f"{b['counts'][c]:>6}" for c in CATS))


if __name__ == "__main__":
    main()
