#!/usr/bin/env python3
"""ISA check for the hand-issued row requests of mineig2_kernel (k_detect.hip).

The kernel issues `buffer_load_ubyte` from inline asm and waits for them with a hand-placed `s_waitcnt vmcnt(5)`; hipcc does
not know that the destination register of such a statement is written LATER, so nothing stops it from copying that register
(a phi move, a tied asm operand) before the byte has landed -- which is what the first version of round 4's run loop ran
into.  This script compiles k_detect.hip to gfx950 assembly and checks, for every mineig2_kernel instantiation, that the
hand-issued loads target accumulation registers (a0 .. a5: the fix -- hipcc allocates no AGPR in this kernel, so
nothing it generates can touch a request in flight), that the kernel owns exactly those AGPRs (no AGPR spilling by
hipcc) and that no compiler-generated instruction names an accumulation register.

Exit code 1 on a violation.  tests/test_host_logic.py runs it."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kimera_vio_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
         "-S", "--cuda-device-only", "-Wno-unused-command-line-argument", "-w"]
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def vregs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check(path, prefix="_ZN4kvfe14mineig2_kernel"):
    """{kernel: (request registers, [violations])}.  The requests must target accumulation registers, the kernel must
    own exactly those AGPRs (hipcc allocated none itself: no AGPR spilling), and no instruction outside the hand-written
    asm blocks may name an accumulation register."""
    kernels, cur, body, nagpr = {}, None, [], {}
    for ln in open(path):
        if ln.startswith(prefix) and ":" in ln and ln.split(":")[0].startswith(prefix):
            cur, body = ln.split(":")[0], []
            kernels[cur] = body
        elif cur is not None:
            body.append(ln.rstrip("\n"))
            if ln.startswith(".Lfunc_end"):
                cur = None
        m = re.match(r"\s*\.set (\S+)\.num_agpr, (\d+)", ln)
        if m and m.group(1).startswith(prefix):
            nagpr[m.group(1)] = int(m.group(2))
    AREG = re.compile(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]")
    report = {}
    for k, body in kernels.items():
        in_asm, dests, bad, n_req = False, set(), [], 0
        for ln in body:
            if "#ASMSTART" in ln:
                in_asm = True
                continue
            if "#ASMEND" in ln:
                in_asm = False
                continue
            code = ln.split(";")[0].strip()
            if not code or code.endswith(":") or code.startswith("."):
                continue
            if code.startswith("buffer_load_ubyte"):
                n_req += 1
                d = code.split(None, 1)[1].split(",")[0].strip()
                if in_asm and re.fullmatch(r"a\d+", d):
                    dests.add(d)
                else:
                    bad.append((code, "a hand-issued request must target an accumulation register"))
            elif not in_asm and AREG.search(code.split(None, 1)[1] if " " in code else ""):
                bad.append((code, "compiler-generated code names an accumulation register"))
        if n_req == 0:
            bad.append(("", "no hand-issued request found (the check no longer matches the kernel)"))
        if nagpr.get(k, -1) != len(dests):
            bad.append(("", f"the kernel allocates {nagpr.get(k)} AGPRs, the requests use {sorted(dests)}: hipcc uses AGPRs itself"))
        report[k] = (sorted(dests), bad)
    return report


def check_hip(hip=os.path.join(CSRC, "k_detect.hip")):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run([HIPCC] + FLAGS + ["-o", out, os.path.abspath(hip)], check=True, cwd=CSRC, capture_output=True)
        return check(out)


if __name__ == "__main__":
    rep = check_hip()
    rc = 0
    for k, (dests, bad) in rep.items():
        print(f"{k[:48]}...: request registers {dests}: {len(bad)} violations")
        for b in bad[:8]:
            print("    ", b)
        rc |= 1 if bad else 0
    sys.exit(rc)
