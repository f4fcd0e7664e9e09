#!/usr/bin/env python3
"""ISA check for the gfx9 "VALU write VGPR -> DPP read of that VGPR" hazard (2 wait states required).

hipcc inserts the wait states (s_nop) for the DPP instructions IT emits (builtins, fused v_*_dpp forms); it cannot see
inside inline asm.  This script compiles a .hip file of csrc/ to gfx950 assembly and reports every DPP instruction whose
DPP source operand (src0) is written by a VALU instruction fewer than 2 wait states earlier in straight-line code.

    python tools/check_dpp_hazard.py [file.hip ...]        (default: every k_*.hip of kimera_vio_amd/csrc)

Exit code 1 if a violation is found.  tests/test_host_logic.py runs it on the tracking kernel.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kimera_vio_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
         "-S", "--cuda-device-only", "-Wno-unused-command-line-argument"]

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def split_ops(rest):
    # operands up to the first DPP / SDWA modifier keyword
    rest = re.split(r"\s(?:row_|quad_perm|wave_|bank_mask|row_mask|bound_ctrl|dst_sel|src0_sel|fi:)", " " + rest)[0]
    return [o.strip() for o in rest.split(",") if o.strip()]


def check_asm(path):
    bad, n_dpp = [], 0
    kernel = "?"
    hist = []  # (mnemonic, written vregs, wait states it provides)
    for ln in open(path):
        s = ln.split(";")[0].strip()
        if not s:
            continue
        if s.endswith(":"):
            if not s.startswith(".") and not s.startswith("$"):
                kernel = s[:-1]
            hist = []  # a label: control flow may join here; be conservative and forget (hazards across branches are the compiler's)
            continue
        if s.startswith("."):
            continue
        parts = s.split(None, 1)
        mn, rest = parts[0], (parts[1] if len(parts) > 1 else "")
        is_valu = mn.startswith("v_")
        ops = split_ops(rest) if is_valu else []
        if "_dpp" in mn and ops:
            n_dpp += 1
            # src0 is the DPP-shifted operand: operand 1 (after vdst); v_cmp*_dpp / readlane forms are not used here
            src = regs(ops[1]) if len(ops) > 1 else set()
            ws = 0
            for pmn, pw, pws in reversed(hist):
                if ws >= 2:
                    break
                if pw & src:
                    bad.append((kernel, pmn, s))
                    break
                ws += pws
        if is_valu and ops and not mn.startswith("v_cmp") and not mn.startswith("v_readlane") \
                and not mn.startswith("v_readfirstlane"):
            hist.append((mn, regs(ops[0]), 1))
        elif mn == "s_nop":
            hist.append((mn, set(), int(rest.strip() or 0) + 1))
        else:
            hist.append((mn, set(), 1))
        if len(hist) > 8:
            hist.pop(0)
    return n_dpp, bad


def check_hip(hip):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run([HIPCC] + FLAGS + ["-o", out, os.path.abspath(hip)], check=True, cwd=CSRC, capture_output=True)
        return check_asm(out)


def main(argv):
    files = argv or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.startswith("k_") and f.endswith(".hip"))
    rc = 0
    for f in files:
        n, bad = check_hip(f)
        print(f"{os.path.basename(f)}: {n} DPP instructions, {len(bad)} with a VALU write of the DPP source < 2 wait states before")
        for k, p, s in bad[:10]:
            print(f"   {k}: after `{p}`: {s}")
        rc |= 1 if bad else 0
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
