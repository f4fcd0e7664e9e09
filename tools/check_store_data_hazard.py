#!/usr/bin/env python3
"""ISA check for the "VMEM store of more than 64 bits -> VALU write of its data registers" hazard (2 wait states on gfx950).

A store of three or four dwords reads its data VGPRs after it has issued; a VALU instruction in the very next issue slot
that writes one of them changes what part of the wave stores.  hipcc pads the pair with an s_nop -- except for a buffer
store whose scalar offset is an SGPR, which its hazard rule exempts.  On gfx950 that form is NOT exempt: round 6 measured
the first dword of pyr2_kernel's 16-byte level-0 copy replaced by the following v_perm_b32's result in lanes 12 - 15 of
every row of 16, one run in three, when a wave issues back to back (profiles/r6_analysis.md, tools/r6/gpu_pyr_probe.sh).

This script compiles the .hip files of csrc/ to gfx950 assembly and reports every buffer / global / flat / scratch store
of more than 64 bits that is followed within fewer than two wait states (an instruction = 1, s_nop N = N + 1; hipcc's own
count for the forms it pads on gfx940 and later) by a VALU write of one of its data registers.  Local labels are looked
through (conservative).

    python tools/check_store_data_hazard.py [file.hip ...]       (default: every k_*.hip of kimera_vio_amd/csrc)

Exit code 1 if a violation is found.  tests/test_host_logic.py runs it on every kernel file.
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

WIDE_STORE = re.compile(r"^(buffer_store_dwordx[34]|buffer_store_format_xyzw?|buffer_store_format_d16_xyzw|"
                        r"global_store_dwordx[34]|flat_store_dwordx[34]|scratch_store_dwordx[34])\b")
VRANGE = re.compile(r"\bv\[(\d+):(\d+)\]")
VDST = re.compile(r"^v(\d+)\b|^v\[(\d+):(\d+)\]")


def store_data_regs(mn, rest):
    """the data operand of a wide store: the first v[a:b] for buffer stores, the second vector operand for
    global / flat / scratch (address first)"""
    ops = [o.strip() for o in rest.split(",")]
    cand = [VRANGE.search(o) for o in ops]
    if mn.startswith("buffer_"):
        m = cand[0] if cand else None
    else:   # global_store_dwordx4 vaddr, vdata, saddr | flat_store_dwordx4 vaddr, vdata
        m = cand[1] if len(cand) > 1 else None
    if not m:
        return set()
    return set(range(int(m.group(1)), int(m.group(2)) + 1))


def valu_written(mn, rest):
    if not mn.startswith("v_") or mn.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_nop")):
        return set()
    m = VDST.match(rest.strip())
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


WAIT_STATES = 2   # gfx940 and later (hipcc's own rule for the forms it pads: "VALUWaitStates = hasGFX940Insts() ? 2 : 1")


def check_asm(path):
    """returns (number of wide stores, [(kernel, store, the instruction that writes its data)])"""
    bad, n = [], 0
    kernel = "?"
    pending = []   # [store text, data registers, wait states seen since] of the wide stores still inside their window
    for ln in open(path):
        s = ln.split(";")[0].strip()
        if not s or s.startswith("."):
            continue              # (a local label is looked through: conservative)
        if s.endswith(":"):
            kernel = s[:-1]
            pending = []
            continue
        parts = s.split(None, 1)
        mn, rest = parts[0], (parts[1] if len(parts) > 1 else "")
        w = valu_written(mn, rest)
        for pnd in pending:
            if w & pnd[1]:
                bad.append((kernel, pnd[0], s))
        states = (int(rest.strip() or 0) + 1) if mn == "s_nop" else 1
        for pnd in pending:
            pnd[2] += states
        pending = [pnd for pnd in pending if pnd[2] < WAIT_STATES]
        if WIDE_STORE.match(mn):
            n += 1
            pending.append([s, store_data_regs(mn, rest), 0])
    return n, bad


def check_hip(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run([HIPCC] + FLAGS + ["-x", "hip", src, "-o", out], check=True, cwd=CSRC,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return check_asm(out)


if __name__ == "__main__":
    files = sys.argv[1:] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.startswith("k_") and f.endswith(".hip"))
    rc = 0
    for f in files:
        n, bad = check_asm(f) if f.endswith(".s") else check_hip(f)
        print("%s: %d stores of more than 64 bits, %d followed by a VALU write of their data" % (os.path.basename(f), n, len(bad)))
        for k, a, b in bad[:20]:
            print("   ", k, "|", a, "||", b)
        rc |= 1 if bad else 0
    sys.exit(rc)
