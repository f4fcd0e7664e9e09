#!/usr/bin/env python3
"""round 6: one stream, synchronous use (upload the pair, step, wait, read the record) -- where the ~0.25 ms go.
usage: KVFE_PROF_TIMELINE=8 python tools/r6/spin_probe.py [n]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from kimera_vio_amd import frontend as F  # noqa: E402
from kimera_vio_amd import workloads as WL  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    assert torch.cuda.is_available()
    wl = WL.build("c2", mode="nominal")
    ctx = F.Context(wl.left, wl.right, wl.params, batch=1, device=0)
    plan = [(st[0], wl.batch_inputs(ctx, st)) for st in wl.plan(n)]
    lefts = [np.ascontiguousarray(wl.lefts[t]) for t in range(wl.ring)]
    rights = [np.ascontiguousarray(wl.rights[t]) for t in range(wl.ring)]
    t_step, t_out, kf = [], [], 0
    for i, (t, inp) in enumerate(plan):
        if i == n - 40:
            ctx.profile_enable(1)
        a = time.perf_counter()
        ctx.step_host(lefts[t], rights[t], inp)
        b = time.perf_counter()
        o = ctx.get_output(0)
        c = time.perf_counter()
        if i >= 20:
            t_step.append(b - a)
            t_out.append(c - b)
            kf += int(o["is_keyframe"])
    ctx.profile_read()
    print(f"spins {len(t_step)} keyframes {kf}: step_host call {1e3 * np.mean(t_step):.4f} ms (median {1e3 * np.median(t_step):.4f}), "
          f"get_output {1e3 * np.mean(t_out):.4f} ms (median {1e3 * np.median(t_out):.4f}), sum {1e3 * (np.mean(t_step) + np.mean(t_out)):.4f} ms")
    ctx.close()


if __name__ == "__main__":
    main()
