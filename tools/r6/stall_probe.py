#!/usr/bin/env python3
"""round 6: which history of the process makes the 6.5 ms stalls inside the output hipMemcpyAsync go away?
usage: KVFE_HOST_PROF=1 python tools/r6/stall_probe.py <pre> ; pre = none | create | steps30 | steps120 | steps120sync
(the per-call times are printed by the library when a context is destroyed)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from kimera_vio_amd import frontend as F  # noqa: E402
from kimera_vio_amd import workloads as WL  # noqa: E402


def run(wl, d_left, d_right, n, sync_every=0, tag=""):
    ctx = F.Context(wl.left, wl.right, wl.params, batch=wl.batch, device=0, device_frames_persist=0)
    plan = [(st[0], wl.batch_inputs(ctx, st)) for st in wl.plan(n)]
    t0 = time.perf_counter()
    for i, (t, inp) in enumerate(plan):
        ctx.step_device(d_left[t].data_ptr(), d_right[t].data_ptr(), inp)
        if sync_every and (i + 1) % sync_every == 0:
            ctx.synchronize()
    te = time.perf_counter()
    ctx.synchronize()
    t1 = time.perf_counter()
    print(f"{tag}: {n} steps, enqueue {1e3 * (te - t0) / max(n, 1):.3f} ms/step, total {1e3 * (t1 - t0) / max(n, 1):.3f} ms/step", file=sys.stderr, flush=True)
    ctx.close()


def main():
    pre = sys.argv[1] if len(sys.argv) > 1 else "none"
    dev = torch.device("cuda", 0)
    wl = WL.build("c3", mode="kf", rank=0)
    lefts, rights = wl.replicated()
    d_left, d_right = torch.from_numpy(lefts).to(dev), torch.from_numpy(rights).to(dev)
    torch.cuda.synchronize()
    if pre == "create":
        run(wl, d_left, d_right, 0, tag="pre (no steps)")
    elif pre == "steps30":
        run(wl, d_left, d_right, 30, tag="pre 30")
    elif pre == "steps120":
        run(wl, d_left, d_right, 120, tag="pre 120")
    elif pre == "steps120sync":
        run(wl, d_left, d_right, 120, sync_every=1, tag="pre 120 sync every step")
    run(wl, d_left, d_right, 166, sync_every=52, tag="main")


if __name__ == "__main__":
    main()
