#!/usr/bin/env python3
"""round 6: do two independent front-end pipelines overlap on one GPU?
K contexts of 64 / K streams each (the headline's streams dealt out), every context driven by its own host thread
(ctypes releases the GIL), against one context of 64 streams.  usage: python tools/r6/two_ctx_probe.py [K ...]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from kimera_vio_amd import frontend as F  # noqa: E402
from kimera_vio_amd import workloads as WL  # noqa: E402

N_WARM, N_STEPS = 12, 104


def run(K, same_thread=False):
    dev = torch.device("cuda", 0)
    B = 64 // K
    wl = WL.build("c3", mode="kf", rank=0, batch=B)
    lefts, rights = wl.replicated()
    d_left, d_right = torch.from_numpy(lefts).to(dev), torch.from_numpy(rights).to(dev)
    torch.cuda.synchronize()
    ctxs = [F.Context(wl.left, wl.right, wl.params, batch=B, device=0, device_frames_persist=0) for _ in range(K)]
    plans = [[(st[0], wl.batch_inputs(c, st)) for st in wl.plan(N_WARM + N_STEPS)] for c in ctxs]

    def drive(k, a, b):
        c = ctxs[k]
        for t, inp in plans[k][a:b]:
            c.step_device(d_left[t].data_ptr(), d_right[t].data_ptr(), inp)

    def phase(a, b):
        if same_thread:
            for i in range(a, b):
                for k in range(K):
                    drive(k, i, i + 1)
        else:
            th = [threading.Thread(target=drive, args=(k, a, b)) for k in range(K)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        for c in ctxs:
            c.synchronize()

    phase(0, N_WARM)
    t0 = time.perf_counter()
    phase(N_WARM, N_WARM + N_STEPS)
    dt = time.perf_counter() - t0
    print(f"K={K} contexts x {B} streams ({'one host thread' if same_thread else 'a host thread each'}): "
          f"{64 * N_STEPS / dt:9.1f} pairs/s, {1e3 * dt / N_STEPS:.4f} ms per 64 pairs", flush=True)
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    ks = [int(a) for a in sys.argv[1:]] or [1, 2, 4]
    for K in ks:
        run(K)
        if K > 1:
            run(K, same_thread=True)
