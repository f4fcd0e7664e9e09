"""round 5: how evenly are the new corners of a real-frame step spread over the 64 streams?  (The grouped cornerSubPix
kernel gives every stream the same number of blocks; a stream with more corners than the others ends the launch.)
Runs the `kf_realistic` workload (c3e) step by step, synchronised, and prints per step: new corners of every unique
window, min / mean / max over the streams, and the step time.  tools/r5/gpu_x.sh."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from kimera_vio_amd import frontend as F, workloads as WL  # noqa: E402

N = int(os.environ.get("PROBE_N", "12"))
dev = torch.device("cuda", 0)
wl = WL.build("c3e", mode="kf", use_ransac=1)
B = wl.batch
lefts, rights = wl.replicated()
d_left, d_right = torch.from_numpy(lefts).to(dev), torch.from_numpy(rights).to(dev)
torch.cuda.synchronize()
ctx = F.Context(wl.left, wl.right, wl.params, batch=B, device=0)
plan = [(st[0], wl.batch_inputs(ctx, st)) for st in wl.plan(20 + N)]
for i in range(20):
    t, inp = plan[i]
    ctx.step_device(d_left[t].data_ptr(), d_right[t].data_ptr(), inp)
ctx.synchronize()
for k in range(N):
    t, inp = plan[20 + k]
    a = time.perf_counter()
    ctx.step_device(d_left[t].data_ptr(), d_right[t].data_ptr(), inp)
    ctx.synchronize()
    dt = time.perf_counter() - a
    nd = np.array([ctx.get_output(s)["n_detected"] for s in range(B)])
    print("step %2d  %.3f ms  new corners: first %d streams %s | min %d mean %.1f max %d sum %d  (max / mean %.2f)"
          % (k, dt * 1e3, 2 * wl.unique, nd[:2 * wl.unique].tolist(), nd.min(), nd.mean(), nd.max(), nd.sum(), nd.max() / max(nd.mean(), 1)),
          flush=True)
ctx.close()
