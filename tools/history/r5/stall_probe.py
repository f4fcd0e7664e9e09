"""round 5: is the periodic slow region of the headline loop (tools/r5/transient_probe.py: regions of 10 steps read 55 / 45 / 55 /
55 / 46 ... k pairs/s, r4: every fourth 20-step region 12 % faster) a host-side block inside an enqueue call?  Times every
kvfe_frontend_step_device call on the host over N steps after a warm-up, prints the calls that took more than 3 x the
median with their step index, the throughput of the whole run and of windows of 10 steps by COMPLETION (a polling thread is
not used: the window rate comes from one synchronisation per window in a second pass).  Run under different runtime
environment settings by the caller (tools/r5/gpu_b.sh)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from kimera_vio_amd import frontend as F, workloads as WL  # noqa: E402

N = int(os.environ.get("PROBE_N", "160"))
dev = torch.device("cuda", 0)
wl = WL.build("c3", mode="kf", use_ransac=1)
if os.environ.get("PROBE_MAX_AGE"):   # (hypothesis: the period is maxFeatureAge -- every feature born at the bootstrap frame ages out together)
    wl.params.tracker.max_feature_track_age = int(os.environ["PROBE_MAX_AGE"])
B = wl.batch
lefts, rights = wl.replicated()
d_left, d_right = torch.from_numpy(lefts).to(dev), torch.from_numpy(rights).to(dev)
torch.cuda.synchronize()
ctx = F.Context(wl.left, wl.right, wl.params, batch=B, device=0, device_frames_persist=int(os.environ.get("PROBE_PERSIST", "0")))
plan = [(st[0], wl.batch_inputs(ctx, st)) for st in wl.plan(60 + 2 * N)]
for i in range(60):
    t, inp = plan[i]
    ctx.step_device(d_left[t].data_ptr(), d_right[t].data_ptr(), inp)
ctx.synchronize()
dur = np.zeros(N)
t0 = time.perf_counter()
for k in range(N):
    t, inp = plan[60 + k]
    a = time.perf_counter()
    ctx.step_device(d_left[t].data_ptr(), d_right[t].data_ptr(), inp)
    dur[k] = time.perf_counter() - a
t_enq = time.perf_counter() - t0
ctx.synchronize()
t_all = time.perf_counter() - t0
med = float(np.median(dur))
slow = [(int(k), round(float(dur[k]) * 1e3, 3)) for k in range(N) if dur[k] > 3 * med]
print("[%s] %d steps: %.1f k pairs/s, %.3f ms per step; host enqueue median %.3f ms, total enqueue %.1f ms of %.1f ms; calls > 3 x median: %s"
      % (os.environ.get("PROBE_TAG", "default"), N, B * N / t_all / 1e3, t_all / N * 1e3, med * 1e3, t_enq * 1e3, t_all * 1e3, slow), flush=True)
# second pass: windows of 10 steps, each synchronised
w = []
for r in range(N // 10):
    ctx.synchronize()
    a = time.perf_counter()
    for k in range(10):
        t, inp = plan[60 + N + r * 10 + k]
        ctx.step_device(d_left[t].data_ptr(), d_right[t].data_ptr(), inp)
    ctx.synchronize()
    w.append(round(B * 10 / (time.perf_counter() - a) / 1e3, 1))
print("   windows of 10 steps:", w, flush=True)
# third pass: every step synchronised, with the number of new corners of stream 0 (the detection-side work of the step)
per = []
for k in range(60):
    t, inp = plan[k]
    a = time.perf_counter()
    ctx.step_device(d_left[t].data_ptr(), d_right[t].data_ptr(), inp)
    ctx.synchronize()
    per.append((round((time.perf_counter() - a) * 1e3, 2), int(ctx.get_output(0)["n_detected"])))
print("   per step (ms, new corners of stream 0):", per, flush=True)
ctx.close()
