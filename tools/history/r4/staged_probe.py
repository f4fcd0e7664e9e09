"""round 4: the staged (PCIe-inclusive) loop of bench.py's pcie leg on its own, for A/B runs of library variants
(KVFE_LIB) -- tools/r4/gpu_y.sh"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kimera_vio_amd import frontend as F, workloads as WL

wl = WL.build("c3", mode="kf", use_ransac=1)
B = wl.batch
lefts, rights = wl.replicated()
hl = np.ascontiguousarray(lefts[:3])
hr = np.ascontiguousarray(rights[:3])
kw = {}
for a in sys.argv[1:]:
    k, v = a.split("=")
    kw[k] = int(v)
ctx = F.Context(wl.left, wl.right, wl.params, batch=B, **kw)
plan = [wl.batch_inputs(ctx, st) for st in wl.plan(40)]
for sl in range(3):
    a, b = ctx.staging_buffers(sl)
    a[:] = hl[sl]
    b[:] = hr[sl]
for i in range(3):
    ctx.step_staged(i % 3, plan[i])
ctx.synchronize()
n_h = 30
th = time.perf_counter()
for i in range(3, 3 + n_h):
    ctx.step_staged(i % 3, plan[i])
ctx.synchronize()
dt = time.perf_counter() - th
print("staged: %.0f pairs/s, %.3f ms per step  (%s, %s)" % (B * n_h / dt, dt / n_h * 1e3, os.environ.get("KVFE_LIB", "libkvfe.so").split("/")[-1], kw))
ctx.close()
