"""The staged-input path must OVERLAP a step's host-to-device upload with the previous step's kernels (VERDICT r5
item 10).  kvfe_frontend_step_staged (the data provider's hand-over, EurocDataProvider.cpp:146-195 ->
StereoVisionImuFrontend.cpp:283) puts the upload of step k + 1 on a copy stream created in another hardware-queue pool
than the compute streams (csrc/kvfe_api.cpp, ensure_staging / create_stream_in_other_pool): that rests on how the HIP
runtime deals same-priority streams onto hardware queues, which no document promises.  If a runtime changes the pooling,
results stay right and only the speed goes -- this test is what notices.

Measured in one process on the headline workload (bench.py `value`: 64 x 752x480, 600 features):
    t_dev    one step with the frames already in HBM          (kvfe_frontend_step_device, persist = 0)
    t_up     the upload of one step's frames alone            (2 x 64 x 752 x 480 bytes from pinned memory)
    t_staged one step through kvfe_frontend_step_staged        (upload inside the loop)
and asserted:  t_staged < 0.85 x (t_up + t_dev)   -- serial execution gives 1.0 x, full overlap max(t_up, t_dev).
Round 6, MI355X: t_dev 0.79 - 0.87 ms, t_up 0.83 - 0.86 ms, t_staged 1.17 ms = 0.69 - 0.71 x the sum.
Each time is the best of three regions so that one slow region of a shared box does not fail the suite."""
import time

import numpy as np
import pytest

from kimera_vio_amd import workloads
from kimera_vio_amd import frontend as F

pytestmark = pytest.mark.gpu

N_STEPS, N_WARM, N_REG = 26, 8, 3


def _regions(run_step, sync, total):
    best = None
    k = 0
    for _ in range(N_WARM):
        run_step(k)
        k += 1
    sync()
    for _ in range(N_REG):
        t0 = time.perf_counter()
        for _ in range(N_STEPS):
            run_step(k)
            k += 1
        sync()
        dt = (time.perf_counter() - t0) / N_STEPS
        best = dt if best is None else min(best, dt)
    assert k <= total
    return best


def test_staged_upload_overlaps_the_step_at_64_streams():
    import torch
    wl = workloads.build("c3", mode="kf")
    B = wl.batch
    assert B == 64
    total = N_WARM + N_REG * N_STEPS
    steps = wl.plan(total)
    lefts, rights = wl.replicated()
    dev = torch.device("cuda", 0)

    # frames resident in HBM
    d_left = torch.from_numpy(lefts).to(dev)
    d_right = torch.from_numpy(rights).to(dev)
    torch.cuda.synchronize()
    ctx = F.Context(wl.left, wl.right, wl.params, batch=B, device_frames_persist=0)
    try:
        plan = [wl.batch_inputs(ctx, st) for st in steps]
        t_dev = _regions(lambda k: ctx.step_device(d_left[steps[k][0]].data_ptr(), d_right[steps[k][0]].data_ptr(), plan[k]),
                         ctx.synchronize, total)
    finally:
        ctx.close()
    del d_left, d_right

    # the link alone
    nbytes = 2 * B * wl.width * wl.height
    h = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    t_up = _regions(lambda k: d.copy_(h, non_blocking=True), torch.cuda.synchronize, total)
    del h, d

    # frames uploaded inside the loop
    ctx = F.Context(wl.left, wl.right, wl.params, batch=B)
    try:
        for sl in range(wl.ring):
            a, b = ctx.staging_buffers(sl)
            a[:] = lefts[sl]
            b[:] = rights[sl]
        plan = [wl.batch_inputs(ctx, st) for st in steps]
        t_staged = _regions(lambda k: ctx.step_staged(steps[k][0], plan[k]), ctx.synchronize, total)
    finally:
        ctx.close()
    ratio = t_staged / (t_up + t_dev)
    print("staged overlap: t_dev %.3f ms  t_up %.3f ms  t_staged %.3f ms  ratio %.3f" % (1e3 * t_dev, 1e3 * t_up, 1e3 * t_staged, ratio))
    assert t_staged < 0.85 * (t_up + t_dev), (
        "the staged step no longer overlaps its upload: %.3f ms against %.3f (upload alone) + %.3f (device-resident step)"
        % (1e3 * t_staged, 1e3 * t_up, 1e3 * t_dev))
