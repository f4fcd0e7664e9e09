"""round 5: what is the start-up transient of a front-end context (the first ~30 steps after context
creation run 20-35 % slow, every fourth 20-step region 12 % faster)?  Regions of `R` steps of the headline workload, each
bracketed by a full synchronisation, under conditions that separate the candidates:
  fresh      : right after kvfe_create (seconds of host set-up during which the GPU idles)
  idle2s     : the SAME warm context after 2 s of doing nothing        -> an idle device (clock / power state) is slow again?
  busy       : 2 s idle, then 150 ms of an unrelated torch kernel loop, then the regions -> a busy device is fast at once?
  smalldma   : fresh context with KVFE_OUT_TRANSFER_BYTES=4096 (no 5 MB output transfer per step) -> the output path?
Prints per condition the k pairs/s of every region, and the engine clock (rocm-smi / sysfs) around them."""
import glob
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from kimera_vio_amd import frontend as F, workloads as WL  # noqa: E402


def sclk():
    out = []
    for p in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            cur = [l.strip() for l in open(p) if "*" in l]
            out.append(cur[0] if cur else "?")
        except OSError:
            pass
    return out[:2]


def regions(ctx, d_left, d_right, plan, i0, n_regions, R, B):
    vals = []
    for r in range(n_regions):
        ctx.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(i0 + r * R, i0 + (r + 1) * R):
            t, inp = plan[i]
            ctx.step_device(d_left[t].data_ptr(), d_right[t].data_ptr(), inp)
        ctx.synchronize()
        vals.append(round(B * R / (time.perf_counter() - t0) / 1e3, 1))
    return vals, i0 + n_regions * R


def main():
    R = int(os.environ.get("PROBE_R", "10"))
    dev = torch.device("cuda", 0)
    wl = WL.build("c3", mode="kf", use_ransac=1)
    B = wl.batch
    lefts, rights = wl.replicated()
    d_left, d_right = torch.from_numpy(lefts).to(dev), torch.from_numpy(rights).to(dev)
    torch.cuda.synchronize()
    persist = int(os.environ.get("PROBE_PERSIST", "0"))

    def make():
        c = F.Context(wl.left, wl.right, wl.params, batch=B, device=0, device_frames_persist=persist)
        plan = [(st[0], wl.batch_inputs(c, st)) for st in wl.plan(400)]
        return c, plan

    t_c = time.perf_counter()
    ctx, plan = make()
    print("create + plan: %.2f s, sclk %s" % (time.perf_counter() - t_c, sclk()), flush=True)
    v, i = regions(ctx, d_left, d_right, plan, 0, 8, R, B)
    print("fresh    (regions of %d steps):" % R, v, "sclk", sclk(), flush=True)
    time.sleep(2.0)
    print("  after 2 s idle: sclk", sclk())
    v, i = regions(ctx, d_left, d_right, plan, i, 8, R, B)
    print("idle2s   :", v, "sclk", sclk(), flush=True)
    time.sleep(2.0)
    a = torch.randn(4096, 4096, device=dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        a = (a @ a).clamp_(-1, 1)
    torch.cuda.synchronize()
    v, i = regions(ctx, d_left, d_right, plan, i, 8, R, B)
    print("busy     :", v, "sclk", sclk(), flush=True)
    # back to back without any gap, as one long run split by events only at the end: do regions differ when nothing idles?
    v, i = regions(ctx, d_left, d_right, plan, i, 12, R, B)
    print("steady   :", v, flush=True)
    ctx.close()
    os.environ["KVFE_OUT_TRANSFER_BYTES"] = "4096"
    ctx, plan = make()
    v, i = regions(ctx, d_left, d_right, plan, 0, 8, R, B)
    print("smalldma :", v, "sclk", sclk(), flush=True)
    ctx.close()
    del os.environ["KVFE_OUT_TRANSFER_BYTES"]
    # a fresh context created while the device is kept busy by another stream
    a = torch.randn(4096, 4096, device=dev)
    ctx, plan = make()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        a = (a @ a).clamp_(-1, 1)
    torch.cuda.synchronize()
    v, i = regions(ctx, d_left, d_right, plan, 0, 8, R, B)
    print("fresh+busy:", v, "sclk", sclk(), flush=True)
    ctx.close()
    try:
        print(subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20).stdout[-600:])
    except Exception as e:  # noqa: BLE001
        print("rocm-smi:", e)


if __name__ == "__main__":
    main()
