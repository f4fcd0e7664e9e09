#!/usr/bin/env python3
"""bench.py — stereo-pairs/sec of the MI355X-native stereo front-end (detect + track + match).

One "step" = one pass of the hot path over one batch of synthetic input: every one of the `batch`
independent 752x480 stereo streams of this GPU advances by one stereo pair through
   pyramid -> (gyro-predicted) pyramidal LK track -> keyframe decision -> masked Shi-Tomasi detect
   + ANMS + cornerSubPix -> undistort-rectify L/R -> epipolar SSD stereo match + depth
with the inputs already resident in HBM (a ring of pre-generated frames).  N GPUs = N processes
(torchrun), each with its own context and its own `batch` streams (weak scaling); the only
collective is the RCCL barrier that brackets the timed region.

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for the roofline accounting.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=64, help="independent stereo streams per GPU")
    ap.add_argument("--width", type=int, default=752)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--features", type=int, default=600)
    ap.add_argument("--klt-max-level", type=int, default=2, help="2 = 3-level pyramid")
    ap.add_argument("--mode", choices=["kf", "nominal"], default="kf",
                    help="kf: every frame is a keyframe (all stages every pair, headline); "
                         "nominal: reference cadence (keyframe every 0.2 s = 4th frame)")
    ap.add_argument("--ransac", type=int, default=1, choices=[0, 1],
                    help="useRANSAC of params/Euroc/FrontendParams.yaml (1 = as shipped: 2-point mono + "
                         "1-point stereo geometric outlier rejection on every keyframe)")
    ap.add_argument("--mono-2point", type=int, default=1, choices=[0, 1],
                    help="ransac_use_2point_mono (0: the 5-point problem of params/D455)")
    ap.add_argument("--stereo-1point", type=int, default=1, choices=[0, 1],
                    help="ransac_use_1point_stereo (0: the 3-point Arun problem of params/D455)")
    ap.add_argument("--ring", type=int, default=6, help="distinct frames per stream (ping-pong)")
    ap.add_argument("--unique-streams", type=int, default=8)
    ap.add_argument("--groups", type=int, default=0,
                    help="stream groups per context (0 = library default for the batch size)")
    ap.add_argument("--cpu-baseline-frames", type=int, default=600)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-events", action="store_true",
                    help="do not record per-stage HIP events in the timed region (no roofline object)")
    ap.add_argument("--stage-event-stride", type=int, default=4,
                    help="every N-th timed step records per-stage HIP events (roofline / stage breakdown)")
    ap.add_argument("--no-single-stream", action="store_true")
    ap.add_argument("--no-dense", action="store_true", help="skip the dense-stereo (SGBM) leg")
    return ap.parse_args()


def make_cameras(P, G, w, h):
    L = P.load_camera_params(os.path.join(G, "params_euroc", "LeftCameraParams.yaml"))
    R = P.load_camera_params(os.path.join(G, "params_euroc", "RightCameraParams.yaml"))
    if (w, h) != (L.width, L.height):
        sx = w / L.width
        for cam in (L, R):
            cx0, cy0 = cam.width / 2.0, cam.height / 2.0
            cam.intrinsics[0] *= sx
            cam.intrinsics[1] *= sx
            cam.intrinsics[2] = w / 2.0 + (cam.intrinsics[2] - cx0) * sx
            cam.intrinsics[3] = h / 2.0 + (cam.intrinsics[3] - cy0) * sx
            cam.width, cam.height = w, h
    return L, R


def ping_pong(i, n):
    if n == 1:
        return 0
    period = 2 * (n - 1)
    j = i % period
    return j if j < n else period - j


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    from kimera_vio_amd import frontend as F
    from kimera_vio_amd import params as P
    from kimera_vio_amd import sharding, synth

    rank, local_rank, world = sharding.env_rank()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libkvfe has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    G = os.path.join(ROOT, "tests", "golden")
    W, H, B = args.width, args.height, args.batch
    L, R = make_cameras(P, G, W, H)
    p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=args.ransac)
    p.detector.max_features_per_frame = args.features
    p.tracker.klt_max_level = args.klt_max_level
    p.tracker.ransac_use_2point_mono = args.mono_2point
    p.tracker.ransac_use_1point_stereo = args.stereo_1point

    # ---- synthetic input ring, resident in HBM ---------------------------------------------------
    U = max(1, min(args.unique_streams, B))
    T = args.ring
    R1 = np.array(F.compute_rectification(L, R).R1).reshape(3, 3)
    streams = [synth.RigStream(L, R, seed=100 * rank + u, rect_R1=R1) for u in range(U)]
    lefts = np.empty((T, B, H, W), np.uint8)
    rights = np.empty((T, B, H, W), np.uint8)
    for t in range(T):
        for u in range(U):
            l, r = streams[u].frame(t)
            lefts[t, u::U] = l
            rights[t, u::U] = r
    d_left = torch.from_numpy(lefts).to(dev)
    d_right = torch.from_numpy(rights).to(dev)
    torch.cuda.synchronize()

    ctx = F.Context(L, R, p, batch=B, device=local_rank, stream_groups=args.groups)
    dt_ns = 50_000_000  # 20 Hz

    def frame_inputs(step_idx, kf_t):
        t = ping_pong(step_idx, T)
        Rs = [synth.rig_keyframe_R_cur(streams[s % U], kf_t, t) for s in range(B)]
        force = 1 if args.mode == "kf" else 0
        return t, ctx.make_inputs([step_idx * dt_ns] * B, Rs, [force] * B)

    # pre-compute inputs for all steps (host work outside the timed region)
    total = args.warmup + args.steps
    plan = []
    kf_t = 0
    last_kf_step = 0
    for i in range(total):
        t, inp = frame_inputs(i, kf_t)
        plan.append((t, inp))
        is_kf = (args.mode == "kf") or i == 0 or (i - last_kf_step) * dt_ns >= p.min_intra_keyframe_time_ns
        if is_kf:
            kf_t, last_kf_step = t, i

    def run(i0, i1):
        for i in range(i0, i1):
            t, inp = plan[i]
            ctx.step_device(d_left[t].data_ptr(), d_right[t].data_ptr(), inp)

    def barrier():
        sharding.barrier(dist, world)  # RCCL: lock-step only, no data-path collective
        ctx.synchronize()
        torch.cuda.synchronize()

    run(0, args.warmup)
    barrier()
    if not args.no_stage_events:
        ctx.profile_enable(args.stage_event_stride)
    t0 = time.perf_counter()
    run(args.warmup, total)
    t_enq = time.perf_counter()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    elapsed, pairs = sharding.reduce_timing(dist, world, elapsed, B * args.steps, device=dev)

    # sanity of the measured work: every stream produced keypoints and stereo matches
    out0 = ctx.get_output(0)
    outl = ctx.get_output(B - 1)
    assert out0["n_keypoints"] > 0 and outl["n_keypoints"] > 0, "front-end produced no keypoints"
    n_valid = int((out0["right_status"] == 0).sum()) if out0["is_keyframe"] else -1

    # PCIe-inclusive rates (host buffers handed over at the boundary): reported, never `value`.
    #  staged : the data provider writes into the context's pinned slots, the upload runs on a copy
    #           stream and overlaps the previous step (kvfe_frontend_step_staged)
    #  pageable: kvfe_frontend_step_host from ordinary host memory, copies on the compute stream
    pcie = None
    if rank == 0 and world == 1 and not args.no_single_stream:
        hl = np.ascontiguousarray(lefts[:3])
        hr = np.ascontiguousarray(rights[:3])
        ctx.reset()
        n_h = 12
        for sl in range(3):  # "decoded" frames sit in the pinned slots before the clock starts
            a, b = ctx.staging_buffers(sl)
            a[:] = hl[sl]
            b[:] = hr[sl]
        for i in range(3):
            ctx.step_staged(i % 3, plan[i][1])
        ctx.synchronize()
        th = time.perf_counter()
        for i in range(3, 3 + n_h):
            ctx.step_staged(i % 3, plan[i][1])
        ctx.synchronize()
        staged = B * n_h / (time.perf_counter() - th)
        ctx.reset()
        for i in range(2):
            ctx.step_host(hl[i % 2], hr[i % 2], plan[i][1])
        ctx.synchronize()
        th = time.perf_counter()
        for i in range(2, 8):
            ctx.step_host(hl[i % 2], hr[i % 2], plan[i][1])
        ctx.synchronize()
        pcie = {"value": round(staged, 2), "unit": "stereo-pairs/s",
                "note": "kvfe_frontend_step_staged: pinned staging slots, H2D upload of every frame inside the "
                        "timed region on a copy stream overlapping the previous step",
                "pageable_value": round(B * 6 / (time.perf_counter() - th), 2)}
    ctx.close()

    value = pairs / elapsed

    # ---- roofline, from HIP events recorded on the context's own stream inside the timed region ----
    # Every dense (image-sized) kernel is priced against the HBM roofline with its ALGORITHMIC bytes
    # per launch; `roofline` is the dominant dense kernel by time, `roofline_kernels`
    # lists all of them.  `traffic` = HBM-side bytes per launch from the rocprofv3 PMC passes of the
    # same command (profiles/pmc_traffic_latest.json: 2 x FETCH_SIZE + WRITE_SIZE, the gfx950
    # correction of MI355X_MICROARCH.md for wide coalesced reads -> an upper bound), null when the
    # committed counters were taken on another workload.
    stages = prof["stages"]
    ns = max(prof["n_samples"], 1)          # launches recorded per stage (all stream groups)
    groups = max(prof["n_groups"], 1)       # launches per step
    pmc = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")) as f:
            pmc = json.load(f)
    except OSError:
        pass
    pmc_ok = pmc is not None and (B, W, H, args.features, groups) == (64, 752, 480, 600, 1)
    pmc_names = {"pyramid": ("pyrdown_kernel", 2), "mineig_localmax": ("mineig_localmax_kernel<false>", 1),
                 "rectify": ("rectify_kernel<true>", 1)}

    def traffic_of(stage):
        if not pmc_ok or stage not in pmc_names:
            return None
        k, launches = pmc_names[stage]
        if k not in pmc:
            return None
        return round((2.0 * pmc[k]["fetch_kb"] + pmc[k]["write_kb"]) * 1024.0 * launches)

    dense = {k: v for k, v in stages.items() if v["alg_bytes"] > 0 and v["ms_total"] > 0}
    kernels = []
    for name, v in dense.items():
        avg_ms = v["ms_total"] / ns
        ach = v["alg_bytes"] / (avg_ms * 1e-3) / 1e9
        kernels.append({"kernel": name, "bound": "hbm", "achieved": round(ach, 2), "peak": 8000.0,
                        "unit": "GB/s", "frac": round(ach / 8000.0, 5), "traffic": traffic_of(name),
                        "alg_bytes_per_launch": v["alg_bytes"], "avg_launch_ms": round(avg_ms, 5)})
    kernels.sort(key=lambda r: -r["avg_launch_ms"])
    roofline = dict(kernels[0]) if kernels else None
    # per step: the launches of all stream groups added up (they overlap on the GPU, so the sum
    # exceeds ms_per_step when groups > 1)
    stage_ms = {k: round(v["ms_total"] / ns * groups, 5) for k, v in stages.items()}
    # the sparse (per-keypoint) kernels are VALU-issue / latency bound, not HBM bound: their share of
    # the step and the end-to-end algorithmic traffic are reported for context
    n_px = float(W) * H
    alg_pair = 7.33 * n_px  # SURVEY.md §8d, lambda recomputed instead of materialised
    e2e = {"alg_bytes_per_pair": round(alg_pair), "achieved_GBps": round(alg_pair * (pairs / elapsed) / 1e9, 2),
           "frac_of_hbm_peak": round(alg_pair * (pairs / elapsed) / 8e12, 5)}

    result = {
        "metric": "stereo-pairs/sec front-end (detect+track+match) @752x480",
        "value": round(value, 2), "unit": "stereo-pairs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/int32+f32",
        "data": f"synthetic ({U} seeded streams x {T} frames per GPU rendered through the calibrated stereo rig, "
                f"replicated to {B} streams)",
        "config": {"workload": f"batched {B} synthetic {W}x{H} stereo streams per GPU, {args.features} "
                               f"features, ANMS binning on, {args.klt_max_level + 1}-level LK, "
                               f"useRANSAC={args.ransac}, mode={args.mode}", "batch_per_gpu": B, "width": W, "height": H,
                   "features": args.features, "mode": args.mode, "use_ransac": args.ransac,
                   "stream_groups": groups,
                   "parallelism": f"streams x{world}"},
        "roofline": roofline,
        "roofline_kernels": kernels,
        "end_to_end_traffic": e2e,
        "stage_ms_per_step_summed_over_groups": stage_ms,
        "pcie_inclusive": pcie,
        "host_enqueue_ms_per_step": round(1e3 * (t_enq - t0) / args.steps, 4),
        "check": {"keypoints_stream0": int(out0["n_keypoints"]), "valid_stereo_stream0": n_valid},
    }

    if rank == 0 and world == 1 and not args.no_single_stream:
        result["single_stream"] = single_stream(F, P, synth, L, R, p, torch, dev, args)
    if rank == 0 and world == 1 and not args.no_dense:
        result["dense_stereo"] = dense_stereo(F, P, synth, L, R, p, dev, args)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(P, synth, L, R, p, args)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def single_stream(F, P, synth, L, R, p, torch, dev, args):
    """BASELINE config[1]: one stream, 300 features, 3-level LK (latency-bound: one launch chain
    per pair)."""
    import copy
    p1 = copy.deepcopy(p)
    p1.detector.max_features_per_frame = 300
    st = synth.RigStream(L, R, seed=4242, rect_R1=np.array(F.compute_rectification(L, R).R1).reshape(3, 3))
    T = 6
    fr = [st.frame(t) for t in range(T)]
    dl = torch.from_numpy(np.stack([f[0] for f in fr])[:, None]).to(dev)
    dr = torch.from_numpy(np.stack([f[1] for f in fr])[:, None]).to(dev)
    ctx = F.Context(L, R, p1, batch=1, device=dev.index)
    steps, warm = 200, 20
    plan = []
    kf_t = 0
    for i in range(steps + warm):
        t = ping_pong(i, T)
        plan.append((t, ctx.make_inputs([i * 50_000_000], [synth.rig_keyframe_R_cur(st, kf_t, t)], [1])))
        kf_t = t
    for i in range(warm):
        ctx.step_device(dl[plan[i][0]].data_ptr(), dr[plan[i][0]].data_ptr(), plan[i][1])
    ctx.synchronize()
    t0 = time.perf_counter()
    for i in range(warm, warm + steps):
        ctx.step_device(dl[plan[i][0]].data_ptr(), dr[plan[i][0]].data_ptr(), plan[i][1])
    ctx.synchronize()
    el = time.perf_counter() - t0
    n = ctx.get_output(0)["n_keypoints"]
    ctx.close()
    return {"workload": "single synthetic 752x480 stream, 300 features, 3-level LK, mode=kf",
            "value": round(steps / el, 2), "unit": "stereo-pairs/s", "ms_per_pair": round(1e3 * el / steps, 4),
            "keypoints": int(n)}


def dense_stereo(F, P, synth, L, R, p, dev, args):
    """SURVEY.md §8 a29 / BASELINE config[4] "dense stereo row": StereoMatcher::denseStereoReconstruction
    (cv::StereoSGBM MODE_HH with the reference's DenseStereoParams) on 8 rectified pairs per call.
    Time = HIP events around the kernel sequence inside libkvfe (the component call takes host
    buffers; its H2D / D2H copies are outside the events)."""
    from kimera_vio_amd import _abi as abi
    ctx = F.Context(L, R, p, batch=1, device=dev.index)
    n = 8
    st = [synth.RigStream(L, R, seed=900 + i) for i in range(2)]
    pairs = []
    for i in range(n):
        l, r = st[i % 2].frame(i // 2)
        pairs.append((ctx.undistort_rectify_image(0, l), ctx.undistort_rectify_image(1, r)))
    dp = abi.dense_stereo_params_default()
    lefts, rights = [a for a, _ in pairs], [b for _, b in pairs]
    ctx.dense_stereo_reconstruction(lefts, rights, dp)
    ctx.dense_profile_read()
    reps = 5
    for _ in range(reps):
        disp = ctx.dense_stereo_reconstruction(lefts, rights, dp)
    ms, cnt = ctx.dense_profile_read()
    ctx.close()
    W, H, D = args.width, args.height, dp.num_disparities
    w1 = W - (dp.min_disparity + D)
    vol = H * w1 * D * 2.0                      # one int16 cost volume
    npx = float(W) * H
    # algorithmic bytes per pair: pixel records, cost -> row sums -> C (write + read each), eight
    # path sweeps (C read, one sum write, seven sum read-modify-writes), selection, the small disparity passes
    agg_bytes = 8 * vol + vol + 7 * 2 * vol
    alg = (2 * npx + 16 * npx) + (16 * npx + vol) + 2 * vol + 2 * vol + agg_bytes + (vol + 2 * npx) + 24 * npx
    traffic = None
    try:   # HBM-side bytes of the same kernels from the committed rocprofv3 PMC passes (2 x FETCH_SIZE + WRITE_SIZE)
        with open(os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")) as f:
            pmc = json.load(f)
        tb = sum((2.0 * v["fetch_kb"] + v["write_kb"]) * 1024.0 for k, v in pmc.items()
                 if k.startswith(("dense_", "speckle_")))
        if tb > 0 and (W, H) == (752, 480):
            traffic = round(tb / n)
    except OSError:
        pass
    ach = alg / (ms / cnt * 1e-3) / 1e9
    valid = float(np.mean(disp[0] != (dp.min_disparity - 1) * 16))
    return {"workload": f"cv::StereoSGBM MODE_HH, block {dp.sad_window_size}, {D} disparities, {W}x{H}, "
                        f"{n} rectified pairs per call",
            "value": round(cnt / (ms * 1e-3), 2), "unit": "stereo-pairs/s", "ms_per_pair": round(ms / cnt, 4),
            "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(ach / 8000.0, 4), "traffic": traffic,
                         "alg_bytes_per_pair": round(alg), "alg_bytes_per_pair_aggregation": round(agg_bytes),
                         "note": "whole kernel sequence of one pair (HIP events inside libkvfe); traffic = PMC bytes "
                                 "per pair of the dense_* / speckle_* kernels"},
            "valid_fraction_pair0": round(valid, 3)}


def cpu_baseline(P, synth, L, R, p, args):
    """The CPU oracle (OpenCV-faithful scalar restatement, 1 thread = the reference's one front-end
    thread) on a bounded sample of the same workload: one stream, same parameters and mode."""
    import oracle_lib as O  # checker / baseline only
    from kimera_vio_amd import _abi as abi
    n = args.cpu_baseline_frames
    from kimera_vio_amd import frontend as F
    st = synth.RigStream(L, R, seed=100, rect_R1=np.array(F.compute_rectification(L, R).R1).reshape(3, 3))
    T = args.ring
    frames = [st.frame(t) for t in range(T)]
    lefts = np.stack([frames[ping_pong(i, T)][0] for i in range(n)])
    rights = np.stack([frames[ping_pong(i, T)][1] for i in range(n)])
    inputs = []
    kf_t = 0
    last_kf = 0
    for i in range(n):
        t = ping_pong(i, T)
        fi = abi.FrameInput()
        fi.timestamp_ns = i * 50_000_000
        Rm = synth.rig_keyframe_R_cur(st, kf_t, t).reshape(9)
        for k in range(9):
            fi.keyframe_R_cur_frame[k] = float(Rm[k])
        fi.force_keyframe = 1 if args.mode == "kf" else 0
        inputs.append(fi)
        if args.mode == "kf" or i == 0 or (i - last_kf) * 50_000_000 >= p.min_intra_keyframe_time_ns:
            kf_t, last_kf = t, i
    fe = O.Frontend(L, R, p)
    secs = fe.time_sequence(lefts, rights, inputs)
    out = {"value": round(n / secs, 3), "unit": "stereo-pairs/s", "cores": 1, "kind": "port",
           "sample": f"{n} consecutive stereo pairs of one synthetic {args.width}x{args.height} stream, "
                     f"{args.features} features, mode={args.mode}, {secs:.1f} s on one host core "
                     f"(scalar OpenCV-faithful restatement, no SIMD/IPP; host has {os.cpu_count()} cores)"}
    # independent streams on the host's cores (one single-threaded front-end per core, the
    # many-sequence mode on the CPU), bounded to a few seconds
    try:
        import multiprocessing as mp
        C_ = max(1, min(os.cpu_count() or 1, 64))
        m = max(20, min(n, 80))
        ctx = mp.get_context("fork")
        with ctx.Pool(C_, initializer=_cpu_worker_init, initargs=(L, R, p, lefts[:m], rights[:m], inputs[:m])) as pool:
            pool.map(_cpu_worker_run, range(C_))            # warm: library load, first-touch
            t0 = time.perf_counter()
            pool.map(_cpu_worker_run, range(C_))
            wall = time.perf_counter() - t0
        out["all_cores"] = {"value": round(C_ * m / wall, 2), "unit": "stereo-pairs/s", "cores": C_,
                            "sample": f"{C_} processes x {m} pairs of the same stream, {wall:.1f} s wall"}
    except Exception as e:  # the single-core figure above is the contract; this one is informative
        out["all_cores"] = {"error": repr(e)}
    return out


_CPU_W = {}


def _cpu_worker_init(L, R, p, lefts, rights, inputs):
    import oracle_lib as O
    _CPU_W.update(L=L, R=R, p=p, lefts=lefts, rights=rights, inputs=inputs, O=O)


def _cpu_worker_run(_):
    w = _CPU_W
    fe = w["O"].Frontend(w["L"], w["R"], w["p"])
    return fe.time_sequence(w["lefts"], w["rights"], w["inputs"])


if __name__ == "__main__":
    main()
