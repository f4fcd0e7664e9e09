#!/usr/bin/env python3
"""bench.py — stereo-pairs/sec of the MI355X-native stereo front-end (detect + track + match).

One "step" = one pass of the hot path over one batch of synthetic input: every one of the `batch`
independent stereo streams of this GPU advances by one stereo pair through
   pyramid -> (gyro-predicted) pyramidal LK track -> keyframe decision -> geometric outlier rejection
   -> masked Shi-Tomasi detect + ANMS + cornerSubPix -> undistort-rectify L/R -> epipolar SSD stereo
   match + depth -> smart stereo measurements
with the inputs already resident in HBM (a ring of pre-generated frames).  N GPUs = N processes
(torch.distributed.run), each with its own context and its own streams; the only collective is the
RCCL barrier that brackets the timed region.  `python bench.py --gpus N` without a torchrun
environment re-executes itself under `python -m torch.distributed.run --nproc-per-node N`.

The workloads are built by kimera_vio_amd/workloads.py — the same call the parity tests
(tests/test_gpu_bench_configs.py) use, so what is timed here is compared with the oracle there.

Prints ONE JSON line (rank 0): `value` = BASELINE configs[2] (64 x 752x480, 600 features, every frame
a keyframe) unless --config says otherwise; the other BASELINE configurations ride along as legs
(`nominal`, `single_stream` = configs[1], `c5` = configs[4] incl. its dense-stereo row) at N = 1.
See DESIGN.md "Measurement" for the roofline accounting.
"""
import argparse
import json
import os
import socket
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ALL_LEGS = ("persist", "first_steps", "nominal", "single_stream", "c5", "klt4", "kf_realistic", "outputs", "spinonce", "alone", "dense", "dense_c5",
            "pcie", "input", "cpu")
HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s
PREWARM = 0                  # (round 4 ran 40 untimed steps in front of every leg: see run_frontend_leg)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=52,
                    help="steps per timed region; the default is two periods of the workload's feature-age cycle "
                         "(maxFeatureAge + 1 = 26 steps, see run_frontend_leg), so every region holds the same work")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=["c2", "c3", "c4", "c5"], default="c3",
                    help="which BASELINE config `value` is measured on: c3 = configs[2] (headline), c4 = configs[3] "
                         "(8 EuRoC sequences sharded over the GPUs, one stream per sequence, strong scaling), "
                         "c5 = configs[4]")
    ap.add_argument("--batch", type=int, default=None, help="independent stereo streams per GPU (config default)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--features", type=int, default=None)
    ap.add_argument("--klt-max-level", type=int, default=None, help="2 = 3-level pyramid")
    ap.add_argument("--mode", choices=["kf", "nominal"], default="kf",
                    help="kf: every frame is a keyframe (all stages every pair, headline); "
                         "nominal: reference cadence (keyframe every 0.2 s = 4th frame)")
    ap.add_argument("--ransac", type=int, default=1, choices=[0, 1],
                    help="useRANSAC of params/Euroc/FrontendParams.yaml (1 = as shipped)")
    ap.add_argument("--mono-2point", type=int, default=1, choices=[0, 1],
                    help="ransac_use_2point_mono (0: the 5-point problem of params/D455)")
    ap.add_argument("--stereo-1point", type=int, default=1, choices=[0, 1],
                    help="ransac_use_1point_stereo (0: the 3-point Arun problem of params/D455)")
    ap.add_argument("--ring", type=int, default=None, help="distinct frames per stream (ping-pong)")
    ap.add_argument("--unique-streams", type=int, default=None)
    ap.add_argument("--groups", type=int, default=0,
                    help="stream groups per context (0 = library default for the batch size)")
    ap.add_argument("--repeats", type=int, default=3,
                    help="the timed region of exactly --steps steps is run this many times; `value` is the median")
    ap.add_argument("--legs", default="all",
                    help="comma list of the extra N=1 legs to run: " + ",".join(ALL_LEGS) + " | all | none")
    ap.add_argument("--cpu-baseline-frames", type=int, default=600)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-events", action="store_true",
                    help="do not record per-stage HIP events in the timed region (no roofline object)")
    ap.add_argument("--stage-event-stride", type=int, default=8,
                    help="every N-th timed step records per-stage HIP events (roofline / stage breakdown)")
    ap.add_argument("--no-single-stream", action="store_true")
    ap.add_argument("--no-dense", action="store_true", help="skip the dense-stereo (SGBM) legs")
    ap.add_argument("--frames-persist", action="store_true",
                    help="kvfe_config.device_frames_persist = 1 on every leg: tracking reads frame k-1 from the caller's "
                         "resident ring instead of the context's own copy (round 4's headline configuration; default: the "
                         "C ABI default 0, with one `frames_persist` leg beside `value`)")
    ap.add_argument("--single-hip-stream", action="store_true",
                    help="kvfe_config.single_hip_stream = 1: every kernel of a step on one HIP stream (no side / output "
                         "stream) -- the stage times are then those of the kernels ALONE")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch plumbing only (tests/test_multi_rank.py, no GPU): become --gpus ranks, build each "
                         "rank's workload shard, barrier + timing reduction over gloo, print the shard map; "
                         "nothing is computed or timed and no `value` is printed")
    a = ap.parse_args()
    legs = set(ALL_LEGS) if a.legs == "all" else set(x for x in a.legs.split(",") if x and x != "none")
    unknown = legs - set(ALL_LEGS)
    if unknown:
        ap.error(f"unknown legs {sorted(unknown)}")
    if a.no_cpu_baseline:
        legs.discard("cpu")
    if a.no_single_stream:
        legs -= {"single_stream", "pcie"}
    if a.no_dense:
        legs -= {"dense", "dense_c5"}
    a.legs = legs
    return a


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def maybe_reexec_distributed(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: become N ranks, one process per GPU."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    os.execvpe(sys.executable, cmd, env)


def load_pmc():
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def load_valu():
    """{kernel: {SQ_INSTS_VALU: wave-instructions per launch, ...}} of the committed SQ pass of the c3 leg
    (tools/save_profiles_r2.sh -> profiles/pmc_valu_latest.json)"""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_valu_latest.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def valu_issue(valu, prefix, launch_ms, n_cu, clock_ghz):
    """what the kernel's own instruction count allows: a wave64 VALU instruction occupies one of the CU's four
    16-lane SIMDs for 4 cycles, so `insts x 4 / (4 n_cu)` cycles is the floor of a launch; frac = floor / measured"""
    insts = sum(v.get("SQ_INSTS_VALU", 0.0) for k, v in valu.items() if k.startswith(prefix))
    if not insts or launch_ms <= 0:
        return None
    floor_ms = insts * 4.0 / (4.0 * n_cu) / (clock_ghz * 1e6)
    res = {"wave_insts_per_launch": round(insts), "floor_ms": round(floor_ms, 5), "frac": round(floor_ms / launch_ms, 4),
           "note": "SQ_INSTS_VALU of the committed c3 PMC pass x 4 cycles / (4 SIMDs x CUs) at the peak engine clock"}
    # all instruction categories: SQ_ACTIVE_INST_ANY (quad-cycles a wave spends issuing anything) against the
    # SIMD issue slots of the measured launch; ~1 = the SIMDs issue one instruction per 4-cycle slot, the bound the
    # dense kernels and LK actually run into (profiles/r2_v5_lk_analysis.md); > 1 = categories issued side by side
    anyq = sum(v.get("SQ_ACTIVE_INST_ANY", 0.0) for k, v in valu.items() if k.startswith(prefix))
    if anyq:
        slots = launch_ms * 1e-3 * clock_ghz * 1e9 / 4.0 * (4.0 * n_cu)
        res["issue_slots_all_categories"] = round(anyq / slots, 4)
    return res


# kernel behind every dense (image-sized) stage: (rocprof kernel name, launches per step)
# third entry: FETCH_SIZE correction.  2 for kernels that read with wide (dwordx4) coalesced loads, the gfx950 correction
# of MI355X_MICROARCH.md; 1 for mineig2_kernel, whose source reads are byte loads (64 B requests: the counter is exact
# for those -- with x2 it would claim 3.7 x the image for a kernel whose row / column overlap is 16 %)
PMC_NAMES = {"pyramid": ("pyr", 1, 2.0), "mineig_localmax": ("mineig", 1, 1.0),
             "rectify": ("rectify_", 1, 2.0)}


def pmc_traffic(pmc_leg, stage):
    """HBM-side bytes per step of the stage's kernel(s) from the committed rocprofv3 PMC passes of the same
    leg alone (tools/rocpd_traffic.py: FETCH_SIZE and WRITE_SIZE in separate passes; per launch shape, the
    shapes of one step added up).  f x FETCH_SIZE + WRITE_SIZE with f from PMC_NAMES: MI355X_MICROARCH.md's gfx950
    correction (2) for wide coalesced reads, 1 for byte loads.  None when no counters are committed for
    this leg."""
    if not pmc_leg or stage not in PMC_NAMES:
        return None
    prefix, launches, ffac = PMC_NAMES[stage]
    tot, hit = 0.0, False
    for k, v in pmc_leg.items():
        if k.startswith(prefix):
            tot += (ffac * v["fetch_kb"] + v["write_kb"]) * 1024.0
            hit = True
    if not hit:
        return None
    return round(tot)


valu_ctx = None   # (per-kernel SQ counters, number of CUs, peak engine clock in GHz), set by main()
# kvfe_config execution options of every front-end leg.  Round 5: the C ABI's DEFAULTS (device_frames_persist = 0: the
# context keeps its own copy of every left frame and no caller pointer outlives a step) -- `value` describes what a caller
# gets without opting into anything.  The benchmark's frames do sit in a device ring that is never rewritten, so the
# caller's guarantee of `device_frames_persist` would hold: the `frames_persist` leg (and --frames-persist for every
# leg) runs with it, which is the configuration round 4's headline was measured in.
DEFAULT_CTX_KW = {"device_frames_persist": 0}


def run_frontend_leg(torch, F, dist, sharding, wl, dev, world, steps, warmup, repeats, groups, stage_stride,
                     pmc_leg=None, read_outputs=False, ctx_kw=None):
    """times `repeats` regions of exactly `steps` steps of the workload on this rank's GPU; returns the
    result dict of the leg (rank-reduced: MAX time over ranks, SUM of pairs)."""
    B, W, H = wl.batch, wl.width, wl.height
    lefts, rights = wl.replicated()
    d_left = torch.from_numpy(lefts).to(dev)
    d_right = torch.from_numpy(rights).to(dev)
    torch.cuda.synchronize()
    ctx = F.Context(wl.left, wl.right, wl.params, batch=B, device=dev.index, stream_groups=groups,
                    **(DEFAULT_CTX_KW if ctx_kw is None else ctx_kw))
    # Round 4 saw "the first ~30 steps run 20-35 % slow" and "every fourth 20-step region is 12 % faster" and put 40
    # untimed steps in front of every leg.  Round 5 (tools/r5/stall_probe.py): neither is a transient of the device or
    # of the runtime -- it is the workload.  Every feature of a synthetic stream is born at the bootstrap frame, so all of
    # them reach maxFeatureAge (25, Tracker.cpp:167-180) in the SAME step: every 26th step re-detects ~320 corners per
    # stream (2.7 ms instead of 1.3, and the two steps after it 1.8 / 1.65 ms); with maxFeatureAge = 1000 all regions of
    # 10 steps read 57.7 k +- 0.3 k pairs/s, with 13 the slow steps come twice as often.  A region of 20 or 40 steps holds
    # one or two of these bursts (hence the pattern); the default region is now 52 steps = two periods, PREWARM is gone,
    # and the rate over a context's first 30 steps (bootstrap frame + the first burst) is its own figure (`first_steps`).
    warmup = warmup + PREWARM
    total = warmup + repeats * steps
    plan = [(st[0], wl.batch_inputs(ctx, st)) for st in wl.plan(total)]   # host work outside the timed region
    # read_outputs: the consumer side inside the timed region -- after enqueuing step i the host copies the complete
    # output records of step i-1 of ALL streams out of the pinned ring (kvfe_frontend_get_outputs), the last step's after
    # the loop: every frame's keypoints, stereo arrays and measurements reach caller-owned memory
    obuf = ctx.output_buffers() if read_outputs else None
    n_read = [0]

    def run(i0, i1):
        for i in range(i0, i1):
            t, inp = plan[i]
            ctx.step_device(d_left[t].data_ptr(), d_right[t].data_ptr(), inp)
            if obuf is not None and i > i0:
                o = obuf.read(1)
                n_read[0] += o[0].n_keypoints
        if obuf is not None:
            o = obuf.read(0)
            n_read[0] += o[0].n_keypoints

    def barrier():
        sharding.barrier(dist, world)  # RCCL: lock-step only, no data-path collective
        ctx.synchronize()
        torch.cuda.synchronize()

    run(0, warmup)
    barrier()
    if stage_stride:
        ctx.profile_enable(stage_stride)
    times, enq = [], []
    new_corners, kf_steps = 0, 0
    for r in range(repeats):
        i0 = warmup + r * steps
        barrier()
        t0 = time.perf_counter()
        run(i0, i0 + steps)
        t_enq = time.perf_counter()
        barrier()
        t1 = time.perf_counter()
        el, pairs = sharding.reduce_timing(dist, world, t1 - t0, B * steps, device=dev)
        times.append(el)
        enq.append(t_enq - t0)
        o = ctx.get_output(0)             # outside the timed region: detection-side work per keyframe
        if o["is_keyframe"]:
            new_corners += o["n_detected"]
            kf_steps += 1
    prof = ctx.profile_read() if stage_stride else None
    ctx.profile_enable(False)
    out0, outl = ctx.get_output(0), ctx.get_output(B - 1)
    assert out0["n_keypoints"] > 0 and outl["n_keypoints"] > 0, "front-end produced no keypoints"
    ctx.close()
    del d_left, d_right

    med = statistics.median(times)
    res = {"value": round(pairs / med, 2), "unit": "stereo-pairs/s", "ms_per_step": round(1e3 * med / steps, 4),
           "repeats": {"n": repeats, "steps_each": steps,
                       "values": [round(pairs / t, 2) for t in times],
                       "best": round(pairs / min(times), 2), "median": round(pairs / med, 2)},
           "host_enqueue_ms_per_step": round(1e3 * statistics.median(enq) / steps, 4),
           "check": {"keypoints_stream0": int(out0["n_keypoints"]), "tracked_stream0": int(out0["n_tracked"]),
                     "new_corners_per_keyframe_stream0": (round(new_corners / kf_steps, 1) if kf_steps else None),
                     "valid_stereo_stream0": int((out0["right_status"] == 0).sum()) if out0["is_keyframe"] else -1}}
    n_px = float(W) * H
    rate = pairs / med
    res["end_to_end_traffic"] = {
        "note": "SURVEY.md 8d algorithmic bytes per pair x pairs/s vs the 8 TB/s HBM peak; 7.33 N = lambda "
                "recomputed (this design), 14.33 N = materialised fp32 lambda map",
        "alg_bytes_per_pair_7p33N": round(7.33 * n_px), "achieved_GBps_7p33N": round(7.33 * n_px * rate / 1e9, 2),
        "frac_of_hbm_peak_7p33N": round(7.33 * n_px * rate / 8e12, 5),
        "alg_bytes_per_pair_14p33N": round(14.33 * n_px), "achieved_GBps_14p33N": round(14.33 * n_px * rate / 1e9, 2),
        "frac_of_hbm_peak_14p33N": round(14.33 * n_px * rate / 8e12, 5)}
    if prof:
        stages = prof["stages"]
        ns = max(prof["n_samples"], 1)          # launches recorded per stage (all stream groups)
        g = max(prof["n_groups"], 1)            # launches per step
        kernels = []
        for name, v in stages.items():
            if v["alg_bytes"] > 0 and v["ms_active"] > 0 and v["active_launches"] > 0:
                # per launch that DID work: a keyframe-only stage is launched on every step but works only for the
                # streams whose flags say so (kvfe_stage_times: the flags of every sampled step are read back), so
                # both the time and the algorithmic bytes are averaged over the launches with >= 1 active stream
                nl = v["active_launches"]
                avg_ms = v["ms_active"] / nl
                alg = v["alg_bytes_per_stream"] * v["active_streams"] / nl
                ach = alg / (avg_ms * 1e-3) / 1e9
                kernels.append({"kernel": name, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBPS,
                                "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 5),
                                "traffic": pmc_traffic(pmc_leg, name),
                                "traffic_source": ("committed profiles/pmc_traffic_latest.json: rocprofv3 --pmc FETCH_SIZE / "
                                                   "WRITE_SIZE passes of this leg on the committed build, per launch -- "
                                                   "NOT measured in this run" if pmc_leg else None),
                                "alg_bytes_per_launch": round(alg), "avg_launch_ms": round(avg_ms, 5),
                                "launches_sampled": ns, "launches_with_work": nl,
                                "active_streams_per_working_launch": round(v["active_streams"] / nl, 2)})
                if pmc_leg is not None and valu_ctx and name in PMC_NAMES and B == 64 and W == 752:
                    vi = valu_issue(valu_ctx[0], PMC_NAMES[name][0], avg_ms, valu_ctx[1], valu_ctx[2])   # (all launches of the stage)
                    if vi:
                        kernels[-1]["valu_issue"] = vi
        # FIXED RULE (rounds 1, 2 and 4 on; round 3's last commit had switched it): `roofline` = the dense kernel with the
        # LONGEST launch inside the step; `roofline_dense_weighted` = sum of algorithmic bytes / sum of launch times of
        # all dense kernels of the step (the time-weighted fraction); `roofline_kernels` lists every one of them
        kernels.sort(key=lambda r: (-r["avg_launch_ms"], -r["alg_bytes_per_launch"]))
        res["roofline"] = dict(kernels[0]) if kernels else None
        res["roofline_kernels"] = kernels
        if kernels:
            tb = sum(k["alg_bytes_per_launch"] for k in kernels)
            tt = sum(k["avg_launch_ms"] for k in kernels)
            ach = tb / (tt * 1e-3) / 1e9
            res["roofline_dense_weighted"] = {
                "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBPS, 5), "alg_bytes": round(tb), "sum_launch_ms": round(tt, 5),
                "kernels": [k["kernel"] for k in kernels]}
        res["stage_ms_per_step_summed_over_groups"] = {k: round(v["ms_total"] / ns * g, 5) for k, v in stages.items()}
        if pmc_leg is not None and valu_ctx and B == 64 and W == 752 and "lk_track" in stages:
            lk_ms = stages["lk_track"]["ms_total"] / ns
            # (lk4_kernel from round 6 on; lk_kernel_sys with counters committed before it)
            vi = (valu_issue(valu_ctx[0], "lk4_kernel", lk_ms, valu_ctx[1], valu_ctx[2]) or
                  valu_issue(valu_ctx[0], "lk_kernel", lk_ms, valu_ctx[1], valu_ctx[2]))
            if vi:   # the largest kernel of the step is sparse (no HBM roofline); its VALU issue floor is reported, but the
                # launch is bound by more than that count (profiles/r6_analysis.md)
                res["largest_kernel"] = {"kernel": "lk_track", "avg_launch_ms": round(lk_ms, 5),
                                         "bound": "vector instruction issue weighted by instruction class (2.4 cycles for plain "
                                                  "VOP2 float / integer, 3.7-3.9 for everything else: profiles/r3_ubench_valu_rate.txt); "
                                                  "four points per wavefront, the sequential float chains OpenCV's summation order "
                                                  "dictates in every lane of a quad (k_lk4.hip); the launch ends on the waves whose "
                                                  "points run all 30 iterations on every level; profiles/r6_analysis.md",
                                         "valu_issue": vi}
        res["stream_groups"] = g
    return res


LINE_MAX_BYTES = 4096       # the driver's parser keeps a bounded line: everything else goes to bench_detail.json
LEG_SCALARS = ("frames_persist", "nominal", "kf_realistic", "outputs_inclusive", "single_stream",
               "single_stream_spinonce", "c5", "klt_max_level_4", "dense_stereo", "dense_stereo_c5", "pcie_inclusive",
               "first_steps")


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[: n - 1] + "~"


def compact_line(result):
    """The ONE stdout line of a run (round 5: round 4's line had grown to 22 kB and the driver's parser dropped it):
    the contract keys, ONE roofline dict, the weighted dense figure, the CPU baseline and one scalar per leg
    (stereo-pairs/s).  Always < LINE_MAX_BYTES; the complete result goes to bench_detail.json and to stderr."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "prewarm_steps_untimed", "ms_per_step",
            "higher_is_better", "scaling", "vs_baseline", "dtype")
    line = {k: result[k] for k in keep if k in result}
    line["data"] = _short(result.get("data", "synthetic"), 120)
    cfg = dict(result.get("config", {}))
    if "workload" in cfg:
        cfg["workload"] = _short(cfg["workload"], 160)
    line["config"] = cfg
    rf = result.get("roofline")
    if rf:
        line["roofline"] = {k: rf[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic",
                                                 "traffic_source", "alg_bytes_per_launch", "avg_launch_ms",
                                                 "frac_of_copy") if k in rf}
    rw = result.get("roofline_dense_weighted")
    if rw:
        line["roofline_dense_weighted"] = {k: rw[k] for k in ("bound", "achieved", "peak", "unit", "frac", "alg_bytes",
                                                                "sum_launch_ms", "frac_of_copy") if k in rw}
    if result.get("roofline_kernels"):
        line["roofline_kernels_frac"] = {k["kernel"]: k["frac"] for k in result["roofline_kernels"]}
        if all("frac_of_copy" in k for k in result["roofline_kernels"]):
            line["roofline_kernels_frac_of_copy"] = {k["kernel"]: k["frac_of_copy"] for k in result["roofline_kernels"]}
    if "hbm_copy_GBps" in result:
        line["hbm_copy_GBps"] = result["hbm_copy_GBps"]
    kr = result.get("kf_realistic")
    if isinstance(kr, dict) and "value" in kr:   # real frames: the honest workload, beside `value`
        line["kf_realistic_value"] = kr["value"]
        line["kf_realistic_ms_per_step"] = kr.get("ms_per_step")
    cb = result.get("cpu_baseline")
    if cb:
        c2 = {k: cb[k] for k in ("value", "unit", "cores", "kind") if k in cb}
        c2["sample"] = _short(cb.get("sample", ""), 200)
        ac = cb.get("all_cores") or {}
        if "value" in ac:
            c2["all_cores_value"], c2["all_cores"] = ac["value"], ac.get("cores")
            if "usable_cpus" in ac:   # (processes started vs processors the container's CPU quota lets them use)
                c2["all_cores_usable_cpus"] = ac["usable_cpus"]
        line["cpu_baseline"] = c2
    legs = {}
    for k in LEG_SCALARS:
        v = result.get(k)
        if isinstance(v, dict) and "value" in v:
            legs[k] = v["value"]
    d1 = (result.get("dense_stereo") or {}).get("one_pair_per_call") if isinstance(result.get("dense_stereo"), dict) else None
    if isinstance(d1, dict) and "value" in d1:   # (the single pair upstream calls the method with)
        legs["dense_stereo_1pair"] = d1["value"]
    ins = result.get("input_side")
    if isinstance(ins, dict) and "stereo_pairs_per_s_all_threads" in ins:   # host side: PNG decode on the usable CPUs
        legs["input_side_host_decode"] = max(ins["stereo_pairs_per_s_all_threads"],
                                             ins.get("stereo_pairs_per_s_all_threads_128_files_per_call", 0.0))   # (64 / 128 files per call)
    if legs:
        line["legs_pairs_per_s"] = legs
    for k in ("value_is", "device_warm_up_ok", "collective_backend", "detail"):
        if k in result:
            line[k] = _short(result[k], 120) if isinstance(result[k], str) else result[k]
    for v in line.values():   # every remaining free-text field of a nested dict is bounded too
        if isinstance(v, dict):
            for k2, v2 in v.items():
                if isinstance(v2, str):
                    v[k2] = _short(v2, 200)
    out = json.dumps(line)
    if len(out) >= LINE_MAX_BYTES:   # never: every free-text field above is cut; belt and braces for the parser
        for k in ("roofline_kernels_frac", "value_is", "collective_backend", "data"):
            line.pop(k, None)
            out = json.dumps(line)
            if len(out) < LINE_MAX_BYTES:
                break
    assert len(out) < LINE_MAX_BYTES, len(out)
    return out


def emit(result):
    """full result -> bench_detail.json (repo root; also gpurun_out/ when it exists) and stderr; compact line -> stdout"""
    full = json.dumps(result)
    wrote = []
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    f.write(full + "\n")
                wrote.append(os.path.relpath(os.path.join(d, "bench_detail.json"), ROOT))
            except OSError:
                pass
    result = dict(result, detail="full result: " + (", ".join(wrote) if wrote else "stderr") + " (and stderr)")
    print(full, file=sys.stderr, flush=True)
    print(compact_line(result), flush=True)


def main():
    args = parse()
    maybe_reexec_distributed(args)
    import torch
    import torch.distributed as dist

    from kimera_vio_amd import frontend as F
    from kimera_vio_amd import sharding
    from kimera_vio_amd import workloads as WL

    rank, local_rank, world = sharding.env_rank()
    if args.dry_run:
        return dry_run(args, dist, sharding, WL, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libkvfe has no CPU fallback")
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # first touch of a freshly provisioned box in a throw-away process (kimera_vio_amd/_warmup.py): EVERY local rank
    # warms its own device (on an 8-GPU node each device is touched for the first time by its own rank)
    from kimera_vio_amd._warmup import warm_up_device
    warmed = warm_up_device(attempts=2, device=local_rank)
    torch.cuda.set_device(local_rank)
    # under torchrun (also with ONE rank) the process group is RCCL: the barrier and the timing reduction then run the
    # same collective code on 1 GPU as on 8 (tests/test_gpu_rccl_r3.py runs exactly this at world size 1)
    under_torchrun = "WORLD_SIZE" in os.environ and "MASTER_ADDR" in os.environ
    if world > 1 or under_torchrun:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    DEFAULT_CTX_KW["device_frames_persist"] = 1 if args.frames_persist else 0
    if args.single_hip_stream:
        DEFAULT_CTX_KW["single_hip_stream"] = 1
    ctx_kw = None
    pmc = load_pmc()
    global valu_ctx
    props = torch.cuda.get_device_properties(dev)
    valu_ctx = (load_valu(), int(props.multi_processor_count), float(getattr(props, "clock_rate", 2400000)) / 1e6)

    kw = dict(use_ransac=args.ransac, mono_2point=args.mono_2point, stereo_1point=args.stereo_1point,
              batch=args.batch, features=args.features, klt_max_level=args.klt_max_level,
              unique=args.unique_streams, ring=args.ring, width=args.width, height=args.height)
    scaling = "weak"
    if args.config == "c4":
        # configs[3]: 8 EuRoC sequences, sequence q on rank q mod world (one per GPU at 8 GPUs); no collective
        seqs = sharding.shard_streams(8, world, rank)
        wl = WL.build("c4", mode=args.mode, rank=rank, sequences=seqs, **kw)
        scaling = "strong"
    else:
        wl = WL.build(args.config, mode=args.mode, rank=rank, **kw)
    # what a plain streaming copy reaches on THIS box in THIS process (kvfe_hbm_copy_probe: 1 GiB, 16 bytes per lane and
    # trip), before and after the main leg -- boxes of the pool differ by up to 2 x on the same binary
    probe0 = F.hbm_copy_probe(1 << 30, 10)
    stride = 0 if args.no_stage_events else args.stage_event_stride
    overridden = any(v is not None for v in (args.batch, args.features, args.klt_max_level, args.width, args.height))
    pmc_leg = None if (overridden or args.groups > 1) else pmc.get(f"{args.config}_{args.mode}")
    main_leg = run_frontend_leg(torch, F, dist, sharding, wl, dev, world, args.steps, args.warmup, args.repeats,
                                args.groups, stride, pmc_leg)
    probe1 = F.hbm_copy_probe(1 << 30, 10)
    copy_gbps = max(probe0["read_plus_write_GBps"], probe1["read_plus_write_GBps"])
    for r in [main_leg.get("roofline"), main_leg.get("roofline_dense_weighted")] + list(main_leg.get("roofline_kernels") or []):
        if r:   # the same achieved figure against the copy rate of this box (the `frac` keys stay against 8 TB/s)
            r["frac_of_copy"] = round(r["achieved"] / copy_gbps, 5)
    p = wl.params
    B, W, H = wl.batch, wl.width, wl.height
    src = (f"{wl.unique} seeded streams x {wl.ring} frames per GPU rendered through the calibrated stereo rig "
           f"(synth.RigStream), replicated to {B} streams, ring walked ping-pong" if wl.source == "rig" else
           f"{wl.unique} window(s) of {wl.ring} MicroEuroc frames per GPU (tests/golden/micro_euroc_f10_18.npz), "
           f"walked ping-pong")
    result = {
        "metric": f"stereo-pairs/sec front-end (detect+track+match) @{W}x{H}",
        "value": main_leg["value"], "unit": "stereo-pairs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "prewarm_steps_untimed": PREWARM, "ms_per_step": main_leg["ms_per_step"],
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "u8/int32+f32",
        "data": f"synthetic ({src})",
        "config": {"workload": f"BASELINE {args.config}: batched {B} {W}x{H} stereo streams per GPU, "
                               f"{p.detector.max_features_per_frame} features, ANMS binning on, "
                               f"{p.tracker.klt_max_level + 1}-level LK, useRANSAC={p.use_ransac}, mode={args.mode}",
                   "batch_per_gpu": B, "width": W, "height": H, "features": p.detector.max_features_per_frame,
                   "mode": args.mode, "use_ransac": p.use_ransac, "stream_groups": main_leg.get("stream_groups", 1),
                   "device_frames_persist": DEFAULT_CTX_KW["device_frames_persist"],
                   "parallelism": f"streams x{world}"},
        "value_is": f"median of {args.repeats} timed regions of exactly {args.steps} steps each",
        "device_warm_up_ok": bool(warmed),
        "hbm_copy_GBps": round(copy_gbps, 1),
        "hbm_copy_probe": {"what": "kvfe_hbm_copy_probe: 1 GiB device-to-device copy kernel (one 16-byte load + store per lane "
                                   "and trip), 10 copies under HIP events, read + write bytes / time; before and after the "
                                   "main leg, in this process; `frac_of_copy` = a kernel's achieved GB/s over the better of the two",
                           "before_main_leg_GBps": round(probe0["read_plus_write_GBps"], 1),
                           "after_main_leg_GBps": round(probe1["read_plus_write_GBps"], 1),
                           "frac_of_8TBps": round(copy_gbps / HBM_PEAK_GBPS, 4)},
    }
    for k in ("repeats", "roofline", "roofline_dense_weighted", "roofline_kernels", "largest_kernel", "end_to_end_traffic",
              "stage_ms_per_step_summed_over_groups", "host_enqueue_ms_per_step", "check"):
        if k in main_leg:
            result[k] = main_leg[k]

    solo = rank == 0 and world == 1
    if solo and "persist" in args.legs and not args.frames_persist:
        leg = run_frontend_leg(torch, F, dist, sharding, wl, dev, 1, args.steps, args.warmup, args.repeats, args.groups,
                               0, None, ctx_kw=dict(DEFAULT_CTX_KW, device_frames_persist=1))
        leg["workload"] = ("as `value` with kvfe_config.device_frames_persist = 1 (the caller guarantees that a step's frames "
                           "stay valid until the next step has completed: no level-0 copy, the rectify / match chain is "
                           "joined one step later) -- the configuration of round 4's headline")
        result["frames_persist"] = leg
    if solo and "first_steps" in args.legs:
        result["first_steps"] = first_steps_leg(torch, F, wl, dev)
    if solo and args.config == "c3" and args.mode == "kf" and "nominal" in args.legs:
        import dataclasses
        # (keyframes fall on every 4th step: a stride of 3 samples keyframe and non-keyframe steps alike)
        leg = run_frontend_leg(torch, F, dist, sharding, dataclasses.replace(wl, mode="nominal"), dev, 1, args.steps,
                               args.warmup, args.repeats, args.groups, 3 if stride else 0, pmc.get("c3_nominal"))
        leg["workload"] = ("same streams, reference cadence: track every frame, detect + rectify + match only on "
                           "keyframes (every min_intra_keyframe_time = 0.2 s = 4th frame)")
        result["nominal"] = leg
    if solo and "single_stream" in args.legs:
        w2 = WL.build("c2", mode="kf", use_ransac=args.ransac)
        leg = run_frontend_leg(torch, F, dist, sharding, w2, dev, 1, 200, 20, args.repeats, 0, 0)
        leg["workload"] = ("BASELINE configs[1]: single EuRoC 752x480 stream (MicroEuroc frames), 300 features, "
                           "3-level LK, mode=kf; latency-bound: one launch chain per pair")
        leg["ms_per_pair"] = leg["ms_per_step"]
        result["single_stream"] = leg
    if solo and "c5" in args.legs and args.config != "c5":
        w5 = WL.build("c5", mode="kf", use_ransac=args.ransac)
        leg = run_frontend_leg(torch, F, dist, sharding, w5, dev, 1, max(20, args.steps // 2), args.warmup, args.repeats,
                               0, stride, pmc.get("c5_kf"))
        leg["workload"] = (f"BASELINE configs[4]: batched {w5.batch} {w5.width}x{w5.height} streams ({w5.unique} "
                           f"unique), 1000 features, 4-level LK, useRANSAC={args.ransac}, mode=kf")
        result["c5"] = leg
    if solo and "klt4" in args.legs and args.config == "c3":
        # "also one run at the shipped value" -- params/Euroc/FrontendParams.yaml klt_max_level: 4 (five levels)
        w4 = WL.build("c3", mode="kf", use_ransac=args.ransac, klt_max_level=4)
        leg = run_frontend_leg(torch, F, dist, sharding, w4, dev, 1, max(20, args.steps // 2), args.warmup, args.repeats,
                               0, stride, None)
        leg["workload"] = "BASELINE configs[2] at the shipped klt_max_level: 4 (5-level LK pyramid), otherwise as `value`"
        result["klt_max_level_4"] = leg
    if solo and "kf_realistic" in args.legs and args.config == "c3":
        we = WL.build("c3e", mode="kf", use_ransac=args.ransac)
        leg = run_frontend_leg(torch, F, dist, sharding, we, dev, 1, max(20, args.steps // 2), args.warmup, args.repeats,
                               0, stride, None)
        leg["workload"] = (f"configs[2] on real frames: {we.batch} streams replaying {we.unique} offset windows of "
                           f"{we.ring} MicroEuroc frames, 600 features, 3-level LK, every frame a keyframe -- detection "
                           f"and cornerSubPix at the loss rate of real images (see check.new_corners_per_keyframe_stream0)")
        result["kf_realistic"] = leg
    if solo and "outputs" in args.legs and args.config == "c3":
        # the `value` workload with every frame's outputs read by the host inside the timed region
        leg = run_frontend_leg(torch, F, dist, sharding, wl, dev, 1, args.steps, args.warmup, args.repeats, args.groups,
                               0, None, read_outputs=True, ctx_kw=ctx_kw)
        ref = run_frontend_leg(torch, F, dist, sharding, wl, dev, 1, args.steps, args.warmup, args.repeats, args.groups,
                               0, None, ctx_kw=ctx_kw)
        leg["workload"] = ("as `value`, plus: the complete kvfe_frame_output of every stream and every step (frame tables, "
                           "stereo arrays, smart stereo measurements, TrackerStatusSummary) is copied into caller-owned "
                           "host memory inside the timed region -- step i-1's records while step i runs "
                           "(kvfe_frontend_get_outputs, steps_back = 1), the last step's at the end")
        leg["no_readback_value"] = ref["value"]
        leg["vs_no_readback"] = round(leg["value"] / ref["value"], 4)
        result["outputs_inclusive"] = leg
    if solo and "spinonce" in args.legs:
        result["single_stream_spinonce"] = spinonce_leg(torch, F, dist, sharding, WL, dev, args)
    if solo and "alone" in args.legs and args.config == "c3":
        # the same step with every kernel in order on ONE HIP stream (kvfe_config.single_hip_stream): what the three
        # dense kernels take when nothing runs beside them -- `roofline` above prices them inside the overlapped step
        leg = run_frontend_leg(torch, F, dist, sharding, wl, dev, 1, 12, 4, 1, 0, 1, pmc.get("c3_" + args.mode),
                               ctx_kw=dict(DEFAULT_CTX_KW, single_hip_stream=1))
        result["dense_kernels_alone"] = {
            "workload": "as `value`, kvfe_config.single_hip_stream = 1 (no side stream, no output stream), HIP events "
                        "around every stage of 12 steps",
            "ms_per_step": leg["ms_per_step"],
            "stage_ms_per_step": leg.get("stage_ms_per_step_summed_over_groups"),
            "kernels": [{k: r[k] for k in ("kernel", "avg_launch_ms", "achieved", "frac", "alg_bytes_per_launch") if k in r}
                        for r in leg.get("roofline_kernels", [])],
            "weighted": leg.get("roofline_dense_weighted")}
    if solo and "dense" in args.legs:
        result["dense_stereo"] = dense_stereo(F, WL, 752, 480, dev, pmc.get("dense"))
    if solo and "dense_c5" in args.legs:
        result["dense_stereo_c5"] = dense_stereo(F, WL, 1280, 720, dev, pmc.get("dense_c5"))
    if solo and "pcie" in args.legs and args.config == "c3":
        result["pcie_inclusive"] = pcie_inclusive(F, wl, torch)
    if solo and "input" in args.legs:
        result["input_side"] = input_side()
    if solo and "cpu" in args.legs:
        result["cpu_baseline"] = cpu_baseline(wl, args)
    if dist.is_initialized():
        result["collective_backend"] = f"{dist.get_backend()} (RCCL), world {dist.get_world_size()}: barrier + timing all_reduce"
    if rank == 0:
        emit(result)
    if dist.is_initialized():
        dist.destroy_process_group()


def first_steps_leg(torch, F, wl, dev):
    """the first 30 steps of a fresh context: the bootstrap frame (600 corners per stream detected and refined), 24 tracked
    frames, the first maxFeatureAge burst -- what a front-end that has just been started delivers"""
    lefts, rights = wl.replicated()
    d_left, d_right = torch.from_numpy(lefts).to(dev), torch.from_numpy(rights).to(dev)
    torch.cuda.synchronize()
    ctx = F.Context(wl.left, wl.right, wl.params, batch=wl.batch, device=dev.index, **DEFAULT_CTX_KW)
    plan = [(st[0], wl.batch_inputs(ctx, st)) for st in wl.plan(30)]
    ctx.synchronize()
    t0 = time.perf_counter()
    for t, inp in plan:
        ctx.step_device(d_left[t].data_ptr(), d_right[t].data_ptr(), inp)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    ctx.close()
    return {"value": round(wl.batch * 30 / dt, 2), "unit": "stereo-pairs/s", "ms_per_step": round(1e3 * dt / 30, 4),
            "workload": "steps 0..29 of a fresh context on the `value` workload (no warm-up): bootstrap frame + first feature-age burst"}


def dry_run(args, dist, sharding, WL, rank, world):
    """--dry-run: the N-rank launch path without a GPU (gloo).  No front-end call is made."""
    import torch
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1:
        dist.init_process_group("gloo")
    if args.config == "c4":
        seqs = sharding.shard_streams(8, world, rank)
        wl = WL.build("c4", mode=args.mode, rank=rank, sequences=seqs, rect_R1=np.eye(3))
    else:
        seqs = None
        wl = WL.build(args.config, mode=args.mode, rank=rank, batch=args.batch or 2, unique=2, ring=2, width=64,
                      height=48, rect_R1=np.eye(3))
    sharding.barrier(dist, world)
    el, pairs = sharding.reduce_timing(dist, world, 1.0 + rank, wl.batch * args.steps)
    mine = {"rank": rank, "batch": wl.batch, "sequences": seqs, "first_pixel_sum": int(wl.lefts[0].sum())}
    gathered = [None] * world
    if world > 1:
        dist.all_gather_object(gathered, mine)
    else:
        gathered = [mine]
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "elapsed_max": el, "pairs_sum": pairs,
                          "scaling": "strong" if args.config == "c4" else "weak", "ranks": gathered}))
    if world > 1:
        dist.destroy_process_group()


def spinonce_leg(torch, F, dist, sharding, WL, dev, args):
    """BASELINE configs[1] through the reference's own call: shim::StereoVisionImuFrontend::spinOnce(StereoImuSyncPacket&&)
    -> StereoFrontendOutput (include/kvfe_kimera_shim.hpp), one stream, host images, the front-end's own keyframe
    decisions, EVERY call returning its complete output -- a synchronous call per frame, as Kimera-VIO's front-end thread
    makes it.  Driven by the plain-g++ host program tests/cpp/shim_check (bench mode).  Beside it the same stream and
    cadence through kvfe_frontend_step_device with no read-back and no per-frame synchronisation."""
    import subprocess
    import tempfile
    from kimera_vio_amd import _abi as abi
    cpp = os.path.join(ROOT, "tests", "cpp")
    subprocess.run(["make", "-C", cpp, "shim_check"], check=True, capture_output=True)
    w2 = WL.build("c2", mode="nominal", use_ransac=args.ransac)
    z = np.load(os.path.join(WL.GOLDEN, "micro_euroc_f10_18.npz"))
    body_R = z["body_R"][: w2.ring]
    cfg = abi.Config()
    cfg.left, cfg.right, cfg.params, cfg.batch, cfg.device = w2.left, w2.right, w2.params, 1, dev.index or 0
    n_spin = 600
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            f.write(bytes(cfg))
            f.write(np.array([w2.ring, w2.width, w2.height], np.int32).tobytes())
            for i in range(w2.ring):
                fi = abi.FrameInput()
                fi.timestamp_ns = i * WL.DT_NS
                if i > 0:   # body-frame gyro rate of the interval i-1 -> i: Log(R_{i-1}^T R_i) / dt
                    from scipy.spatial.transform import Rotation
                    w = Rotation.from_matrix(body_R[i - 1].T @ body_R[i]).as_rotvec() / (WL.DT_NS * 1e-9)
                    for k in range(3):
                        fi.keyframe_R_cur_frame[k] = float(w[k])
                f.write(bytes(fi))
                f.write(np.ascontiguousarray(w2.lefts[i, 0]).tobytes())
                f.write(np.ascontiguousarray(w2.rights[i, 0]).tobytes())
        r = subprocess.run([os.path.join(cpp, "shim_check"), fin, fout, "bench", str(n_spin)], capture_output=True,
                           text=True, timeout=600)
    if r.returncode != 0:
        return {"error": (r.stderr or r.stdout)[-500:]}
    res = json.loads(r.stdout.strip().splitlines()[-1])
    ref = run_frontend_leg(torch, F, dist, sharding, w2, dev, 1, 200, 20, args.repeats, 0, 0)
    return {"value": res["pairs_per_s"], "unit": "stereo-pairs/s", "ms_per_pair": res["ms_per_pair"],
            "spins": res["spins"], "keyframes": res["keyframes"], "measurements_per_spin": res["measurements_per_spin"],
            "workload": "BASELINE configs[1] (single EuRoC stream, 300 features, 3-level LK), reference cadence (the front-end's "
                        "own keyframe decisions), host images, through shim::StereoVisionImuFrontend::spinOnce: every call "
                        "uploads the pair, runs the step, waits for it and builds the StereoFrontendOutput",
            "no_readback_value": ref["value"], "no_readback_ms_per_pair": ref["ms_per_step"],
            "no_readback_workload": "the same stream and cadence, frames resident in HBM, kvfe_frontend_step_device back to "
                                    "back, no output read and no synchronisation inside the timed region",
            "vs_no_readback": round(res["pairs_per_s"] / ref["value"], 4)}


def usable_cpus():
    """processors this process may use: hardware threads cut down to the affinity mask and the cgroup CPU quota (the GPU
    boxes of the pool: 256 hardware threads, cpu.max = 16 -- tools/r5/gpu_ad.sh)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            q, per = open(path).read().split()[:2]
            if q != "max":
                n = min(n, max(1, -(-int(q) // int(per))))
        except (OSError, ValueError):
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and per > 0:
            n = min(n, max(1, -(-q // per)))
    except (OSError, ValueError):
        pass
    return n


def input_side():
    """SURVEY 8 f3, host side: PNG -> grey decode rate of the data provider (kvfe_png_decode_gray_batch into one
    buffer per frame, the role of the pinned staging slots) on the committed 752x480 EuRoC frames, one host thread and
    all of them, ~1 s each; PIL's decoder on one thread beside it.  Host code: no GPU time, never part of `value`."""
    import io
    from kimera_vio_amd import dataprovider as DP
    files = []
    for name in ("left_img_0.png", "right_img_0.png"):
        with open(os.path.join(ROOT, "tests", "golden", name), "rb") as f:
            files.append(f.read())
    files = files * 32
    out = np.empty((len(files), 480, 752), np.uint8)
    DP.decode_png_gray_batch(files[:2], out[:2], 1)
    res = {"workload": "64 PNG files per call (32 x tests/golden/{left,right}_img_0.png, 752x480 8-bit grey, "
                       "362 kB each)", "unit": "frames/s", "host_cores": os.cpu_count(), "usable_cpus": usable_cpus()}
    for key, threads in (("decode_1_thread", 1), ("decode_all_threads", 0)):
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < 1.0:
            DP.decode_png_gray_batch(files, out, threads)
            n += len(files)
        res[key] = round(n / (time.perf_counter() - t0), 1)
    try:
        from PIL import Image
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < 0.5:
            np.asarray(Image.open(io.BytesIO(files[n % 2])))
            n += 1
        res["pil_1_thread"] = round(n / (time.perf_counter() - t0), 1)
    except ImportError:
        res["pil_1_thread"] = None
    res["stereo_pairs_per_s_all_threads"] = round(res["decode_all_threads"] / 2.0, 1)
    # one step of the headline configuration: the 64 stereo pairs of 64 streams = 128 files in one call
    files2 = files * 2
    out2 = np.empty((len(files2), 480, 752), np.uint8)
    DP.decode_png_gray_batch(files2, out2, 0)
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < 1.0:
        DP.decode_png_gray_batch(files2, out2, 0)
        n += len(files2)
    res["decode_all_threads_128_files_per_call"] = round(n / (time.perf_counter() - t0), 1)
    res["stereo_pairs_per_s_all_threads_128_files_per_call"] = round(res["decode_all_threads_128_files_per_call"] / 2.0, 1)
    return res


def pcie_inclusive(F, wl, torch=None):
    """PCIe-inclusive rates (host buffers handed over at the boundary): reported, never `value`.
      staged  : the `value` workload with every frame arriving from the host -- the data provider's decoded frames sit in
                the context's pinned slots (one per frame of the ring), every step uploads its pair of frames on a copy
                stream beside the previous step (kvfe_frontend_step_staged)
      cyclic3 : round 4's form of this leg -- three frames in a cycle (the wrap-around loses most tracks: a much heavier
                step than `value`'s, 1.9 ms where `value` takes 1.1) -- kept as a scalar for continuity
      pageable: kvfe_frontend_step_host from ordinary host memory, copies on the compute stream
    Beside them the link's own ceiling: the upload of one step's frames alone (2 x batch x W x H bytes)."""
    B = wl.batch
    lefts, rights = wl.replicated()
    ring = lefts.shape[0]
    ctx = F.Context(wl.left, wl.right, wl.params, batch=B)
    n_h, n_w = 52, 12        # (two maxFeatureAge periods of 26 steps, like `value`'s regions)
    steps = wl.plan(n_w + n_h)
    plan = [wl.batch_inputs(ctx, st) for st in steps]
    res = {"unit": "stereo-pairs/s"}
    if ring <= 8:            # KVFE_STAGING_SLOTS
        for sl in range(ring):   # "decoded" frames sit in the pinned slots before the clock starts
            a, b = ctx.staging_buffers(sl)
            a[:] = lefts[sl]
            b[:] = rights[sl]
        for i in range(n_w):
            ctx.step_staged(steps[i][0], plan[i])
        ctx.synchronize()
        th = time.perf_counter()
        for i in range(n_w, n_w + n_h):
            ctx.step_staged(steps[i][0], plan[i])
        t_enq = time.perf_counter() - th
        ctx.synchronize()
        dt = time.perf_counter() - th
        res["value"] = round(B * n_h / dt, 2)
        res["ms_per_step"] = round(1e3 * dt / n_h, 4)
        res["host_enqueue_ms_per_step"] = round(1e3 * t_enq / n_h, 4)
        ctx.reset()
    # round 4's leg: three frames in a cycle, 30 steps
    hl = np.ascontiguousarray(lefts[:3])
    hr = np.ascontiguousarray(rights[:3])
    for sl in range(3):
        a, b = ctx.staging_buffers(sl)
        a[:] = hl[sl]
        b[:] = hr[sl]
    for i in range(3):
        ctx.step_staged(i % 3, plan[i])
    ctx.synchronize()
    th = time.perf_counter()
    for i in range(3, 33):
        ctx.step_staged(i % 3, plan[i])
    ctx.synchronize()
    res["cyclic3_value"] = round(B * 30 / (time.perf_counter() - th), 2)
    ctx.reset()
    for i in range(2):
        ctx.step_host(hl[i % 2], hr[i % 2], plan[i])
    ctx.synchronize()
    th = time.perf_counter()
    for i in range(2, 8):
        ctx.step_host(hl[i % 2], hr[i % 2], plan[i])
    ctx.synchronize()
    res["pageable_value"] = round(B * 6 / (time.perf_counter() - th), 2)
    ctx.close()
    if torch is not None:    # the link alone: one step's frames from pinned memory, 20 times
        nbytes = 2 * B * wl.width * wl.height
        h = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        d = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        d.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        th = time.perf_counter()
        for _ in range(20):
            d.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        up = (time.perf_counter() - th) / 20
        res["link"] = {"bytes_per_step": nbytes, "upload_alone_ms": round(1e3 * up, 4), "GBps": round(nbytes / up / 1e9, 1),
                       "ceiling_pairs_per_s": round(B / up, 1)}
    res["note"] = ("value: the `value` workload (same frames, same ping-pong order) through kvfe_frontend_step_staged -- every "
                   "step's frames uploaded from pinned host slots inside the timed region, %d steps; cyclic3_value: round "
                   "4's form of the leg (three frames in a cycle, a heavier step); link: the upload of one step's frames "
                   "alone" % n_h)
    return res


def dense_stereo(F, WL, W, H, dev, pmc_leg):
    """SURVEY.md §8 a29 / BASELINE config[4] "dense stereo row": StereoMatcher::denseStereoReconstruction
    (cv::StereoSGBM MODE_HH with the reference's DenseStereoParams) on 8 rectified pairs per call.
    Time = HIP events around the kernel sequence inside libkvfe (the component call takes host
    buffers; its H2D / D2H copies are outside the events)."""
    from kimera_vio_amd import _abi as abi
    n = 8
    wl = WL.build("c5", batch=2, unique=2, ring=n // 2, width=W, height=H, seed_base=900)
    ctx = F.Context(wl.left, wl.right, wl.params, batch=1, device=dev.index)
    pairs = []
    for i in range(n):
        l, r = wl.lefts[i // 2, i % 2], wl.rights[i // 2, i % 2]
        pairs.append((ctx.undistort_rectify_image(0, l), ctx.undistort_rectify_image(1, r)))
    dp = abi.dense_stereo_params_default()
    lefts, rights = [a for a, _ in pairs], [b for _, b in pairs]
    ctx.dense_stereo_reconstruction(lefts, rights, dp)
    ctx.dense_profile_read()
    vals = []
    for _ in range(5):
        disp = ctx.dense_stereo_reconstruction(lefts, rights, dp)
        ms, cnt = ctx.dense_profile_read()
        vals.append(ms / cnt)
    # what upstream calls the method with: ONE pair per call (StereoMatcher.cpp:32-121) -- the eight direction sweeps, not the
    # two-pass launch (its critical path is W + 2H dependent steps whatever the tiling: profiles/r6_analysis.md section 13)
    one = []
    if W == 752:
        ctx.dense_stereo_reconstruction(lefts[:1], rights[:1], dp)
        ctx.dense_profile_read()
        for _ in range(5):
            ctx.dense_stereo_reconstruction(lefts[:1], rights[:1], dp)
            ms1, cnt1 = ctx.dense_profile_read()
            one.append(ms1 / cnt1)
    ctx.close()
    ms_pair = statistics.median(vals)
    D = dp.num_disparities
    w1 = W - (dp.min_disparity + D)
    vol = H * w1 * D * 2.0                      # one int16 cost volume
    npx = float(W) * H
    # bytes this design moves per pair: pixel records, C(p,d) written once by the fused cost kernel (round
    # 4), the two aggregation passes of round 5 (C read twice, one partial sum written per pass, the block-to-block
    # hand-over words written and read: 528 bytes per column and boundary between 15-row blocks; rounds 1-4: eight
    # sweeps = 8 C reads, 1 sum write, 7 read-modify-writes = 23 volumes), selection (both partial sums), the small
    # disparity passes
    bands = (H + 14) // 15
    agg_bytes = 2 * vol + 2 * vol + 2 * 2 * (bands - 1) * w1 * 528.0
    alg = (2 * npx + 16 * npx) + (16 * npx + vol) + agg_bytes + (2 * vol + 2 * npx) + 24 * npx
    traffic = None
    if pmc_leg:   # HBM-side bytes of the same kernels from the committed rocprofv3 PMC passes
        tb = sum((2.0 * v["fetch_kb"] + v["write_kb"]) * 1024.0 for k, v in pmc_leg.items()
                 if k.startswith(("dense_", "speckle_")))
        traffic = round(tb / n) if tb > 0 else None
    # what the ALGORITHM needs: cv::StereoSGBM MODE_HH is two passes over the image, each aggregating four directions
    # with its running L_r rows kept on chip -- cost volume C written once and read by both passes, the sum volume S
    # written by the first pass, read and finished by the second: ~6 volume transfers + the images.  `frac` is priced
    # against this minimum; the design's own traffic (the same two passes since round 5, with both partial sums written
    # and read by the selection: 7 volumes + the hand-over words) is listed beside it.
    alg_min = 6 * vol + 4 * npx + 24 * npx
    ach = alg_min / (ms_pair * 1e-3) / 1e9
    ach_design = alg / (ms_pair * 1e-3) / 1e9
    valid = float(np.mean(disp[0] != (dp.min_disparity - 1) * 16))
    return {"workload": f"cv::StereoSGBM MODE_HH, block {dp.sad_window_size}, {D} disparities, {W}x{H}, "
                        f"{n} rectified pairs per call",
            "value": round(1e3 / ms_pair, 2), "unit": "stereo-pairs/s", "ms_per_pair": round(ms_pair, 4),
            "ms_per_pair_min": round(min(vals), 4),
            **({"one_pair_per_call": {"value": round(1e3 / statistics.median(one), 2), "unit": "stereo-pairs/s",
                                      "ms_per_pair": round(statistics.median(one), 4)}} if one else {}),
            "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "alg_bytes_per_pair": round(alg_min),
                         "design_traffic_bytes_per_pair": round(alg), "design_traffic_aggregation": round(agg_bytes),
                         "design_traffic_GBps": round(ach_design, 1),
                         "design_traffic_frac_of_peak": round(ach_design / HBM_PEAK_GBPS, 4),
                         "note": "whole kernel sequence of one pair (HIP events inside libkvfe); alg_bytes_per_pair = the "
                                 "two-pass minimum of cv::StereoSGBM MODE_HH (~6 cost-volume transfers), which `achieved` "
                                 "and `frac` are priced against; design_traffic_* = what this implementation's two "
                                 "aggregation passes move (rounds 1-4: eight sweeps); traffic = PMC bytes per pair of the "
                                 "dense_* / speckle_* kernels, 2 x FETCH_SIZE + WRITE_SIZE (the guide's gfx950 rule, the "
                                 "one rule used for every kernel of this file and of profiles/r6_analysis.md)"},
            "parity": "unpinned -- HIP == oracle bit for bit (tests/test_gpu_dense_twopass.py, tests/test_gpu_parity.py), but "
                      "the oracle's SGBM / BM restate OpenCV 4.2's published algorithm and are pinned on an independent numpy "
                      "statement only: the reference's own test of this method is a stub (tests/testStereoMatcher.cpp:131) "
                      "and OpenCV is not in this image",
            "valid_fraction_pair0": round(valid, 3)}


def cpu_baseline(wl, args):
    """The CPU oracle (OpenCV-faithful scalar restatement, 1 thread = the reference's one front-end
    thread, Pipeline.cpp:331) on a bounded sample of the same workload: one stream, same parameters and mode."""
    import oracle_lib as O  # checker / baseline only
    from kimera_vio_amd import _abi as abi
    n = args.cpu_baseline_frames
    plan = wl.plan(n)
    lefts = np.stack([wl.lefts[st[0], 0] for st in plan])
    rights = np.stack([wl.rights[st[0], 0] for st in plan])
    inputs = []
    for (t, ts, Rs, force) in plan:
        fi = abi.FrameInput()
        fi.timestamp_ns = ts
        Rm = np.asarray(Rs[0], np.float64).reshape(9)
        for k in range(9):
            fi.keyframe_R_cur_frame[k] = float(Rm[k])
        fi.force_keyframe = force
        inputs.append(fi)
    fe = O.Frontend(wl.left, wl.right, wl.params)
    secs = fe.time_sequence(lefts, rights, inputs)
    p = wl.params
    out = {"value": round(n / secs, 3), "unit": "stereo-pairs/s", "cores": 1, "kind": "port",
           "sample": f"{n} consecutive stereo pairs of one {wl.width}x{wl.height} stream of the `value` workload, "
                     f"{p.detector.max_features_per_frame} features, mode={wl.mode}, {secs:.1f} s on one host core "
                     f"(scalar OpenCV-faithful restatement, no SIMD/IPP; host has {os.cpu_count()} hardware threads, "
                     f"{usable_cpus()} usable under the container's CPU quota)"}
    # independent streams on the host's cores (one single-threaded front-end per core, the
    # many-sequence mode on the CPU), bounded to a few seconds; best of 3
    try:
        import multiprocessing as mp
        C_ = max(1, min(os.cpu_count() or 1, 64))
        m = max(20, min(n, 60))
        mctx = mp.get_context("fork")
        with mctx.Pool(C_, initializer=_cpu_worker_init,
                       initargs=(wl.left, wl.right, p, lefts[:m], rights[:m], inputs[:m])) as pool:
            pool.map(_cpu_worker_run, range(C_))            # warm: library load, first-touch
            walls = []
            for _ in range(3):
                t0 = time.perf_counter()
                pool.map(_cpu_worker_run, range(C_))
                walls.append(time.perf_counter() - t0)
        out["all_cores"] = {"value": round(C_ * m / min(walls), 2), "unit": "stereo-pairs/s", "cores": C_,
                            "usable_cpus": usable_cpus(),
                            "sample": f"{C_} processes x {m} pairs of the same stream, best of 3 "
                                      f"({min(walls):.1f} s wall; all: {[round(w, 1) for w in walls]})"}
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
