"""Parity ON WHAT IS BENCHMARKED: every BASELINE.json configuration that `bench.py` times is built here
through the same `kimera_vio_amd.workloads.build()` call (same cameras, parameters, rendered frames,
rotation plan, replication of the unique streams over the batch, device-resident input ring and
`kvfe_frontend_step_device` entry point) and every step of it is compared field by field, tolerance 0,
with the CPU oracle (`StereoVisionImuFrontend::processStereoFrame` restatement,
/root/reference/src/frontend/StereoVisionImuFrontend.cpp:283-481).

    C2  single EuRoC stream (MicroEuroc frames), 300 features, 3-level LK      kf + nominal cadence
    C3  64 streams (8 unique RigStream seeds replicated), 600 features, 3-level LK, useRANSAC 1,
        ring of 6 frames walked ping-pong incl. the turn-around              kf + nominal cadence
    C5  1280x720, 1000 features, 4-level LK, batched                          kf + nominal cadence
    C5 dense row: cv::StereoSGBM MODE_HH on rectified 1280x720 pairs (component call)

Run on the MI355X box:  python -m pytest tests -m gpu -x -q
"""
import os
import numpy as np
import pytest

import oracle_lib as O
from kimera_vio_amd import _abi as abi
from kimera_vio_amd import frontend as F
from kimera_vio_amd import workloads as WL
from parity_util import assert_step_equal

pytestmark = pytest.mark.gpu

_CACHE = {}


def _build(config, mode, **kw):
    """frames are rendered once per configuration; the two cadences share them"""
    import dataclasses
    key = (config, tuple(sorted(kw.items())))
    if key not in _CACHE:
        _CACHE[key] = WL.build(config, mode="kf", **kw)
    return dataclasses.replace(_CACHE[key], mode=mode)


def _run_workload(wl, n_steps, check_streams, replicas_equal_at_end=True, persist=1):
    """drives the GPU context exactly like bench.py's timed loop and the per-unique-stream oracles in
    lock-step; returns [(step, is_keyframe, n_tracked, n_detected)] of unique stream 0"""
    import torch
    dev = torch.device("cuda", 0)
    lefts, rights = wl.replicated()
    d_left = torch.from_numpy(lefts).to(dev)
    d_right = torch.from_numpy(rights).to(dev)
    torch.cuda.synchronize()
    # bench.py's contexts: kvfe_config.device_frames_persist = 1 (the frames sit in a device ring that is never rewritten)
    ctx = F.Context(wl.left, wl.right, wl.params, batch=wl.batch, device_frames_persist=persist)
    fes = [O.Frontend(wl.left, wl.right, wl.params) for _ in range(wl.unique)]
    kinds = []
    try:
        for i, step in enumerate(wl.plan(n_steps)):
            t, ts, Rs, force = step
            ctx.step_device(d_left[t].data_ptr(), d_right[t].data_ptr(), wl.batch_inputs(ctx, step))
            exp = [fes[u].process(wl.lefts[t, u], wl.rights[t, u], ts, Rs[u], bool(force))
                   for u in range(wl.unique)]
            for s in check_streams:
                assert_step_equal(ctx.get_output(s), exp[wl.unique_of(s)], (wl.name, wl.mode, "step", i, "stream", s))
            kinds.append((i, exp[0]["is_keyframe"], exp[0]["n_tracked"], exp[0]["n_detected"]))
        if replicas_equal_at_end and wl.batch > wl.unique:
            # size-independent property over the WHOLE batch: replica s is bit-identical to stream s mod U
            ref = [ctx.get_output(u) for u in range(wl.unique)]
            for s in range(wl.unique, wl.batch):
                got = ctx.get_output(s)
                for k in ("n_keypoints", "n_measurements", "is_keyframe"):
                    assert got[k] == ref[wl.unique_of(s)][k], (s, k)
                for k in ("keypoints", "landmarks", "depth", "meas_uL_uR_v"):
                    assert np.array_equal(got[k], ref[wl.unique_of(s)][k], equal_nan=True), (s, k)
        # kvfe_frontend_get_outputs (all streams in one call, copied by several threads from 16 streams up) against the
        # per-stream accessor, for the last step and the one before it
        bufs = ctx.output_buffers()
        for back in ((0, 1) if n_steps > 1 else (0,)):
            for arrs in bufs.arrays:      # (a frame without stereo data leaves the stereo arrays of the caller untouched)
                for v in arrs.values():
                    v[...] = 0
            structs = bufs.read(back)
            for s in range(wl.batch):
                one = ctx.get_output(s, steps_back=back)
                assert structs[s].n_keypoints == one["n_keypoints"] and structs[s].n_measurements == one["n_measurements"]
                assert structs[s].frame_id == one["frame_id"] and structs[s].is_keyframe == one["is_keyframe"]
                n, m = one["n_keypoints"], one["n_measurements"]
                for k, _, _ in bufs.FIELDS:
                    cnt = m if k.startswith("meas_") else n
                    assert np.array_equal(bufs.arrays[s][k][:cnt], one[k], equal_nan=True), (back, s, k)
    finally:
        ctx.close()
    return kinds


@pytest.mark.parametrize("mode", ["kf", "nominal"])
def test_c3_headline_64_streams_600_features(mode):
    """BASELINE configs[2], the workload of bench.py's `value`: all 8 unique streams + replica 63 on every
    step of 12 (the 6-frame ring is walked 0..5 and back: the turn-around re-tracks into frames seen before)."""
    wl = _build("c3", mode)
    assert (wl.batch, wl.unique, wl.ring, wl.width, wl.height) == (64, 8, 6, 752, 480)
    assert wl.params.detector.max_features_per_frame == 600 and wl.params.tracker.klt_max_level == 2
    assert wl.params.use_ransac == 1 and wl.params.detector.non_max_suppression_type == abi.ANMS_BINNING
    kinds = _run_workload(wl, 12, list(range(8)) + [63])
    if mode == "kf":
        assert all(k[1] == 1 for k in kinds)
        assert all(k[2] > 400 for k in kinds[1:]), kinds        # the headline really tracks ~600 points
    else:
        assert [k[1] for k in kinds[:9]] == [1, 0, 0, 0, 1, 0, 0, 0, 1]


def test_c3_with_the_context_owned_level0_copy():
    """bench.py --copy-level0 (kvfe_config.device_frames_persist = 0, the default of the C ABI): the pyramid launch also
    writes the context's own copy of the left frame, and the next step tracks from it"""
    wl = _build("c3", "kf")
    _run_workload(wl, 5, [0, 7, 63], persist=0)


def test_lk_launch_forms_agree_on_the_headline_batch():
    """round 6: kvfe_config.lk_impl -- 0 = four points per wavefront (k_lk4.hip, the default), 1 = one wavefront per point
    (lk_kernel_sys).  Both are compared with the oracle elsewhere; here the whole 64-stream batch, every field of every
    stream, on 8 steps of the headline workload and of its five-level (klt_max_level 4: a 47 x 30 top level, every
    window of it on a border) variant."""
    import torch
    dev = torch.device("cuda", 0)
    for kw in (dict(), dict(klt_max_level=4, batch=16)):
        wl = _build("c3", "kf", **kw)
        lefts, rights = wl.replicated()
        d_left = torch.from_numpy(lefts).to(dev)
        d_right = torch.from_numpy(rights).to(dev)
        torch.cuda.synchronize()
        ctxs = [F.Context(wl.left, wl.right, wl.params, batch=wl.batch, device_frames_persist=1, lk_impl=i) for i in (0, 1)]
        try:
            for i, step in enumerate(wl.plan(8)):
                t = step[0]
                for c in ctxs:
                    c.step_device(d_left[t].data_ptr(), d_right[t].data_ptr(), wl.batch_inputs(c, step))
                for s in range(wl.batch):
                    a, b = ctxs[0].get_output(s), ctxs[1].get_output(s)
                    for k in ("n_keypoints", "n_tracked", "n_detected", "is_keyframe", "tracking_status_mono",
                              "tracking_status_stereo", "n_measurements"):
                        assert a[k] == b[k], (kw, i, s, k, a[k], b[k])
                    for k in ("keypoints", "landmarks", "landmarks_age", "versors", "depth", "meas_uL_uR_v"):
                        assert np.array_equal(a[k], b[k], equal_nan=True), (kw, i, s, k)
        finally:
            for c in ctxs:
                c.close()


_BURST = {}


def _burst_expectation(wl, n_steps):
    """the oracle over all unique streams of the headline workload for `n_steps` steps (shared by the two burst tests:
    ~6 s of CPU): [step][unique stream] -> oracle record"""
    key = (wl.name, wl.mode, n_steps)
    if key not in _BURST:
        fes = [O.Frontend(wl.left, wl.right, wl.params) for _ in range(wl.unique)]
        out = []
        for (t, ts, Rs, force) in wl.plan(n_steps):
            out.append([fes[u].process(wl.lefts[t, u], wl.rights[t, u], ts, Rs[u], bool(force))
                        for u in range(wl.unique)])
        _BURST[key] = out
    return _BURST[key]


def _assert_burst_shape(wl, exp):
    """Tracker.cpp:167-180: every feature of a synthetic stream is born in frame 0, so at step maxFeatureAge + 1 (the
    first step where age > maxFeatureAge) the whole list passes the age limit at once, only the corners detected
    after frame 0 survive the tracker and the detector refills the frame"""
    age = wl.params.tracker.max_feature_track_age
    assert age == 25, age
    for u in range(wl.unique):
        before, burst = exp[age][u], exp[age + 1][u]
        assert before["n_tracked"] > 400, (u, before["n_tracked"])
        # the survivors of the burst step are the few corners detected after frame 0, everything else expired
        assert burst["n_tracked"] < 0.5 * before["n_tracked"], (u, burst["n_tracked"], before["n_tracked"])
        assert burst["n_detected"] > 250, (u, burst["n_detected"])


def test_c3_headline_through_a_feature_age_burst():
    """the headline's timed loop (bench.py `value`: c3, device_frames_persist = 0,
    kvfe_frontend_step_device) runs through the step where every feature of a stream passes maxFeatureAge at once
    (step 26, the 27th: lk.skip_age skips the whole list, ~440 corners per stream re-detected, the grouped cornerSubPix kernel
    chosen by count, two heavy steps behind it).  30 steps, all 8 unique streams + replica 63 against the oracle on
    every step, tolerance 0 (Tracker.cpp:167-180, StereoVisionImuFrontend.cpp:283-481)."""
    import torch
    wl = _build("c3", "kf")
    n_steps = 30
    exp = _burst_expectation(wl, n_steps)
    _assert_burst_shape(wl, exp)
    dev = torch.device("cuda", 0)
    lefts, rights = wl.replicated()
    d_left = torch.from_numpy(lefts).to(dev)
    d_right = torch.from_numpy(rights).to(dev)
    torch.cuda.synchronize()
    ctx = F.Context(wl.left, wl.right, wl.params, batch=wl.batch, device_frames_persist=0)
    try:
        for i, step in enumerate(wl.plan(n_steps)):
            t = step[0]
            ctx.step_device(d_left[t].data_ptr(), d_right[t].data_ptr(), wl.batch_inputs(ctx, step))
            for s in list(range(8)) + [63]:
                assert_step_equal(ctx.get_output(s), exp[i][wl.unique_of(s)], ("c3 burst", "step", i, "stream", s))
    finally:
        ctx.close()


def test_c3_headline_burst_through_the_staged_input_path():
    """the same 30 steps through kvfe_frontend_step_staged at batch 64 (bench.py's `pcie_inclusive` leg: the frames of
    the ring sit in the context's pinned slots, every step uploads its pair on the copy stream behind the previous
    step's kernels and is only enqueued); outputs read back one step late like a pipelined caller does"""
    wl = _build("c3", "kf")
    n_steps = 30
    exp = _burst_expectation(wl, n_steps)
    lefts, rights = wl.replicated()
    ctx = F.Context(wl.left, wl.right, wl.params, batch=wl.batch)
    check = list(range(8)) + [63]
    try:
        for sl in range(wl.ring):
            a, b = ctx.staging_buffers(sl)
            a[:] = lefts[sl]
            b[:] = rights[sl]
        plan = wl.plan(n_steps)
        for i, step in enumerate(plan):
            ctx.step_staged(step[0], wl.batch_inputs(ctx, step))         # enqueue only
            if i > 0:                                                    # the previous step's record, one step back
                for s in check:
                    assert_step_equal(ctx.get_output(s, steps_back=1), exp[i - 1][wl.unique_of(s)],
                                      ("c3 burst staged", "step", i - 1, "stream", s))
        for s in check:
            assert_step_equal(ctx.get_output(s), exp[n_steps - 1][wl.unique_of(s)],
                              ("c3 burst staged", "step", n_steps - 1, "stream", s))
    finally:
        ctx.close()


@pytest.mark.parametrize("mode", ["kf", "nominal"])
def test_c2_single_euroc_stream_300_features_3_level_lk(mode):
    """BASELINE configs[1] as stated: EuRoC frames (MicroEuroc 10..18), 300 features, klt_max_level 2, shipped
    Euroc parameters with useRANSAC 1; 14 steps = the ring forwards and most of the way back."""
    wl = _build("c2", mode)
    assert (wl.batch, wl.ring) == (1, 9) and wl.params.detector.max_features_per_frame == 300
    assert wl.params.tracker.klt_max_level == 2
    kinds = _run_workload(wl, 14, [0])
    assert all(k[2] > 100 for k in kinds[1:])


@pytest.mark.parametrize("mode", ["kf", "nominal"])
def test_c5_1280x720_1000_features_4_level_lk_batched(mode):
    """BASELINE configs[4] shape (bench.py's `c5` leg runs it with batch 32 = 4 unique streams x 8): here batch 8
    = 4 unique streams x 2, every unique stream and one replica compared on 8 steps."""
    wl = _build("c5", mode, batch=8)
    assert (wl.width, wl.height, wl.unique) == (1280, 720, 4)
    assert wl.params.detector.max_features_per_frame == 1000 and wl.params.tracker.klt_max_level == 3
    kinds = _run_workload(wl, 8, [0, 1, 2, 3, 7])
    assert all(k[2] > 600 for k in kinds[1:]), kinds


def test_c5_dense_stereo_row_1280x720():
    """the "dense stereo row" of configs[4]: StereoMatcher::denseStereoReconstruction (StereoMatcher.cpp:32-121,
    cv::StereoSGBM MODE_HH with the reference's DenseStereoParams) on rectified 1280x720 pairs, bit-exact;
    bench.py's `dense_stereo_c5` leg times this call."""
    wl = WL.build("c5", batch=2, unique=2, ring=1)
    ctx = F.Context(wl.left, wl.right, wl.params, batch=1)
    ocam = O.Camera(wl.left, wl.right)
    try:
        dp = abi.dense_stereo_params_default()
        pairs = []
        for u in range(2):
            lr = ctx.undistort_rectify_image(0, wl.lefts[0, u])
            rr = ctx.undistort_rectify_image(1, wl.rights[0, u])
            assert np.array_equal(lr, ocam.rectify_image(0, wl.lefts[0, u]))
            pairs.append((lr, rr))
        got = ctx.dense_stereo_reconstruction([a for a, _ in pairs], [b for _, b in pairs], dp)
        for u in range(2):
            exp = O.dense_stereo_reconstruction(pairs[u][0], pairs[u][1], dp)
            assert np.array_equal(got[u], exp)
            assert np.mean(exp != (dp.min_disparity - 1) * 16) > 0.5      # a real disparity map, not all invalid
    finally:
        ctx.close()


def test_c4_rank_windows_are_distinct_sequences():
    """configs[3]: 8 sequences, one per GPU, batch 1 — rank r replays its own window of MicroEuroc; rank 3's
    window on this GPU matches its oracle (the 8-GPU launch itself is the driver's)."""
    wl = WL.build("c4", mode="nominal", rank=3)
    wl0 = WL.build("c4", mode="nominal", rank=0)
    assert not np.array_equal(wl.lefts[0], wl0.lefts[0])
    _run_workload(wl, 9, [0])


def test_c3_at_the_shipped_klt_max_level_4():
    """bench.py's `klt_max_level_4` leg ("also one run at the shipped value 4"): configs[2] with the
    five-level pyramid of params/Euroc/FrontendParams.yaml:5 -- two pyr2 launches for levels 1-2, the tile kernel for
    levels 3-4 (188 x 120 is not a multiple of 16 wide)."""
    wl = _build("c3", "kf", klt_max_level=4, batch=16)
    assert wl.params.tracker.klt_max_level == 4
    kinds = _run_workload(wl, 8, list(range(8)) + [15])
    assert all(k[2] > 400 for k in kinds[1:]), kinds


def test_c3e_real_frames_64_streams():
    """bench.py's `kf_realistic` leg: 64 streams replaying 4 offset windows of MicroEuroc, 600 features, every frame a
    keyframe; all unique windows + one replica on 8 steps"""
    wl = _build("c3e", "kf")
    assert (wl.batch, wl.unique, wl.source) == (64, 4, "euroc")
    kinds = _run_workload(wl, 8, [0, 1, 2, 3, 63])
    assert all(k[1] == 1 for k in kinds) and all(k[3] > 20 for k in kinds[1:]), kinds


def test_grouped_corner_subpix_kernel_forced():
    """round 4: two cornerSubPix kernels share the launch slot -- one corner per block for launches with few new corners,
    SPG_G corners per block (one lane per float64 chain) for launches with many (>= 3000 over all streams: the real-frame
    configuration above takes it by itself).  KVFE_SUBPIX_GROUP=1 (read once per process, hence the sub-process) forces the
    grouped kernel onto the small configurations too: single stream, 64 synthetic streams, the pipelined host steps."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(root, "tests", "test_gpu_bench_configs.py") + "::test_c2_single_euroc_stream_300_features_3_level_lk",
                        os.path.join(root, "tests", "test_gpu_bench_configs.py") + "::test_c3_headline_64_streams_600_features",
                        os.path.join(root, "tests", "test_gpu_pipelined_r3.py") + "::test_pipelined_host_steps_no_sync"],
                       env=dict(os.environ, KVFE_SUBPIX_GROUP="1"), capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
