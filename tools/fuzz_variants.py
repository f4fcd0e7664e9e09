"""Randomised GPU-vs-oracle comparison, part 2 (not part of the test suite): mono / RGBD front-ends and
multi-stream batches whose streams see different sequences and keyframe cadences; dense stereo parameters.
Usage: python tools/fuzz_variants.py [n_configs] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
# FUZZ_ODD_SIZES=1: widths / heights that are no multiples of 4 / 8 / 16 (the same number of random draws, so every other
# parameter of a configuration stays what it is without the switch) -- for runs under KVFE_GUARD_ALLOC, where an access
# past a row's or an image's end faults
ODD = int(os.environ.get("FUZZ_ODD_SIZES", "0"))
ODD_W, ODD_H = [321, 377, 481, 750], [241, 290, 479]
import oracle_lib as O
from kimera_vio_amd import _abi as abi, frontend as F, params as P, synth, workloads

G = os.path.join(ROOT, "tests", "golden")
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def depth_image(h, w, t, f32, seed):
    r = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    d = 3.0 + 1.8 * np.sin(xx / 83.0 + 0.07 * t) * np.cos(yy / 59.0)
    holes = r.randint(0, 100, (h, w)) < 5
    if f32:
        d = d.astype(np.float32)
        d[holes] = np.nan
        return d
    d = np.clip(np.rint(d * 1000.0), 0, 65535).astype(np.uint16)
    d[holes] = 0
    return d


bad = 0
for ci in range(n_cfg):
    kind = int(rng.choice([0, 1, 2]))   # stereo / mono / rgbd
    w = int(rng.choice(ODD_W if ODD else [320, 376, 480, 752]))
    h = int(rng.choice(ODD_H if ODD else [240, 288, 480]))
    B = int(rng.choice([1, 2, 3]))
    L, R = workloads.make_cameras(w, h)
    p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=int(rng.randint(0, 2)))
    p.detector.max_features_per_frame = int(rng.choice([60, 150, 300]))
    p.detector.non_max_suppression_type = int(rng.choice([0, 4, 6]))
    p.tracker.klt_max_level = int(rng.choice([2, 3]))
    p.tracker.ransac_use_1point_stereo = int(rng.randint(0, 2))
    p.tracker.ransac_threshold_stereo = float(rng.choice([0.1, 0.5, 1.0]))
    p.min_intra_keyframe_time_ns = float(rng.choice([0.0, 1.0e8, 2.0e8]))
    p.use_stereo_tracking = int(rng.randint(0, 2)) if kind != 1 else p.use_stereo_tracking
    f32 = int(rng.randint(0, 2))
    dp = abi.depth_params_default(abi.DEPTH_F32 if f32 else abi.DEPTH_U16)
    dp.virtual_baseline = float(rng.choice([0.02, 0.1]))
    dp.min_depth = 0.5
    dp.max_depth = float(rng.choice([3.5, 10.0]))
    if not f32:
        dp.depth_to_meters = 0.001
    desc = dict(kind=["stereo", "mono", "rgbd"][kind], w=w, h=h, B=B, feats=p.detector.max_features_per_frame,
                anms=p.detector.non_max_suppression_type, ransac=p.use_ransac, one=p.tracker.ransac_use_1point_stereo,
                kf_ns=p.min_intra_keyframe_time_ns, ust=p.use_stereo_tracking, f32=f32)
    R1 = np.array(F.compute_rectification(L, R).R1).reshape(3, 3) if kind == 0 else np.eye(3)
    seeds = [int(rng.randint(0, 1000)) for _ in range(B)]
    starts = [int(rng.randint(0, 3)) for _ in range(B)]
    streams = [synth.RigStream(L, R, seed=sd, rect_R1=R1) for sd in seeds]
    if kind == 0:
        fe = [O.Frontend(L, R, p) for _ in range(B)]
        c = F.Context(L, R, p, batch=B)
    elif kind == 1:
        fe = [O.Frontend(L, L, p, mono=True) for _ in range(B)]
        c = F.Context(L, L, p, batch=B, frontend_type=abi.FRONTEND_MONO)
    else:
        fe = [O.Frontend(L, L, p, depth=dp) for _ in range(B)]
        c = F.Context(L, L, p, batch=B, frontend_type=abi.FRONTEND_RGBD, depth=dp)
    ok = True
    kf = [starts[s] for s in range(B)]
    try:
        for i in range(6):
            idx = [starts[s] + i for s in range(B)]
            fr = [streams[s].frame(idx[s]) for s in range(B)]
            lefts = np.stack([f[0] for f in fr])
            if kind == 2:
                rights = np.stack([depth_image(h, w, idx[s], f32, seeds[s] + idx[s]) for s in range(B)])
            else:
                rights = np.stack([f[1] for f in fr])
            Rs = [synth.rig_keyframe_R_cur(streams[s], kf[s], idx[s]) for s in range(B)]
            if kind != 0:   # unrectified camera frame
                Rs = [streams[s].rotation(kf[s]).T @ streams[s].rotation(idx[s]) for s in range(B)]
            force = [int(rng.randint(0, 4) == 0) for _ in range(B)]
            ts = [i * 70_000_000 for _ in range(B)]
            c.step_host(lefts, None if kind == 1 else rights, c.make_inputs(ts, Rs, force))
            for s in range(B):
                exp = fe[s].process(lefts[s], rights[s] if kind != 1 else lefts[s], ts[s], Rs[s], bool(force[s]))
                got = c.get_output(s)
                keys = ["n_keypoints", "is_keyframe", "n_tracked", "n_detected", "n_measurements",
                        "tracking_status_mono", "tracking_status_stereo", "nr_stereo_putatives", "nr_stereo_inliers"]
                arrs = ["landmarks", "landmarks_age", "keypoints", "versors", "lkf_T_k_mono", "lkf_T_k_stereo"]
                if exp["is_keyframe"]:
                    arrs += ["left_rect_xy", "left_status", "meas_landmark", "meas_uL_uR_v"]
                    if kind != 1:
                        arrs += ["right_rect_xy", "right_status", "depth", "keypoints_3d"]
                    kf[s] = idx[s]
                for k in keys:
                    if got[k] != exp[k]:
                        ok = False
                        print(ci, "frame", i, "stream", s, "MISMATCH", k, got[k], exp[k])
                for k in arrs:
                    if not np.array_equal(got[k], exp[k], equal_nan=True):
                        ok = False
                        print(ci, "frame", i, "stream", s, "MISMATCH array", k)
            if not ok:
                break
    except F.KvfeError as e:
        ok = False
        print(ci, "DEVICE ERROR", e)
    finally:
        c.close()
    print(ci, "ok" if ok else "FAILED", desc, flush=True)
    bad += 0 if ok else 1

# dense stereo parameters
c = F.Context(*workloads.make_cameras(376, 240), P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=0))
dbad = 0
for ci in range(n_cfg):
    dp = abi.dense_stereo_params_default()
    dp.use_sgbm = int(rng.randint(0, 3) > 0)
    dp.use_mode_hh = int(rng.randint(0, 2))
    dp.num_disparities = int(rng.choice([16, 32, 48, 64]))
    dp.min_disparity = int(rng.choice([0, 1, 2, 7]))
    dp.sad_window_size = int(rng.choice([5, 7, 9, 11])) if dp.use_sgbm else int(rng.choice([5, 9, 15, 21]))
    dp.p1 = int(rng.choice([8, 50, 120]))
    dp.p2 = int(rng.choice([32, 200, 240]))
    dp.uniqueness_ratio = int(rng.choice([0, 5, 15]))
    dp.speckle_window_size = int(rng.choice([0, 50, 500]))
    dp.speckle_range = int(rng.choice([1, 3, 8]))
    dp.disp_12_max_diff = int(rng.choice([-1, 1, 3])) if dp.use_sgbm else -1
    dp.pre_filter_cap = int(rng.choice([5, 31, 63]))
    dp.texture_threshold = int(rng.choice([0, 10, 200]))
    dp.median_blur_disparity = int(rng.randint(0, 2))
    tex = synth.base_texture(376 + 80, 240, int(rng.randint(0, 100)))
    base = np.clip(np.rint(tex[96:96 + 240, 60:60 + 376 + 70]), 0, 255).astype(np.uint8)
    sh = int(rng.randint(2, 40))
    left, right = np.ascontiguousarray(base[:, :376]), np.ascontiguousarray(base[:, sh:sh + 376])
    if rng.randint(0, 3) == 0:
        right = np.clip(right.astype(np.int32) + rng.randint(-20, 20, right.shape), 0, 255).astype(np.uint8)
    desc = {f: getattr(dp, f) for f, _ in dp._fields_ if f != "reserved0"}
    try:
        got = c.dense_stereo_reconstruction(left, right, dp)
    except F.KvfeError as e:
        print("dense", ci, "refused:", e.status, desc)
        continue
    exp = O.dense_stereo_reconstruction(left, right, dp, list(c.rect.roi1), list(c.rect.roi2))
    if not np.array_equal(got, exp):
        dbad += 1
        print("dense", ci, "MISMATCH", np.count_nonzero(got != exp), desc)
c.close()
print("front-end configs failed:", bad, "of", n_cfg, "; dense configs failed:", dbad, "of", n_cfg)
