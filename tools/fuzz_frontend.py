"""Randomised GPU-vs-oracle comparison of the stereo front-end (not part of the test suite): random image
sizes, feature counts, window sizes, pyramid depths, ANMS types, stereo template sizes; a few frames each.
Usage: python tools/fuzz_frontend.py [n_configs] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
# FUZZ_ODD_SIZES=1: widths / heights that are no multiples of 4 / 8 / 16 (the same number of random draws, so every other
# parameter of a configuration stays what it is without the switch) -- for runs under KVFE_GUARD_ALLOC, where an access
# past a row's or an image's end faults
ODD = int(os.environ.get("FUZZ_ODD_SIZES", "0"))
ODD_W, ODD_H = [250, 321, 377, 481, 642, 750], [193, 241, 290, 363, 479]
import oracle_lib as O
from kimera_vio_amd import _abi as abi, frontend as F, params as P, synth, workloads

G = os.path.join(ROOT, "tests", "golden")
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ORACLE_ONLY = int(sys.argv[3]) if len(sys.argv) > 3 else -1   # config index: print the oracle's outputs, no GPU
bad = 0
for ci in range(n_cfg):
    w = int(rng.choice(ODD_W if ODD else [256, 320, 376, 480, 640, 752]))
    h = int(rng.choice(ODD_H if ODD else [192, 240, 288, 360, 480]))
    L, R = workloads.make_cameras(w, h)
    p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=int(rng.randint(0, 2)))
    d, t, s = p.detector, p.tracker, p.stereo
    d.max_features_per_frame = int(rng.choice([40, 100, 200, 400]))
    d.min_distance = int(rng.choice([1, 3, 5, 10, 20, 33, 50]))
    d.quality_level = float(rng.choice([0.0002, 0.001, 0.01, 0.05]))
    d.max_nr_keypoints_before_anms = int(rng.choice([150, 2000, 6000]))
    d.non_max_suppression_type = int(rng.choice([0, 1, 2, 3, 4, 5, 6]))
    d.enable_subpixel_corner_refinement = int(rng.randint(0, 2))
    t.klt_win_size = int(rng.choice([16, 24, 32, 21, 15]))
    t.klt_max_level = int(rng.choice([1, 2, 3, 4]))
    t.klt_max_iter = int(rng.choice([10, 30]))
    t.max_feature_track_age = int(rng.choice([3, 25]))
    s.templ_cols = int(rng.choice([41, 61, 101]))
    s.templ_rows = int(rng.choice([5, 11]))
    s.subpixel_refinement = int(rng.randint(0, 2))
    p.tracker.ransac_use_1point_stereo = int(rng.randint(0, 2))
    p.tracker.ransac_use_2point_mono = int(rng.randint(0, 2))
    desc = dict(w=w, h=h, feats=d.max_features_per_frame, md=d.min_distance, q=d.quality_level,
                anms=d.non_max_suppression_type, subpix=d.enable_subpixel_corner_refinement, win=t.klt_win_size,
                lvl=t.klt_max_level, it=t.klt_max_iter, age=t.max_feature_track_age, tc=s.templ_cols, tr=s.templ_rows,
                ssub=s.subpixel_refinement, ransac=p.use_ransac, one=p.tracker.ransac_use_1point_stereo,
                two=p.tracker.ransac_use_2point_mono)
    stream_seed = int(rng.randint(0, 1000))
    forces = [bool(rng.randint(0, 2)) for _ in range(5)]
    R1 = np.array(F.compute_rectification(L, R).R1).reshape(3, 3)
    if ORACLE_ONLY >= 0:
        if ci != ORACLE_ONLY:
            continue
        fe = O.Frontend(L, R, p)
        st = synth.RigStream(L, R, seed=stream_seed, rect_R1=R1)
        kf = 0
        print(desc)
        for i in range(5):
            l, r = st.frame(i)
            exp = fe.process(l, r, i * 60_000_000, synth.rig_keyframe_R_cur(st, kf, i), forces[i])
            print(i, {k: exp[k] for k in ("n_keypoints", "is_keyframe", "n_tracked", "n_detected", "n_measurements")})
            if exp["is_keyframe"]:
                kf = i
        continue
    try:
        c = F.Context(L, R, p, batch=1)
    except F.KvfeError as e:
        print(ci, "create refused:", e, desc)
        continue
    fe = O.Frontend(L, R, p)
    st = synth.RigStream(L, R, seed=stream_seed, rect_R1=R1)
    kf = 0
    ok = True
    try:
        for i in range(5):
            l, r = st.frame(i)
            Rk = synth.rig_keyframe_R_cur(st, kf, i)
            force = forces[i]
            ts = i * 60_000_000
            c.step_host(l[None], r[None], c.make_inputs([ts], [Rk], [int(force)]))
            exp = fe.process(l, r, ts, Rk, force)
            try:
                got = c.get_output(0)
            except F.KvfeError as e:
                # kvfe.h, kvfe_config.candidate_capacity: the minimum-distance filter holds 8192 accepted corners, the tree /
                # range / SSC suppression 4096 keypoints -- reachable only with max_nr_keypoints_before_anms beyond that
                if e.status == -5 and d.max_nr_keypoints_before_anms > 4096:
                    print(ci, "frame", i, "refused by a documented capacity:", e)
                    break
                ok = False
                print(ci, "frame", i, "DEVICE ERROR", e, "oracle:", {k: exp[k] for k in ("n_keypoints", "n_tracked", "n_detected")})
                break
            keys = ["n_keypoints", "is_keyframe", "n_tracked", "n_detected", "n_measurements",
                    "tracking_status_mono", "tracking_status_stereo"]
            arrs = ["landmarks", "landmarks_age", "keypoints", "versors"]
            if exp["is_keyframe"]:
                arrs += ["left_rect_xy", "left_status", "right_rect_xy", "right_status", "depth", "keypoints_3d",
                         "meas_landmark", "lkf_T_k_mono", "lkf_T_k_stereo"]
                kf = i
            for k in keys:
                if got[k] != exp[k]:
                    ok = False
                    print(ci, "frame", i, "MISMATCH", k, got[k], exp[k])
            for k in arrs:
                if not np.array_equal(got[k], exp[k], equal_nan=True):
                    ok = False
                    print(ci, "frame", i, "MISMATCH array", k, np.count_nonzero(got[k] != exp[k]) if got[k].shape == exp[k].shape else (got[k].shape, exp[k].shape))
            if not ok:
                break
    finally:
        c.close()
    print(ci, "ok" if ok else "FAILED", desc, flush=True)
    bad += 0 if ok else 1
print("configs failed:", bad, "of", n_cfg)
