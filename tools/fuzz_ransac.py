"""Randomised GPU-vs-oracle comparison of the geometric outlier rejection components (not part of the test
suite): 5-point (Nister) and 2-point mono RANSAC, 3-point (Arun) RANSAC and 1-point voting, on random
motions, match counts (incl. fewer than a sample), outlier ratios, noise levels, thresholds, iteration caps.
Every field must be identical: status, inlier set, iteration count, pose, information matrix.
Usage: python tools/fuzz_ransac.py [n_configs] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
import test_oracle_ransac as TR
from kimera_vio_amd import frontend as F, params as P, workloads

G = os.path.join(ROOT, "tests", "golden")
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed0)
L, R_ = workloads.make_cameras(752, 480)
ocam = O.Camera(L, R_)
bad = 0


def same(name, got, exp, desc):
    global bad
    ok = (got["status"] == exp["status"] and list(got["inliers"]) == list(exp["inliers"])
          and got["iterations"] == exp["iterations"] and np.array_equal(got["pose"], exp["pose"])
          and np.array_equal(got["info"], exp["info"]))
    if not ok:
        bad += 1
        print("MISMATCH", name, desc, "status", got["status"], exp["status"], "inliers", len(got["inliers"]),
              len(exp["inliers"]), "iterations", got["iterations"], exp["iterations"], flush=True)


for ci in range(n_cfg):
    p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=None)
    t = p.tracker
    t.ransac_use_2point_mono = 0
    t.ransac_use_1point_stereo = 0
    t.ransac_max_iterations = int(rng.choice([20, 100, 500]))
    t.ransac_threshold_mono = float(rng.choice([1e-6, 1e-5, 1e-4]))
    t.ransac_threshold_stereo = float(rng.choice([1e-3, 0.05, 1.0]))
    t.ransac_probability = float(rng.choice([0.9, 0.995]))
    t.min_nr_mono_inliers = int(rng.choice([10, 40]))
    t.min_nr_stereo_inliers = int(rng.choice([5, 40]))
    n_in = int(rng.choice([0, 4, 7, 8, 9, 30, 120, 400]))
    n_out = int(rng.choice([0, 0, 5, 60, 300])) if n_in else int(rng.choice([0, 12]))
    planar = bool(rng.integers(0, 2))
    noise_b = float(rng.choice([0.0, 1e-5, 1e-3]))
    w = rng.normal(size=3) * float(rng.choice([0.0, 0.01, 0.2]))
    Rm = TR.expmap(w)
    T = rng.normal(size=3) * float(rng.choice([0.05, 1.0]))
    if np.linalg.norm(T) < 1e-3:
        T = np.array([0.1, 0.0, 0.0])
    desc = dict(ci=ci, n_in=n_in, n_out=n_out, planar=planar, noise=noise_b, it=t.ransac_max_iterations,
                thr=t.ransac_threshold_mono, thr3=t.ransac_threshold_stereo, prob=t.ransac_probability)
    c = F.Context(L, R_, p)
    try:
        srng = np.random.default_rng(1000 * seed0 + ci)
        f_ref, f_cur = TR.mono_scene(ocam, srng, Rm, T, planar, n_in, n_out)
        if noise_b and len(f_cur):
            f_cur = f_cur + srng.normal(0, noise_b, f_cur.shape)
            f_cur /= np.linalg.norm(f_cur, axis=1, keepdims=True)
        if len(f_ref):   # geometricOutlierRejection2d2d CHECKs for a non-empty match set (Tracker.cpp:243)
            same("5-point", c.outlier_rejection_2d2d(f_ref, f_cur), O.outlier_rejection_2d2d(f_ref, f_cur, t), desc)
            same("2-point", c.outlier_rejection_2d2d_given_rotation(f_ref, f_cur, Rm),
                 O.outlier_rejection_2d2d_given_rotation(f_ref, f_cur, Rm, t), desc)
        else:            # ... for both solvers: the CHECK abort is KVFE_ERR_INVALID_ARG at the C boundary
            for call in (lambda: c.outlier_rejection_2d2d(f_ref, f_cur),
                         lambda: c.outlier_rejection_2d2d_given_rotation(f_ref, f_cur, Rm)):
                try:
                    call()
                    bad += 1
                    print("MISMATCH empty match set accepted", desc)
                except F.KvfeError as e:
                    assert e.status == -1, e
        Ts = np.array([ocam.rect.baseline, 0.0, 0.0]) * float(rng.choice([1.0, 3.0]))
        rl, rr, p_ref, cl, cr, p_cur = TR.stereo_scene(ocam, srng, Rm, Ts, planar, n_in, n_out,
                                                       float(rng.choice([0.0, 0.002])))
        same("3-point", c.outlier_rejection_3d3d(p_ref, p_cur), O.outlier_rejection_3d3d(p_ref, p_cur, t), desc)
        g = c.outlier_rejection_3d3d_given_rotation(rl, rr, p_ref, cl, cr, p_cur, Rm)
        e = O.outlier_rejection_3d3d_given_rotation(ocam, rl, rr, p_ref, cl, cr, p_cur, Rm, t)
        g["iterations"] = e["iterations"] = 1
        same("1-point", g, e, desc)
    except F.KvfeError as e:
        bad += 1
        print(ci, "DEVICE ERROR", e, desc)
    finally:
        c.close()
    print(ci, "done", desc, flush=True)
print("mismatching checks:", bad)
