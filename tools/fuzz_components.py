"""Randomised GPU-vs-oracle comparison of the component calls (not part of the test suite): rectification,
LK (random points incl. the image border and outside, random window / levels / images), sub-pixel refinement,
epipolar stereo search, flow predictor, undistortion, on random image sizes incl. small ones.
Usage: python tools/fuzz_components.py [n_configs] [seed] [only]     (only = k: configuration k alone, with the random
draws of the ones before it replayed -- to reproduce a finding of a long run)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
# FUZZ_ODD_SIZES=1: widths / heights that are no multiples of 4 / 8 / 16 (the same number of random draws, so every other
# parameter of a configuration stays what it is without the switch) -- for runs under KVFE_GUARD_ALLOC, where an access
# past a row's or an image's end faults
ODD = int(os.environ.get("FUZZ_ODD_SIZES", "0"))
ODD_W, ODD_H = [98, 130, 163, 250, 321, 377, 751], [81, 97, 121, 193, 241, 363]
import oracle_lib as O
from kimera_vio_amd import _abi as abi, frontend as F, params as P, synth, workloads

G = os.path.join(ROOT, "tests", "golden")
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ONLY = int(sys.argv[3]) if len(sys.argv) > 3 else -1
bad = 0


TRACE = os.environ.get("FUZZ_TRACE") is not None


def check(name, a, b, desc):
    global bad
    if TRACE:
        print("   checked", name, flush=True)
    if not np.array_equal(a, b, equal_nan=True):
        bad += 1
        n = np.count_nonzero(a != b) if getattr(a, "shape", None) == getattr(b, "shape", None) else "shape"
        print("MISMATCH", name, n, desc, flush=True)
        return False
    return True


for ci in range(n_cfg):
    w = int(rng.choice(ODD_W if ODD else [96, 160, 256, 320, 377, 480, 752]))
    h = int(rng.choice(ODD_H if ODD else [80, 120, 192, 241, 360, 480]))
    L, R = workloads.make_cameras(w, h)
    p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=0)
    p.tracker.klt_win_size = int(rng.choice([9, 15, 16, 21, 24, 32]))
    p.tracker.klt_max_level = int(rng.choice([0, 1, 2, 4]))
    p.tracker.klt_max_iter = int(rng.choice([5, 30]))
    p.tracker.klt_eps = float(rng.choice([0.1, 0.01]))
    p.tracker.optical_flow_predictor_type = int(rng.randint(0, 2))
    p.stereo.templ_cols = int(rng.choice([21, 41, 101]))
    p.stereo.templ_rows = int(rng.choice([3, 11]))
    p.stereo.stripe_extra_rows = int(rng.choice([0, 2]))
    p.stereo.subpixel_refinement = int(rng.randint(0, 2))
    p.stereo.min_point_dist = float(rng.choice([0.3, 0.5, 1.0]))
    desc = dict(w=w, h=h, win=p.tracker.klt_win_size, lvl=p.tracker.klt_max_level, it=p.tracker.klt_max_iter,
                tc=p.stereo.templ_cols, tr=p.stereo.templ_rows, ex=p.stereo.stripe_extra_rows,
                ssub=p.stereo.subpixel_refinement)
    if p.stereo.templ_cols >= w:
        continue
    if ONLY >= 0 and ci != ONLY:   # the draws of a configuration that is skipped (same order as below)
        n = 200
        rng.randint(0, 1000); rng.randint(0, 256, (h, w)); rng.uniform(-6, w + 6, n); rng.uniform(-6, h + 6, n)
        rng.uniform(0, w - 1, n); rng.uniform(0, h - 1, n); rng.normal(0, 2.0, (n, 2))
        rng.normal(size=3); rng.uniform(0, 3); rng.uniform(-2, w + 2, n); rng.uniform(-2, h + 2, n); rng.randint(0, 10, n)
        continue
    print(ci, "start", desc, flush=True)
    try:
        c = F.Context(L, R, p)
    except F.KvfeError as e:
        print(ci, "create refused", e, desc)
        continue
    oc = O.Camera(L, R)
    try:
        st = synth.RigStream(L, R, seed=int(rng.randint(0, 1000)), rect_R1=np.array(c.rect.R1).reshape(3, 3))
        (l0, r0), (l1, _) = st.frame(0), st.frame(1)
        noise = rng.randint(0, 256, (h, w)).astype(np.uint8)
        # rectification
        for cam, img in ((0, l0), (1, r0), (0, noise)):
            check("rectify", c.undistort_rectify_image(cam, img), oc.rectify_image(cam, img), desc)
        # keypoints: inside, on the border, outside
        n = 200
        pts = np.stack([rng.uniform(-6, w + 6, n), rng.uniform(-6, h + 6, n)], 1).astype(np.float32)
        pts[:20] = np.rint(pts[:20])
        for cam in (0, 1):
            for useR, useP in ((1, 1), (1, 0), (0, 0)):
                check("undistort", c.undistort_rectify_keypoints(cam, pts, useR, useP),
                      oc.undistort_keypoints(cam, pts, useR, useP), desc)
            check("versors", c.get_bearing_vectors(cam, pts), oc.bearing_vectors(cam, pts), desc)
        # LK: prev -> cur with initial guesses around the truth, some wild
        inside = np.stack([rng.uniform(0, w - 1, n), rng.uniform(0, h - 1, n)], 1).astype(np.float32)
        init = inside + rng.normal(0, 2.0, inside.shape).astype(np.float32)
        init[:10] += 40
        inside[-10:] = pts[-10:]
        for prev, cur in ((l0, l1), (noise, np.roll(noise, 3, 1))):
            g = c.calc_optical_flow_pyr_lk(prev, cur, inside, init)
            e = O.calc_optical_flow_pyr_lk(prev, cur, inside, init, win=p.tracker.klt_win_size,
                                           max_level=p.tracker.klt_max_level, max_iter=p.tracker.klt_max_iter,
                                           eps=p.tracker.klt_eps)
            okk = check("lk status", g[1], e[1], desc)
            m = e[1] != 0
            check("lk xy", g[0][m], e[0][m], desc)
            check("lk err", g[2][m], e[2][m], desc)
        # sub-pixel refinement
        for win, zz, its, eps in ((10, -1, 40, 0.001), (3, -1, 10, 0.01), (5, 1, 5, 0.1)):
            if 2 * win + 5 >= min(w, h):
                continue
            sp = inside[:100]
            check("subpix", c.corner_subpix(l0, sp, win, zz, its, eps), O.corner_subpix(l0, sp, win, zz, its, eps), desc)
        # predictor
        Rm = synth.rot_from_axis_angle(rng.normal(size=3), np.deg2rad(rng.uniform(0, 3)))
        check("predict", c.predict_sparse_flow(inside, Rm),
              O.predict_sparse_flow(p.tracker.optical_flow_predictor_type, L, inside, Rm), desc)
        # stereo search on rectified images
        lr, rr = oc.rectify_image(0, l0), oc.rectify_image(1, r0)
        lxy = np.stack([rng.uniform(-2, w + 2, n), rng.uniform(-2, h + 2, n)], 1).astype(np.float32)
        lst = (rng.randint(0, 10, n) == 0).astype(np.uint8)
        g = c.get_right_keypoints_rectified(lr, rr, lxy, lst)
        e = oc.get_right_keypoints_rectified(lr, rr, lxy, lst, p.stereo)
        check("stereo status", g[1], e[1], desc)
        check("stereo xy", g[0], e[0], desc)
        # full sparse stereo from raw images
        kps = inside[:150]
        g = c.sparse_stereo_reconstruction(l0, r0, kps)
        e = oc.sparse_stereo(l0, r0, kps, p.stereo)
        for k in ("left_rect_xy", "left_status", "right_rect_xy", "right_status", "depth", "keypoints_3d"):
            if k in e:
                check("sparse " + k, g[k], e[k], desc)
        check("equalize", c.equalize_hist(l0), O.equalize_hist(l0), desc)
    except F.KvfeError as e:
        bad += 1
        print(ci, "DEVICE ERROR", e, desc)
    finally:
        c.close()
    print(ci, "done", desc, flush=True)
print("mismatching checks:", bad)
