"""Randomised GPU-vs-oracle comparison, part 3 (round 6; not part of the test suite, a fixed-seed slice is): BATCHED
stereo front-ends -- 5 to 24 streams that see different sequences at different keyframe cadences -- through every entry
point a caller can feed frames with (kvfe_frontend_step_host, _device with device_frames_persist 0 / 1, _staged), random
image sizes, feature counts, pyramid depths, tracking windows (24 = the four-points-per-wave kernel, others = the
one-point kernel), ANMS types and outlier-rejection options.  The fuzzers of parts 1 and 2 stop at three streams; the
batched code paths (one fork per step instead of the few-stream arrangement, the grouped cornerSubPix kernel chosen by
the number of new corners, interleaved dispatch of the tracking blocks, the staged input's copy stream) were covered by
the fixed bench workloads only.
Usage: python tools/fuzz_batched.py [n_configs] [seed] [only]"""
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
PYR_PROBE = int(os.environ.get("FUZZ_PYR_PROBE", "0"))   # read the context's own pyramid back after every device0 step

KEYS = ["n_keypoints", "is_keyframe", "n_tracked", "n_detected", "n_measurements", "tracking_status_mono",
        "tracking_status_stereo", "nr_stereo_putatives", "nr_stereo_inliers"]
ARRS = ["landmarks", "landmarks_age", "keypoints", "versors", "lkf_T_k_mono", "lkf_T_k_stereo"]
KF_ARRS = ["left_rect_xy", "left_status", "meas_landmark", "meas_uL_uR_v", "right_rect_xy", "right_status", "depth",
           "keypoints_3d"]


def run(n_cfg, seed, only=-1, verbose=True):
    """returns the number of configurations that failed"""
    rng = np.random.RandomState(seed)
    bad = 0
    for ci in range(n_cfg):
        w = int(rng.choice(ODD_W if ODD else [256, 320, 376, 480, 752]))
        h = int(rng.choice(ODD_H if ODD else [192, 240, 288, 480]))
        B = int(rng.choice([5, 6, 8, 9, 12, 16, 17, 24]))
        entry = int(rng.choice([0, 1, 2, 3]))        # host | device persist 0 | device persist 1 | staged
        n_steps = int(rng.choice([5, 7, 9]))
        p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=int(rng.randint(0, 2)))
        d, t = p.detector, p.tracker
        d.max_features_per_frame = int(rng.choice([60, 150, 300, 600]))
        d.non_max_suppression_type = int(rng.choice([0, 4, 6, 6]))
        d.min_distance = int(rng.choice([5, 10, 20]))
        t.klt_win_size = int(rng.choice([24, 24, 24, 16, 21]))
        t.klt_max_level = int(rng.choice([1, 2, 3, 4]))
        t.klt_max_iter = int(rng.choice([10, 30]))
        t.max_feature_track_age = int(rng.choice([3, 4, 25]))
        t.ransac_use_1point_stereo = int(rng.randint(0, 2))
        t.ransac_use_2point_mono = int(rng.randint(0, 2))
        p.min_intra_keyframe_time_ns = float(rng.choice([0.0, 1.0e8, 2.0e8]))
        p.stereo.subpixel_refinement = int(rng.randint(0, 2))
        force_p = int(rng.choice([1, 2, 4, 1000]))   # a stream's frame is a forced keyframe with probability 1 / force_p
        desc = dict(w=w, h=h, B=B, entry=["host", "device0", "device1", "staged"][entry], steps=n_steps,
                    feats=d.max_features_per_frame, anms=d.non_max_suppression_type, md=d.min_distance, win=t.klt_win_size,
                    lvl=t.klt_max_level, it=t.klt_max_iter, age=t.max_feature_track_age, ransac=p.use_ransac,
                    one=t.ransac_use_1point_stereo, two=t.ransac_use_2point_mono, kf_ns=p.min_intra_keyframe_time_ns,
                    ssub=p.stereo.subpixel_refinement, force_p=force_p)
        groups = 0
        if seed >= 100:   # extensions of the second campaign, drawn from their own generator: the configurations of the
            r2 = np.random.RandomState(seed * 1000 + ci)   # seeds below 100 (a regression test replays one) stay what they were
            if r2.randint(0, 4) == 0:
                B = int(r2.choice([3, 4, 32, 40]))          # few streams in the batched call (fork_swap) / more than three groups of 8
            if r2.randint(0, 5) == 0:
                t.klt_max_level = 0                         # no pyramid launch: the level-0 copy is a 2-D memcpy
            if r2.randint(0, 6) == 0:
                p.stereo.equalize_image = 1                 # context-owned equalised slots in front of every entry point
            if r2.randint(0, 5) == 0 and entry in (0, 1, 2) and B >= 6:
                groups = int(r2.choice([2, 3]))             # kvfe_config.stream_groups: child contexts on their own streams
            if r2.randint(0, 12) == 0 and B <= 8:
                w, h = 1280, 720                            # two waves along x in the pyramid launch
            desc.update(B=B, lvl=t.klt_max_level, eq=p.stereo.equalize_image, groups=groups, w=w, h=h)
        seeds = [int(rng.randint(0, 1000)) for _ in range(B)]
        starts = [int(rng.randint(0, 3)) for _ in range(B)]
        forces = rng.randint(0, force_p, (n_steps, B)) == 0
        if only >= 0 and ci != only:
            continue
        L, R = workloads.make_cameras(w, h)
        R1 = np.array(F.compute_rectification(L, R).R1).reshape(3, 3)
        streams = [synth.RigStream(L, R, seed=sd, rect_R1=R1) for sd in seeds]
        try:
            c = F.Context(L, R, p, batch=B, stream_groups=groups,
                          **({"device_frames_persist": entry - 1} if entry in (1, 2) else {}))
        except F.KvfeError as e:
            if verbose:
                print(ci, "create refused", e, desc, flush=True)
            continue
        fe = [O.Frontend(L, R, p) for _ in range(B)]
        ok = True
        kf = list(starts)
        keep = []     # device buffers of the frames a persisting context may still read
        try:
            for i in range(n_steps):
                idx = [starts[s] + i for s in range(B)]
                fr = [streams[s].frame(idx[s]) for s in range(B)]
                lefts = np.ascontiguousarray(np.stack([f[0] for f in fr]))
                rights = np.ascontiguousarray(np.stack([f[1] for f in fr]))
                Rs = [synth.rig_keyframe_R_cur(streams[s], kf[s], idx[s]) for s in range(B)]
                force = [int(forces[i, s]) for s in range(B)]
                ts = [i * 70_000_000 for _ in range(B)]
                inp = c.make_inputs(ts, Rs, force)
                if entry == 0:
                    c.step_host(lefts, rights, inp)
                elif entry in (1, 2):
                    import torch
                    dl, dr = torch.from_numpy(lefts).cuda(), torch.from_numpy(rights).cuda()
                    torch.cuda.synchronize()
                    keep.append((dl, dr))
                    c.step_device(dl.data_ptr(), dr.data_ptr(), inp)
                else:
                    a, b = c.staging_buffers(i % 3)
                    a[:] = lefts
                    b[:] = rights
                    c.step_staged(i % 3, inp)
                if PYR_PROBE and entry == 1 and not groups and t.klt_max_level > 0:   # the context's own pyramid of this frame, checked in place
                    lv, cp = c.debug_pyramid(0, True)
                    for s in range(B):
                        exp_l, src = [], lefts[s]
                        for _ in lv[s]:
                            src = O.pyr_down(src)
                            exp_l.append(src)
                        for name, g_, e_ in [("level0_copy", cp[s], lefts[s])] + [("level%d" % (k_ + 1), lv[s][k_], exp_l[k_])
                                                                                    for k_ in range(len(lv[s]))]:
                            if not np.array_equal(g_, e_):
                                ys, xs = np.nonzero(g_ != e_)
                                print(ci, "frame", i, "stream", s, "PYRAMID", name, "differs at", len(ys), "pixels: rows",
                                      int(ys.min()), "-", int(ys.max()), "cols", int(xs.min()), "-", int(xs.max()),
                                      "got", g_[ys[:8], xs[:8]].tolist(), "expected", e_[ys[:8], xs[:8]].tolist(),
                                      "| cols of row", int(ys[0]), ":", xs[ys == ys[0]][:24].tolist(), flush=True)
                for s in range(B):
                    exp = fe[s].process(lefts[s], rights[s], ts[s], Rs[s], bool(force[s]))
                    got = c.get_output(s)
                    arrs = list(ARRS)
                    if exp["is_keyframe"]:
                        arrs += KF_ARRS
                        kf[s] = idx[s]
                    for k in KEYS:
                        if got[k] != exp[k]:
                            ok = False
                            print(ci, "frame", i, "stream", s, "MISMATCH", k, got[k], exp[k], flush=True)
                    for k in arrs:
                        if not np.array_equal(got[k], exp[k], equal_nan=True):
                            ok = False
                            print(ci, "frame", i, "stream", s, "MISMATCH array", k, flush=True)
                            if only >= 0 and getattr(got[k], "shape", None) == getattr(exp[k], "shape", None):
                                ga, ea = np.asarray(got[k]), np.asarray(exp[k])
                                rows = np.nonzero(np.any((ga != ea).reshape(len(ga), -1), axis=1))[0]
                                for r_ in rows[:8]:
                                    print("      row", int(r_), "of", len(ga), "got", ga[r_], "expected", ea[r_],
                                          "| landmark", got["landmarks"][r_], "age", got["landmarks_age"][r_],
                                          "n_tracked", got["n_tracked"], flush=True)
                if not ok:
                    break
        except F.KvfeError as e:
            ok = False
            print(ci, "DEVICE ERROR", e, flush=True)
        finally:
            c.close()
        if verbose or not ok:
            print(ci, "ok" if ok else "FAILED", desc, flush=True)
        bad += 0 if ok else 1
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    on = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    rep = int(os.environ.get("FUZZ_REPEAT", "1"))   # the `only` configuration again and again (an intermittent finding)
    b = sum(run(n, sd, on, verbose=(rep == 1)) for _ in range(rep))
    print("configs failed:", b, "of", n)
    sys.exit(1 if b else 0)
