"""Known-answer tests of the CPU oracle's geometric outlier rejection (oracle/kimera_ransac.cpp,
oracle/opengv_re.cpp), restating the reference's own tests of these call sites:

  tests/testTracker.cpp:804-895   geometricOutlierRejection2d2dGivenRotation
  tests/testTracker.cpp:898-1039  geometricOutlierRejection3d3d (3-point Arun)
  tests/testTracker.cpp:1042-1185 geometricOutlierRejection3d3dGivenRotation
  tests/testTracker.cpp:1188-1233 getPoint3AndCovariance (Monte-Carlo covariance)

The reference draws its synthetic scenes with libc rand(); the scenes here are drawn with numpy with
the same construction (random pixels, depth ranges, outlier rule cos(angle) <= 0.9) and the same
assertions (every inlier kept, every outlier rejected, translation recovered).
"""
import os

import numpy as np
import pytest

import oracle_lib as O
from kimera_vio_amd import _abi as abi
from kimera_vio_amd import params as P

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
W, H = 752, 480


@pytest.fixture(scope="module")
def cam():
    L = P.load_camera_params(os.path.join(G, "sensorLeft.yaml"))
    R = P.load_camera_params(os.path.join(G, "sensorRight.yaml"))
    return O.Camera(L, R)


def expmap(w):
    w = np.asarray(w, float)
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def tracker_params(**kw):
    tp = P.default_frontend_params().tracker  # class defaults (VisionImuTrackerParams.h)
    tp.ransac_randomize = 0
    for k, v in kw.items():
        setattr(tp, k, v)
    return tp


def rand_px(rng, n):
    return np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], 1).astype(np.float32)


def mono_scene(cam, rng, R, T, planar, n_in, n_out):
    """AddNonPlanarInliersToFrame / AddPlanarInliersToFrame / AddOutliersToFrame (testTracker.cpp:212-330)."""
    v_ref = cam.bearing_vectors(0, rand_px(rng, n_in))
    tn = np.linalg.norm(T)
    if planar:
        N = np.array([0.1, -0.1, 1.0])
        pts = v_ref * (tn / (v_ref @ N))[:, None]
    else:
        depth = tn + (10 * tn - tn) * rng.random(n_in)
        pts = v_ref * depth[:, None]
    v_cur = (R.T @ (pts - T).T).T
    v_cur /= np.linalg.norm(v_cur, axis=1, keepdims=True)
    fr, fc = [v_ref], [v_cur]
    k = 0
    while k < n_out:
        a = cam.bearing_vectors(0, rand_px(rng, 1))[0]
        b = cam.bearing_vectors(0, rand_px(rng, 1))[0]
        proj = R.T @ (a * tn - T)
        proj /= np.linalg.norm(proj)
        if proj @ b > 0.9:
            continue
        fr.append(a[None])
        fc.append(b[None])
        k += 1
    return np.concatenate(fr), np.concatenate(fc)


@pytest.mark.parametrize("planar,n_in,n_out", [(False, 80, 0), (False, 80, 20), (True, 80, 20)])
@pytest.mark.parametrize("policy", [abi.RNG_LIBSTDCXX_PRE11, abi.RNG_LIBSTDCXX_11])
def test_2d2d_given_rotation(cam, planar, n_in, n_out, policy):
    rng = np.random.default_rng(3 + n_out + int(planar))
    R, T = np.eye(3), np.array([1.0, 0, 0])  # testTracker.cpp:813-815
    f_ref, f_cur = mono_scene(cam, rng, R, T, planar, n_in, n_out)
    r = O.outlier_rejection_2d2d_given_rotation(f_ref, f_cur, R, tracker_params(ransac_rng_policy=policy))
    assert r["status"] == abi.TRACKING_VALID
    assert list(r["inliers"]) == list(range(n_in))  # every inlier kept, every outlier rejected
    assert np.allclose(r["pose"][:, :3], R)
    t = r["pose"][:, 3]
    assert abs(np.linalg.norm(t) - 1) < 1e-12 and abs(abs(t @ T) - 1) < 1e-6  # translation direction


def test_2d2d_with_rotation_and_few_matches(cam):
    rng = np.random.default_rng(11)
    R, T = expmap([0.02, -0.01, 0.03]), np.array([0.3, 0.1, -0.05])
    f_ref, f_cur = mono_scene(cam, rng, R, T, False, 60, 15)
    r = O.outlier_rejection_2d2d_given_rotation(f_ref, f_cur, R, tracker_params())
    assert r["status"] == abi.TRACKING_VALID and list(r["inliers"]) == list(range(60))
    d = T / np.linalg.norm(T)
    assert abs(abs(r["pose"][:, 3] @ d) - 1) < 1e-6
    # fewer inliers than minNrMonoInliers -> FEW_MATCHES (Tracker.cpp:285-289)
    r = O.outlier_rejection_2d2d_given_rotation(f_ref[:6], f_cur[:6], R, tracker_params())
    assert r["status"] == abi.TRACKING_FEW_MATCHES and r["n_inliers"] == 6
    # one correspondence cannot be sampled: RANSAC fails -> INVALID, identity pose
    r = O.outlier_rejection_2d2d_given_rotation(f_ref[:1], f_cur[:1], R, tracker_params())
    assert r["status"] == abi.TRACKING_INVALID and r["n_inliers"] == 0
    assert np.array_equal(r["pose"], np.eye(3, 4))


def stereo_project(cam, p):
    """gtsam::StereoCamera(Pose3(), Cal3_S2Stereo(P1, baseline)).project"""
    P1 = np.array(cam.rect.P1).reshape(3, 4)
    fx, fy, cx, cy, b = P1[0, 0], P1[1, 1], P1[0, 2], P1[1, 2], cam.rect.baseline
    uL = fx * p[:, 0] / p[:, 2] + cx
    uR = fx * (p[:, 0] - b) / p[:, 2] + cx
    v = fy * p[:, 1] / p[:, 2] + cy
    return np.stack([uL, v], 1).astype(np.float32), uR.astype(np.float32)


def stereo_scene(cam, rng, R, T, planar, n_in, n_out, noise=0.0):
    """AddNonPlanarInliersToStereoFrame / AddPlanarInliersToStereoFrame / AddOutliersToStereoFrame
    (testTracker.cpp:393-505)."""
    tn = np.linalg.norm(T)
    lo, hi = 10 * tn, 20 * tn
    v = cam.bearing_vectors(0, rand_px(rng, n_in))
    if planar:
        p_ref = v * ((20 * tn) / v[:, 2])[:, None]
    else:
        p_ref = v * (lo + (hi - lo) * rng.random(n_in))[:, None]
    p_cur = (R.T @ (p_ref - T).T).T
    pr, pc = [p_ref], [p_cur]
    k = 0
    while k < n_out:
        a = cam.bearing_vectors(0, rand_px(rng, 1))[0] * (lo + (hi - lo) * rng.random())
        b = cam.bearing_vectors(0, rand_px(rng, 1))[0] * (lo + (hi - lo) * rng.random())
        proj = R.T @ (a - T)
        if proj @ b / np.linalg.norm(proj) / np.linalg.norm(b) > 0.9:
            continue
        pr.append(a[None])
        pc.append(b[None])
        k += 1
    p_ref, p_cur = np.concatenate(pr), np.concatenate(pc)
    rl, rr = stereo_project(cam, p_ref)
    cl, cr = stereo_project(cam, p_cur)
    if noise:
        # AddNoiseToStereoFrame (testTracker.cpp:507-517) seeds a fresh default_random_engine per call,
        # so the reference adds the SAME noise sequence to both frames
        nz = rng.normal(0, noise, p_ref.shape)
        p_ref = p_ref + nz
        p_cur = p_cur + nz
    return rl, rr, p_ref, cl, cr, p_cur


@pytest.mark.parametrize("case,planar,n_in,n_out,noise", [(0, False, 3, 0, 0.0), (1, False, 40, 0, 0.0),
                                                          (2, False, 80, 40, 0.0), (3, True, 80, 40, 0.01)])
def test_3d3d_given_rotation(cam, case, planar, n_in, n_out, noise):
    rng = np.random.default_rng(100 + case)
    R = expmap([0.1, 0.1, 0.1])                     # testTracker.cpp:1047-1049
    T = np.array([cam.rect.baseline, 0.0, 0.0])
    rl, rr, p_ref, cl, cr, p_cur = stereo_scene(cam, rng, R, T, planar, n_in, n_out, noise)
    tp = tracker_params()
    r = O.outlier_rejection_3d3d_given_rotation(cam, rl, rr, p_ref, cl, cr, p_cur, R, tp)
    assert list(r["inliers"]) == list(range(n_in))
    assert r["status"] == (abi.TRACKING_VALID if n_in >= tp.min_nr_stereo_inliers else abi.TRACKING_FEW_MATCHES)
    t = r["pose"][:, 3]
    if case < 2:
        assert np.allclose(t, T, atol=1e-3)          # :1164-1169
    tol = 1e-3 if case < 2 else 1e-2
    exp = (R.T @ (p_ref[:n_in] - t).T).T             # :1174-1181
    assert np.all(np.linalg.norm(exp - p_cur[:n_in], axis=1) < tol)
    assert np.allclose(r["pose"][:, :3], R)
    info = r["info"]
    assert np.allclose(info, info.T, rtol=1e-9) and np.all(np.linalg.eigvalsh(info) > 0)


def test_3d3d_given_rotation_no_solution(cam):
    rng = np.random.default_rng(5)
    R, T = expmap([0.1, 0.1, 0.1]), np.array([cam.rect.baseline, 0.0, 0.0])
    tp = tracker_params()
    # a single match: the coherent set has size 1 < 2 -> INVALID, Pose3(), zero information (Tracker.cpp:553-558)
    rl, rr, p_ref, cl, cr, p_cur = stereo_scene(cam, rng, R, T, False, 1, 0)
    r = O.outlier_rejection_3d3d_given_rotation(cam, rl, rr, p_ref, cl, cr, p_cur, R, tp)
    assert r["status"] == abi.TRACKING_INVALID and r["n_inliers"] == 0
    assert np.array_equal(r["pose"], np.eye(3, 4)) and not r["info"].any()
    r = O.outlier_rejection_3d3d_given_rotation(cam, rl[:0], rr[:0], p_ref[:0], cl[:0], cr[:0], p_cur[:0], R, tp)
    assert r["status"] == abi.TRACKING_INVALID


def test_get_point3_and_covariance_monte_carlo(cam):
    """testTracker.cpp:1188-1233: the propagated covariance against 10^6 Monte-Carlo samples, tol 0.2."""
    P1 = np.array(cam.rect.P1).reshape(3, 4)
    fx, fy, cx, cy, b = P1[0, 0], P1[1, 1], P1[0, 2], P1[1, 2], cam.rect.baseline
    xL, v = 379.999 / 2, 255.238 / 2
    xR = xL - 10

    def backproject(uL, uR, vv):
        z = b * fx / (uL - uR)
        return np.stack([z * (uL - cx) / fx, z * (vv - cy) / fy, z], -1)

    p3 = backproject(xL, xR, v)
    point, cov = O.get_point3_and_covariance(cam, xL, xR, v, p3)
    assert np.array_equal(point, p3)
    rng = np.random.default_rng(0)
    n = 1_000_000
    s = backproject(xL + rng.normal(0, 1, n), xR + rng.normal(0, 1, n), v + rng.normal(0, 1, n))
    d = s - p3
    mc = d.T @ d / (n - 1)
    assert np.allclose(s.mean(0), point, atol=0.2)
    assert np.allclose(mc, cov, atol=0.2)
    # rotated into another frame: R cov R^T
    Rm = expmap([0.1, -0.2, 0.05])
    point_r, cov_r = O.get_point3_and_covariance(cam, xL, xR, v, p3, Rm)
    assert np.allclose(point_r, Rm @ p3) and np.allclose(cov_r, Rm @ cov @ Rm.T, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("case,n_in,n_out", [(0, 3, 0), (1, 40, 0), (2, 80, 40)])
def test_3d3d_arun_ransac(cam, case, n_in, n_out):
    """testTracker.cpp:898-1039: opengv PointCloudSacProblem (3-point Arun), threshold 0.1 there is
    tracker default ransac_threshold_stereo; here the noiseless scenes with a tight threshold."""
    rng = np.random.default_rng(200 + case)
    R = expmap([0.1, 0.1, 0.1])
    T = np.array([cam.rect.baseline, 0.0, 0.0])
    _, _, p_ref, _, _, p_cur = stereo_scene(cam, rng, R, T, False, n_in, n_out)
    r = O.ransac_point_cloud(p_ref, p_cur, 1e-3, 100, 0.995)
    assert r is not None
    assert list(r["inliers"]) == list(range(n_in))
    assert np.allclose(r["pose"][:, :3], R, atol=1e-6) and np.allclose(r["pose"][:, 3], T, atol=1e-6)


@pytest.mark.parametrize("planar,n_in,n_out", [(False, 82, 0), (False, 80, 40), (True, 80, 40)])
def test_2d2d_five_point_nister(cam, planar, n_in, n_out):
    """testTracker.cpp:704-802 (ransac_use_2point_mono_ = false, 1000 iterations): opengv
    CentralRelativePoseSacProblem(NISTER) keeps every synthetic inlier and rejects every outlier; rotation and
    translation direction recovered (the translation norm is the scale of the essential matrix: arbitrary)."""
    rng = np.random.default_rng(5 + n_out + int(planar))
    R, T = expmap([0.01, 0.01, 0.01]), np.array([1.0, 0.0, 0.0])
    f_ref, f_cur = mono_scene(cam, rng, R, T, planar, n_in, n_out)
    r = O.outlier_rejection_2d2d(f_ref, f_cur, tracker_params(ransac_max_iterations=1000, ransac_use_2point_mono=0))
    assert r["status"] == abi.TRACKING_VALID
    assert list(r["inliers"]) == list(range(n_in))
    assert np.allclose(r["pose"][:, :3], R, atol=1e-9)
    t = r["pose"][:, 3]
    assert abs(abs(t @ T) / np.linalg.norm(t) - 1) < 1e-9
    if n_out == 0:
        assert r["iterations"] == 1


def test_five_point_solver_returns_the_true_essential_matrix():
    """relative_pose::fivept_nister on five exact correspondences: every returned E satisfies the epipolar
    constraints, det E = 0 and 2 E E^T E - tr(E E^T) E = 0, and one of them is [t]x R of the scene."""
    import ctypes as C
    rng = np.random.default_rng(0)
    for trial in range(20):
        R = expmap(rng.normal(0, 0.2, 3))
        T = rng.normal(0, 1, 3)
        P3 = np.stack([rng.uniform(-2, 2, 5), rng.uniform(-2, 2, 5), rng.uniform(3, 8, 5)], 1)
        f1 = np.ascontiguousarray(P3 / np.linalg.norm(P3, axis=1, keepdims=True))
        pc = (R.T @ (P3 - T).T).T
        f2 = np.ascontiguousarray(pc / np.linalg.norm(pc, axis=1, keepdims=True))
        idx = np.arange(5, dtype=np.int32)
        E = np.zeros((10, 9))
        L = O.lib()
        L.kvo_fivept_nister.argtypes = [C.c_void_p] * 4
        n = L.kvo_fivept_nister(f1.ctypes.data, f2.ctypes.data, idx.ctypes.data, E.ctypes.data)
        assert 1 <= n <= 10
        skew = np.array([[0, -T[2], T[1]], [T[2], 0, -T[0]], [-T[1], T[0], 0]])
        Et = skew @ R
        Et /= np.linalg.norm(Et)
        best = 1.0
        for k in range(n):
            Ek = E[k].reshape(3, 3)
            sc = np.linalg.norm(Ek)
            assert max(abs(f1[i] @ Ek @ f2[i]) for i in range(5)) < 1e-10 * sc
            assert abs(np.linalg.det(Ek)) < 1e-6 * sc ** 3
            assert np.abs(2 * Ek @ Ek.T @ Ek - np.trace(Ek @ Ek.T) * Ek).max() < 1e-6 * sc ** 3
            En = Ek / sc
            best = min(best, np.abs(En - Et).max(), np.abs(En + Et).max())
        assert best < 1e-7, (trial, best)


def test_sampler_stream_matches_std_mt19937():
    """SampleConsensusProblem::rnd(): std::mt19937(12345) through uniform_int_distribution<int>(0, INT_MAX);
    numpy's legacy seeding is the same init_genrand, so its raw stream is the reference stream."""
    rs = np.random.RandomState(12345)
    st = rs.get_state()
    bg = np.random.MT19937()
    bg.state = {"bit_generator": "MT19937", "state": {"key": st[1], "pos": st[2]}}
    raw = bg.random_raw(4000)
    assert np.array_equal(O.mt19937_draws(abi.RNG_LIBSTDCXX_11, 4000), (raw >> 1).astype(np.int32))
    acc = raw[raw < 2 ** 31].astype(np.int32)
    assert np.array_equal(O.mt19937_draws(abi.RNG_LIBSTDCXX_PRE11, 1500), acc[:1500])


# ---------------------------------------------------------------------------------------------
# Tracker helper functions on their own (tests/testTracker.cpp:1320-1533)
# ---------------------------------------------------------------------------------------------
def _landmark_frames(rng, with_status):
    """the synthetic landmark lists of FindMatchingKeypoints / FindMatchingStereoKeypoints
    (testTracker.cpp:1327-1366, :1386-1451): 100 common ids 3i, 90 only in ref (3i+1), 80 only in cur
    (3i+2), 70 invalid (-1) in both, shuffled."""
    ref = [3 * i for i in range(100)] + [3 * i + 1 for i in range(90)] + [-1] * 70
    cur = [3 * i for i in range(100)] + [3 * i + 2 for i in range(80)] + [-1] * 70
    ref, cur = np.array(ref, np.int64), np.array(cur, np.int64)
    rng.shuffle(ref)
    rng.shuffle(cur)
    if not with_status:
        return ref, cur, None, None
    VALID, NO_RIGHT_RECT = 0, 2   # KeypointStatus (vio_types.h:38-44)
    rs = np.where(ref % 6 == 0, VALID, NO_RIGHT_RECT).astype(np.uint8)
    cs = np.where(cur % 6 == 0, VALID, NO_RIGHT_RECT).astype(np.uint8)
    return ref, cur, rs, cs


def test_find_matching_keypoints():
    """TestTracker.FindMatchingKeypoints (:1320-1383): exactly the 100 common landmarks, each once,
    matched index to index."""
    ref, cur, _, _ = _landmark_frames(np.random.default_rng(7), False)
    m = O.find_matching_keypoints(ref, cur)
    assert len(m) == 100
    assert np.array_equal(ref[m[:, 0]], cur[m[:, 1]])
    assert len(set(ref[m[:, 0]].tolist())) == 100 and (ref[m[:, 0]] != -1).all()
    # order: the current frame's keypoint order (Tracker.cpp:934-945 walks cur.landmarks_)
    assert np.all(np.diff(m[:, 1]) > 0)


def test_find_matching_stereo_keypoints():
    """TestTracker.FindMatchingStereoKeypoints (:1386-1474): of the 100 common landmarks only those with
    a VALID right keypoint in both frames (ids divisible by 6) survive: (100 + 1) / 2 = 50."""
    ref, cur, rs, cs = _landmark_frames(np.random.default_rng(8), True)
    m = O.find_matching_keypoints(ref, cur, rs, cs)
    assert len(m) == (100 + 1) // 2
    assert np.array_equal(ref[m[:, 0]], cur[m[:, 1]])
    assert (ref[m[:, 0]] % 6 == 0).all()
    assert len(set(ref[m[:, 0]].tolist())) == len(m)


def test_mahalanobis_distance_three_ways():
    """TestTracker.MahalanobisDistance (:1477-1533): the hand-expanded float32 formula the voting loop uses
    (Tracker.cpp:499-523) against an LLT solve and against v' O^-1 v, 1000 random cases, tolerance 1e-2."""
    rng = np.random.default_rng(9)
    for _ in range(1000):
        mm = rng.uniform(-1, 1, (3, 5))
        Om = (mm @ mm.T).astype(np.float32)
        v = rng.uniform(-1, 1, 3).astype(np.float32)
        d3 = O.mahalanobis_f(v, Om, np.zeros(3, np.float32), np.zeros(9, np.float32))
        d1 = float(v.astype(np.float64) @ np.linalg.solve(Om.astype(np.float64), v.astype(np.float64)))
        d2 = float(v @ np.linalg.inv(Om) @ v)
        scale = max(1.0, abs(d1))  # the reference's absolute 1e-2 on O(1) values; relative for ill-conditioned draws
        assert abs(d3 - d1) < 1e-2 * scale, (d3, d1)
        assert abs(d3 - d2) < 1e-2 * scale, (d3, d2)


def test_compute_median_disparity():
    """Tracker::computeMedianDisparity (Tracker.cpp:991-1018; exercised by testDisparityCheck,
    testStereoVisionImuFrontend.cpp:674-926): pixel distance of the matched keypoints, std::nth_element
    median at index n / 2, sqrt of it; false and 0 for an empty match set."""
    rng = np.random.default_rng(10)
    ref = rng.uniform(0, 700, (50, 2)).astype(np.float32)
    shift = rng.uniform(-30, 30, (50, 2)).astype(np.float32)
    cur = ref + shift
    pairs = np.stack([np.arange(50), np.arange(50)], 1).astype(np.int32)[rng.permutation(50)[:31]]
    ok, med = O.compute_median_disparity(ref, cur, pairs)
    d = cur[pairs[:, 1]] - ref[pairs[:, 0]]
    d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float64)   # float products, widened
    assert ok and med == np.sqrt(np.sort(d2)[len(d2) // 2])
    ok, med = O.compute_median_disparity(ref, cur, np.zeros((0, 2), np.int32))
    assert not ok and med == 0.0
