"""CPU-only tests of the product library's host side: the C-ABI shared library loads, exports
every symbol include/kvfe.h declares, its init-time calibration math (stereoRectify, maps) agrees
with the oracle, and it refuses to run without a GPU (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as O
from kimera_vio_amd import _abi as abi
from kimera_vio_amd import frontend as F
from kimera_vio_amd import lib as L
from kimera_vio_amd import params as P

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = L.load()
    header = open(os.path.join(ROOT, "include", "kvfe.h")).read()
    declared = set(re.findall(r"KVFE_API\s+[\w\s\*]+?\b(kvfe_\w+)\s*\(", header))
    assert declared, "no KVFE_API declarations found"
    assert declared == set(L.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), f"libkvfe.so does not export {name}"
    assert lib.kvfe_version().decode().startswith("libkvfe")


def test_abi_struct_sizes_match_header():
    # sizes computed from the C declaration order (natural alignment); guards _abi.py drift
    assert C.sizeof(abi.CameraParams) == 8 + 32 + 8 + 64 + 128
    assert C.sizeof(abi.FrameInput) == 8 + 72 + 8
    assert C.sizeof(abi.Rectification) == (9 + 9 + 12 + 12 + 16) * 8 + 32 + 8
    assert C.sizeof(abi.DetectorParams) % 8 == 0 and C.sizeof(abi.FrontendParams) % 8 == 0


def test_default_params_match_python_mirror():
    lib = L.load()
    a = abi.FrontendParams()
    lib.kvfe_default_frontend_params(C.byref(a))
    b = P.default_frontend_params()
    assert bytes(a) == bytes(b)


@pytest.mark.parametrize("pair", [("sensorLeft.yaml", "sensorRight.yaml"),
                                  ("ForStereoTracker/camLeft.yaml", "ForStereoTracker/camRight.yaml"),
                                  ("params_euroc/LeftCameraParams.yaml", "params_euroc/RightCameraParams.yaml")])
def test_rectification_matches_oracle(pair):
    """StereoCamera::computeRectificationParameters: the product's host math vs the oracle's
    independent restatement of cv::stereoRectify (both float64, same operation order)."""
    Lc = P.load_camera_params(os.path.join(G, pair[0]))
    Rc = P.load_camera_params(os.path.join(G, pair[1]))
    r = F.compute_rectification(Lc, Rc)
    o = O.Camera(Lc, Rc).rect
    for name in ("R1", "R2", "P1", "P2", "Q"):
        assert np.allclose(np.array(getattr(r, name)), np.array(getattr(o, name)), rtol=0, atol=1e-12), name
    assert list(r.roi1) == list(o.roi1) and list(r.roi2) == list(o.roi2)
    assert abs(r.baseline - o.baseline) < 1e-15


def test_baseline_kat():
    """tests/testStereoMatcher.cpp:148 on the product's own host math."""
    Lc = P.load_camera_params(os.path.join(G, "sensorLeft.yaml"))
    Rc = P.load_camera_params(os.path.join(G, "sensorRight.yaml"))
    assert abs(F.compute_rectification(Lc, Rc).baseline - 0.110078) < 1e-5


def test_maps_match_oracle_bit_exact():
    """UndistorterRectifier::initUndistortRectifyMaps: float32 maps identical to the oracle's."""
    Lc = P.load_camera_params(os.path.join(G, "sensorLeft.yaml"))
    Rc = P.load_camera_params(os.path.join(G, "sensorRight.yaml"))
    cam = O.Camera(Lc, Rc)
    for c, cp in ((0, Lc), (1, Rc)):
        R = cam.rect.R1 if c == 0 else cam.rect.R2
        Pm = cam.rect.P1 if c == 0 else cam.rect.P2
        mx, my = F.compute_undistort_rectify_maps(cp, np.array(R), np.array(Pm))
        ox, oy = cam.maps(c)
        assert np.array_equal(mx, ox) and np.array_equal(my, oy)


def test_create_fails_loudly_without_gpu():
    """No CPU fallback: on a machine without a gfx950 device kvfe_create returns NO_DEVICE."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    Lc = P.load_camera_params(os.path.join(G, "sensorLeft.yaml"))
    Rc = P.load_camera_params(os.path.join(G, "sensorRight.yaml"))
    p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=0)
    with pytest.raises(L.KvfeError) as e:
        F.Context(Lc, Rc, p)
    assert e.value.status == abi.KVFE_ERR_NO_DEVICE


def test_unsupported_configurations_are_rejected():
    Lc = P.load_camera_params(os.path.join(G, "sensorLeft.yaml"))
    Rc = P.load_camera_params(os.path.join(G, "sensorRight.yaml"))
    p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=None)
    assert p.use_ransac == 1 and p.tracker.ransac_use_2point_mono == 1 and p.tracker.ransac_randomize == 0
    for field, value in (("ransac_randomize", 1),):        # time-seeded sampling
        q = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=None)
        setattr(q.tracker, field, value)
        with pytest.raises(L.KvfeError) as e:
            F.Context(Lc, Rc, q)
        assert e.value.status == abi.KVFE_ERR_UNSUPPORTED, field
    # FeatureDetectorType: GFTT and FAST exist; ORB does not, AGAST is LOG(FATAL) upstream (FeatureDetector.cpp:67-70)
    for det in (abi.DET_ORB, abi.DET_AGAST, 7):
        q = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=None)
        q.detector.feature_detector_type = det
        with pytest.raises(L.KvfeError) as e:
            F.Context(Lc, Rc, q)
        assert e.value.status == abi.KVFE_ERR_UNSUPPORTED, det
    # the 5-point problem is implemented for 2d2d_algorithm: 1 (NISTER, every shipped YAML) only
    q = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=None)
    assert q.tracker.pose_2d2d_algorithm == 1
    q.tracker.ransac_use_2point_mono = 0
    q.tracker.pose_2d2d_algorithm = 0   # STEWENIUS
    with pytest.raises(L.KvfeError) as e:
        F.Context(Lc, Rc, q)
    assert e.value.status == abi.KVFE_ERR_UNSUPPORTED


def test_yaml_parsing_euroc():
    p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=0)
    assert p.detector.max_features_per_frame == 300 and p.detector.non_max_suppression_type == 6
    assert p.detector.nr_horizontal_bins == 7 and p.detector.nr_vertical_bins == 5
    assert sum(p.detector.binning_mask) == 35
    assert p.tracker.klt_max_level == 4 and p.tracker.klt_win_size == 24 and abs(p.tracker.klt_eps - 0.1) < 1e-15
    assert p.stereo.templ_cols == 101 and p.stereo.min_point_dist == 0.5
    assert p.min_intra_keyframe_time_ns == 0.2e9 and p.max_intra_keyframe_time_ns == 5e9


def test_yaml_pnp_settings():
    """use_pnp_tracking / pnp_algorithm / min_pnp_inliers / ransac_threshold_pnp (VisionImuFrontendParams.cpp,
    VisionImuTrackerParams.cpp) are carried next to the C struct for kvfe_pnp; no YAML is rejected for them"""
    p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=None)
    assert p.use_pnp_tracking == 0 and p.pnp.pnp_algorithm == abi.PNP_EPNP
    assert p.pnp.min_pnp_inliers == 20 and p.pnp.ransac_threshold_pnp == 1.0 and p.pnp.optimize_2d3d_pose_from_inliers == 0
    # the class defaults (VisionImuTrackerParams.h:55-76), also what a YAML without the keys falls back to and what
    # kvfe_default_frontend_params fills in: one set of numbers everywhere
    d = abi.pnp_params_default()
    assert (d.pnp_algorithm, d.min_pnp_inliers, d.ransac_threshold_pnp) == (3, 10, 1.0)
    import tempfile
    keys = ("pnp_algorithm", "min_pnp_inliers", "ransac_threshold_pnp", "optimize_2d3d_pose_from_inliers")
    src = open(os.path.join(G, "params_euroc", "FrontendParams.yaml")).read().splitlines(True)
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
        f.writelines(l for l in src if not l.startswith(keys))
    q = P.load_frontend_params(f.name, use_ransac=0)
    os.unlink(f.name)
    assert (q.pnp.pnp_algorithm, q.pnp.min_pnp_inliers, q.pnp.ransac_threshold_pnp,
            q.pnp.optimize_2d3d_pose_from_inliers) == (3, 10, 1.0, 0)
    from kimera_vio_amd import lib as L
    dp = abi.FrontendParams()
    L.load().kvfe_default_frontend_params(C.byref(dp))
    assert dp.pnp.min_pnp_inliers == d.min_pnp_inliers and dp.pnp.pnp_algorithm == d.pnp_algorithm


def test_cpp_adapter_program_builds_with_plain_gxx():
    """tests/cpp/adapter_sequence.cpp (the reference-shaped C++ host program over include/kvfe_adapter.hpp)
    compiles with g++ alone and links libkvfe.so; without arguments it prints its usage and exits 2.  The GPU
    suite runs it against the oracle."""
    import subprocess
    cpp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp")
    r = subprocess.run(["make", "-C", cpp], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([os.path.join(cpp, "adapter_sequence")], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


def test_device_std_sort_restatement_equals_libstdcxx():
    """kimera_vio_amd/csrc/kvfe_stdsort.inl -- the introsort the select kernel runs for BrownANMS -- compiled for
    the host and compared with the real std::sort (comparator of anms/anms.h sort_pred) on tie-heavy inputs up
    to 8192 elements, heap-sort fallback included: same permutation."""
    import subprocess
    cpp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp")
    r = subprocess.run(["make", "-C", cpp, "stdsort_check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([os.path.join(cpp, "stdsort_check"), "1200"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr


def test_cpp_adapter_input_side_classes():
    """tests/cpp/input_side.cpp: the C++ adapter's utils::ThreadsafeImuBuffer / StereoDataProviderModule /
    ReadAndConvertToGrayScale (include/kvfe_adapter.hpp) through reference cases (testStereoProvider.cpp:526-560
    dropRightFrame, testThreadsafeImuBuffer.cpp:194-279) and a committed frame; host only, no GPU"""
    import subprocess
    import numpy as np
    from PIL import Image
    cpp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp")
    r = subprocess.run(["make", "-C", cpp, "input_side"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    png = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "left_img_0.png")
    r = subprocess.run([os.path.join(cpp, "input_side"), png], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    a = np.asarray(Image.open(png)).reshape(-1).astype(np.uint64)
    checksum = int((a * (np.arange(a.size, dtype=np.uint64) % 251 + 1)).sum())
    assert r.stdout.splitlines() == [
        "first: none action=5",                                            # KVFE_SYNC_DROP_FIRST_FRAME
        "no right: none action=8",                                         # KVFE_SYNC_DROP_NO_RIGHT
        "valid: packet action=0 t=8 tags=2,2 imu=1,2,3,4,6,7,8",
        "empty: none action=1",
        "borders(21,29): result=0 cols=3 stamps=21,25,29 values=21,25,29",
        "borders(40,51): result=1 cols=0",                                 # kDataNotYetAvailable
        "between(21,24): result=4 cols=0",                                 # kTooFewMeasurementsAvailable
        "frame: valid=200 valid_kps=200 lmk_of_px7=5 idx=7 missing=-1",     # testFrame.cpp:82-99, :186-200
        f"png: 752x480 checksum={checksum}",
    ]


def test_input_side_under_sanitizers():
    """tests/cpp/input_fuzz.cpp: host_input.cpp and host_jpeg.cpp compiled with -fsanitize=address,undefined and fed
    with damaged JPEG files (4:2:0, 4:2:2 with restart markers, 4:4:4 with optimised tables; byte damage, truncation,
    forged markers), damaged PNG files (CRCs repaired so that the damage reaches inflate / unfilter / expansion), 3000 well-formed containers
    around random scan lines of every colour type / depth / interlacing, mutated index files and random
    synchroniser traffic with too-small output arrays: no report, no write past the destination, nothing accepted
    that kvfe_png_info refuses"""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    cpp = os.path.join(here, "cpp")
    r = subprocess.run(["make", "-C", cpp, "input_fuzz"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    import io
    import tempfile
    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(2)
    jpegs = []
    with tempfile.TemporaryDirectory() as td:
        for i, (ss, kw) in enumerate([(2, {}), (1, {"restart_marker_blocks": 2}), (0, {"optimize": True})]):
            a = np.clip(np.kron(rng.integers(0, 256, (12, 16, 3)), np.ones((4, 4, 1))) + rng.normal(0, 5, (48, 64, 3)),
                        0, 255).astype(np.uint8)
            path = os.path.join(td, f"f{i}.jpg")
            Image.fromarray(a[:45, :61]).save(path, "JPEG", quality=85, subsampling=ss, **kw)
            jpegs.append(path)
        r = subprocess.run([os.path.join(cpp, "input_fuzz"), os.path.join(here, "golden", "left_img_0.png"),
                            os.path.join(here, "golden", "chessboard.png")] + jpegs, capture_output=True, text=True,
                           env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1].startswith("OK"), r.stdout + r.stderr


def test_yaml_loader_reference_param_tests():
    """kimera_vio_amd/params.py on the fixtures and with the expectations of the reference's own parser tests:
    tests/testFeatureDetectorParams.cpp:20-74 and tests/testVisionImuFrontendParams.cpp:32-70
    (ForTracker/trackerParameters.yaml, ForFeatureDetector/frontendParams-NMS-Binning2.yaml)"""
    from kimera_vio_amd import params as P
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    p = P.load_frontend_params(os.path.join(g, "ForTracker", "trackerParameters.yaml"))
    d, t, s = p.detector, p.tracker, p.stereo
    assert d.feature_detector_type == 0 and d.max_features_per_frame == 200
    assert d.enable_subpixel_corner_refinement == 1
    assert (d.subpix_max_iters, d.subpix_epsilon, d.subpix_window_size, d.subpix_zero_zone) == (42, 0.201, 12, 2)
    assert d.enable_non_max_suppression == 1 and d.non_max_suppression_type == 4
    assert (d.nr_horizontal_bins, d.nr_vertical_bins) == (5, 2)
    assert list(d.binning_mask[:10]) == [1] * 10                      # "binning_mask: []" = all ones (2 x 5)
    assert (d.quality_level, d.block_size, d.use_harris_detector, d.k, d.fast_thresh) == (0.5, 3, 0, 0.04, 52)
    assert (t.klt_win_size, t.klt_max_iter, t.klt_max_level, t.klt_eps, t.max_feature_track_age) == \
        (24, 30, 2, 0.001, 10)
    assert (t.min_nr_mono_inliers, t.min_nr_stereo_inliers) == (2000, 1000)
    assert (t.ransac_threshold_mono, t.ransac_threshold_stereo) == (1e-06, 0.3)
    assert (t.ransac_use_1point_stereo, t.ransac_use_2point_mono, t.ransac_max_iterations, t.ransac_probability,
            t.ransac_randomize, t.disparity_threshold) == (0, 1, 100, 0.995, 0, 1)
    assert (s.equalize_image, s.tolerance_template_matching, s.templ_cols, s.templ_rows, s.stripe_extra_rows) == \
        (1, 0.17, 103, 5, 2)
    assert p.min_intra_keyframe_time_ns == 0.5 * 1e9 and p.min_number_features == 100
    assert (p.use_stereo_tracking, p.use_ransac, p.use_pnp_tracking) == (1, 0, 1)
    d2 = P.default_frontend_params().detector                         # FeatureDetectorParams::parseYAML alone
    P.load_detector_params(os.path.join(g, "ForFeatureDetector", "frontendParams-NMS-Binning2.yaml"), d2)
    assert d2.enable_non_max_suppression == 1 and d2.non_max_suppression_type == 6
    assert (d2.nr_horizontal_bins, d2.nr_vertical_bins) == (4, 5)
    expected = [1] * 20                                                # Ones(5, 4) with six zeros, row-major
    for r, c in ((0, 0), (0, 1), (0, 2), (0, 3), (3, 0), (3, 2)):
        expected[r * 4 + c] = 0
    assert list(d2.binning_mask[:20]) == expected
    assert P.default_frontend_params().detector.fast_thresh == 10      # FeatureDetectorParams.h:105


def test_camera_yaml_loader_reference_param_test():
    """load_camera_params on tests/data/sensor.yaml with the expectations of tests/testCameraParams.cpp:34-95
    (resolution, intrinsics, body_Pose_cam within gtsam::assert_equal's 1e-9, the four radtan coefficients)"""
    from kimera_vio_amd import params as P
    c = P.load_camera_params(os.path.join(G, "sensor.yaml"))
    assert (c.width, c.height) == (752, 480)
    assert list(c.intrinsics) == [458.654, 457.296, 367.215, 248.375]
    assert c.distortion_model == abi.DIST_RADTAN and c.n_distortion == 4
    assert list(c.distortion)[:4] == [-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05]
    T = np.array(c.body_pose_cam).reshape(4, 4)
    R_expected = np.array([[0.0148655429818, -0.999880929698, 0.00414029679422],
                           [0.999557249008, 0.0149672133247, 0.025715529948],
                           [-0.0257744366974, 0.00375618835797, 0.999660727178]])
    assert np.abs(T[:3, :3] - R_expected).max() < 1e-9
    assert np.abs(T[:3, 3] - np.array([-0.0216401454975, -0.064676986768, 0.00981073058949])).max() < 1e-9
    assert T[3].tolist() == [0.0, 0.0, 0.0, 1.0]


def test_cpp_adapter_rectified_body_poses_and_relative_pose(tmp_path):
    """include/kvfe_adapter.hpp: StereoCamera::B_Pose_camLrect_ / B_Pose_camRrect_ (StereoCamera.cpp:55-66, the right
    one composed with the LEFT body pose as upstream), the Cal3_S2Stereo of :73-80 and StereoVisionImuFrontend::
    getRelativePoseBodyMono / Stereo (StereoVisionImuFrontend.cpp:699-716) against numpy on the EuRoC calibration"""
    import subprocess
    Lc = P.load_camera_params(os.path.join(G, "sensorLeft.yaml"))
    Rc = P.load_camera_params(os.path.join(G, "sensorRight.yaml"))
    txt = tmp_path / "cams.txt"
    with open(txt, "w") as f:
        for c in (Lc, Rc):
            f.write(f"{c.width} {c.height} " + " ".join(repr(float(x)) for x in list(c.intrinsics)[:4]) + " " +
                    " ".join(repr(float(x)) for x in list(c.distortion)[:4]) + " " +
                    " ".join(repr(float(x)) for x in c.body_pose_cam) + "\n")
    cpp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp")
    assert subprocess.run(["make", "-C", cpp, "input_side"], capture_output=True, text=True).returncode == 0
    r = subprocess.run([os.path.join(cpp, "input_side"), "", "", str(txt)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = {ln.split(":")[0]: np.array([float(x) for x in ln.split(":")[1].split()]) for ln in r.stdout.splitlines()
           if ln.startswith(("B_Pose", "relative_pose_body", "stereo_calib"))}
    rect = F.compute_rectification(Lc, Rc)
    T = np.array(Lc.body_pose_cam).reshape(4, 4)

    def pose(R1):
        M = np.eye(4)
        M[:3, :3] = T[:3, :3] @ np.array(R1).reshape(3, 3).T
        M[:3, 3] = T[:3, 3]
        return M
    for name, R in (("B_Pose_camLrect", rect.R1), ("B_Pose_camRrect", rect.R2)):
        M = pose(R)
        assert np.abs(got[name][:9].reshape(3, 3) - M[:3, :3]).max() < 1e-15 and np.array_equal(got[name][9:], M[:3, 3])
    lkf = np.eye(4)
    lkf[:3] = np.array([0.9998, -0.01, 0.015, 0.05, 0.0101, 0.99995, -0.002, -0.02, -0.01498, 0.00215, 0.99989,
                        0.3]).reshape(3, 4)
    bl = pose(rect.R1)
    binv = np.eye(4)
    binv[:3, :3] = bl[:3, :3].T
    binv[:3, 3] = -bl[:3, :3].T @ bl[:3, 3]
    rel = bl @ lkf @ binv
    assert np.abs(got["relative_pose_body"][:9].reshape(3, 3) - rel[:3, :3]).max() < 1e-14
    assert np.abs(got["relative_pose_body"][9:] - rel[:3, 3]).max() < 1e-14
    P1 = np.array(rect.P1).reshape(3, 4)
    assert got["stereo_calib"].tolist() == [P1[0, 0], P1[1, 1], P1[0, 1], P1[0, 2], P1[1, 2], rect.baseline]
    assert abs(rect.baseline - 0.110078) < 1e-5                      # tests/testStereoMatcher.cpp:148


def test_device_warm_up_is_harmless_without_a_device():
    """kimera_vio_amd/_warmup.py: the throw-away first touch of a GPU box reports failure (it does not raise, and it
    does not hide anything: the caller's own kvfe_create still fails loudly) when there is no device"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    from kimera_vio_amd._warmup import warm_up_device
    assert warm_up_device(attempts=1, timeout_s=120) is False


def test_no_valu_to_dpp_hazard_in_inline_asm():
    """gfx9 needs 2 wait states between a VALU write of a VGPR and a DPP read of it; hipcc provides them
    for its own DPP instructions but cannot see inside inline asm.  tools/check_dpp_hazard.py scans the compiled gfx950
    ISA of the tracking kernels (the only file with hand-written DPP arithmetic on freshly computed carries)."""
    import importlib.util
    import shutil
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_dpp_hazard", os.path.join(root, "tools", "check_dpp_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n, bad = mod.check_hip(os.path.join(root, "kimera_vio_amd", "csrc", "k_track.hip"))
    assert n > 100, "the tracking kernels are expected to hold their DPP chains"
    assert not bad, bad[:5]
    # round 6: the four-points-per-wave kernel adds its chain terms with quad-broadcast DPP additions right behind
    # hand-scheduled blocks (v_dot2 / v_cvt in inline asm, ending in their own s_nop)
    n, bad = mod.check_hip(os.path.join(root, "kimera_vio_amd", "csrc", "k_lk4.hip"))
    assert n > 400
    assert not bad, bad[:5]
    # round 5: the row minimum of the dense two-pass aggregation is four hand-written v_min_u32_dpp with their own s_nop
    n, bad = mod.check_hip(os.path.join(root, "kimera_vio_amd", "csrc", "k_dense.hip"))
    assert n > 100
    assert not bad, bad[:5]


def test_no_valu_write_of_the_data_of_a_wide_store_in_the_next_issue_slot():
    """round 6 (tools/fuzz_batched.py 120 2 79): gfx950 reads the data registers of a VMEM store of more than 64 bits
    after the store has issued; a VALU write of one of them within two wait states changes what lanes 12 - 15 of every row
    of 16 store.  hipcc pads the pair except behind a buffer store with an SGPR offset -- pyr2_kernel's level-0 copy was
    that form, and a wave that ran alone on its SIMD stored the following v_perm_b32's result.  tools/
    check_store_data_hazard.py scans the compiled gfx950 code of every kernel file for the pair."""
    import importlib.util
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_store_data_hazard",
                                                  os.path.join(root, "tools", "check_store_data_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    csrc = os.path.join(root, "kimera_vio_amd", "csrc")
    total = 0
    for f in sorted(os.listdir(csrc)):
        if f.startswith("k_") and f.endswith(".hip"):
            n, bad = mod.check_hip(os.path.join(csrc, f))
            total += n
            assert not bad, (f, bad[:5])
    assert total > 100, "the kernels are expected to hold wide stores (pyr2_kernel's level-0 copy alone has 100)"
    # the checker itself: the pair it looks for, and the padded form
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as t:
        t.write("k:\n\tbuffer_store_dwordx4 v[12:15], v45, s[8:11], s3 offen\n\tv_perm_b32 v12, v8, v8, s7\n"
                "\tbuffer_store_dwordx4 v[8:11], v45, s[8:11], 0 offen\n\ts_nop 1\n\tv_pk_add_u16 v8, v16, v20\n"
                "\tglobal_store_dwordx4 v[0:1], v[4:7], off\n\ts_add_i32 s3, s3, s6\n\tv_mov_b32_e32 v5, 0\n"
                "\tbuffer_store_dwordx4 v[20:23], v45, s[8:11], 0 offen\n\ts_add_i32 s3, s3, s6\n\ts_nop 0\n\tv_mov_b32_e32 v20, 0\n")
    try:
        n, bad = mod.check_asm(t.name)
    finally:
        os.unlink(t.name)
    # (two wait states: the instruction behind the store and the one behind that; an s_nop N counts N + 1)
    assert n == 4 and len(bad) == 2 and "v_perm_b32" in bad[0][2] and "v_mov_b32_e32 v5" in bad[1][2], (n, bad)


def test_hand_issued_row_requests_target_accumulation_registers():
    """round 4: the min-eigenvalue kernel issues its source-row loads from inline asm and waits for them by hand; hipcc
    does not know that such a statement's destination is written late, and the first run-walking version of the kernel
    read copies of registers whose byte had not landed (a phi move at a loop header).  The requests now target AGPRs,
    which hipcc never allocates in this kernel; tools/check_inflight_regs.py checks the compiled ISA for that."""
    import importlib.util
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_inflight_regs", os.path.join(root, "tools", "check_inflight_regs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rep = mod.check_hip()
    assert len(rep) >= 2
    for k, (dests, bad) in rep.items():
        assert dests == ["a0", "a1", "a2", "a3", "a4", "a5"] and not bad, (k, dests, bad[:3])
