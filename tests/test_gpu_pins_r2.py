"""GPU side of the round-2 pins (tests/test_oracle_pins_r2.py): the HIP path on the same reference-held vectors,
compared with the oracle field by field AND with the reference's own expected values."""
import os

import numpy as np
import pytest
from PIL import Image

import oracle_lib as O
import test_oracle_pins_r2 as PIN
from kimera_vio_amd import _abi as abi
from kimera_vio_amd import frontend as F
from kimera_vio_amd import params as P
from parity_util import assert_step_equal

pytestmark = pytest.mark.gpu
G = PIN.G


def test_disparity_check_keyframe_pattern_on_gpu():
    """tests/testStereoVisionImuFrontend.cpp:674-926 through libkvfe: keyframes T,T,T,T,T,F,T, every step identical
    to the oracle (LK -> shouldBeKeyframe -> 2-point RANSAC -> LOW_DISPARITY bookkeeping -> detection -> stereo)."""
    z = np.load(os.path.join(G, "ForStereoTracker", "frames_0_1_8.npz"))
    idx = {0: 0, 1: 1, 8: 2}
    L, R = PIN._euroc_tracker_cams()
    p = PIN._disparity_check_params()
    fe = O.Frontend(L, R, p)
    c = F.Context(L, R, p, batch=1)
    frames = [0, 0, 1, 8, 8, 8, 8]
    stamps = [0, 3.1e6, 6.1e6, 6.11e6, 9.11e6, 12.11e6, 52.11e9]
    kfs = []
    try:
        for i, (f, t) in enumerate(zip(frames, stamps)):
            Rk = np.eye(3) if i == 0 else PIN._small_rotation(i)
            l, r = z["lefts"][idx[f]], z["rights"][idx[f]]
            c.step_host(l[None], r[None], c.make_inputs([int(t)], [Rk], [0]))
            got = c.get_output(0)
            assert_step_equal(got, fe.process(l, r, int(t), Rk, False), ("disparity_check", i))
            kfs.append(bool(got["is_keyframe"]))
    finally:
        c.close()
    assert kfs == [True, True, True, True, True, False, True]


def test_lost_feature_track_on_gpu():
    """tests/testStereoVisionImuFrontend.cpp:928+ testLostFeatureTrack through libkvfe"""
    L, R = PIN._euroc_tracker_cams()
    H, W = L.height, L.width
    sl, sr = np.zeros((H, W), np.uint8), np.zeros((H, W), np.uint8)
    sl[2:4, 4:6] = 255
    sr[2:4, 2:4] = 255
    bl, br = np.zeros((H, W), np.uint8), np.zeros((H, W), np.uint8)
    bl[300:320, 300:310] = 255
    br[300:320, 290:300] = 255
    p = PIN._disparity_check_params()
    fe = O.Frontend(L, R, p)
    c = F.Context(L, R, p, batch=1)
    try:
        for i, (l, r, t) in enumerate(((sl, sr, 0), (bl, br, int(3.1e6)))):
            c.step_host(l[None], r[None], c.make_inputs([t], [np.eye(3)], [0]))
            got = c.get_output(0)
            assert_step_equal(got, fe.process(l, r, t, np.eye(3), False), ("lost_track", i))
        assert not got["is_keyframe"] and got["n_tracked"] == 0 and got["n_detected"] > 0
    finally:
        c.close()


def test_fisheye_rectified_pair_matches_reference_opencv_image_on_gpu():
    """tests/testUndistortRectifier.cpp:270-347: the GPU remap of the fisheye pair against the reference's
    real-OpenCV image (same criterion as the oracle pin) and bit-exact against the oracle."""
    L = P.load_camera_params(os.path.join(G, "left_sensor_fisheye.yaml"))
    R = P.load_camera_params(os.path.join(G, "right_sensor_fisheye.yaml"))
    p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=0)
    cam = O.Camera(L, R)
    imgs = [PIN.gray("left_fisheye_img_0.png"), PIN.gray("right_fisheye_img_0.png")]
    golden = np.array(Image.open(os.path.join(G, "sidebyside_ref_img_0_gray.png")))
    H, W = imgs[0].shape
    c = F.Context(L, R, p, batch=1)
    try:
        for k in (0, 1):
            mine = c.undistort_rectify_image(k, imgs[k])
            assert np.array_equal(mine, cam.rectify_image(k, imgs[k]))
            mx, my = cam.maps(k)
            sx = np.rint(mx.astype(np.float64) * 32).astype(np.int64) >> 5
            sy = np.rint(my.astype(np.float64) * 32).astype(np.int64) >> 5
            strict = (sx >= 0) & (sx + 1 <= W - 1) & (sy >= 0) & (sy + 1 <= H - 1)
            d = np.abs(mine.astype(int) - golden[:, k * W:(k + 1) * W].astype(int))
            assert ((d > 0) & strict).sum() <= 40 and d[strict].max() <= 6
    finally:
        c.close()


@pytest.mark.parametrize("use_ransac", [0, 1])
def test_rgbd_frontend_on_reference_rgbd_data(use_ransac):
    """RgbdVisionImuFrontend on the reference's own RGBD test data (tests/data/ForRgbd: two colour + float depth
    frames, camera with the depth block of sensorLeft.yaml, is_registered forced to 1 as FillStereoFrame's test
    does): GPU == oracle on both frames."""
    z = np.load(os.path.join(G, "ForRgbd", "rgbd_0_1.npz"))
    cam = P.load_camera_params(os.path.join(G, "ForRgbd", "sensorLeft.yaml"))
    dp = abi.depth_params_default(abi.DEPTH_F32)
    dp.virtual_baseline, dp.min_depth, dp.max_depth, dp.depth_to_meters, dp.is_registered = 0.3, 0.0, 10.0, 1.0, 1
    p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=use_ransac)
    p.detector.max_features_per_frame = 200
    fe = O.Frontend(cam, cam, p, depth=dp)
    c = F.Context(cam, cam, p, batch=1, frontend_type=abi.FRONTEND_RGBD, depth=dp)
    try:
        for i in range(4):
            j = i % 2
            ts = i * 100_000_000
            c.step_host(z["lefts"][j][None], z["depths"][j][None], c.make_inputs([ts], [np.eye(3)], [1]))
            exp = fe.process(z["lefts"][j], z["depths"][j], ts, np.eye(3), True)
            got = c.get_output(0)
            assert_step_equal(got, exp, ("rgbd", i))
        assert got["n_keypoints"] > 50 and (got["right_status"] == abi.KP_VALID).sum() > 20
    finally:
        c.close()
