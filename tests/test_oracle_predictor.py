"""Pins the oracle's optical-flow predictor (OpticalFlowPredictor.cpp:27-126) on the reference's own
tests/testOpticalFlowPredictor.cpp: the fixture's synthetic scene (unit cube seen by a distortion-free
camera with the Euroc focal length, :46-75), every assertion of its TEST_Fs with their 0.1 px
tolerance, and the golden pixel values the reference pasted from its own output (:590-612) --
the one place where the reference's tests hold numbers produced by the reference itself.
CPU only; the GPU predictor is compared with this oracle bit for bit in test_gpu_parity.py."""
import numpy as np
import pytest

import oracle_lib as O
from kimera_vio_amd import _abi as abi

FX = 458.654          # tests/data/EurocParams/LeftCameraParams.yaml: intrinsics[0]
W, H = 752, 480
TOL = 1e-1            # compareKeypoints(..., 1e-1) in every test of the reference


def sim_cam():
    c = abi.CameraParams()
    c.width, c.height = W, H
    for i, v in enumerate((FX, FX, W // 2, H // 2)):   # gtsam::Cal3DS2(fx, fx, 0, w/2, h/2, 0, 0)
        c.intrinsics[i] = float(v)
    c.distortion_model, c.n_distortion = abi.DIST_RADTAN, 4
    for i in range(16):
        c.body_pose_cam[i] = float(i % 5 == 0)
    return c


K = np.array([[FX, 0, W // 2], [0, FX, H // 2], [0, 0, 1.0]])
# buildSceneLandmarks(&lmks_, 1.0): bottom then top face of the unit cube (:88-103)
LMKS = np.array([[0, 0, 0], [0, 1, 0], [1, 0, 0], [1, 1, 0], [0, 0, 1], [0, 1, 1], [1, 0, 1], [1, 1, 1]], float)
CAM1_T = np.array([0.0, 0.0, -4.0])      # cam_1_pose_ = Pose3(Rot3(), (0, 0, -4))


def quat_R(w, x, y, z):
    """gtsam::Rot3(w, x, y, z) = Eigen::Quaterniond(w, x, y, z).toRotationMatrix(), not normalised."""
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def project(R_wc, t_wc):
    """PinholeCamera(pose, calib).projectSafe for every landmark (all in front in these scenes)."""
    pc = (LMKS - t_wc) @ R_wc          # R^T (X - t)
    return (pc[:, :2] / pc[:, 2:3]) * FX + np.array([W // 2, H // 2])


CAM1_KPTS = project(np.eye(3), CAM1_T).astype(np.float32)


def predict(R, ptype=abi.FLOW_ROTATIONAL, kpts=CAM1_KPTS):
    return O.predict_sparse_flow(ptype, sim_cam(), kpts, R)


def close(a, b):
    return np.max(np.abs(np.asarray(a, float) - np.asarray(b, float))) < TOL


def test_no_prediction_returns_the_keypoints():
    """DefaultNoPredictionOpticalFlowPrediction / ...Invariance (:318-368)."""
    R = quat_R(0.924, 0, 0, 0.383)
    out = predict(R, abi.FLOW_NO_PREDICTION)
    assert np.array_equal(out, CAM1_KPTS)
    assert np.array_equal(predict(R, abi.FLOW_NO_PREDICTION, out), CAM1_KPTS)


def test_default_rotational_prediction_is_the_homography():
    """DefaultRotationalOpticalFlowPrediction (:371-404): H = K R^-1 K^-1 in double."""
    R = quat_R(0.924, 0, 0, 0.383)
    Hm = K @ np.linalg.inv(R) @ np.linalg.inv(K)
    ph = (Hm @ np.c_[CAM1_KPTS.astype(float), np.ones(8)].T).T
    assert close(predict(R), ph[:, :2] / ph[:, 2:3])


def test_rotational_prediction_forward_and_back():
    """RotationalOpticalFlowPredictionInvariance (:407-432)."""
    R = quat_R(0.924, 0, 0, 0.383)
    fwd = predict(R)
    assert not close(fwd, CAM1_KPTS)
    assert close(predict(np.linalg.inv(R), kpts=fwd), CAM1_KPTS)


def test_rotation_only_matches_the_projection_into_cam2():
    """RotationalOpticalFlowPredictionRotationOnly (:436-458): 20 degrees about z, no translation."""
    R = quat_R(0.985, 0, 0, 0.174)
    assert close(predict(R), project(R, CAM1_T))


def test_small_rotation_is_treated_as_no_rotation():
    """RotationalOpticalFlowPredictionSmallRotationOnly (:463-497): |1 - |q.w|| < 1e-4 -> identity."""
    assert close(predict(quat_R(0.99999999, 0, 0, 0)), CAM1_KPTS)
    assert close(predict(quat_R(-0.99999999, 0, 0, 0)), CAM1_KPTS)   # double cover of the quaternion


@pytest.mark.parametrize("q", [(0.924, 0.383, 0.0, 0.0),     # ...OutOfImage (:501-527): 45 degrees about x
                               (0.0, 0.0, 1.0, 0.0)])        # ...BehindCam (:531-561): 180 degrees about y
def test_out_of_image_or_behind_camera_keeps_the_old_keypoints(q):
    assert close(predict(quat_R(*q)), CAM1_KPTS)


def test_rotation_and_translation_golden_values():
    """RotationalOpticalFlowPredictionRotationAndTranslation (:563-615): the reference's own output for
    the 20-degree rotation, asserted there with 1e-1.  The oracle lands within 0.009 px of them (the
    residual is how gtsam stores the un-normalised quaternion (0.985, 0, 0, 0.174): normalising it first
    is three times further away, 0.027 px)."""
    golden = np.array([[376.00003051757812, 239.99998474121094], [415.302001953125, 347.7138671875],
                       [483.71389770507812, 200.69801330566406], [523.015869140625, 308.41189575195312],
                       [376.00003051757812, 239.99998474121094], [407.44161987304688, 326.17108154296875],
                       [462.17111206054688, 208.55841064453125], [493.61270141601562, 294.7294921875]])
    out = predict(quat_R(0.985, 0, 0, 0.174))
    assert out.shape == (8, 2)
    err = np.max(np.abs(out.astype(float) - golden))
    assert err < TOL
    assert err < 1e-2, err
