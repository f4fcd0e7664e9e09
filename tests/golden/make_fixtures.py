"""Regenerates tests/golden/micro_euroc_f10_18.npz from the reference's vendored test data
(/root/reference/tests/data/MicroEurocDataset, 752x480 EuRoC frames).  Run in the build container
only (the GPU box has no /root/reference); the .npz is committed.

Contents: 9 consecutive stereo pairs (frames 10..18 of the 95-frame clip, the range the reference's
own pipeline test uses starts at frame 10: tests/testStereoImuPipeline.cpp:37-64), their timestamps,
and the gyro-integrated body rotation at each frame time (first-order integration of imu0/data.csv,
no bias), from which the tests derive keyframe_R_cur_frame.
"""
import csv
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/tests/data/MicroEurocDataset/mav0"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro_euroc_f10_18.npz")
FIRST, COUNT = 10, 9


def expm_so3(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def main():
    rows = [r for r in csv.reader(open(os.path.join(SRC, "cam0", "data.csv"))) if not r[0].startswith("#")]
    rows = rows[FIRST:FIRST + COUNT]
    ts = np.array([int(r[0]) for r in rows], np.int64)
    lefts = np.stack([np.array(Image.open(os.path.join(SRC, "cam0", "data", r[1])).convert("L")) for r in rows])
    rights = np.stack([np.array(Image.open(os.path.join(SRC, "cam1", "data", r[1])).convert("L")) for r in rows])
    imu = np.array([[float(v) for v in r[:4]] for r in csv.reader(open(os.path.join(SRC, "imu0", "data.csv")))
                    if not r[0].startswith("#")])
    R = np.eye(3)
    Rs = []
    j = 0
    t_prev = ts[0]
    for t in ts:
        while j < len(imu) and imu[j, 0] < t:
            if imu[j, 0] >= t_prev:
                dt = (min(imu[j + 1, 0], t) - imu[j, 0]) * 1e-9 if j + 1 < len(imu) else 0.0
                R = R @ expm_so3(imu[j, 1:4] * dt)
            j += 1
        t_prev = t
        Rs.append(R.copy())
    np.savez_compressed(OUT, lefts=lefts, rights=rights, timestamps=ts, body_R=np.stack(Rs))
    print("wrote", OUT, os.path.getsize(OUT) / 1e6, "MB")
    # the IMU rows around those frames, verbatim (imu0/data.csv: 60 ms before frame 10 .. 15 ms after frame 18): with the
    # frames above they make a complete EuRoC-layout clip for the end-to-end replay test (tests/test_gpu_replay_r3.py)
    lo, hi = ts[0] - 60_000_000, ts[-1] + 15_000_000
    rows_imu = [ln for ln in open(os.path.join(SRC, "imu0", "data.csv"))
                if ln.startswith("#") or lo <= int(ln.split(",")[0]) <= hi]
    with open(os.path.join(os.path.dirname(OUT), "micro_euroc_imu_f10_18.csv"), "w") as f:
        f.writelines(rows_imu)


def bgr2gray_opencv(rgb: np.ndarray) -> np.ndarray:
    """cv::cvtColor(COLOR_BGR2GRAY) on 8-bit data: fixed point, 15 fractional bits in OpenCV 4 (RY15 9798, GY15 19235,
    BY15 3735; OpenCV 3 used 14 bits: 4899 / 9617 / 1868), rounded -- what UtilsOpenCV::ReadAndConvertToGrayScale
    (src/utils/UtilsOpenCV.cpp:390-403) applies to a 3-channel file and what kvfe_png_decode_gray does.  The JPEG
    frames converted below carry R = G = B, for which every such weighting returns the value itself.
    (PIL's convert("L") uses other constants.)"""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return ((r * 9798 + g * 19235 + b * 3735 + (1 << 14)) >> 15).astype(np.uint8)


REF = "/root/reference/tests/data"
HERE = os.path.dirname(os.path.abspath(__file__))


def fisheye_golden():
    """tests/testUndistortRectifier.cpp:270-347 (DISABLED_undistortFisheyeStereoFrame): the reference's own
    real-OpenCV rectified fisheye pair.  The PNG is RGB with R = G = B; stored as one grey channel."""
    import shutil
    shutil.copy(os.path.join(REF, "ForStereoFrame", "right_fisheye_img_0.png"), HERE)
    sb = np.array(Image.open(os.path.join(REF, "ForStereoFrame", "sidebyside_ref_img_0.png")))
    assert sb.ndim == 3 and np.array_equal(sb[..., 0], sb[..., 1]) and np.array_equal(sb[..., 0], sb[..., 2])
    Image.fromarray(sb[..., 0]).save(os.path.join(HERE, "sidebyside_ref_img_0_gray.png"), optimize=True)


def stereo_tracker_frames():
    """tests/testStereoVisionImuFrontend.cpp:674-926 (testDisparityCheck): the three JPEG stereo pairs it replays,
    decoded here (PIL / libjpeg-turbo, the decoder family OpenCV's imread uses) and converted to grey with OpenCV's
    BGR2GRAY arithmetic; the camera YAMLs of that test."""
    import shutil
    d = os.path.join(REF, "ForStereoTracker")
    out = {}
    for side in ("left", "right"):
        out[side] = np.stack([bgr2gray_opencv(np.array(Image.open(os.path.join(d, f"{side}_frame{n:04d}.jpg")).convert("RGB")))
                              for n in (0, 1, 8)])
    np.savez_compressed(os.path.join(HERE, "ForStereoTracker", "frames_0_1_8.npz"), lefts=out["left"],
                        rights=out["right"], frame_numbers=np.array([0, 1, 8]))
    for f in ("camLeftEuroc.yaml", "camRightEuroc.yaml"):
        shutil.copy(os.path.join(d, f), os.path.join(HERE, "ForStereoTracker"))


def rgbd_frames():
    """tests/testDepthFrame.cpp / tests/testRgbdFrame.cpp data: two colour images (grey via BGR2GRAY) and their
    float32 depth images (metres), the camera YAML with the depth block."""
    import shutil
    d = os.path.join(REF, "ForRgbd")
    os.makedirs(os.path.join(HERE, "ForRgbd"), exist_ok=True)
    lefts = np.stack([bgr2gray_opencv(np.array(Image.open(os.path.join(d, f"left_img_{i}.png")).convert("RGB")))
                      for i in (0, 1)])
    depths = np.stack([np.array(Image.open(os.path.join(d, f"depth_img_{i}.tiff"))) for i in (0, 1)])
    assert depths.dtype == np.float32
    np.savez_compressed(os.path.join(HERE, "ForRgbd", "rgbd_0_1.npz"), lefts=lefts, depths=depths)
    shutil.copy(os.path.join(d, "sensorLeft.yaml"), os.path.join(HERE, "ForRgbd"))


def parser_fixtures():
    """tests/testFeatureDetectorParams.cpp / testVisionImuFrontendParams.cpp: the YAML file both parse"""
    import shutil
    os.makedirs(os.path.join(HERE, "ForTracker"), exist_ok=True)
    shutil.copy(os.path.join(REF, "ForTracker", "trackerParameters.yaml"), os.path.join(HERE, "ForTracker"))


if __name__ == "__main__":
    main()
    fisheye_golden()
    stereo_tracker_frames()
    rgbd_frames()
    parser_fixtures()
