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


if __name__ == "__main__":
    main()
