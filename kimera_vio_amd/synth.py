"""Seeded synthetic stereo streams for the benchmark and the size-independent parity tests
(SURVEY.md §8d): band-limited noise texture + random rectangles (so that Shi-Tomasi finds plenty of
corners), a right view displaced by a smooth disparity field, and frame-to-frame motion by a small
rotation homography whose rotation is also what the optical-flow predictor receives.

Pure numpy, deterministic for a given (seed, stream).  This is input generation only; it is not part
of the measured path.
"""
from __future__ import annotations

import numpy as np

from . import _abi as abi


def _box_blur(a: np.ndarray, r: int) -> np.ndarray:
    """separable box blur (three passes ~ Gaussian), float32"""
    if r <= 0:
        return a
    k = 2 * r + 1
    for axis in (0, 1):
        c = np.cumsum(np.pad(a, [(r + 1, r) if ax == axis else (0, 0) for ax in (0, 1)], mode="reflect"),
                      axis=axis, dtype=np.float64)
        if axis == 0:
            a = ((c[k:] - c[:-k]) / k).astype(np.float32)
        else:
            a = ((c[:, k:] - c[:, :-k]) / k).astype(np.float32)
    return a


def base_texture(w: int, h: int, seed: int, n_rect: int = 200) -> np.ndarray:
    """float32 image in [0,255] on a canvas larger than (h, w) by a 96 px margin on every side."""
    m = 96
    H, W = h + 2 * m, w + 2 * m
    rng = np.random.RandomState(1000 + seed)
    a = rng.uniform(0, 255, size=(H, W)).astype(np.float32)
    for _ in range(3):
        a = _box_blur(a, 2)
    a = (a - a.mean()) * (60.0 / max(a.std(), 1e-6)) + 128.0
    rr = np.random.RandomState(2000 + seed)
    for _ in range(n_rect + 150):
        x0, y0 = rr.randint(0, W - 8), rr.randint(0, H - 8)
        rw, rh = rr.randint(8, 60), rr.randint(8, 60)
        a[y0:y0 + rh, x0:x0 + rw] = rr.uniform(20, 235)
    return np.clip(a, 0, 255)


def _sample_bilinear(img: np.ndarray, x: np.ndarray, y: np.ndarray) -> np.ndarray:
    H, W = img.shape
    x = np.clip(x, 0, W - 1.001)
    y = np.clip(y, 0, H - 1.001)
    x0 = np.floor(x).astype(np.int32)
    y0 = np.floor(y).astype(np.int32)
    fx = (x - x0).astype(np.float32)
    fy = (y - y0).astype(np.float32)
    p00 = img[y0, x0]
    p01 = img[y0, x0 + 1]
    p10 = img[y0 + 1, x0]
    p11 = img[y0 + 1, x0 + 1]
    return (p00 * (1 - fx) + p01 * fx) * (1 - fy) + (p10 * (1 - fx) + p11 * fx) * fy


def rot_from_axis_angle(axis, angle):
    axis = np.asarray(axis, np.float64)
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K


class SyntheticStream:
    """One stereo stream.  frame(t) -> (left u8, right u8); rotation(t) = camera rotation at frame t
    relative to frame 0 (world_R_cam)."""

    def __init__(self, cam: abi.CameraParams, seed: int, max_step_deg: float = 0.3):
        self.w, self.h = cam.width, cam.height
        self.K = np.array([[cam.intrinsics[0], 0, cam.intrinsics[2]],
                           [0, cam.intrinsics[1], cam.intrinsics[3]], [0, 0, 1.0]])
        self.tex = base_texture(self.w, self.h, seed)
        rng = np.random.RandomState(3000 + seed)
        self.axis = rng.normal(size=3)
        self.axis[2] *= 0.3
        self.step = np.deg2rad(rng.uniform(0.3, 1.0) * max_step_deg)
        # smooth disparity field in [6, 60] px
        d = rng.uniform(0, 1, size=(self.h // 16 + 2, self.w // 16 + 2)).astype(np.float32)
        d = np.kron(d, np.ones((16, 16), np.float32))[: self.h, : self.w]
        d = _box_blur(_box_blur(d, 8), 8)
        d = (d - d.min()) / max(d.max() - d.min(), 1e-6)
        self.disp = 6.0 + 54.0 * d
        yy, xx = np.mgrid[0:self.h, 0:self.w]
        self.xx = xx.astype(np.float32)
        self.yy = yy.astype(np.float32)

    def rotation(self, t: int) -> np.ndarray:
        return rot_from_axis_angle(self.axis, self.step * t)

    def frame(self, t: int):
        m = 96
        # pixel p in frame t sees texture point H p with H = K R(t) K^-1 (pure rotation)
        Hm = self.K @ self.rotation(t) @ np.linalg.inv(self.K)
        X = Hm[0, 0] * self.xx + Hm[0, 1] * self.yy + Hm[0, 2]
        Y = Hm[1, 0] * self.xx + Hm[1, 1] * self.yy + Hm[1, 2]
        Z = Hm[2, 0] * self.xx + Hm[2, 1] * self.yy + Hm[2, 2]
        u, v = X / Z + m, Y / Z + m
        left = _sample_bilinear(self.tex, u, v)
        right = _sample_bilinear(self.tex, u + self.disp * (X / Z * 0 + 1.0), v)
        return (np.clip(np.rint(left), 0, 255).astype(np.uint8),
                np.clip(np.rint(right), 0, 255).astype(np.uint8))


def keyframe_R_cur(stream: SyntheticStream, t_kf: int, t_cur: int) -> np.ndarray:
    """keyframe_R_cur_frame for the predictor: rotation taking current-frame vectors to the keyframe
    (consistent with frame(): p_cur ~ K R_cur^T R_kf K^-1 p_kf)."""
    return stream.rotation(t_kf).T @ stream.rotation(t_cur)
