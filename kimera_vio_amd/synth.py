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


# ---------------------------------------------------------------------------------------------
# physically consistent rig: both cameras of the calibrated pair look at one textured plane
# ---------------------------------------------------------------------------------------------
def _undistort_normalized(xd, yd, k1, k2, p1, p2, iters=25):
    """inverse radial-tangential model on normalized coordinates (fixed-point iteration)"""
    x, y = xd.copy(), yd.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        icdist = 1.0 / (1.0 + (k2 * r2 + k1) * r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x = (xd - dx) * icdist
        y = (yd - dy) * icdist
    return x, y


def _pixel_rays(cam: abi.CameraParams) -> np.ndarray:
    """unit ray of every pixel in the camera frame, [H, W, 3] float64 (distortion inverted)"""
    fx, fy, cx, cy = (cam.intrinsics[i] for i in range(4))
    yy, xx = np.mgrid[0:cam.height, 0:cam.width].astype(np.float64)
    xd, yd = (xx - cx) / fx, (yy - cy) / fy
    if cam.distortion_model == abi.DIST_RADTAN:
        k1, k2, p1, p2 = (cam.distortion[i] for i in range(4))
        xd, yd = _undistort_normalized(xd, yd, k1, k2, p1, p2)
    r = np.stack([xd, yd, np.ones_like(xd)], -1)
    return r / np.linalg.norm(r, axis=-1, keepdims=True)


class RigStream:
    """One stereo stream rendered through the calibrated camera pair (intrinsics, radial-tangential
    distortion and extrinsics of both kvfe_camera_params): a textured plane at 1.5-3 m, slightly
    tilted, seen from a rig that translates a few centimetres and rotates a fraction of a degree per
    frame.  Bearing vectors of tracked points therefore satisfy the epipolar geometry of
    (rotation(t), position(t)) and stereo disparities follow from the plane depth, so the geometric
    outlier rejection of keyframes sees a consistent scene (SURVEY.md §8d asks for the forward
    distortion model; a shift-only right view would not rectify consistently).

    frame(t) -> (left u8, right u8).  keyframe_R_cur(stream, t_kf, t_cur) applies as for
    SyntheticStream; `rect_R1` (3x3, from kvfe_compute_rectification) expresses it in the rectified
    left frame, which is the frame of the front-end's bearing vectors."""

    def __init__(self, left: abi.CameraParams, right: abi.CameraParams, seed: int, rect_R1=None,
                 max_step_deg: float = 0.3, max_step_m: float = 0.04):
        self.w, self.h = left.width, left.height
        self.tex = base_texture(self.w, self.h, seed)
        self.rays = [_pixel_rays(left), _pixel_rays(right)]
        TL = np.array(left.body_pose_cam, np.float64).reshape(4, 4)
        TR = np.array(right.body_pose_cam, np.float64).reshape(4, 4)
        T_lr = np.linalg.inv(TL) @ TR            # pose of the right camera in the left camera frame
        self.R_lr, self.t_lr = T_lr[:3, :3], T_lr[:3, 3]
        self.R1 = np.eye(3) if rect_R1 is None else np.asarray(rect_R1, np.float64).reshape(3, 3)
        rng = np.random.RandomState(3000 + seed)
        self.axis = rng.normal(size=3)
        self.axis[2] *= 0.3
        self.step = np.deg2rad(rng.uniform(0.3, 1.0) * max_step_deg)
        v = rng.normal(size=3)
        v[2] *= 0.5
        self.vel = v / np.linalg.norm(v) * rng.uniform(0.4, 1.0) * max_step_m
        n = np.array([rng.uniform(-0.25, 0.25), rng.uniform(-0.25, 0.25), 1.0])
        self.n = n / np.linalg.norm(n)
        self.d0 = rng.uniform(1.5, 3.0) * self.n[2]          # plane n.X = d0; optical axis hits it at z = d0/n_z
        self.X0 = np.array([0.0, 0.0, self.d0 / self.n[2]])
        e1 = np.cross([0.0, 1.0, 0.0], self.n)
        self.e1 = e1 / np.linalg.norm(e1)
        self.e2 = np.cross(self.n, self.e1)
        self.f = left.intrinsics[0]
        self.c0 = (left.intrinsics[2], left.intrinsics[3])

    def rotation(self, t: int) -> np.ndarray:
        return rot_from_axis_angle(self.axis, self.step * t)

    def position(self, t: int) -> np.ndarray:
        return self.vel * t

    def _render(self, cam: int, t: int) -> np.ndarray:
        R = self.rotation(t)
        o = self.position(t)
        if cam == 1:
            o = o + R @ self.t_lr
            R = R @ self.R_lr
        d = self.rays[cam] @ R.T                                  # world ray of every pixel
        lam = (self.d0 - self.n @ o) / (d @ self.n)
        X = o + lam[..., None] * d - self.X0
        s = self.f / self.X0[2]                                   # texture px per metre on the plane
        u = (X @ self.e1) * s + self.c0[0] + 96
        v = (X @ self.e2) * s + self.c0[1] + 96
        img = _sample_bilinear(self.tex, u.astype(np.float32), v.astype(np.float32))
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)

    def frame(self, t: int):
        return self._render(0, t), self._render(1, t)


def rig_keyframe_R_cur(stream: RigStream, t_kf: int, t_cur: int) -> np.ndarray:
    """keyframe_R_cur_frame in the rectified left frame (camLrectLkf_R_camLrectK)."""
    return stream.R1 @ stream.rotation(t_kf).T @ stream.rotation(t_cur) @ stream.R1.T
