"""The benchmark's workloads, built in ONE place so that `bench.py` (which times them) and
`tests/test_gpu_bench_configs.py` (which compares them with the oracle field by field) run exactly
the same contexts, frames, rotations and keyframe plan.

BASELINE.json configs:
    c2  single EuRoC 752x480 stream, 300 features, 3-level LK          (configs[1])
    c3  64 synthetic 752x480 streams, 600 features, ANMS on            (configs[2], the headline)
    c4  8 independent sequences sharded one per GPU, batch 1 per GPU   (configs[3])
    c5  1280x720, 1000 features, 4-level LK (+ the dense-stereo row)   (configs[4])

Input generation only (numpy); nothing here is part of the measured path.
"""
from __future__ import annotations

import copy
import os
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

from . import _abi as abi
from . import params as P
from . import synth

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
DT_NS = 50_000_000  # 20 Hz, the EuRoC camera rate

CONFIGS = {
    #        batch  W     H    features klt_max_level unique ring  source
    "c2": dict(batch=1, width=752, height=480, features=300, klt_max_level=2, unique=1, ring=9, source="euroc"),
    "c3": dict(batch=64, width=752, height=480, features=600, klt_max_level=2, unique=8, ring=6, source="rig"),
    "c4": dict(batch=1, width=752, height=480, features=300, klt_max_level=2, unique=1, ring=6, source="euroc"),
    "c5": dict(batch=32, width=1280, height=720, features=1000, klt_max_level=3, unique=4, ring=6, source="rig"),
    # c3 on REAL frames: 64 streams replaying 4 offset windows of MicroEuroc (600 features, every frame a keyframe): the
    # detection side sees real image statistics (~50 new corners per keyframe instead of ~11 on the rendered plane)
    "c3e": dict(batch=64, width=752, height=480, features=600, klt_max_level=2, unique=4, ring=6, source="euroc"),
}


def make_cameras(w: int, h: int) -> Tuple[abi.CameraParams, abi.CameraParams]:
    """params/Euroc/{Left,Right}CameraParams.yaml, intrinsics scaled about the image centre when
    (w, h) differs from 752x480 (BASELINE's C5 is a synthetic scale-up)."""
    L = P.load_camera_params(os.path.join(GOLDEN, "params_euroc", "LeftCameraParams.yaml"))
    R = P.load_camera_params(os.path.join(GOLDEN, "params_euroc", "RightCameraParams.yaml"))
    if (w, h) != (L.width, L.height):
        sx = w / L.width
        for cam in (L, R):
            cx0, cy0 = cam.width / 2.0, cam.height / 2.0
            cam.intrinsics[0] *= sx
            cam.intrinsics[1] *= sx
            cam.intrinsics[2] = w / 2.0 + (cam.intrinsics[2] - cx0) * sx
            cam.intrinsics[3] = h / 2.0 + (cam.intrinsics[3] - cy0) * sx
            cam.width, cam.height = w, h
    return L, R


def ping_pong(i: int, n: int) -> int:
    """frame index of step i in a ring of n frames walked forwards and backwards"""
    if n == 1:
        return 0
    period = 2 * (n - 1)
    j = i % period
    return j if j < n else period - j


@dataclass
class Workload:
    name: str
    left: abi.CameraParams
    right: abi.CameraParams
    params: abi.FrontendParams
    batch: int
    unique: int
    ring: int
    mode: str                       # "kf": every frame a keyframe; "nominal": reference cadence
    source: str                     # "rig" (synth.RigStream) or "euroc" (tests/golden/micro_euroc_f10_18.npz)
    lefts: np.ndarray = None        # [ring, unique, H, W] uint8
    rights: np.ndarray = None
    streams: list = field(default_factory=list)
    cam_R: list = field(default_factory=list)   # euroc: per unique sequence, camLrect_R at every ring frame

    @property
    def width(self):
        return self.left.width

    @property
    def height(self):
        return self.left.height

    def unique_of(self, s: int) -> int:
        """stream s of the batch replays unique stream s mod U (bench.py replicates 8 rendered streams)"""
        return s % self.unique

    def replicated(self) -> Tuple[np.ndarray, np.ndarray]:
        """[ring, batch, H, W] arrays as the benchmark uploads them"""
        idx = [self.unique_of(s) for s in range(self.batch)]
        return np.ascontiguousarray(self.lefts[:, idx]), np.ascontiguousarray(self.rights[:, idx])

    def keyframe_R_cur(self, u: int, t_kf: int, t: int) -> np.ndarray:
        if self.source == "rig":
            return synth.rig_keyframe_R_cur(self.streams[u], t_kf, t)
        return self.cam_R[u][t_kf].T @ self.cam_R[u][t]

    def plan(self, total: int) -> List[Tuple[int, int, List[np.ndarray], int]]:
        """[(ring frame t, timestamp_ns, [keyframe_R_cur_frame per unique stream], force_keyframe)] for
        `total` consecutive steps.  The rotation handed to the predictor / RANSAC is relative to the
        last frame the PLAN expects to be a keyframe (every frame in "kf" mode, every
        min_intra_keyframe_time in "nominal" mode, like the reference's cadence on EuRoC)."""
        out = []
        kf_t, last_kf = 0, 0
        for i in range(total):
            t = ping_pong(i, self.ring)
            Rs = [self.keyframe_R_cur(u, kf_t, t) for u in range(self.unique)]
            out.append((t, i * DT_NS, Rs, 1 if self.mode == "kf" else 0))
            if self.mode == "kf" or i == 0 or (i - last_kf) * DT_NS >= self.params.min_intra_keyframe_time_ns:
                kf_t, last_kf = t, i
        return out

    def batch_inputs(self, ctx, step) -> "abi.FrameInput array":
        t, ts, Rs, force = step
        B = self.batch
        return ctx.make_inputs([ts] * B, [Rs[self.unique_of(s)] for s in range(B)], [force] * B)


def _euroc_cam_R(L, body_R, R1):
    """camLrect_R_body^T . body_R . camLrect_R_body (StereoVisionImuFrontend.cpp:143-150)"""
    TL = np.array(L.body_pose_cam).reshape(4, 4)
    body_R_cam = TL[:3, :3] @ np.asarray(R1, np.float64).reshape(3, 3).T
    return [body_R_cam.T @ Rb @ body_R_cam for Rb in body_R]


def build(config: str, mode: str = "kf", rank: int = 0, use_ransac: int = 1, batch: int = None,
          features: int = None, klt_max_level: int = None, unique: int = None, ring: int = None,
          width: int = None, height: int = None, mono_2point: int = 1, stereo_1point: int = 1,
          rect_R1=None, seed_base: int = None, sequences=None) -> Workload:
    """Builds the workload of a BASELINE config; keyword arguments override single entries.
    `rect_R1` = R1 of kvfe_compute_rectification (host math; passed in so that this module does not
    need the GPU library).  `rank` offsets the stream seeds so that every GPU works on its own streams.
    `sequences` (c4): the ids of the EuRoC sequences this process owns (sharding.shard_streams); the batch
    is then one stream per sequence."""
    c = dict(CONFIGS[config])
    for k, v in dict(batch=batch, features=features, klt_max_level=klt_max_level, unique=unique, ring=ring,
                     width=width, height=height).items():
        if v is not None:
            c[k] = v
    p = P.load_frontend_params(os.path.join(GOLDEN, "params_euroc", "FrontendParams.yaml"), use_ransac=use_ransac)
    p.detector.max_features_per_frame = c["features"]
    p.tracker.klt_max_level = c["klt_max_level"]
    p.tracker.ransac_use_2point_mono = mono_2point
    p.tracker.ransac_use_1point_stereo = stereo_1point
    B = c["batch"]
    U = max(1, min(c["unique"], B))
    if c["source"] == "euroc":
        L = P.load_camera_params(os.path.join(GOLDEN, "sensorLeft.yaml"))
        R = P.load_camera_params(os.path.join(GOLDEN, "sensorRight.yaml"))
        z = np.load(os.path.join(GOLDEN, "micro_euroc_f10_18.npz"))
        n = len(z["lefts"])
        T = min(c["ring"], n)
        if rect_R1 is None:
            from . import frontend as F
            rect_R1 = F.compute_rectification(L, R).R1
        # "sequences" = offset windows of MicroEuroc (SURVEY.md §8c "Real-data fixtures"; the full EuRoC
        # sequences are not in the container): sequence q replays the T-frame window starting at frame
        # q mod (n-T+1), every second lap of the offsets in reverse.  c2 is sequence 0 with T = n.
        seqs = list(sequences) if sequences is not None else [rank]
        if config == "c2":
            seqs = [0]
        elif sequences is None and U > 1:   # c3e: U windows replicated to B streams
            seqs = [rank * U + u for u in range(U)]
        U = len(seqs)
        wl = Workload(config, L, R, p, U if sequences is not None else B, U, T, mode, "euroc")
        span = n - T + 1
        wl.lefts = np.empty((T, U) + z["lefts"].shape[1:], np.uint8)
        wl.rights = np.empty_like(wl.lefts)
        for u, q in enumerate(seqs):
            order = [q % span + k for k in range(T)]
            if (q // span) % 2:
                order = order[::-1]
            wl.lefts[:, u] = z["lefts"][order]
            wl.rights[:, u] = z["rights"][order]
            wl.cam_R.append(_euroc_cam_R(L, [z["body_R"][k] for k in order], rect_R1))
        return wl
    L, R = make_cameras(c["width"], c["height"])
    if rect_R1 is None:
        from . import frontend as F
        rect_R1 = F.compute_rectification(L, R).R1
    R1 = np.asarray(rect_R1, np.float64).reshape(3, 3)
    T = c["ring"]
    wl = Workload(config, L, R, p, B, U, T, mode, "rig")
    base = (100 * rank) if seed_base is None else seed_base
    wl.streams = [synth.RigStream(L, R, seed=base + u, rect_R1=R1) for u in range(U)]
    H, W = L.height, L.width
    wl.lefts = np.empty((T, U, H, W), np.uint8)
    wl.rights = np.empty((T, U, H, W), np.uint8)
    for t in range(T):
        for u in range(U):
            wl.lefts[t, u], wl.rights[t, u] = wl.streams[u].frame(t)
    return wl


def with_params(wl: Workload, **det) -> Workload:
    """copy of a workload with detector parameters replaced (frames shared)"""
    w2 = copy.copy(wl)
    w2.params = copy.deepcopy(wl.params)
    for k, v in det.items():
        setattr(w2.params.detector, k, v)
    return w2
