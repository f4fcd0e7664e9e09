"""Host-side mirror of the reference interface for the hot path, over the C ABI of libkvfe.

Names follow the reference classes so the parity tests read like the reference's own tests:

    StereoCamera            src/frontend/StereoCamera.cpp            (rectification constants)
    UndistorterRectifier    src/frontend/UndistorterRectifier.cpp
    FeatureDetector         src/frontend/feature-detector/FeatureDetector.cpp
    Tracker                 src/frontend/Tracker.cpp                 (featureTracking's numeric core)
    StereoMatcher           src/frontend/StereoMatcher.cpp
    StereoVisionImuFrontend src/frontend/StereoVisionImuFrontend.cpp (batched; useRANSAC with the
                            2-point mono / 1-point stereo problems)

Every method goes straight to the GPU library; nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _abi as abi
from .lib import KvfeError, load


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _img(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim != 2:
        raise ValueError("expected a 2-D uint8 image")
    return a


def _pts(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 2)


class Context:
    """A kvfe_ctx: StereoCamera + FeatureDetector + Tracker + StereoMatcher for `batch` streams."""

    def __init__(self, left: abi.CameraParams, right: abi.CameraParams, params: abi.FrontendParams,
                 batch: int = 1, device: int = 0, hip_stream: int | None = None,
                 candidate_capacity: int = 0, stream_groups: int = 0, frontend_type: int = 0,
                 depth: "abi.DepthParams | None" = None, device_frames_persist: int = 0,
                 single_hip_stream: int = 0, copy_inputs: int = 0, ssd_impl: int = 0, lk_impl: int = 0):
        self.lib = load()
        cfg = abi.Config()
        cfg.left, cfg.right, cfg.params = left, right, params
        cfg.batch, cfg.device = batch, device
        cfg.hip_stream = hip_stream
        cfg.candidate_capacity = candidate_capacity
        cfg.stream_groups = stream_groups
        cfg.frontend_type = frontend_type
        cfg.device_frames_persist = device_frames_persist
        cfg.single_hip_stream = single_hip_stream
        cfg.copy_inputs = copy_inputs
        cfg.ssd_impl = ssd_impl
        cfg.lk_impl = lk_impl
        if depth is not None:   # RgbdVisionImuFrontend: CameraParams::DepthParams of `left`
            cfg.depth = depth
        self.depth_params = depth
        self.cfg = cfg
        self.left, self.right, self.params = left, right, params
        self.batch = batch
        self.w, self.h = left.width, left.height
        h = C.c_void_p()
        st = self.lib.kvfe_create(C.byref(cfg), C.byref(h))
        if st != abi.KVFE_OK:
            raise KvfeError(st, "kvfe_create", self.lib.kvfe_status_string(st).decode())
        self._h = h
        self.rect = abi.Rectification()
        self._chk(self.lib.kvfe_get_rectification(self._h, C.byref(self.rect)), "get_rectification")
        d = params.detector
        self.kcap = d.max_features_per_frame + d.max_nr_keypoints_before_anms + 64

    def _chk(self, st, where):
        if st != abi.KVFE_OK:
            raise KvfeError(st, where, self.lib.kvfe_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self.lib.kvfe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- UndistorterRectifier ----------------------------------------------------------------
    def undistort_rectify_image(self, cam: int, img) -> np.ndarray:
        img = _img(img)
        out = np.empty_like(img)
        self._chk(self.lib.kvfe_undistort_rectify_image(self._h, cam, _p(img), img.strides[0], _p(out),
                                                        out.strides[0]), "undistort_rectify_image")
        return out

    def undistort_rectify_keypoints(self, cam: int, xy, use_R=True, use_P=True) -> np.ndarray:
        p = _pts(xy)
        out = np.zeros_like(p)
        self._chk(self.lib.kvfe_undistort_rectify_keypoints(self._h, cam, _p(p), len(p), int(use_R),
                                                            int(use_P), _p(out)), "undistort_keypoints")
        return out

    def get_bearing_vectors(self, cam: int, xy) -> np.ndarray:
        p = _pts(xy)
        out = np.zeros((len(p), 3), np.float64)
        self._chk(self.lib.kvfe_get_bearing_vectors(self._h, cam, _p(p), len(p), _p(out)),
                  "get_bearing_vectors")
        return out

    def equalize_hist(self, img) -> np.ndarray:
        """cv::equalizeHist as UtilsOpenCV::ReadAndConvertToGrayScale(img, equalize=True) applies it."""
        img = _img(img)
        out = np.empty_like(img)
        self._chk(self.lib.kvfe_equalize_hist(self._h, _p(img), img.strides[0], _p(out), out.strides[0]),
                  "equalize_hist")
        return out

    # ---- StereoMatcher::denseStereoReconstruction / StereoCamera::backProjectDisparityTo3D ---------
    def dense_stereo_reconstruction(self, left_rect, right_rect, params: abi.DenseStereoParams = None):
        """StereoMatcher::denseStereoReconstruction (StereoMatcher.cpp:32-121) on one rectified pair or
        a list of pairs -> int16 disparity (x16; (min_disparity - 1) * 16 where invalid), same shape"""
        single = not isinstance(left_rect, (list, tuple))
        lefts = [_img(a) for a in ([left_rect] if single else left_rect)]
        rights = [_img(a) for a in ([right_rect] if single else right_rect)]
        dp = params if params is not None else abi.dense_stereo_params_default()
        n = len(lefts)
        outs = [np.empty(a.shape, np.int16) for a in lefts]
        lp = (C.c_void_p * n)(*[a.ctypes.data for a in lefts])
        rp = (C.c_void_p * n)(*[a.ctypes.data for a in rights])
        op = (C.c_void_p * n)(*[a.ctypes.data for a in outs])
        w = lefts[0].shape[1]
        self._chk(self.lib.kvfe_dense_stereo_reconstruction(self._h, C.byref(dp), n, lp, rp, w, op, w),
                  "dense_stereo_reconstruction")
        return outs[0] if single else outs

    def dense_debug_volume(self, which: int, shape) -> np.ndarray:
        out = np.empty(shape, np.int16)
        self._chk(self.lib.kvfe_dense_debug_volume(self._h, which, _p(out), out.size), "dense_debug_volume")
        return out

    def dense_profile_read(self):
        ms, n = C.c_double(0), C.c_int64(0)
        self._chk(self.lib.kvfe_dense_profile_read(self._h, C.byref(ms), C.byref(n)), "dense_profile_read")
        return ms.value, n.value

    def backproject_disparity_to_3d(self, disparity_f32) -> np.ndarray:
        """StereoCamera::backProjectDisparityTo3D: float disparity (int16 / 16) -> [h, w, 3] float32"""
        d = np.ascontiguousarray(disparity_f32, np.float32)
        out = np.empty(d.shape + (3,), np.float32)
        self._chk(self.lib.kvfe_backproject_disparity_to_3d(self._h, _p(d), d.shape[1], _p(out)),
                  "backproject_disparity_to_3d")
        return out

    # ---- FeatureDetector -----------------------------------------------------------------------
    def raw_feature_detection(self, img, mask=None) -> np.ndarray:
        img = _img(img)
        cap = 8192
        out = np.zeros((cap, 2), np.float32)
        n = C.c_int32(0)
        m = _img(mask) if mask is not None else None
        self._chk(self.lib.kvfe_raw_feature_detection(self._h, _p(img), img.strides[0],
                                                      _p(m) if m is not None else None,
                                                      m.strides[0] if m is not None else 0, _p(out),
                                                      cap, C.byref(n)), "raw_feature_detection")
        return out[: n.value].copy()

    def feature_detection(self, img, tracked_xy, need: int) -> np.ndarray:
        img = _img(img)
        tr = _pts(tracked_xy)
        cap = 8192
        out = np.zeros((cap, 2), np.float32)
        n = C.c_int32(0)
        self._chk(self.lib.kvfe_feature_detection(self._h, _p(img), img.strides[0], _p(tr), len(tr),
                                                  need, _p(out), cap, C.byref(n)), "feature_detection")
        return out[: n.value].copy()

    def corner_subpix(self, img, xy, win=10, zero_zone=-1, max_iters=40, eps=0.001) -> np.ndarray:
        img = _img(img)
        p = _pts(xy).copy()
        self._chk(self.lib.kvfe_corner_subpix(self._h, _p(img), img.strides[0], _p(p), len(p), win,
                                              zero_zone, max_iters, eps), "corner_subpix")
        return p

    # ---- Tracker ---------------------------------------------------------------------------------
    def calc_optical_flow_pyr_lk(self, prev, cur, prev_xy, init_xy):
        prev, cur = _img(prev), _img(cur)
        p = _pts(prev_xy)
        q = _pts(init_xy).copy()
        n = len(p)
        status = np.zeros(n, np.uint8)
        err = np.zeros(n, np.float32)
        self._chk(self.lib.kvfe_calc_optical_flow_pyr_lk(self._h, _p(prev), _p(cur), prev.strides[0],
                                                         _p(p), _p(q), n, _p(status), _p(err)),
                  "calc_optical_flow_pyr_lk")
        return q, status, err

    def build_optical_flow_pyramid(self, imgs, with_level0_copy=False):
        """cv::buildOpticalFlowPyramid (the cv::pyrDown chain of cv::calcOpticalFlowPyrLK, Tracker.cpp:137-146) of a
        batch of images [n, H, W] (any row / image strides).  Returns (levels, copy): levels[s] = list of the
        levels 1..L of image s as 2-D arrays; copy = the dense level-0 copy [n, H, W] or None."""
        a = np.asarray(imgs)
        if a.ndim == 2:
            a = a[None]
        assert a.dtype == np.uint8 and a.ndim == 3 and a.strides[2] == 1
        n, H, W = a.shape
        row_stride, img_stride = a.strides[1], (a.strides[0] if n > 1 else a.strides[1] * H)
        sizes = np.zeros(2 * 16, np.int32)
        nlev = np.zeros(1, np.int32)
        cap = n * H * W
        out = np.zeros(cap, np.uint8)
        cp = np.zeros((n, H, W), np.uint8) if with_level0_copy else None
        self._chk(self.lib.kvfe_build_optical_flow_pyramid(self._h, a.ctypes.data, row_stride, img_stride, n, _p(out),
                                                           cap, _p(sizes), _p(nlev), _p(cp) if cp is not None else None),
                  "build_optical_flow_pyramid")
        L = int(nlev[0])
        per = sum(int(sizes[2 * l]) * int(sizes[2 * l + 1]) for l in range(1, L))
        levels = []
        for s in range(n):
            off, lv = s * per, []
            for l in range(1, L):
                w, h = int(sizes[2 * l]), int(sizes[2 * l + 1])
                lv.append(out[off:off + w * h].reshape(h, w).copy())
                off += w * h
            levels.append(lv)
        return levels, cp

    def debug_pyramid(self, steps_back=0, with_level0_copy=True):
        """test hook (kvfe_frontend_debug_pyramid): the front-end's own pyramid of the last step's frame (or the one
        before): (levels, copy) as build_optical_flow_pyramid returns them, for every stream of the batch."""
        sizes = np.zeros(2 * 16, np.int32)
        nlev = np.zeros(1, np.int32)
        H, W, n = self.h, self.w, self.batch
        probe = np.zeros((1, H, W), np.uint8)
        tmp = np.zeros(H * W, np.uint8)
        self._chk(self.lib.kvfe_build_optical_flow_pyramid(self._h, probe.ctypes.data, W, H * W, 1, _p(tmp), tmp.size,
                                                           _p(sizes), _p(nlev), None), "build_optical_flow_pyramid")
        L = int(nlev[0])
        per = sum(int(sizes[2 * l]) * int(sizes[2 * l + 1]) for l in range(1, L))
        out = np.zeros(max(1, n * per), np.uint8)
        cp = np.zeros((n, H, W), np.uint8) if with_level0_copy else None
        self._chk(self.lib.kvfe_frontend_debug_pyramid(self._h, steps_back, _p(cp) if cp is not None else None, _p(out),
                                                       out.size), "debug_pyramid")
        levels = []
        for s in range(n):
            off, lv = s * per, []
            for l in range(1, L):
                w, h = int(sizes[2 * l]), int(sizes[2 * l + 1])
                lv.append(out[off:off + w * h].reshape(h, w).copy())
                off += w * h
            levels.append(lv)
        return levels, cp

    def predict_sparse_flow(self, prev_xy, ref_R_cur) -> np.ndarray:
        p = _pts(prev_xy)
        R = np.ascontiguousarray(ref_R_cur, np.float64).reshape(9)
        out = np.zeros_like(p)
        self._chk(self.lib.kvfe_predict_sparse_flow(self._h, _p(p), len(p), _p(R), _p(out)),
                  "predict_sparse_flow")
        return out

    # ---- StereoMatcher ---------------------------------------------------------------------------
    def get_right_keypoints_rectified(self, left_rect, right_rect, left_xy, left_status):
        left_rect, right_rect = _img(left_rect), _img(right_rect)
        p = _pts(left_xy)
        st = np.ascontiguousarray(left_status, np.uint8)
        n = len(p)
        rxy = np.zeros((n, 2), np.float32)
        rst = np.zeros(n, np.uint8)
        score = np.zeros(n, np.float64)
        self._chk(self.lib.kvfe_get_right_keypoints_rectified(self._h, _p(left_rect), _p(right_rect),
                                                              left_rect.strides[0], _p(p), _p(st), n,
                                                              _p(rxy), _p(rst), _p(score)),
                  "get_right_keypoints_rectified")
        return rxy, rst, score

    def sparse_stereo_reconstruction(self, left, right, left_xy, want_images=False) -> dict:
        left, right = _img(left), _img(right)
        p = _pts(left_xy)
        n = len(p)
        res = dict(left_rect_xy=np.zeros((n, 2), np.float32), left_status=np.zeros(n, np.uint8),
                   right_rect_xy=np.zeros((n, 2), np.float32), right_status=np.zeros(n, np.uint8),
                   depth=np.zeros(n, np.float64), right_xy=np.zeros((n, 2), np.float32),
                   keypoints_3d=np.zeros((n, 3), np.float64))
        if want_images:
            res["left_rect_img"] = np.zeros((self.h, self.w), np.uint8)
            res["right_rect_img"] = np.zeros((self.h, self.w), np.uint8)
        so = abi.StereoOutput()
        for k, v in res.items():
            setattr(so, k, v.ctypes.data)
        self._chk(self.lib.kvfe_sparse_stereo_reconstruction(self._h, _p(left), _p(right),
                                                             left.strides[0], _p(p), n, C.byref(so)),
                  "sparse_stereo_reconstruction")
        return res

    # ---- Tracker: geometric outlier rejection ----------------------------------------------------
    @staticmethod
    def _ransac_result(out, inl):
        return dict(status=out.status, n_inliers=out.n_inliers, iterations=out.iterations,
                    pose=np.array(out.pose, np.float64).reshape(3, 4),
                    info=np.array(out.info, np.float64).reshape(3, 3), inliers=inl[: out.n_inliers].copy())

    def outlier_rejection_2d2d_given_rotation(self, f_ref, f_cur, R) -> dict:
        """Tracker::geometricOutlierRejection2d2d with the rotation given (2-point RANSAC)."""
        a = np.ascontiguousarray(f_ref, np.float64).reshape(-1, 3)
        b = np.ascontiguousarray(f_cur, np.float64).reshape(-1, 3)
        Rm = np.ascontiguousarray(R, np.float64).reshape(9)
        inl = np.zeros(max(len(a), 1), np.int32)
        out = abi.RansacOutput()
        self._chk(self.lib.kvfe_outlier_rejection_2d2d_given_rotation(self._h, _p(a), _p(b), len(a), _p(Rm),
                                                                      _p(inl), C.byref(out)),
                  "outlier_rejection_2d2d_given_rotation")
        return self._ransac_result(out, inl)

    def outlier_rejection_3d3d_given_rotation(self, ref_left_xy, ref_right_x, ref_p3, cur_left_xy,
                                              cur_right_x, cur_p3, R) -> dict:
        """Tracker::geometricOutlierRejection3d3dGivenRotation (1-point voting)."""
        rl, cl = _pts(ref_left_xy), _pts(cur_left_xy)
        rr = np.ascontiguousarray(ref_right_x, np.float32).reshape(-1)
        cr = np.ascontiguousarray(cur_right_x, np.float32).reshape(-1)
        rp = np.ascontiguousarray(ref_p3, np.float64).reshape(-1, 3)
        cp = np.ascontiguousarray(cur_p3, np.float64).reshape(-1, 3)
        Rm = np.ascontiguousarray(R, np.float64).reshape(9)
        n = len(rl)
        inl = np.zeros(max(n, 1), np.int32)
        out = abi.RansacOutput()
        self._chk(self.lib.kvfe_outlier_rejection_3d3d_given_rotation(
            self._h, _p(rl), _p(rr), _p(rp), _p(cl), _p(cr), _p(cp), n, _p(Rm), _p(inl), C.byref(out)),
            "outlier_rejection_3d3d_given_rotation")
        return self._ransac_result(out, inl)

    def outlier_rejection_2d2d(self, f_ref, f_cur) -> dict:
        """Tracker::geometricOutlierRejection2d2d without rotation prior (5-point Nister RANSAC)."""
        a = np.ascontiguousarray(f_ref, np.float64).reshape(-1, 3)
        b = np.ascontiguousarray(f_cur, np.float64).reshape(-1, 3)
        n = len(a)
        inl = np.zeros(max(n, 1), np.int32)
        out = abi.RansacOutput()
        self._chk(self.lib.kvfe_outlier_rejection_2d2d(self._h, _p(a), _p(b), n, _p(inl), C.byref(out)),
                  "outlier_rejection_2d2d")
        return self._ransac_result(out, inl)

    def outlier_rejection_3d3d(self, ref_p3, cur_p3) -> dict:
        """Tracker::geometricOutlierRejection3d3d (3-point Arun RANSAC) on matched 3-D points."""
        rp = np.ascontiguousarray(ref_p3, np.float64).reshape(-1, 3)
        cp = np.ascontiguousarray(cur_p3, np.float64).reshape(-1, 3)
        n = len(rp)
        inl = np.zeros(max(n, 1), np.int32)
        out = abi.RansacOutput()
        self._chk(self.lib.kvfe_outlier_rejection_3d3d(self._h, _p(rp), _p(cp), n, _p(inl), C.byref(out)),
                  "outlier_rejection_3d3d")
        return self._ransac_result(out, inl)

    def pnp(self, cam_bearing_vectors, F_points, pnp_params: abi.PnpParams = None) -> dict:
        """Tracker::pnp(bearings, F_points, ...) (Tracker.cpp:1122-1288, EPNP RANSAC) + outlierRejectionPnP's status;
        result["success"] = Tracker::pnp's return value, pose = F_Pose_cam."""
        f = np.ascontiguousarray(cam_bearing_vectors, np.float64).reshape(-1, 3)
        pw = np.ascontiguousarray(F_points, np.float64).reshape(-1, 3)
        assert len(f) == len(pw)
        pp = pnp_params if pnp_params is not None else abi.pnp_params_default()
        inl = np.zeros(max(len(f), 1), np.int32)
        out = abi.RansacOutput()
        self._chk(self.lib.kvfe_pnp(self._h, C.byref(pp), _p(f), _p(pw), len(f), _p(inl), C.byref(out)), "pnp")
        r = self._ransac_result(out, inl)
        r["success"] = bool(out.reserved0)
        return r

    # ---- UndistorterRectifier / StereoCamera / StereoMatcher keypoint methods on their own -------
    def check_undistorted_rectified_left_keypoints(self, cam: int, distorted_xy, undistorted_xy, pixel_tol=2.0):
        d, u = _pts(distorted_xy), _pts(undistorted_xy)
        assert len(d) == len(u)
        out, st = np.zeros_like(d), np.zeros(len(d), np.uint8)
        self._chk(self.lib.kvfe_check_undistorted_rectified_left_keypoints(
            self._h, cam, _p(d), _p(u), len(d), float(pixel_tol), _p(out), _p(st)), "check_undistorted_rectified")
        return out, st

    def distort_unrectify_keypoints(self, cam: int, rect_xy, status) -> np.ndarray:
        r = _pts(rect_xy)
        st = np.ascontiguousarray(status, np.uint8)
        out = np.zeros_like(r)
        self._chk(self.lib.kvfe_distort_unrectify_keypoints(self._h, cam, _p(r), _p(st), len(r), _p(out)),
                  "distort_unrectify_keypoints")
        return out

    def undistort_rectify_left_keypoints(self, xy):
        k = _pts(xy)
        out, st = np.zeros_like(k), np.zeros(len(k), np.uint8)
        self._chk(self.lib.kvfe_undistort_rectify_left_keypoints(self._h, _p(k), len(k), _p(out), _p(st)),
                  "undistort_rectify_left_keypoints")
        return out, st

    def distort_unrectify_right_keypoints(self, rect_xy, status) -> np.ndarray:
        r = _pts(rect_xy)
        st = np.ascontiguousarray(status, np.uint8)
        out = np.zeros_like(r)
        self._chk(self.lib.kvfe_distort_unrectify_right_keypoints(self._h, _p(r), _p(st), len(r), _p(out)),
                  "distort_unrectify_right_keypoints")
        return out

    def undistort_rectify_stereo_frame(self, left, right):
        l, r = _img(left), _img(right)
        lo, ro = np.empty_like(l), np.empty_like(r)
        self._chk(self.lib.kvfe_undistort_rectify_stereo_frame(self._h, _p(l), _p(r), l.shape[1], _p(lo), _p(ro),
                                                               l.shape[1]), "undistort_rectify_stereo_frame")
        return lo, ro

    def get_depth_from_rectified_matches(self, left_xy, left_status, right_xy, right_status):
        l, r = _pts(left_xy), _pts(right_xy)
        ls = np.ascontiguousarray(left_status, np.uint8)
        rs = np.ascontiguousarray(right_status, np.uint8).copy()
        depth = np.zeros(len(l), np.float64)
        self._chk(self.lib.kvfe_get_depth_from_rectified_matches(self._h, _p(l), _p(ls), _p(r), _p(rs), len(l),
                                                                 _p(depth)), "get_depth_from_rectified_matches")
        return depth, rs

    # ---- Frame-level FeatureDetector / Tracker calls ------------------------------------------------
    @staticmethod
    def _frame_struct(cap, n, kps, lmk, age, ver):
        f = abi.Frame()
        f.capacity, f.n_keypoints = cap, n
        f.keypoints, f.landmarks, f.landmarks_age, f.versors = (kps.ctypes.data, lmk.ctypes.data, age.ctypes.data,
                                                                ver.ctypes.data)
        return f

    def _frame_arrays(self, frame: dict | None, cap: int):
        kps = np.zeros((cap, 2), np.float32)
        lmk = np.zeros(cap, np.int64)
        age = np.zeros(cap, np.int32)
        ver = np.zeros((cap, 3), np.float64)
        n = 0
        if frame is not None:
            n = len(frame["landmarks"])
            kps[:n] = _pts(frame["keypoints"])
            lmk[:n] = frame["landmarks"]
            age[:n] = frame["landmarks_age"]
            if "versors" in frame and frame["versors"] is not None:
                ver[:n] = np.asarray(frame["versors"], np.float64).reshape(-1, 3)
        return n, kps, lmk, age, ver

    @staticmethod
    def _frame_dict(f, kps, lmk, age, ver) -> dict:
        n = f.n_keypoints
        return dict(keypoints=kps[:n].copy(), landmarks=lmk[:n].copy(), landmarks_age=age[:n].copy(),
                    versors=ver[:n].copy())

    def feature_detection_frame(self, img, frame: dict | None, landmark_counter: int = 0):
        """FeatureDetector::featureDetection(Frame*, R): returns (frame dict, new landmark counter)"""
        img = _img(img)
        n, kps, lmk, age, ver = self._frame_arrays(frame, self.kcap)
        f = self._frame_struct(self.kcap, n, kps, lmk, age, ver)
        ctr = C.c_int64(int(landmark_counter))
        self._chk(self.lib.kvfe_feature_detection_frame(self._h, _p(img), img.shape[1], C.byref(f), C.byref(ctr)),
                  "feature_detection_frame")
        return self._frame_dict(f, kps, lmk, age, ver), int(ctr.value)

    def feature_tracking_frame(self, ref_img, cur_img, ref_frame: dict, ref_R_cur=None):
        """Tracker::featureTracking: returns (ref landmarks after the call, cur frame dict)"""
        a, b = _img(ref_img), _img(cur_img)
        n, kps, lmk, age, ver = self._frame_arrays(ref_frame, self.kcap)
        fr = self._frame_struct(self.kcap, n, kps, lmk, age, ver)
        _, ck, cl, ca, cv = self._frame_arrays(None, self.kcap)
        fc = self._frame_struct(self.kcap, 0, ck, cl, ca, cv)
        R = np.ascontiguousarray(np.eye(3) if ref_R_cur is None else ref_R_cur, np.float64).reshape(9)
        self._chk(self.lib.kvfe_feature_tracking_frame(self._h, _p(a), _p(b), a.shape[1], C.byref(fr), C.byref(fc),
                                                       _p(R)), "feature_tracking_frame")
        return lmk[:n].copy(), self._frame_dict(fc, ck, cl, ca, cv)

    # ---- StereoVisionImuFrontend (batched) -----------------------------------------------------
    def make_inputs(self, timestamps_ns, Rs=None, force_keyframe=None):
        arr = (abi.FrameInput * self.batch)()
        for s in range(self.batch):
            arr[s].timestamp_ns = int(timestamps_ns[s])
            R = np.eye(3) if Rs is None else np.asarray(Rs[s], np.float64).reshape(3, 3)
            for i in range(9):
                arr[s].keyframe_R_cur_frame[i] = float(R.reshape(9)[i])
            arr[s].force_keyframe = 0 if force_keyframe is None else int(force_keyframe[s])
        return arr

    def step_host(self, lefts, rights, inputs):
        """lefts/rights: uint8 arrays [batch, H, W] in host memory (rights = None for the mono front-end;
        the RGBD front-end takes the depth images, uint16 or float32 per DepthParams.depth_type)."""
        lefts = np.ascontiguousarray(lefts, np.uint8)
        assert lefts.shape == (self.batch, self.h, self.w)
        if self.cfg.frontend_type == abi.FRONTEND_RGBD:
            dt = np.float32 if self.depth_params.depth_type == abi.DEPTH_F32 else np.uint16
            depths = np.ascontiguousarray(rights, dt)
            assert depths.shape == lefts.shape
            self._chk(self.lib.kvfe_frontend_step_host(self._h, _p(lefts), _p(depths), self.w, self.w * self.h,
                                                       inputs), "frontend_step_host")
            return
        if rights is not None:
            rights = np.ascontiguousarray(rights, np.uint8)
            assert rights.shape == lefts.shape
        self._chk(self.lib.kvfe_frontend_step_host(self._h, _p(lefts), _p(rights) if rights is not None else None,
                                                   self.w, self.w * self.h, inputs), "frontend_step_host")

    def step_device(self, left_ptr: int, right_ptr: int, inputs, row_stride=None, image_stride=None):
        """left_ptr/right_ptr: device pointers to `batch` images, read by this step only (valid until it has
        completed); no caller pointer is kept past the call."""
        self._chk(self.lib.kvfe_frontend_step_device(self._h, C.c_void_p(left_ptr), C.c_void_p(right_ptr),
                                                     row_stride or self.w,
                                                     image_stride or self.w * self.h, inputs),
                  "frontend_step_device")

    def update_map(self, stream: int, landmark_ids, xyz):
        """Tracker::updateMap for one stream: landmark id -> world position (use_pnp_tracking)."""
        ids = np.ascontiguousarray(landmark_ids, np.int64).reshape(-1)
        pts = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
        assert len(ids) == len(pts)
        self._chk(self.lib.kvfe_frontend_update_map(self._h, stream, _p(ids), _p(pts), len(ids)), "frontend_update_map")

    def staging_buffers(self, slot: int):
        """numpy views [batch, H, W] of the pinned left / right staging slot `slot`."""
        pl, pr = C.c_void_p(), C.c_void_p()
        self._chk(self.lib.kvfe_frontend_staging_buffer(self._h, slot, C.byref(pl), C.byref(pr)),
                  "frontend_staging_buffer")
        n = self.batch * self.h * self.w
        shape = (self.batch, self.h, self.w)
        left = np.ctypeslib.as_array((C.c_uint8 * n).from_address(pl.value)).reshape(shape)
        right = np.ctypeslib.as_array((C.c_uint8 * n).from_address(pr.value)).reshape(shape)
        return left, right

    def staging_wait(self, slot: int):
        self._chk(self.lib.kvfe_frontend_staging_wait(self._h, slot), "frontend_staging_wait")

    def step_staged(self, slot: int, inputs):
        """upload staging slot `slot` on the copy stream (overlapping the previous step) and step."""
        self._chk(self.lib.kvfe_frontend_step_staged(self._h, slot, inputs), "frontend_step_staged")

    def synchronize(self):
        self._chk(self.lib.kvfe_synchronize(self._h), "synchronize")

    def reset(self):
        self._chk(self.lib.kvfe_frontend_reset(self._h), "frontend_reset")

    def get_output(self, stream: int, steps_back: int = 0) -> dict:
        """kvfe_frontend_get_output_at: the output record of the step `steps_back` steps before the latest one"""
        cap = self.kcap
        arrs = dict(landmarks=np.zeros(cap, np.int64), landmarks_age=np.zeros(cap, np.int32),
                    keypoints=np.zeros((cap, 2), np.float32), versors=np.zeros((cap, 3), np.float64),
                    left_rect_xy=np.zeros((cap, 2), np.float32), left_status=np.zeros(cap, np.uint8),
                    right_rect_xy=np.zeros((cap, 2), np.float32), right_status=np.zeros(cap, np.uint8),
                    depth=np.zeros(cap, np.float64), right_xy=np.zeros((cap, 2), np.float32),
                    keypoints_3d=np.zeros((cap, 3), np.float64), meas_landmark=np.zeros(cap, np.int64),
                    meas_uL_uR_v=np.zeros((cap, 3), np.float64))
        out = abi.FrameOutput()
        out.capacity = cap
        for k, v in arrs.items():
            setattr(out, k, v.ctypes.data)
        self._chk(self.lib.kvfe_frontend_get_output_at(self._h, stream, steps_back, C.byref(out)), "frontend_get_output")
        n = min(out.n_keypoints, cap)
        m = min(out.n_measurements, cap)
        d = dict(n_keypoints=out.n_keypoints, is_keyframe=out.is_keyframe, n_tracked=out.n_tracked,
                 n_detected=out.n_detected, n_measurements=out.n_measurements, frame_id=out.frame_id,
                 tracking_status_mono=out.tracking_status_mono,
                 tracking_status_stereo=out.tracking_status_stereo,
                 tracking_status_pnp=out.tracking_status_pnp, nr_pnp_inliers=out.nr_pnp_inliers,
                 W_T_k_pnp=np.array(out.W_T_k_pnp, np.float64).reshape(3, 4),
                 lkf_T_k_mono=np.array(out.lkf_T_k_mono, np.float64).reshape(3, 4),
                 lkf_T_k_stereo=np.array(out.lkf_T_k_stereo, np.float64).reshape(3, 4),
                 info_mat_stereo_translation=np.array(out.info_mat_stereo_translation, np.float64).reshape(3, 3),
                 nr_mono_putatives=out.nr_mono_putatives, nr_mono_inliers=out.nr_mono_inliers,
                 nr_stereo_putatives=out.nr_stereo_putatives, nr_stereo_inliers=out.nr_stereo_inliers)
        for k, v in arrs.items():
            d[k] = v[:m].copy() if k.startswith("meas_") else v[:n].copy()
        return d

    def output_buffers(self) -> "OutputBuffers":
        """caller-owned output storage for all streams, allocated once (kvfe_frontend_get_outputs fills it)"""
        return OutputBuffers(self)

    def profile_enable(self, on=1):
        """on = N: every N-th step records per-stage HIP events (0 / False: off)."""
        self._chk(self.lib.kvfe_profile_enable(self._h, int(on)), "profile_enable")

    def profile_read(self) -> dict:
        st = abi.StageTimes()
        st.struct_size = C.sizeof(abi.StageTimes)
        self._chk(self.lib.kvfe_profile_read(self._h, C.byref(st)), "profile_read")
        return dict(n_samples=st.n_samples, n_groups=st.n_groups,
                    stages={st.name[i].decode(): dict(ms_total=st.ms_total[i], alg_bytes=st.alg_bytes[i],
                                                      ms_active=st.ms_active[i], active_streams=st.active_streams[i],
                                                      alg_bytes_per_stream=st.alg_bytes_per_stream[i],
                                                      active_launches=st.active_launches[i])
                            for i in range(st.n_stages)})


def hbm_copy_probe(nbytes=1 << 30, iters=10) -> dict:
    """kvfe_hbm_copy_probe on the current device: read + write GB/s of a plain 16-byte-per-lane streaming copy of
    `nbytes` bytes (measurement hook: tells a slow box from a regression; no reference counterpart)"""
    L = load()
    g, ms = C.c_double(0.0), C.c_double(0.0)
    st = L.kvfe_hbm_copy_probe(int(nbytes), int(iters), C.byref(g), C.byref(ms))
    if st != 0:
        raise KvfeError(st, "hbm_copy_probe", L.kvfe_status_string(st).decode())
    return dict(read_plus_write_GBps=g.value, ms_per_copy=ms.value, bytes=int(nbytes), iters=int(iters))


# reference-shaped aliases ------------------------------------------------------------------------
class OutputBuffers:
    """kvfe_frame_output[batch] with their arrays, allocated once: `read(steps_back)` = kvfe_frontend_get_outputs -- one
    C call copies every stream's record of that step out of the pinned ring slot (a consumer that reads frame k while
    frame k+1 is being processed passes steps_back = 1)."""
    FIELDS = (("landmarks", np.int64, 1), ("landmarks_age", np.int32, 1), ("keypoints", np.float32, 2),
              ("versors", np.float64, 3), ("left_rect_xy", np.float32, 2), ("left_status", np.uint8, 1),
              ("right_rect_xy", np.float32, 2), ("right_status", np.uint8, 1), ("depth", np.float64, 1),
              ("right_xy", np.float32, 2), ("keypoints_3d", np.float64, 3), ("meas_landmark", np.int64, 1),
              ("meas_uL_uR_v", np.float64, 3))

    def __init__(self, ctx: "Context"):
        self.ctx = ctx
        B, cap = ctx.batch, ctx.kcap
        self.structs = (abi.FrameOutput * B)()
        self.arrays = []
        for s in range(B):
            arrs = {k: np.zeros((cap, w) if w > 1 else cap, dt) for k, dt, w in self.FIELDS}
            self.structs[s].capacity = cap
            for k, v in arrs.items():
                setattr(self.structs[s], k, v.ctypes.data)
            self.arrays.append(arrs)

    def read(self, steps_back: int = 0):
        self.ctx._chk(self.ctx.lib.kvfe_frontend_get_outputs(self.ctx._h, steps_back, self.structs), "frontend_get_outputs")
        return self.structs


class StereoVisionImuFrontend(Context):
    """Batched StereoVisionImuFrontend::spinOnce for `batch` independent streams."""

    def spin_once_host(self, lefts, rights, timestamps_ns, Rs=None, force_keyframe=None):
        self.step_host(lefts, rights, self.make_inputs(timestamps_ns, Rs, force_keyframe))
        return [self.get_output(s) for s in range(self.batch)]


def compute_rectification(left: abi.CameraParams, right: abi.CameraParams) -> abi.Rectification:
    """StereoCamera::computeRectificationParameters (host math, no device needed)."""
    r = abi.Rectification()
    st = load().kvfe_compute_rectification(C.byref(left), C.byref(right), C.byref(r))
    if st != abi.KVFE_OK:
        raise KvfeError(st, "kvfe_compute_rectification")
    return r


def compute_undistort_rectify_maps(cam: abi.CameraParams, R, P):
    """UndistorterRectifier::initUndistortRectifyMaps (host math, no device needed)."""
    Rm = np.ascontiguousarray(R, np.float64).reshape(9)
    Pm = np.ascontiguousarray(P, np.float64).reshape(12)
    mx = np.zeros((cam.height, cam.width), np.float32)
    my = np.zeros((cam.height, cam.width), np.float32)
    st = load().kvfe_compute_undistort_rectify_maps(C.byref(cam), _p(Rm), _p(Pm), _p(mx), _p(my))
    if st != abi.KVFE_OK:
        raise KvfeError(st, "kvfe_compute_undistort_rectify_maps")
    return mx, my
