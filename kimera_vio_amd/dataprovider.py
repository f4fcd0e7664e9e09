"""Input side of the front-end over the C ABI of libkvfe: the host-side mirror of

* ``utils::ThreadsafeImuBuffer``            (include/kimera-vio/utils/ThreadsafeImuBuffer.h:46-196)
* ``StereoDataProviderModule``              (src/dataprovider/StereoDataProviderModule.cpp:35-91, over
  MonoDataProviderModule.cpp:44-118 and DataProviderModule.cpp:80-181)
* ``EurocDataProvider``                     (src/dataprovider/EurocDataProvider.cpp:146-195, 229-306, 437-482)
* ``UtilsOpenCV::ReadAndConvertToGrayScale`` (src/utils/UtilsOpenCV.cpp:390-403) for PNG files

Same names and argument meaning as upstream; the work (PNG decoding, the IMU buffer queries, the
left / right / IMU synchronisation, the index parsing) is done by the library -- this file only marshals.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _abi as abi
from .lib import KvfeError, load


def _check(st: int, where: str):
    if st != 0:
        raise KvfeError(st, where)


# ---------------------------------------------------------------------------------------------------------
# images
# ---------------------------------------------------------------------------------------------------------
def png_info(data: bytes):
    """(width, height, channels of the cv::Mat cv::imread(IMREAD_ANYCOLOR) returns)"""
    w, h, c = C.c_int32(), C.c_int32(), C.c_int32()
    _check(load().kvfe_png_info(data, len(data), C.byref(w), C.byref(h), C.byref(c)), "kvfe_png_info")
    return w.value, h.value, c.value


def decode_png_gray(data: bytes, out: np.ndarray | None = None) -> np.ndarray:
    """cv::imread(IMREAD_ANYCOLOR) + cv::cvtColor(BGR2GRAY) when the file has colour: 8-bit grey image"""
    w, h, _ = png_info(data)
    if out is None:
        out = np.empty((h, w), np.uint8)
    if out.shape != (h, w) or out.dtype != np.uint8 or out.strides[1] != 1:
        raise ValueError("output must be a (height, width) uint8 array with unit column stride")
    _check(load().kvfe_png_decode_gray(data, len(data), out.ctypes.data_as(C.c_void_p), out.strides[0], w, h),
           "kvfe_png_decode_gray")
    return out


def jpeg_info(data: bytes):
    """(width, height, components of the file)"""
    w, h, c = C.c_int32(), C.c_int32(), C.c_int32()
    _check(load().kvfe_jpeg_info(data, len(data), C.byref(w), C.byref(h), C.byref(c)), "kvfe_jpeg_info")
    return w.value, h.value, c.value


def decode_jpeg_gray(data: bytes, out: np.ndarray | None = None) -> np.ndarray:
    """cv::imread(IMREAD_ANYCOLOR) of a baseline JPEG (libjpeg's defaults) + cv::cvtColor(BGR2GRAY) for colour files"""
    w, h, _ = jpeg_info(data)
    if out is None:
        out = np.empty((h, w), np.uint8)
    if out.shape != (h, w) or out.dtype != np.uint8 or out.strides[1] != 1:
        raise ValueError("output must be a (height, width) uint8 array with unit column stride")
    _check(load().kvfe_jpeg_decode_gray(data, len(data), out.ctypes.data_as(C.c_void_p), out.strides[0], w, h),
           "kvfe_jpeg_decode_gray")
    return out


def decode_png_gray_batch(files: list[bytes], out: np.ndarray, threads: int = 0) -> np.ndarray:
    """n PNG files into out[n, height, width] (e.g. a view of a pinned staging slot) by `threads` host threads"""
    n = len(files)
    if out.ndim != 3 or out.shape[0] != n or out.dtype != np.uint8 or out.strides[2] != 1:
        raise ValueError("output must be (n, height, width) uint8 with unit column stride")
    h, w = out.shape[1], out.shape[2]
    # (pointer tables without a Python-level loop over ctypes objects: at 128 files per call that loop was ~0.4 ms,
    # a third of a frame's decode time, spent with every worker thread idle)
    bufs = (C.c_char_p * n)(*files)
    sizes = np.fromiter(map(len, files), np.uint64, n)
    dsts = out.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(out.strides[0])
    status = (C.c_int32 * n)()
    st = load().kvfe_png_decode_gray_batch(C.cast(bufs, C.POINTER(C.c_void_p)), sizes.ctypes.data_as(C.POINTER(C.c_size_t)),
                                           dsts.ctypes.data_as(C.POINTER(C.c_void_p)), out.strides[1], w, h, n, threads,
                                           status)
    if st != 0:
        bad = [i for i in range(n) if status[i] != 0]
        raise KvfeError(st, "kvfe_png_decode_gray_batch", f"files {bad}")
    return out


def ReadAndConvertToGrayScale(img_name: str, equalize: bool = False, ctx=None) -> np.ndarray:
    """UtilsOpenCV::ReadAndConvertToGrayScale for a PNG or baseline JPEG file; `equalize` needs a front-end context (cv::equalizeHist
    runs on the device: kvfe_equalize_hist)"""
    with open(img_name, "rb") as f:
        data = f.read()
    img = decode_jpeg_gray(data) if data[:2] == b"\xff\xd8" else decode_png_gray(data)
    if equalize:
        if ctx is None:
            raise ValueError("equalize=True needs ctx (kvfe_equalize_hist is a device kernel)")
        img = ctx.equalize_hist(img)
    return img


# ---------------------------------------------------------------------------------------------------------
# IMU buffer
# ---------------------------------------------------------------------------------------------------------
class ThreadsafeImuBuffer:
    """utils::ThreadsafeImuBuffer.  Queries return (QueryResult, stamps int64[n], acc_gyr float64[6, n])."""
    kDataAvailable, kDataNotYetAvailable, kDataNeverAvailable, kQueueShutdown, kTooFewMeasurementsAvailable = range(5)

    def __init__(self, buffer_length_ns: int = -1):
        self._lib = load()
        self._h = self._lib.kvfe_imu_buffer_create(buffer_length_ns)
        if not self._h:
            raise MemoryError("kvfe_imu_buffer_create")

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.kvfe_imu_buffer_destroy(self._h)
            self._h = None

    def addMeasurement(self, timestamp_ns: int, acc_gyr):
        a = np.ascontiguousarray(acc_gyr, np.float64).reshape(6)
        self._lib.kvfe_imu_buffer_add(self._h, int(timestamp_ns), a.ctypes.data_as(C.POINTER(C.c_double)))

    def size(self) -> int:
        return int(self._lib.kvfe_imu_buffer_size(self._h))

    def shutdown(self):
        self._lib.kvfe_imu_buffer_shutdown(self._h)

    def _query(self, fn, *args):
        cap = 64
        while True:
            stamps = np.empty(cap, np.int64)
            vals = np.empty((cap, 6), np.float64)
            n = C.c_int32()
            r = fn(self._h, *args, stamps.ctypes.data_as(C.POINTER(C.c_int64)),
                   vals.ctypes.data_as(C.POINTER(C.c_double)), cap, C.byref(n))
            if r == -1:
                cap = n.value
                continue
            return r, stamps[:n.value].copy(), vals[:n.value].T.copy()

    def getImuDataBtwTimestamps(self, t_from: int, t_to: int, get_lower_bound: bool = False):
        return self._query(lambda h, *a: self._lib.kvfe_imu_buffer_between(h, int(t_from), int(t_to),
                                                                           int(get_lower_bound), *a))

    def getImuDataInterpolatedUpperBorder(self, t_from: int, t_to: int):
        return self._query(lambda h, *a: self._lib.kvfe_imu_buffer_interpolated_upper_border(h, int(t_from),
                                                                                             int(t_to), *a))

    def getImuDataInterpolatedBorders(self, t_from: int, t_to: int):
        return self._query(lambda h, *a: self._lib.kvfe_imu_buffer_interpolated_borders(h, int(t_from), int(t_to),
                                                                                        *a))

    @staticmethod
    def linearInterpolate(t0: int, y0, t1: int, y1, t: int) -> np.ndarray:
        a = np.ascontiguousarray(y0, np.float64).reshape(6)
        b = np.ascontiguousarray(y1, np.float64).reshape(6)
        y = np.empty(6, np.float64)
        p = C.POINTER(C.c_double)
        load().kvfe_imu_linear_interpolate(int(t0), a.ctypes.data_as(p), int(t1), b.ctypes.data_as(p), int(t),
                                           y.ctypes.data_as(p))
        return y


def preintegrate_rotation(stamps, acc_gyr, gyro_bias=(0.0, 0.0, 0.0), deltaRij=None) -> np.ndarray:
    """ImuFrontend::preintegrateImuMeasurements, rotation part: deltaRij (3x3, identity when None) advanced by the
    packet's samples (stamps int64[n], acc_gyr float64[6, n] as the packets carry them)"""
    t = np.ascontiguousarray(stamps, np.int64)
    ag = np.ascontiguousarray(np.asarray(acc_gyr, np.float64).T)          # column k = sample k -> 6 per sample
    R = np.eye(3) if deltaRij is None else np.array(deltaRij, np.float64).reshape(3, 3).copy()
    b = np.ascontiguousarray(gyro_bias, np.float64).reshape(3)
    p = C.POINTER(C.c_double)
    _check(load().kvfe_imu_preintegrate_rotation(t.ctypes.data_as(C.POINTER(C.c_int64)), ag.ctypes.data_as(p), len(t),
                                                 b.ctypes.data_as(p), R.ctypes.data_as(p)),
           "kvfe_imu_preintegrate_rotation")
    return R


def keyframe_R_cur_frame(body_R_camLrect, deltaRij) -> np.ndarray:
    """camLrectLkf_R_camLrectK_imu = cam_R_body . deltaRij . body_R_cam (StereoVisionImuFrontend.cpp:143-150): the
    kvfe_frame_input::keyframe_R_cur_frame of the current frame"""
    a = np.ascontiguousarray(body_R_camLrect, np.float64).reshape(3, 3)
    d = np.ascontiguousarray(deltaRij, np.float64).reshape(3, 3)
    out = np.empty((3, 3), np.float64)
    p = C.POINTER(C.c_double)
    load().kvfe_keyframe_R_cur_frame(a.ctypes.data_as(p), d.ctypes.data_as(p), out.ctypes.data_as(p))
    return out


# ---------------------------------------------------------------------------------------------------------
# left / right / IMU synchronisation
# ---------------------------------------------------------------------------------------------------------
class StereoImuSyncPacket:
    """what StereoDataProviderModule hands to the front-end: the frame pair (by caller tag) and the IMU samples
    since the previous packet's frame, both borders interpolated (StereoImuSyncPacket.h:81-107)"""
    __slots__ = ("timestamp", "left_tag", "right_tag", "imu_stamps", "imu_accgyrs")

    def __init__(self, timestamp, left_tag, right_tag, imu_stamps, imu_accgyrs):
        self.timestamp, self.left_tag, self.right_tag = timestamp, left_tag, right_tag
        self.imu_stamps, self.imu_accgyrs = imu_stamps, imu_accgyrs


class StereoDataProviderModule:
    """StereoDataProviderModule in sequential mode: fill the queues, call getInputPacket() (spinOnce).
    `last_action` is the KVFE_SYNC_* code of the last call (why a frame was dropped, or that the module waits)."""

    MODE = 0   # KVFE_SYNC_MODE_STEREO

    def __init__(self, imu_buffer_length_ns: int = -1):
        self._lib = load()
        self._h = self._lib.kvfe_stereo_sync_create(imu_buffer_length_ns)
        if not self._h:
            raise MemoryError("kvfe_stereo_sync_create")
        _check(self._lib.kvfe_stereo_sync_set_mode(self._h, self.MODE), "kvfe_stereo_sync_set_mode")
        self.last_action = abi.SYNC_EMPTY

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.kvfe_stereo_sync_destroy(self._h)
            self._h = None

    def fillLeftFrameQueue(self, timestamp_ns: int, tag: int):
        self._lib.kvfe_stereo_sync_fill_left(self._h, int(timestamp_ns), int(tag))

    def fillRightFrameQueue(self, timestamp_ns: int, tag: int):
        self._lib.kvfe_stereo_sync_fill_right(self._h, int(timestamp_ns), int(tag))

    def fillImuQueue(self, timestamp_ns: int, acc_gyr):
        a = np.ascontiguousarray(acc_gyr, np.float64).reshape(6)
        self._lib.kvfe_stereo_sync_fill_imu(self._h, int(timestamp_ns), a.ctypes.data_as(C.POINTER(C.c_double)))

    def doCoarseImuCameraTemporalSync(self):
        self._lib.kvfe_stereo_sync_do_coarse_imu_camera_temporal_sync(self._h)

    def setImuTimeShift(self, imu_time_shift_s: float):
        self._lib.kvfe_stereo_sync_set_imu_time_shift(self._h, float(imu_time_shift_s))

    def shutdown(self):
        self._lib.kvfe_stereo_sync_shutdown(self._h)

    def getInputPacket(self):
        """one spin: a StereoImuSyncPacket, or None (see last_action)"""
        cap = 64
        while True:
            pk = abi.SyncPacket()
            stamps = np.empty(cap, np.int64)
            vals = np.empty((cap, 6), np.float64)
            r = self._lib.kvfe_stereo_sync_next(self._h, C.byref(pk), stamps.ctypes.data_as(C.POINTER(C.c_int64)),
                                                vals.ctypes.data_as(C.POINTER(C.c_double)), cap)
            if r == -1:
                cap = pk.n_imu
                continue
            self.last_action = r
            if r != abi.SYNC_PACKET:
                return None
            return StereoImuSyncPacket(pk.timestamp_ns, pk.left_tag, pk.right_tag, stamps[:pk.n_imu].copy(),
                                       vals[:pk.n_imu].T.copy())


class MonoDataProviderModule(StereoDataProviderModule):
    """MonoDataProviderModule (src/dataprovider/MonoDataProviderModule.cpp:30-118): left frames + IMU only; packets
    carry right_tag = -1"""
    MODE = 1   # KVFE_SYNC_MODE_MONO


class RgbdDataProviderModule(StereoDataProviderModule):
    """RgbdDataProviderModule (src/dataprovider/RgbdDataProviderModule.cpp:44-84): the second queue holds the depth
    frames (fillDepthFrameQueue); a frame whose depth image is missing is dropped but still ends the IMU interval"""
    MODE = 2   # KVFE_SYNC_MODE_RGBD

    def fillDepthFrameQueue(self, timestamp_ns: int, tag: int):
        self.fillRightFrameQueue(timestamp_ns, tag)


# ---------------------------------------------------------------------------------------------------------
# EuRoC dataset
# ---------------------------------------------------------------------------------------------------------
def parse_camera_csv(text: bytes) -> np.ndarray:
    """CameraImageLists::parseCamImgList: the frame timestamps of mav0/camN/data.csv"""
    lib = load()
    n = C.c_int32()
    st = lib.kvfe_euroc_parse_camera_csv(text, len(text), None, 0, C.byref(n))
    if st not in (0, -5):
        raise KvfeError(st, "kvfe_euroc_parse_camera_csv")
    ts = np.empty(n.value, np.int64)
    _check(lib.kvfe_euroc_parse_camera_csv(text, len(text), ts.ctypes.data_as(C.POINTER(C.c_int64)), n.value,
                                           C.byref(n)), "kvfe_euroc_parse_camera_csv")
    return ts


def parse_imu_csv(text: bytes):
    """EurocDataProvider::parseImuData: (timestamps int64[n], acc_gyr float64[n, 6], acceleration first)"""
    lib = load()
    n = C.c_int32()
    st = lib.kvfe_euroc_parse_imu_csv(text, len(text), None, None, 0, C.byref(n))
    if st not in (0, -5):
        raise KvfeError(st, "kvfe_euroc_parse_imu_csv")
    ts = np.empty(n.value, np.int64)
    ag = np.empty((n.value, 6), np.float64)
    _check(lib.kvfe_euroc_parse_imu_csv(text, len(text), ts.ctypes.data_as(C.POINTER(C.c_int64)),
                                        ag.ctypes.data_as(C.POINTER(C.c_double)), n.value, C.byref(n)),
           "kvfe_euroc_parse_imu_csv")
    return ts, ag


class EurocDataProvider:
    """EurocDataProvider: parses <dataset_path>/mav0/{cam0,cam1,imu0}/data.csv, then `spin()` sends all IMU data
    and the frame pairs initial_k .. final_k - 1 through the callbacks (sequential mode).  Callback signatures:
    imu_single_callback(timestamp, acc_gyr[6]); left/right_frame_callback(k, timestamp, image uint8[h, w])."""
    kLeftCamName, kRightCamName, kImuName = "cam0", "cam1", "imu0"

    def __init__(self, dataset_path: str, initial_k: int = 0, final_k: int = 1 << 30, equalize_image: bool = False,
                 ctx=None):
        self.dataset_path = dataset_path
        self.equalize_image, self.ctx = equalize_image, ctx
        self.imu_single_callback = self.left_frame_callback = self.right_frame_callback = None
        self._parse()
        self.initial_k = initial_k
        self.final_k = min(final_k, self.getNumImages())   # clipFinalFrame
        if not self.final_k > self.initial_k:
            raise ValueError("final_k must be larger than initial_k")
        self.current_k = initial_k
        self._imu_sent = False

    def _read(self, *parts):
        with open(os.path.join(self.dataset_path, "mav0", *parts), "rb") as f:
            return f.read()

    def _parse(self):
        self.imu_timestamps, self.imu_acc_gyr = parse_imu_csv(self._read(self.kImuName, "data.csv"))
        self.left_timestamps = parse_camera_csv(self._read(self.kLeftCamName, "data.csv"))
        self.right_timestamps = parse_camera_csv(self._read(self.kRightCamName, "data.csv"))

    def getNumImages(self) -> int:
        return int(len(self.left_timestamps))

    def timestampAtFrame(self, k: int) -> int:
        return int(self.left_timestamps[k])

    def getLeftImgName(self, k: int):
        return self._img_name(self.kLeftCamName, self.left_timestamps, k)

    def getRightImgName(self, k: int):
        return self._img_name(self.kRightCamName, self.right_timestamps, k)

    def _img_name(self, cam, stamps, k):
        if k < len(stamps):
            return os.path.join(self.dataset_path, "mav0", cam, "data", f"{int(stamps[k])}.png")
        return None

    def hasData(self) -> bool:
        return self.current_k < self.final_k

    def sendImuData(self):
        for t, ag in zip(self.imu_timestamps, self.imu_acc_gyr):
            self.imu_single_callback(int(t), ag)

    def spinOnce(self) -> bool:
        if self.current_k >= self.final_k:
            return False
        k = self.current_k
        t = self.timestampAtFrame(k)
        ln, rn = self.getLeftImgName(k), self.getRightImgName(k)
        if ln and rn and os.path.exists(ln) and os.path.exists(rn):
            self.left_frame_callback(k, t, ReadAndConvertToGrayScale(ln, self.equalize_image, self.ctx))
            self.right_frame_callback(k, t, ReadAndConvertToGrayScale(rn, self.equalize_image, self.ctx))
        # else: "Missing left/right stereo pair, proceeding to the next one."
        self.current_k += 1
        return True

    def spin(self) -> bool:
        if not self._imu_sent:
            if self.imu_single_callback:
                self.sendImuData()
            self._imu_sent = True
        while self.spinOnce():
            pass
        return False
