"""CPU restatement of the reference's input side -- TEST INFRASTRUCTURE ONLY: imported by tests/,
never by the product (kimera_vio_amd/) or the timed region of bench.py.

Pure-Python statement of
  * utils::ThreadsafeImuBuffer            (src/utils/ThreadsafeImuBuffer.cpp:48-234, ThreadsafeImuBuffer-inl.h:52-70,
                                           ThreadsafeTemporalBuffer-inl.h:198-262, 343-362)
  * DataProviderModule / MonoDataProviderModule / StereoDataProviderModule in sequential mode
                                          (src/dataprovider/DataProviderModule.cpp:80-181, MonoDataProviderModule.cpp:
                                           44-118, StereoDataProviderModule.cpp:35-91, pipeline/QueueSynchronizer.h:79-162)
Pinned on tests/testThreadsafeImuBuffer.cpp and tests/testStereoProvider.cpp (tests/test_input_side.py).
"""
import bisect

K_AVAILABLE, K_NOT_YET, K_NEVER, K_SHUTDOWN, K_TOO_FEW = range(5)


def linear_interpolate(t0, y0, t1, y1, t):
    # *y = t0 == t1 ? y0 : y0 + (y1 - y0) * double(t - t0) / double(t1 - t0)
    if t0 == t1:
        return list(y0)
    return [a + (b - a) * float(t - t0) / float(t1 - t0) for a, b in zip(y0, y1)]


class ImuBuffer:
    def __init__(self, buffer_length_ns=-1):
        self.t, self.v = [], []
        self.buffer_length_ns = buffer_length_ns
        self.shutdown = False

    def add(self, t, acc_gyr):
        if self.t and t <= self.t[-1]:      # "Enforce strict time-wise ordering."
            return
        self.t.append(t)
        self.v.append(list(acc_gyr))
        if self.buffer_length_ns > 0:       # removeOutdatedItems
            thr = self.t[-1] - self.buffer_length_ns
            if self.t[0] < thr:
                k = bisect.bisect_left(self.t, thr)
                del self.t[:k], self.v[:k]

    def available(self, t_from, t_to):
        if self.shutdown:
            return K_SHUTDOWN
        if not self.t or self.t[-1] < t_to:
            return K_NOT_YET
        if t_from < self.t[0]:
            return K_NEVER
        return K_AVAILABLE

    def between(self, t_from, t_to, lower=False):
        q = self.available(t_from, t_to)
        if q != K_AVAILABLE:
            return q, [], []
        ts, vs = [], []
        for t, v in zip(self.t, self.v):
            if t < t_from or t >= t_to or (t == t_from and not lower):
                continue
            ts.append(t)
            vs.append(v)
        if not ts:
            return K_TOO_FEW, [], []
        return q, ts, vs

    def interpolate(self, t):
        k = bisect.bisect_left(self.t, t)
        if k < len(self.t) and self.t[k] == t:
            return list(self.v[k])
        return linear_interpolate(self.t[k - 1], self.v[k - 1], self.t[k], self.v[k], t)

    def interpolated_upper_border(self, t_from, t_to):
        q, ts, vs = self.between(t_from, t_to, True)
        if q != K_AVAILABLE:
            return q, [], []
        return q, ts + [t_to], vs + [self.interpolate(t_to)]

    def interpolated_borders(self, t_from, t_to):
        q, ts, vs = self.between(t_from, t_to, False)
        if q != K_AVAILABLE:
            return q, [], []
        return q, [t_from] + ts + [t_to], [self.interpolate(t_from)] + vs + [self.interpolate(t_to)]


class StereoProvider:
    """one spin() of the reference's sequential mode == one getInputPacket()"""
    (PACKET, EMPTY, WAIT_IMU, DROP_OUT_OF_ORDER, DROP_NO_IMU, DROP_FIRST_FRAME, DROP_IMU_NEVER, DROP_IMU_TOO_FEW,
     DROP_NO_RIGHT, SHUTDOWN) = range(10)

    def __init__(self, mode=0):        # 0 stereo, 1 mono, 2 RGBD (second queue = depth frames)
        self.mode = mode
        self.left, self.right = [], []
        self.imu = ImuBuffer(-1)
        self.cached = None
        self.last = 0                      # InvalidTimestamp
        self.coarse = False
        self.correction = 0
        self.shift = 0

    def spin(self):
        if self.cached is not None:
            lf, self.cached = self.cached, None
        else:
            if not self.left:
                return self.EMPTY, None
            lf = self.left.pop(0)
        t = lf[0]
        if self.last >= t:
            return self.DROP_OUT_OF_ORDER, None
        if not self.imu.t:
            return self.DROP_NO_IMU, None
        if self.last == 0:
            self.last = t
            return self.DROP_FIRST_FRAME, None
        if self.coarse:
            self.correction = self.imu.t[-1] - t
            self.coarse = False
        off = self.correction + self.shift
        q, ts, vs = self.imu.interpolated_borders(self.last + off, t + off)
        if q == K_NOT_YET:
            self.cached = lf
            return self.WAIT_IMU, None
        if q == K_NEVER:
            self.last = t
            return self.DROP_IMU_NEVER, None
        if q != K_AVAILABLE:
            return self.DROP_IMU_TOO_FEW, None
        ts = [x - off for x in ts]
        if self.mode != 0:                 # getMonoImuSyncPacket(cache_timestamp = true)
            self.last = t
        if self.mode == 1:
            return self.PACKET, (t, lf[1], -1, ts, vs)
        rf = None
        while self.right:                  # syncQueue
            cur = self.right[0]
            if cur[0] > t:
                break
            self.right.pop(0)
            if cur[0] == t:
                rf = cur
                break
        if rf is None:
            return self.DROP_NO_RIGHT, None
        self.last = t
        return self.PACKET, (t, lf[1], rf[1], ts, vs)


def _mul3(a, b):
    return [a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j] for i in range(3) for j in range(3)]


def preintegrate_rotation(stamps, gyro, gyro_bias=(0.0, 0.0, 0.0), deltaRij=None):
    """ImuFrontend::preintegrateImuMeasurements (src/imu-frontend/ImuFrontend.cpp:158-173), rotation only:
    deltaRij <- deltaRij . so3::ExpmapFunctor((gyro_i - bias) dt_i).expmap() for i < n - 1, dt = ns / 1e9 (gtsam 4.2
    on-manifold preintegration, NavState::update).  gyro: n rows of 3.  Returns the 9 row-major entries."""
    import math
    R = [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0] if deltaRij is None else [float(x) for x in deltaRij]
    for i in range(len(stamps) - 1):
        dt = float(stamps[i + 1] - stamps[i]) / 1e9
        w = [(float(gyro[i][k]) - float(gyro_bias[k])) * dt for k in range(3)]
        theta2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2]
        W = [0.0, -w[2], w[1], w[2], 0.0, -w[0], -w[1], w[0], 0.0]
        if theta2 <= 2.220446049250313e-16:
            E = list(W)
        else:
            theta = math.sqrt(theta2)
            s, s2 = math.sin(theta), math.sin(theta / 2.0)
            omc = 2.0 * s2 * s2
            K = [x / theta for x in W]
            KK = _mul3(K, K)
            E = [s * K[k] + omc * KK[k] for k in range(9)]
        E[0] += 1.0
        E[4] += 1.0
        E[8] += 1.0
        R = _mul3(R, E)
    return R
