"""Input side of the front-end: PNG decoding, utils::ThreadsafeImuBuffer, the left / right / IMU
synchronisation of StereoDataProviderModule and the EuRoC index files -- host code of libkvfe, no GPU.

Pins:  tests/testThreadsafeImuBuffer.cpp (every case), tests/testStereoProvider.cpp (all 15 cases) of the reference,
restated below with the reference's numbers; PNG decoding against PIL (an independent decoder) on the committed
frames and on files built here with every filter type, bit depth, colour type and Adam7.
"""
import io
import os
import struct
import sys
import zlib

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from kimera_vio_amd import _abi as abi  # noqa: E402
from kimera_vio_amd import dataprovider as dp  # noqa: E402
from kimera_vio_amd.lib import KvfeError, load  # noqa: E402
from oracle import input_side as ora  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
PIL = pytest.importorskip("PIL.Image")


# ---------------------------------------------------------------------------------------------------------------
# PNG
# ---------------------------------------------------------------------------------------------------------------
def _chunk(tag, body):
    return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)


def _filter_row(ft, row, prev, bpp):
    out = bytearray(len(row))
    for i, v in enumerate(row):
        a = row[i - bpp] if i >= bpp else 0
        b = prev[i] if prev is not None else 0
        c = prev[i - bpp] if (prev is not None and i >= bpp) else 0
        if ft == 0:
            p = 0
        elif ft == 1:
            p = a
        elif ft == 2:
            p = b
        elif ft == 3:
            p = (a + b) >> 1
        else:
            pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
            p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
        out[i] = (v - p) & 255
    return bytes(out)


def make_png(rows, width, depth, color, filters=(0,), interlace=False, palette=None, idat_split=1, extra=b""):
    """rows: list of packed scanlines (bytes) of the full image; returns the PNG file.  With interlace the
    sub-images are cut out of `rows` sample by sample (only depths >= 8)."""
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color]
    bits = depth * channels
    bpp = max(1, bits // 8)
    height = len(rows)
    raw = bytearray()

    def emit(sub_rows, k0):
        prev = None
        for k, r in enumerate(sub_rows):
            ft = filters[(k0 + k) % len(filters)]
            raw.append(ft)
            raw.extend(_filter_row(ft, r, prev, bpp))
            prev = r

    if not interlace:
        emit(rows, 0)
    else:
        assert depth >= 8
        x0s, y0s = (0, 4, 0, 2, 0, 1, 0), (0, 0, 4, 0, 2, 0, 1)
        dxs, dys = (8, 8, 4, 4, 2, 2, 1), (8, 8, 8, 4, 4, 2, 2)
        for p in range(7):
            sub = []
            for y in range(y0s[p], height, dys[p]):
                r = b"".join(rows[y][x * bpp:(x + 1) * bpp] for x in range(x0s[p], width, dxs[p]))
                if r:
                    sub.append(r)
            emit(sub, p)
    z = zlib.compress(bytes(raw), 6)
    parts = [z[i * len(z) // idat_split:(i + 1) * len(z) // idat_split] for i in range(idat_split)]
    ihdr = struct.pack(">IIBBBBB", width, height, depth, color, 0, 0, 1 if interlace else 0)
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", ihdr)
    if palette is not None:
        out += _chunk(b"PLTE", bytes(palette))
    out += extra
    for part in parts:
        out += _chunk(b"IDAT", part)
    return out + _chunk(b"IEND", b"")


def cv_gray(rgb):
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return ((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15).astype(np.uint8)


@pytest.mark.parametrize("name", ["left_img_0.png", "right_img_0.png", "left_fisheye_img_0.png", "chessboard.png",
                                  "sidebyside_ref_img_0_gray.png"])
def test_png_golden_frames_equal_pil(name):
    data = open(os.path.join(GOLDEN, name), "rb").read()
    img = PIL.open(io.BytesIO(data))
    got = dp.decode_png_gray(data)
    w, h, c = dp.png_info(data)
    assert (h, w) == got.shape == (img.size[1], img.size[0])
    if img.mode in ("L", "1", "I;16", "I"):
        assert c == 1
        want = np.asarray(img.convert("L")) if img.mode == "L" else None
    else:
        assert c == 3
        want = cv_gray(np.asarray(img.convert("RGB")))
    if want is not None:
        assert np.array_equal(got, want)


def test_png_every_filter_type_and_split_idat():
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    img[10:20] = np.arange(53, dtype=np.uint8)   # smooth rows: the predictors matter
    rows = [img[y].tobytes() for y in range(img.shape[0])]
    for filters in [(0,), (1,), (2,), (3,), (4,), (0, 1, 2, 3, 4), (4, 3, 2, 1)]:
        data = make_png(rows, 53, 8, 0, filters, idat_split=3, extra=_chunk(b"tEXt", b"Comment\x00kvfe"))
        assert np.array_equal(dp.decode_png_gray(data), img), filters
        assert np.array_equal(np.asarray(PIL.open(io.BytesIO(data))), img)   # the encoder above is sound


def test_png_runs_of_average_and_paeth_rows():
    """round 5: runs of Average / Paeth rows of 8-bit grey are undone four (or two) rows at a time as a wavefront
    (host_input.cpp unfilter_wave): every run length 1..9 between rows of the other types, both orders of the two types
    inside a run, widths from below the wavefront's minimum to a full EuRoC row, against PIL"""
    rng = np.random.default_rng(17)
    for w in (1, 3, 7, 8, 9, 31, 752):
        seq = []
        for run in (1, 2, 3, 4, 5, 6, 7, 8, 9):
            seq += [int(t) for t in rng.integers(3, 5, run)] + [int(rng.integers(0, 3))]
        seq += [4] * 11 + [3] * 5 + [4, 3] * 6        # long runs, the image's last rows inside one
        h = len(seq)
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        img[h // 3: h // 2] = (np.arange(w) * 3 % 256).astype(np.uint8)   # smooth rows: the predictors matter
        rows = [img[y].tobytes() for y in range(h)]
        data = make_png(rows, w, 8, 0, tuple(seq), idat_split=2)
        assert np.array_equal(np.asarray(PIL.open(io.BytesIO(data))), img)   # the encoder above is sound
        assert np.array_equal(dp.decode_png_gray(data), img), w


@pytest.mark.parametrize("interlace", [False, True])
def test_png_colour_types_and_depths(interlace):
    rng = np.random.default_rng(11)
    h, w = 19, 23
    # RGB / RGBA 8-bit: cv::cvtColor(BGR2GRAY) weights
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    data = make_png([rgb[y].tobytes() for y in range(h)], w, 8, 2, (0, 4, 1), interlace)
    assert np.array_equal(np.asarray(PIL.open(io.BytesIO(data)).convert("RGB")), rgb)
    assert np.array_equal(dp.decode_png_gray(data), cv_gray(rgb))
    rgba = np.concatenate([rgb, rng.integers(0, 256, (h, w, 1), dtype=np.uint8)], axis=2)
    data = make_png([rgba[y].tobytes() for y in range(h)], w, 8, 6, (3, 2), interlace)
    assert np.array_equal(dp.decode_png_gray(data), cv_gray(rgb))   # alpha dropped, not blended
    assert dp.png_info(data) == (w, h, 3)
    # grey + alpha: the grey channel
    ga = rng.integers(0, 256, (h, w, 2), dtype=np.uint8)
    data = make_png([ga[y].tobytes() for y in range(h)], w, 8, 4, (1,), interlace)
    assert np.array_equal(dp.decode_png_gray(data), ga[..., 0])
    # 16-bit grey / RGB: the high byte (png_set_strip_16)
    g16 = rng.integers(0, 65536, (h, w), dtype=np.uint16)
    data = make_png([g16[y].astype(">u2").tobytes() for y in range(h)], w, 16, 0, (4, 0), interlace)
    assert np.array_equal(dp.decode_png_gray(data), (g16 >> 8).astype(np.uint8))
    rgb16 = rng.integers(0, 65536, (h, w, 3), dtype=np.uint16)
    data = make_png([rgb16[y].astype(">u2").tobytes() for y in range(h)], w, 16, 2, (2, 3), interlace)
    assert np.array_equal(dp.decode_png_gray(data), cv_gray((rgb16 >> 8).astype(np.uint8)))
    # palette, 8-bit indices
    pal = rng.integers(0, 256, (200, 3), dtype=np.uint8)
    idx = rng.integers(0, 200, (h, w), dtype=np.uint8)
    data = make_png([idx[y].tobytes() for y in range(h)], w, 8, 3, (0, 1), interlace, palette=pal.tobytes())
    assert np.array_equal(np.asarray(PIL.open(io.BytesIO(data)).convert("RGB")), pal[idx])
    assert np.array_equal(dp.decode_png_gray(data), cv_gray(pal[idx]))


@pytest.mark.parametrize("depth", [1, 2, 4])
def test_png_packed_grey_and_palette(depth):
    rng = np.random.default_rng(depth)
    h, w = 9, 21   # width not a multiple of the samples per byte
    vals = rng.integers(0, 1 << depth, (h, w), dtype=np.uint8)
    per = 8 // depth

    def pack(row):
        out = bytearray((w + per - 1) // per)
        for x, v in enumerate(row):
            out[x // per] |= int(v) << ((per - 1 - x % per) * depth)
        return bytes(out)

    rows = [pack(vals[y]) for y in range(h)]
    data = make_png(rows, w, depth, 0, (0, 2, 1))
    want = (vals.astype(np.int64) * 255 // ((1 << depth) - 1)).astype(np.uint8)
    assert np.array_equal(dp.decode_png_gray(data), want)
    pil = PIL.open(io.BytesIO(data))
    assert np.array_equal(np.asarray(pil.convert("L")), want)
    pal = rng.integers(0, 256, (1 << depth, 3), dtype=np.uint8)
    data = make_png(rows, w, depth, 3, (0,), palette=pal.tobytes())
    assert np.array_equal(dp.decode_png_gray(data), cv_gray(pal[vals]))


def test_png_pil_written_files():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (60, 80), dtype=np.uint8)
    img[:, 10:50] = (np.arange(40) * 3)[None, :]
    for kw in ({}, {"optimize": True}, {"compress_level": 1}, {"compress_level": 9}):
        b = io.BytesIO()
        PIL.fromarray(img).save(b, "PNG", **kw)
        assert np.array_equal(dp.decode_png_gray(b.getvalue()), img)
    rgb = rng.integers(0, 256, (31, 45, 3), dtype=np.uint8)
    b = io.BytesIO()
    PIL.fromarray(rgb).save(b, "PNG")
    assert np.array_equal(dp.decode_png_gray(b.getvalue()), cv_gray(rgb))


def test_png_errors():
    img = np.arange(64, dtype=np.uint8).reshape(8, 8)
    data = make_png([img[y].tobytes() for y in range(8)], 8, 8, 0)
    assert np.array_equal(dp.decode_png_gray(data), img)
    with pytest.raises(KvfeError):
        dp.decode_png_gray(b"not a png at all, but long enough to hold a header ........")
    with pytest.raises(KvfeError):
        dp.decode_png_gray(data[:40])                      # truncated
    bad = bytearray(data)
    bad[-20] ^= 0x40                                        # a bit inside the IDAT payload: CRC mismatch
    with pytest.raises(KvfeError):
        dp.decode_png_gray(bytes(bad))
    out = np.empty((8, 9), np.uint8)                        # a destination of another size: refused by the library
    assert load().kvfe_png_decode_gray(data, len(data), out.ctypes.data, out.strides[0], 9, 8) == -1
    # a corrupt deflate stream with valid CRCs
    raw = zlib.compress(b"\x00" + bytes(8), 6)[:-3]
    broken = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", 8, 8, 8, 0, 0, 0, 0)) + \
        _chunk(b"IDAT", raw) + _chunk(b"IEND", b"")
    with pytest.raises(KvfeError):
        dp.decode_png_gray(broken)


def test_png_batch_threads_and_strided_destination():
    rng = np.random.default_rng(9)
    imgs = rng.integers(0, 256, (12, 48, 64), dtype=np.uint8)
    files = []
    for i in range(12):
        b = io.BytesIO()
        PIL.fromarray(imgs[i]).save(b, "PNG")
        files.append(b.getvalue())
    for threads in (0, 1, 4):
        out = np.zeros((12, 48, 64), np.uint8)
        dp.decode_png_gray_batch(files, out, threads)
        assert np.array_equal(out, imgs)
    pitched = np.zeros((12, 48, 80), np.uint8)            # rows padded to 80 bytes, e.g. a pitched staging slot
    dp.decode_png_gray_batch(files, pitched[:, :, :64], 3)
    assert np.array_equal(pitched[:, :, :64], imgs) and not pitched[:, :, 64:].any()
    files[5] = files[5][:100]
    with pytest.raises(KvfeError) as e:
        dp.decode_png_gray_batch(files, np.zeros((12, 48, 64), np.uint8), 4)
    assert "[5]" in str(e.value)


# ---------------------------------------------------------------------------------------------------------------
# utils::ThreadsafeImuBuffer -- tests/testThreadsafeImuBuffer.cpp
# ---------------------------------------------------------------------------------------------------------------
B = dp.ThreadsafeImuBuffer


def _filled():
    b = B(-1)
    for t in (10, 15, 20, 25, 30, 40, 50):
        b.addMeasurement(t, np.full(6, float(t)))
    return b


def _expect(res, code, stamps=None):
    q, ts, vs = res
    assert q == code
    if code != B.kDataAvailable:
        assert ts.size == 0 and vs.size == 0
    if stamps is not None:
        assert ts.tolist() == list(stamps)
        assert vs.shape == (6, len(stamps))
        assert vs[0].tolist() == [float(s) for s in stamps]   # the test signal is value == time
        assert (vs == vs[0]).all()


def test_imu_buffer_pop_from_empty_buffer():   # :47-88
    b = B(-1)
    for res in (b.getImuDataBtwTimestamps(50, 100), b.getImuDataBtwTimestamps(50, 100, True),
                b.getImuDataInterpolatedUpperBorder(50, 100), b.getImuDataInterpolatedBorders(50, 100)):
        _expect(res, B.kDataNotYetAvailable)


def test_imu_buffer_linear_interpolate():   # :90-95
    y = B.linearInterpolate(10, np.full(6, 10.0), 20, np.full(6, 50.0), 15)
    assert y.tolist() == [30.0] * 6
    assert B.linearInterpolate(7, np.arange(6.0), 7, np.ones(6), 7).tolist() == list(np.arange(6.0))   # t0 == t1: y0


def test_imu_buffer_get_data_between_timestamps():   # :97-192
    b = _filled()
    _expect(b.getImuDataBtwTimestamps(20, 30), B.kDataAvailable, [25])
    _expect(b.getImuDataBtwTimestamps(20, 30, True), B.kDataAvailable, [20, 25])
    _expect(b.getImuDataBtwTimestamps(19, 31), B.kDataAvailable, [20, 25, 30])
    _expect(b.getImuDataBtwTimestamps(19, 31, True), B.kDataAvailable, [20, 25, 30])
    _expect(b.getImuDataBtwTimestamps(40, 51), B.kDataNotYetAvailable)
    _expect(b.getImuDataBtwTimestamps(60, 61), B.kDataNotYetAvailable)
    _expect(b.getImuDataBtwTimestamps(-1, 20), B.kDataNeverAvailable)
    _expect(b.getImuDataBtwTimestamps(-20, -10), B.kDataNeverAvailable)
    _expect(b.getImuDataBtwTimestamps(21, 24), B.kTooFewMeasurementsAvailable)
    _expect(b.getImuDataBtwTimestamps(21, 24, True), B.kTooFewMeasurementsAvailable)
    _expect(b.getImuDataBtwTimestamps(20, 25), B.kTooFewMeasurementsAvailable)
    _expect(b.getImuDataBtwTimestamps(20, 25, True), B.kDataAvailable, [20])


def test_imu_buffer_interpolated_borders():   # :194-279
    b = _filled()
    _expect(b.getImuDataInterpolatedBorders(20, 30), B.kDataAvailable, [20, 25, 30])
    _expect(b.getImuDataInterpolatedBorders(20, 40), B.kDataAvailable, [20, 25, 30, 40])
    _expect(b.getImuDataInterpolatedBorders(19, 21), B.kDataAvailable, [19, 20, 21])
    _expect(b.getImuDataInterpolatedBorders(40, 51), B.kDataNotYetAvailable)
    _expect(b.getImuDataInterpolatedBorders(60, 61), B.kDataNotYetAvailable)
    _expect(b.getImuDataInterpolatedBorders(-1, 20), B.kDataNeverAvailable)
    _expect(b.getImuDataInterpolatedBorders(-20, -10), B.kDataNeverAvailable)
    _expect(b.getImuDataInterpolatedBorders(21, 29), B.kDataAvailable, [21, 25, 29])


def test_imu_buffer_interpolated_upper_border():   # :281-360
    b = _filled()
    _expect(b.getImuDataInterpolatedUpperBorder(20, 40), B.kDataAvailable, [20, 25, 30, 40])
    _expect(b.getImuDataInterpolatedUpperBorder(19, 21), B.kDataAvailable, [20, 21])
    _expect(b.getImuDataInterpolatedUpperBorder(40, 51), B.kDataNotYetAvailable)
    _expect(b.getImuDataInterpolatedUpperBorder(60, 61), B.kDataNotYetAvailable)
    _expect(b.getImuDataInterpolatedUpperBorder(9, 20), B.kDataNeverAvailable)
    _expect(b.getImuDataInterpolatedUpperBorder(-20, -10), B.kDataNeverAvailable)
    _expect(b.getImuDataInterpolatedUpperBorder(21, 24), B.kTooFewMeasurementsAvailable)
    _expect(b.getImuDataInterpolatedUpperBorder(21, 29), B.kDataAvailable, [25, 29])


def test_imu_buffer_ordering_length_shutdown_and_growth():
    b = B(-1)
    for t in (2, 4, 3, 5, 5, 5):           # not strictly increasing: ignored (ThreadsafeImuBuffer-inl.h:56-63)
        b.addMeasurement(t, np.full(6, float(t)))
    assert b.size() == 3
    assert b.getImuDataBtwTimestamps(2, 5, True)[1].tolist() == [2, 4]
    lim = B(10)                            # ThreadsafeTemporalBuffer::removeOutdatedItems
    for t in range(0, 50, 3):
        lim.addMeasurement(t, np.zeros(6))
    assert lim.size() == 4 and lim.getImuDataBtwTimestamps(39, 48, True)[1].tolist() == [39, 42, 45]
    big = B(-1)                            # more samples than the wrapper's first guess: the count comes back
    for t in range(1000):
        big.addMeasurement(t, np.full(6, 0.5 * t))
    q, ts, vs = big.getImuDataInterpolatedBorders(10, 900)
    assert q == B.kDataAvailable and ts.tolist() == list(range(10, 901)) and np.array_equal(vs[3], 0.5 * ts)
    _expect(big.getImuDataInterpolatedBorders(20, 20), B.kDataNeverAvailable)    # from >= to: CHECK_LT upstream
    _expect(big.getImuDataBtwTimestamps(30, 20), B.kDataNeverAvailable)
    big.shutdown()
    _expect(big.getImuDataInterpolatedBorders(10, 900), B.kQueueShutdown)


def test_imu_buffer_equals_oracle_on_random_streams():
    rng = np.random.default_rng(21)
    for trial in range(20):
        b, o = B(-1), ora.ImuBuffer(-1)
        t = 0
        for _ in range(60):
            t += int(rng.integers(-2, 9))
            v = rng.normal(size=6)
            b.addMeasurement(t, v)
            o.add(t, v)
        assert b.size() == len(o.t)
        for _ in range(60):
            a = int(rng.integers(o.t[0] - 5, o.t[-1] + 5))
            c = a + int(rng.integers(1, 40))
            for fn, ofn in ((b.getImuDataInterpolatedBorders, o.interpolated_borders),
                            (b.getImuDataInterpolatedUpperBorder, o.interpolated_upper_border),
                            (lambda x, y: b.getImuDataBtwTimestamps(x, y, True), lambda x, y: o.between(x, y, True)),
                            (b.getImuDataBtwTimestamps, o.between)):
                q, ts, vs = fn(a, c)
                oq, ots, ovs = ofn(a, c)
                assert q == oq and ts.tolist() == ots
                if ots:
                    assert np.array_equal(vs.T, np.asarray(ovs))   # same IEEE operations: bit-exact


# ---------------------------------------------------------------------------------------------------------------
# StereoDataProviderModule -- tests/testStereoProvider.cpp (sequential mode: one spin() = one getInputPacket())
# ---------------------------------------------------------------------------------------------------------------
class Provider:
    def __init__(self):
        self.m = dp.StereoDataProviderModule(-1)
        self.last_id = 0

    def addImu(self, t):
        self.m.fillImuQueue(t, np.zeros(6))

    def addFrame(self, t):
        self.m.fillLeftFrameQueue(t, self.last_id)
        self.m.fillRightFrameQueue(t, self.last_id)
        self.last_id += 1

    def addLeftFrame(self, t):
        self.m.fillLeftFrameQueue(t, self.last_id)

    def spin(self):
        return self.m.getInputPacket()


def test_stereo_provider_basic_sequential_case():   # :99-129
    p = Provider()
    p.addImu(10)
    p.addFrame(11)
    assert p.spin() is None and p.m.last_action == abi.SYNC_DROP_FIRST_FRAME
    for t in (12, 13, 14):
        p.addImu(t)
    p.addFrame(17)
    p.addImu(18)
    r = p.spin()
    assert r.timestamp == 17 and r.imu_stamps.tolist() == [11, 12, 13, 14, 17]
    assert r.left_tag == 1 and r.right_tag == 1     # getStereoFrame().id_ == 1
    assert r.imu_accgyrs.shape == (6, 5)


def test_stereo_provider_drop_frames_older_than_imu():   # :131-164
    p = Provider()
    p.addImu(10)
    p.addFrame(11)
    assert p.spin() is None
    p.addImu(16)
    for t in range(12, 16):
        p.addFrame(t)
        assert p.spin() is None and p.m.last_action == abi.SYNC_DROP_IMU_TOO_FEW
    p.addFrame(17)
    p.addImu(18)
    r = p.spin()
    assert r.timestamp == 17 and r.imu_stamps.tolist() == [11, 16, 17]


def test_stereo_provider_many_imu():   # :166-199
    p = Provider()
    t_curr, num_imu = 10, 5
    for t in range(num_imu):
        p.addImu(t_curr + t)
    t_curr += num_imu
    p.addFrame(t_curr)
    for _ in range(10):
        for t in range(num_imu):
            p.addImu(t_curr + t)
        t_curr += num_imu
        p.addFrame(t_curr)
    p.addImu(t_curr + 1)
    assert p.spin() is None
    for i in range(10):
        r = p.spin()
        assert r.timestamp == 20 + i * num_imu and r.imu_stamps.size == num_imu + 1


def test_stereo_provider_image_before_imu():   # :201-226, :228-254
    for delayed in (False, True):
        p = Provider()
        script = [("f", 10), ("spin", None), ("i", 11), ("f", 12), ("spin", None), ("i", 13), ("f", 14), ("i", 15)]
        outs = []
        for kind, t in script:
            if kind == "f":
                p.addFrame(t)
            elif kind == "i":
                p.addImu(t)
            elif not delayed:
                outs.append(p.spin())
        if delayed:
            outs = [p.spin(), p.spin()]
        assert outs == [None, None]
        r = p.spin()
        assert r.timestamp == 14 and r.imu_stamps.tolist() == [12, 13, 14]


def test_stereo_provider_valid_and_invalid_imu_sequence():   # :256-295
    p = Provider()
    p.addImu(0)
    p.addFrame(1)
    assert p.spin() is None
    for t in (2, 3, 4):
        p.addImu(t)
    p.addFrame(3)
    r = p.spin()
    assert r.timestamp == 3 and r.imu_stamps.size == 3 and r.imu_accgyrs.shape == (6, 3)
    p = Provider()
    p.addImu(10)
    p.addFrame(1)
    assert p.spin() is None
    for t in (11, 12, 13):
        p.addImu(t)
    p.addFrame(3)
    assert p.spin() is None and p.m.last_action == abi.SYNC_DROP_IMU_NEVER


def test_stereo_provider_partial_imu_sequence():   # :297-325
    p = Provider()
    p.addImu(0)
    p.addFrame(1)
    assert p.spin() is None
    for t in (2, 3, 4):
        p.addImu(t)
    p.addFrame(5)
    assert p.spin() is None and p.m.last_action == abi.SYNC_WAIT_IMU
    assert p.spin() is None and p.m.last_action == abi.SYNC_WAIT_IMU   # (earlier versions would loop forever here)
    p.addImu(5)
    r = p.spin()
    assert r.imu_stamps.tolist() == [1, 2, 3, 4, 5]


def test_stereo_provider_out_of_order_sequences():   # :327-468
    p = Provider()
    p.addImu(0)
    p.addFrame(1)
    assert p.spin() is None
    for t in (2, 4, 3, 5, 5, 5, 5):
        p.addImu(t)
    p.addFrame(5)
    r = p.spin()
    assert r.timestamp == 5 and r.imu_stamps.tolist() == [1, 2, 4, 5]

    p = Provider()                                    # testOutOfOrderManyImageSequence (a superset of :356-383)
    p.addImu(0)
    p.addFrame(3)
    assert p.spin() is None
    for t in (2, 3, 4, 5):
        p.addImu(t)
    p.addFrame(2)
    p.addFrame(5)
    p.addImu(6)
    p.addFrame(7)
    p.addImu(8)
    p.addFrame(9)
    p.addImu(10)
    assert p.spin() is None and p.m.last_action == abi.SYNC_DROP_OUT_OF_ORDER
    assert [p.spin().imu_stamps.tolist() for _ in range(3)] == [[3, 4, 5], [5, 6, 7], [7, 8, 9]]

    p = Provider()                                    # testOutOfOrderImuAndImageSequence
    p.addImu(0)
    p.addFrame(3)
    assert p.spin() is None
    for t in (2, 4, 3, 5, 5, 5):
        p.addImu(t)
    p.addFrame(2)
    p.addImu(5)
    p.addImu(5)
    p.addFrame(5)
    assert p.spin() is None
    r = p.spin()
    assert r.timestamp == 5 and r.imu_stamps.tolist() == [3, 4, 5]


def test_stereo_provider_coarse_correction_and_manual_time_shift():   # :470-524
    for mode in ("coarse", "shift"):
        p = Provider()
        if mode == "coarse":
            p.m.doCoarseImuCameraTemporalSync()
        else:
            p.m.setImuTimeShift(10.0e-9)
        p.addImu(10)
        p.addFrame(1)
        assert p.spin() is None
        for t in (11, 12, 13):
            p.addImu(t)
        p.addFrame(3)
        r = p.spin()
        assert r.timestamp == 3 and r.imu_stamps.tolist() == [1, 2, 3]


def test_stereo_provider_drop_right_frame():   # :526-560
    p = Provider()
    p.addImu(0)
    p.addFrame(1)
    assert p.spin() is None
    for t in (2, 3, 4):
        p.addImu(t)
    p.addLeftFrame(5)
    p.addImu(6)
    p.addImu(7)
    p.addFrame(8)
    p.addImu(9)
    assert p.spin() is None and p.m.last_action == abi.SYNC_DROP_NO_RIGHT
    r = p.spin()
    assert r.timestamp == 8 and r.imu_stamps.tolist() == [1, 2, 3, 4, 6, 7, 8]
    assert p.spin() is None and p.m.last_action == abi.SYNC_EMPTY


def test_stereo_provider_equals_oracle_on_random_traffic():
    rng = np.random.default_rng(77)
    for trial in range(30):
        p, o = dp.StereoDataProviderModule(-1), ora.StereoProvider()
        if trial % 5 == 1:
            p.doCoarseImuCameraTemporalSync()
            o.coarse = True
        if trial % 5 == 2:
            p.setImuTimeShift(3e-9)
            o.shift = 3
        t_imu, t_cam, tag = int(rng.integers(1, 5)), int(rng.integers(1, 9)), 0
        for _ in range(300):
            ev = rng.integers(0, 10)
            if ev < 5:
                t_imu += int(rng.integers(-1, 4))
                v = rng.normal(size=6)
                p.fillImuQueue(t_imu, v)
                o.imu.add(t_imu, v)
            elif ev < 8:
                t_cam += int(rng.integers(-2, 7))
                if t_cam < 1:
                    t_cam = 1
                drop_right = rng.integers(0, 12) == 0
                p.fillLeftFrameQueue(t_cam, tag)
                o.left.append((t_cam, tag))
                if not drop_right:
                    p.fillRightFrameQueue(t_cam, tag)
                    o.right.append((t_cam, tag))
                tag += 1
            else:
                r = p.getInputPacket()
                code, pk = o.spin()
                assert p.last_action == code
                if pk is None:
                    assert r is None
                else:
                    assert (r.timestamp, r.left_tag, r.right_tag) == pk[:3]
                    assert r.imu_stamps.tolist() == pk[3]
                    assert np.array_equal(r.imu_accgyrs.T, np.asarray(pk[4]))


# ---------------------------------------------------------------------------------------------------------------
# EuRoC index files and the data provider end to end
# ---------------------------------------------------------------------------------------------------------------
CAM_CSV = b"#timestamp [ns],filename\n1403715273262142976,1403715273262142976.png\n" \
          b"1403715273312143104,1403715273312143104.png\r\n1403715273362142976,1403715273362142976.png\n"
IMU_CSV = b"#timestamp [ns],w_RS_S_x [rad s^-1],w_RS_S_y [rad s^-1],w_RS_S_z [rad s^-1],a_RS_S_x [m s^-2]," \
          b"a_RS_S_y [m s^-2],a_RS_S_z [m s^-2]\n" \
          b"1403715273262142976,-0.099134701513277898,0.14730578886832138,0.02722713633111154," \
          b"8.1476917083333333,-0.37592158333333331,-2.4026292499999999\n" \
          b"1403715273267142912,-0.099134701513277898,0.14032447186034408,0.029321531433504733," \
          b"8.033280791666666,-0.40861041666666664,-2.4026292499999999\n" \
          b"1403715273272143104,-1e-3,2.5E+0,.5,9.81,0,-0\n"


def test_euroc_index_files():
    ts = dp.parse_camera_csv(CAM_CSV)
    assert ts.tolist() == [1403715273262142976, 1403715273312143104, 1403715273362142976]
    t, ag = dp.parse_imu_csv(IMU_CSV)
    assert t.tolist() == [1403715273262142976, 1403715273267142912, 1403715273272143104]
    rows = [r.split(",") for r in IMU_CSV.decode().splitlines()[1:]]
    want = np.array([[float(r[4]), float(r[5]), float(r[6]), float(r[1]), float(r[2]), float(r[3])] for r in rows])
    assert np.array_equal(ag, want)                       # "Acceleration first!", strtod == Python's float()
    with pytest.raises(KvfeError):                        # not in chronological order
        dp.parse_imu_csv(IMU_CSV + b"1403715273262142976,0,0,0,0,0,0\n")
    with pytest.raises(KvfeError):                        # a short row
        dp.parse_imu_csv(IMU_CSV + b"1403715273282143104,0,0,0\n")
    with pytest.raises(KvfeError):
        dp.parse_camera_csv(b"#header\nabc,abc.png\n")
    assert dp.parse_camera_csv(b"#header only\n").size == 0


def test_euroc_data_provider_feeds_the_synchroniser(tmp_path):
    """EurocDataProvider -> callbacks -> StereoDataProviderModule -> StereoImuSyncPackets, on a dataset written here
    (EuRoC layout, 20 Hz frames, 200 Hz IMU)"""
    rng = np.random.default_rng(1)
    t0, n_frames = 1_000_000_000, 6
    frames = rng.integers(0, 256, (2, n_frames, 24, 32), dtype=np.uint8)
    stamps = [t0 + k * 50_000_000 for k in range(n_frames)]
    for c, cam in enumerate(("cam0", "cam1")):
        d = tmp_path / "mav0" / cam / "data"
        d.mkdir(parents=True)
        lines = ["#timestamp [ns],filename"]
        for k, t in enumerate(stamps):
            if not (cam == "cam1" and k == 3):            # one right frame is missing on disk
                PIL.fromarray(frames[c, k]).save(d / f"{t}.png")
            lines.append(f"{t},{t}.png")
        (tmp_path / "mav0" / cam / "data.csv").write_text("\n".join(lines) + "\n")
    (tmp_path / "mav0" / "imu0").mkdir()
    imu_t = [t0 - 7_000_000 + i * 5_000_000 for i in range(70)]
    imu_rows = ["#timestamp [ns],w_x,w_y,w_z,a_x,a_y,a_z"] + \
               [f"{t},{0.001 * i},{-0.002 * i},0.5,{9.0 + 0.01 * i},0.25,-1.5" for i, t in enumerate(imu_t)]
    (tmp_path / "mav0" / "imu0" / "data.csv").write_text("\n".join(imu_rows) + "\n")

    prov = dp.EurocDataProvider(str(tmp_path), initial_k=0, final_k=100)
    assert prov.getNumImages() == n_frames and prov.final_k == n_frames
    sync = dp.StereoDataProviderModule(-1)
    images = {}

    def on_frame(side):
        def cb(k, t, img):
            images[(side, k)] = img
            (sync.fillLeftFrameQueue if side == 0 else sync.fillRightFrameQueue)(t, k)
        return cb

    prov.imu_single_callback = sync.fillImuQueue
    prov.left_frame_callback, prov.right_frame_callback = on_frame(0), on_frame(1)
    assert prov.spin() is False and not prov.hasData()
    assert sorted(k for s, k in images if s == 0) == [0, 1, 2, 4, 5]     # pair 3 was skipped as a pair
    for (s, k), img in images.items():
        assert np.array_equal(img, frames[s, k])
    packets = []
    while True:
        pk = sync.getInputPacket()
        if pk is None and sync.last_action == abi.SYNC_EMPTY:
            break
        if pk is not None:
            packets.append(pk)
    assert [p.left_tag for p in packets] == [1, 2, 4, 5] and all(p.left_tag == p.right_tag for p in packets)
    imu_t = np.asarray(imu_t)
    for p, prev_k in zip(packets, (0, 1, 2, 4)):
        a, b = stamps[prev_k], p.timestamp
        inner = imu_t[(imu_t > a) & (imu_t < b)]
        assert p.imu_stamps.tolist() == [a] + inner.tolist() + [b]
        assert np.all(p.imu_accgyrs[5] == 0.5) and np.all(p.imu_accgyrs[1] == 0.25)   # constant channels survive
        i_a = (a - imu_t[0]) / 5e6                                                  # linear channels interpolate
        assert abs(p.imu_accgyrs[0, 0] - (9.0 + 0.01 * i_a)) < 1e-12


MICRO_EUROC = "/root/reference/tests/data/MicroEurocDataset"


@pytest.mark.skipif(not os.path.isdir(MICRO_EUROC), reason="the reference checkout (build container only)")
def test_micro_euroc_dataset_of_the_reference():
    """the reference's own MicroEuroc dataset through EurocDataProvider + StereoDataProviderModule: every PNG equals
    PIL's decode, the index files equal a numpy parse, and every frame pair after the first becomes a packet whose
    IMU samples are the 200 Hz samples strictly between the two frames plus the two interpolated borders"""
    prov = dp.EurocDataProvider(MICRO_EUROC)
    cam = np.loadtxt(os.path.join(MICRO_EUROC, "mav0/cam0/data.csv"), delimiter=",", skiprows=1, usecols=0,
                     dtype=np.int64)
    cam1 = np.loadtxt(os.path.join(MICRO_EUROC, "mav0/cam1/data.csv"), delimiter=",", skiprows=1, usecols=0,
                      dtype=np.int64)
    assert np.array_equal(prov.left_timestamps, cam) and np.array_equal(prov.right_timestamps, cam1)
    assert np.array_equal(cam1[:len(cam)], cam)   # (the right list is longer; frames pair up by index, as upstream)
    imu = np.loadtxt(os.path.join(MICRO_EUROC, "mav0/imu0/data.csv"), delimiter=",", skiprows=1)
    imu_t = np.loadtxt(os.path.join(MICRO_EUROC, "mav0/imu0/data.csv"), delimiter=",", skiprows=1, usecols=0,
                       dtype=np.int64)
    assert np.array_equal(prov.imu_timestamps, imu_t)
    assert np.array_equal(prov.imu_acc_gyr, imu[:, [4, 5, 6, 1, 2, 3]])
    sync = dp.StereoDataProviderModule(-1)
    frames = {}
    prov.imu_single_callback = sync.fillImuQueue
    prov.left_frame_callback = lambda k, t, img: (frames.__setitem__((0, k), img), sync.fillLeftFrameQueue(t, k))
    prov.right_frame_callback = lambda k, t, img: (frames.__setitem__((1, k), img), sync.fillRightFrameQueue(t, k))
    prov.spin()
    n = prov.getNumImages()
    assert len(frames) == 2 * n
    for (side, k), img in list(frames.items())[::7]:
        name = prov.getLeftImgName(k) if side == 0 else prov.getRightImgName(k)
        assert np.array_equal(img, np.asarray(PIL.open(name)))
    packets = []
    for _ in range(n + 2):
        pk = sync.getInputPacket()
        if pk is not None:
            packets.append(pk)
    usable = [k for k in range(1, n) if cam[k] <= imu_t[-1] and cam[k - 1] >= imu_t[0]]
    assert [p.left_tag for p in packets] == usable[:len(packets)] and len(packets) >= len(usable) - 1
    for p in packets:
        a, b = int(cam[p.left_tag - 1]), int(cam[p.left_tag])
        inner = imu_t[(imu_t > a) & (imu_t < b)]
        assert p.imu_stamps.tolist() == [a] + inner.tolist() + [b]
        assert np.array_equal(p.imu_accgyrs[:, 1:-1].T, imu[(imu_t > a) & (imu_t < b)][:, [4, 5, 6, 1, 2, 3]])


def test_cpp_euroc_data_provider_equals_python(tmp_path):
    """include/kvfe_adapter.hpp's EurocDataProvider + StereoDataProviderModule (tests/cpp/input_side.cpp) on a dataset
    written here: the same packets, drops and pixels as the Python mirror"""
    import subprocess
    rng = np.random.default_rng(4)
    t0, n_frames = 2_000_000_000, 7
    frames = rng.integers(0, 256, (2, n_frames, 20, 28), dtype=np.uint8)
    stamps = [t0 + k * 50_000_000 for k in range(n_frames)]
    for c, cam in enumerate(("cam0", "cam1")):
        d = tmp_path / "mav0" / cam / "data"
        d.mkdir(parents=True)
        lines = ["#timestamp [ns],filename"]
        for k, t in enumerate(stamps):
            if not (cam == "cam0" and k == 2):
                PIL.fromarray(frames[c, k]).save(d / f"{t}.png")
            lines.append(f"{t},{t}.png")
        (tmp_path / "mav0" / cam / "data.csv").write_text("\n".join(lines) + "\n")
    (tmp_path / "mav0" / "imu0").mkdir()
    imu_t = [t0 + 3_000_000 + i * 5_000_000 for i in range(55)]        # the IMU starts after frame 0, ends before frame 6
    rows = ["#timestamp [ns],w_x,w_y,w_z,a_x,a_y,a_z"] + \
           [f"{t},{0.01 * i},0.2,-0.3,{1.0 + 0.125 * i},0.5,9.81" for i, t in enumerate(imu_t)]
    (tmp_path / "mav0" / "imu0" / "data.csv").write_text("\n".join(rows) + "\n")

    prov = dp.EurocDataProvider(str(tmp_path))
    sync = dp.StereoDataProviderModule(-1)
    pix = [0]

    def cb(side):
        def f(k, t, img):
            pix[0] += int(img.sum()) * (1 if side == 0 else 3)
            (sync.fillLeftFrameQueue if side == 0 else sync.fillRightFrameQueue)(t, k)
        return f
    prov.imu_single_callback = sync.fillImuQueue
    prov.left_frame_callback, prov.right_frame_callback = cb(0), cb(1)
    prov.spin()
    want = [f"dataset: images={n_frames} imu={len(imu_t)} pixel_sum={pix[0]}"]
    while True:
        pk = sync.getInputPacket()
        if pk is None and sync.last_action == abi.SYNC_EMPTY:
            break
        if pk is None:
            want.append(f"dropped: action={sync.last_action}")
            if sync.last_action == abi.SYNC_WAIT_IMU:      # the last frame waits for IMU data that never comes
                break
        else:
            want.append(f"packet: t={pk.timestamp} tag={pk.left_tag},{pk.right_tag} n_imu={pk.imu_stamps.size} "
                        f"first={pk.imu_stamps[0]} last={pk.imu_stamps[-1]} acc0={float(pk.imu_accgyrs[0, 0])!r}")
    assert any(w.startswith("packet") for w in want) and any(w.startswith("dropped") for w in want)
    cpp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp")
    r = subprocess.run(["make", "-C", cpp, "input_side"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([os.path.join(cpp, "input_side"), "", str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = [ln for ln in r.stdout.splitlines() if ln.startswith(("dataset", "packet", "dropped"))]
    norm = lambda ln: " ".join(f"acc0={float(tok[5:])!r}" if tok.startswith("acc0=") else tok for tok in ln.split())  # noqa: E731
    assert [norm(g) for g in got] == [norm(w) for w in want]


# ---------------------------------------------------------------------------------------------------------------
# MonoDataProviderModule (tests/testMonoProvider.cpp: the stereo cases without a right queue) and RgbdDataProviderModule
# ---------------------------------------------------------------------------------------------------------------
def _mono_script(events):
    m = dp.MonoDataProviderModule(-1)
    out, tag = [], 0
    for kind, t in events:
        if kind == "i":
            m.fillImuQueue(t, np.zeros(6))
        elif kind == "f":
            m.fillLeftFrameQueue(t, tag)
            tag += 1
        else:
            pk = m.getInputPacket()
            out.append(None if pk is None else (pk.timestamp, pk.left_tag, pk.right_tag, pk.imu_stamps.tolist()))
    return out


def test_mono_provider_reference_cases():
    S = ("s", None)
    # basicSequentialCase (:77-107)
    assert _mono_script([("i", 10), ("f", 11), S, ("i", 12), ("i", 13), ("i", 14), ("f", 17), ("i", 18), S]) == \
        [None, (17, 1, -1, [11, 12, 13, 14, 17])]
    # dropFramesOlderThanImu (:109-142)
    ev = [("i", 10), ("f", 11), S, ("i", 16)]
    for t in range(12, 16):
        ev += [("f", t), S]
    ev += [("f", 17), ("i", 18), S]
    assert _mono_script(ev) == [None] * 5 + [(17, 5, -1, [11, 16, 17])]
    # imageBeforeImuDelayedSpinTest (:206-232)
    assert _mono_script([("f", 10), ("i", 11), ("f", 12), ("i", 13), ("f", 14), ("i", 15), S, S, S]) == \
        [None, None, (14, 2, -1, [12, 13, 14])]
    # monoPipelineInvalidImuSequence (:257-273), testPartialImuSequence (:275-303)
    assert _mono_script([("i", 10), ("f", 1), S, ("i", 11), ("i", 12), ("i", 13), ("f", 3), S]) == [None, None]
    assert _mono_script([("i", 0), ("f", 1), S, ("i", 2), ("i", 3), ("i", 4), ("f", 5), S, ("i", 5), S]) == \
        [None, None, (5, 1, -1, [1, 2, 3, 4, 5])]
    # testOutOfOrderImuAndImageSequence (:363-393)
    ev = [("i", 0), ("f", 3), S] + [("i", t) for t in (2, 4, 3, 5, 5, 5)] + [("f", 2), ("i", 5), ("i", 5), ("f", 5), S, S]
    assert _mono_script(ev) == [None, None, (5, 2, -1, [3, 4, 5])]


def test_rgbd_provider_missing_depth_frame_ends_the_imu_interval():
    """RgbdDataProviderModule.cpp:44-84 caches the frame's timestamp before it looks for the depth frame (the stereo
    module does not): after a frame without depth image the next packet's IMU data starts at THAT frame"""
    outs = {}
    for cls in (dp.StereoDataProviderModule, dp.RgbdDataProviderModule):
        m = cls(-1)
        fill2 = m.fillDepthFrameQueue if cls is dp.RgbdDataProviderModule else m.fillRightFrameQueue
        m.fillImuQueue(0, np.zeros(6))
        m.fillLeftFrameQueue(1, 0)
        fill2(1, 0)
        assert m.getInputPacket() is None
        for t in (2, 3, 4):
            m.fillImuQueue(t, np.zeros(6))
        m.fillLeftFrameQueue(5, 1)                      # no depth / right frame for this one
        m.fillImuQueue(6, np.zeros(6))
        m.fillImuQueue(7, np.zeros(6))
        m.fillLeftFrameQueue(8, 2)
        fill2(8, 2)
        m.fillImuQueue(9, np.zeros(6))
        assert m.getInputPacket() is None and m.last_action == abi.SYNC_DROP_NO_RIGHT
        outs[cls.__name__] = m.getInputPacket().imu_stamps.tolist()
    assert outs == {"StereoDataProviderModule": [1, 2, 3, 4, 6, 7, 8], "RgbdDataProviderModule": [5, 6, 7, 8]}


def test_mono_and_rgbd_providers_equal_oracle_on_random_traffic():
    rng = np.random.default_rng(5)
    for trial in range(20):
        mode = 1 + trial % 2
        p = (dp.MonoDataProviderModule if mode == 1 else dp.RgbdDataProviderModule)(-1)
        o = ora.StereoProvider(mode)
        t_imu, t_cam, tag = 1, 2, 0
        for _ in range(300):
            ev = rng.integers(0, 10)
            if ev < 5:
                t_imu += int(rng.integers(-1, 4))
                v = rng.normal(size=6)
                p.fillImuQueue(t_imu, v)
                o.imu.add(t_imu, v)
            elif ev < 8:
                t_cam = max(1, t_cam + int(rng.integers(-2, 7)))
                p.fillLeftFrameQueue(t_cam, tag)
                o.left.append((t_cam, tag))
                if mode == 2 and rng.integers(0, 8) != 0:
                    p.fillRightFrameQueue(t_cam, tag)
                    o.right.append((t_cam, tag))
                tag += 1
            else:
                r = p.getInputPacket()
                code, pk = o.spin()
                assert p.last_action == code
                if pk is None:
                    assert r is None
                else:
                    assert (r.timestamp, r.left_tag, r.right_tag, r.imu_stamps.tolist()) == (pk[0], pk[1], pk[2], pk[3])
                    assert np.array_equal(r.imu_accgyrs.T, np.asarray(pk[4]))


def test_read_and_convert_to_gray_scale_reference_case():
    """tests/testUtilsOpenCV.cpp:529-540: ReadAndConvertToGrayScale("chessboard.png") has one channel and equals the
    pattern of cvCreateChessboard(30, 10, 8) (:77-92: 30-pixel squares, 10 rows x 8 columns, white where the square
    indices sum to an even number)"""
    img = dp.ReadAndConvertToGrayScale(os.path.join(GOLDEN, "chessboard.png"))
    r, c = np.mgrid[0:300, 0:240]
    expected = np.where(((r // 30) + (c // 30)) % 2 == 0, 255, 0).astype(np.uint8)
    assert img.shape == (300, 240) and dp.png_info(open(os.path.join(GOLDEN, "chessboard.png"), "rb").read())[2] == 1
    assert np.array_equal(img, expected)


# ---------------------------------------------------------------------------------------------------------------
# JPEG (libjpeg's default pipeline restated: Huffman, integer IDCT, fancy upsampling, YCbCr -> RGB) against PIL
# ---------------------------------------------------------------------------------------------------------------
def _smooth(rng, h, w, c):
    a = rng.integers(0, 256, (h // 4 + 2, w // 4 + 2, c)).astype(np.float64)
    a = np.kron(a, np.ones((4, 4, 1)))[:h, :w] + rng.normal(0, 6, (h, w, c))
    return np.clip(a, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("subsampling", [0, 1, 2])
def test_jpeg_colour_equals_libjpeg(subsampling):
    """4:4:4 / 4:2:2 / 4:2:0 baseline files of PIL's encoder (standard and optimised Huffman tables, restart intervals,
    qualities 30-100, sizes that are not MCU multiples down to one pixel): grey == BGR2GRAY of libjpeg's RGB, bit for
    bit"""
    rng = np.random.default_rng(100 + subsampling)
    for (h, w) in [(8, 8), (37, 53), (1, 1), (7, 1), (1, 9), (64, 80), (33, 17), (100, 3), (2, 2), (17, 16)]:
        for q, kw in [(30, {}), (75, {"optimize": True}), (95, {"restart_marker_blocks": 3}),
                      (100, {"restart_marker_rows": 1})]:
            b = io.BytesIO()
            PIL.fromarray(_smooth(rng, h, w, 3)).save(b, "JPEG", quality=q, subsampling=subsampling, **kw)
            f = b.getvalue()
            assert dp.jpeg_info(f) == (w, h, 3)
            want = cv_gray(np.asarray(PIL.open(io.BytesIO(f)).convert("RGB")))
            assert np.array_equal(dp.decode_jpeg_gray(f), want), (h, w, q, kw)


def test_jpeg_grey_progressive_and_errors():
    rng = np.random.default_rng(8)
    for (h, w) in [(8, 8), (37, 53), (1, 1), (40, 200)]:
        for q in (50, 90):
            b = io.BytesIO()
            PIL.fromarray(_smooth(rng, h, w, 1)[..., 0]).save(b, "JPEG", quality=q)
            f = b.getvalue()
            assert dp.jpeg_info(f) == (w, h, 1)
            assert np.array_equal(dp.decode_jpeg_gray(f), np.asarray(PIL.open(io.BytesIO(f))))
    b = io.BytesIO()
    PIL.fromarray(_smooth(rng, 32, 32, 3)).save(b, "JPEG", progressive=True)
    with pytest.raises(KvfeError) as e:
        dp.decode_jpeg_gray(b.getvalue())
    assert e.value.status == -2                                   # KVFE_ERR_UNSUPPORTED
    b = io.BytesIO()
    PIL.fromarray(_smooth(rng, 24, 24, 3)).save(b, "JPEG", quality=80)
    good = b.getvalue()
    for cut in (3, 20, len(good) // 2):                           # truncated inside the headers: refused
        with pytest.raises(KvfeError):
            dp.decode_jpeg_gray(good[:cut])
    out = np.empty((24, 25), np.uint8)                            # a destination of another size: refused
    assert load().kvfe_jpeg_decode_gray(good, len(good), out.ctypes.data, out.strides[0], 25, 24) == -1
    tail = dp.decode_jpeg_gray(good[:len(good) - 40] + b"\xff\xd9")   # damaged entropy data decodes to SOMETHING
    assert tail.shape == (24, 24)
    with pytest.raises(KvfeError):
        dp.decode_jpeg_gray(b"\x89PNG not a jpeg")


@pytest.mark.skipif(not os.path.isdir("/root/reference/tests/data/ForStereoTracker"),
                    reason="the reference checkout (build container only)")
def test_jpeg_reference_frames_equal_committed_fixture():
    """the six JPEG frames of tests/testStereoVisionImuFrontend.cpp:674-926 through ReadAndConvertToGrayScale ==
    tests/golden/ForStereoTracker/frames_0_1_8.npz (decoded by libjpeg through PIL when the fixture was made)"""
    z = np.load(os.path.join(GOLDEN, "ForStereoTracker", "frames_0_1_8.npz"))
    for side, key in (("left", "lefts"), ("right", "rights")):
        for k, n in enumerate((0, 1, 8)):
            img = dp.ReadAndConvertToGrayScale(f"/root/reference/tests/data/ForStereoTracker/{side}_frame{n:04d}.jpg")
            assert np.array_equal(img, z[key][k])


# ---------------------------------------------------------------------------------------------------------------
# gyro rotation between keyframes -> keyframe_R_cur_frame
# ---------------------------------------------------------------------------------------------------------------
def test_preintegrate_rotation_equals_restatement_and_scipy():
    from scipy.spatial.transform import Rotation as Rot
    rng = np.random.default_rng(12)
    for trial in range(10):
        n = int(rng.integers(2, 60))
        t = np.cumsum(rng.integers(1, 9_000_000, n)).astype(np.int64)
        ag = rng.normal(0, 1, (6, n))
        if trial == 3:
            ag[3:, :] = 0.0                                   # the near-zero branch of so3::ExpmapFunctor
        if trial == 4:
            ag[3:, :] *= 1e-9
        bias = rng.normal(0, 0.01, 3) if trial % 2 else np.zeros(3)
        R0 = Rot.from_rotvec(rng.normal(0, 0.3, 3)).as_matrix() if trial % 3 == 0 else None
        R = dp.preintegrate_rotation(t, ag, bias, R0)
        want = ora.preintegrate_rotation(t.tolist(), ag[3:].T.tolist(), bias.tolist(),
                                         None if R0 is None else R0.reshape(9).tolist())
        assert R.reshape(9).tolist() == want                  # same IEEE operations in the same order
        E = np.eye(3) if R0 is None else R0.copy()
        for i in range(n - 1):
            E = E @ Rot.from_rotvec((ag[3:, i] - bias) * ((t[i + 1] - t[i]) / 1e9)).as_matrix()
        assert np.abs(R - E).max() < 1e-13
    with pytest.raises(KvfeError):                            # "No Imu data found." / "Imu delta is 0!"
        dp.preintegrate_rotation(np.array([5], np.int64), np.zeros((6, 1)))
    with pytest.raises(KvfeError):
        dp.preintegrate_rotation(np.array([5, 5], np.int64), np.zeros((6, 2)))
    bRc = Rot.from_rotvec([0.1, -1.2, 0.4]).as_matrix()
    d = Rot.from_rotvec([0.02, 0.01, -0.03]).as_matrix()
    assert np.abs(dp.keyframe_R_cur_frame(bRc, d) - bRc.T @ d @ bRc).max() < 1e-15


@pytest.mark.skipif(not os.path.isdir(MICRO_EUROC), reason="the reference checkout (build container only)")
def test_micro_euroc_packets_to_keyframe_rotation():
    """dataset -> provider -> synchroniser -> gyro preintegration: the rotation since frame 10 accumulated packet by
    packet stays within a milliradian of the fixture's own coarser integration (tests/golden/make_fixtures.py: raw
    samples, no interpolated borders) over the nine frames the smoke test and the benchmark's single stream replay"""
    from scipy.spatial.transform import Rotation as Rot
    z = np.load(os.path.join(GOLDEN, "micro_euroc_f10_18.npz"))
    prov = dp.EurocDataProvider(MICRO_EUROC, initial_k=9, final_k=19)
    sync = dp.StereoDataProviderModule(-1)
    prov.imu_single_callback = sync.fillImuQueue
    prov.left_frame_callback = lambda k, t, img: sync.fillLeftFrameQueue(t, k)
    prov.right_frame_callback = lambda k, t, img: sync.fillRightFrameQueue(t, k)
    prov.spin()
    R, got = np.eye(3), {}
    while True:
        pk = sync.getInputPacket()
        if pk is None and sync.last_action == abi.SYNC_EMPTY:
            break
        if pk is None:
            continue
        if pk.left_tag >= 11:                                 # packets 11..18 carry the IMU data since frame 10..17
            R = dp.preintegrate_rotation(pk.imu_stamps, pk.imu_accgyrs, deltaRij=R)
            got[pk.left_tag] = R.copy()
    assert sorted(got) == list(range(11, 19)) and z["timestamps"][0] == prov.timestampAtFrame(10)
    for k in range(11, 19):
        err = Rot.from_matrix(z["body_R"][k - 10].T @ got[k]).magnitude()
        assert err < 1e-3, (k, err)


@pytest.mark.skipif(not os.path.isdir(MICRO_EUROC), reason="the reference checkout (build container only)")
def test_replay_tool_dry_run():
    """tools/replay_euroc.py --dry-run: dataset -> packets -> rotation -> batch decode, no device"""
    import subprocess
    tool = os.path.join(os.path.dirname(__file__), "..", "tools", "replay_euroc.py")
    r = subprocess.run([sys.executable, tool, MICRO_EUROC, "--dry-run", "--final-k", "6", "--copies", "3"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert [ln.split()[1] for ln in lines[:-1]] == ["1", "2", "3", "4", "5"]      # frame 0 only sets the previous stamp
    assert all("imu=11" in ln for ln in lines[:-1]) and "5 pairs x 3 stream(s)" in lines[-1]
