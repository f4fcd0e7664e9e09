"""CPU tests of the dense-stereo oracle (oracle/ocv_stereo.cpp): cv::StereoSGBM / StereoBM / filterSpeckles /
medianBlur / reprojectImageTo3D as StereoMatcher::denseStereoReconstruction and
StereoCamera::backProjectDisparityTo3D use them (StereoMatcher.cpp:32-121, StereoCamera.cpp:176-196).

The reference has no numeric test of this path (tests/testStereoMatcher.cpp:131 is a smoke test,
testStereoCamera.cpp:264-362 checks that back-projected points reproject onto their pixels), so parity
with OpenCV is UNPINNED; what is pinned here is that the OpenCV-shaped restatement (row sweep with
rolling Lr buffers) equals an independent numpy statement of the same mathematics, plus the
self-consistency properties the reference's own tests use.
"""
import numpy as np
import pytest
from scipy import ndimage

import oracle_lib as O
from kimera_vio_amd import _abi as abi
from kimera_vio_amd import synth


def _pair(w, h, d, seed=4):
    tex = synth.base_texture(w + 80, h, seed)
    base = np.clip(np.rint(tex[96:96 + h, 60:60 + w + 70]), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(base[:, :w]), np.ascontiguousarray(base[:, d:d + w])


def _params(**kw):
    dp = abi.dense_stereo_params_default()
    for k, v in kw.items():
        setattr(dp, k, v)
    return dp


# ---- independent numpy statement of the SGBM cost volume and path aggregation -------------------
def _bt_rows(img, ftzero):
    """prow channels of calcPixelCostBT and their half-pixel min / max, [2][h][w] each"""
    a = img.astype(np.int32)
    up, dn = np.vstack([a[:1], a[:-1]]), np.vstack([a[1:], a[-1:]])
    g = np.zeros_like(a)
    g[:, 1:-1] = (a[:, 2:] - a[:, :-2]) * 2 + (up[:, 2:] - up[:, :-2]) + (dn[:, 2:] - dn[:, :-2])
    p0 = np.clip(g, -ftzero, ftzero) + ftzero
    p1 = a.copy()
    for p in (p0, p1):
        p[:, 0] = p[:, -1] = ftzero
    out = []
    for p in (p0, p1):
        left = np.concatenate([p[:, :1], (p[:, 1:] + p[:, :-1]) // 2], 1)
        right = np.concatenate([(p[:, :-1] + p[:, 1:]) // 2, p[:, -1:]], 1)
        out.append((p, np.minimum(np.minimum(left, right), p), np.maximum(np.maximum(left, right), p)))
    return out


def _numpy_cost_volume(left, right, dp):
    h, w = left.shape
    minD, D = dp.min_disparity, dp.num_disparities
    minX1 = max(minD + D, 0)
    w1 = w + min(minD, 0) - minX1
    ft = max(dp.pre_filter_cap, 15) | 1
    L, R = _bt_rows(left, ft), _bt_rows(right, ft)
    pix = np.zeros((h, w1, D), np.int64)
    xs = np.arange(minX1, minX1 + w1)
    for ch, shift in ((0, 0), (1, 2)):
        u, u0, u1 = (t[:, xs] for t in L[ch])
        for d in range(D):
            v, v0, v1 = (t[:, xs - d - minD] for t in R[ch])
            c0 = np.maximum(np.maximum(0, u - v1), v0 - u)
            c1 = np.maximum(np.maximum(0, v - u1), u0 - v)
            pix[:, :, d] += np.minimum(c0, c1) >> shift
    r = dp.sad_window_size // 2
    P2 = max(dp.p2, dp.p1 + 1)
    xi = np.clip(np.arange(w1)[:, None] + np.arange(-r, r + 1)[None], 0, w1 - 1)
    hs = pix[:, xi, :].sum(2)
    yi = np.clip(np.arange(h)[:, None] + np.arange(-r, r + 1)[None], 0, h - 1)
    C = P2 + hs[yi].sum(1)
    C[1:, 0, :] = P2              # OpenCV's recurrence never touches column 0 after the first row ...
    C[max(h - r, 1):, :, :] = P2  # ... nor the last SH2 rows (k = y + SH2 >= height)
    return C


def _numpy_aggregate(C, P1, P2):
    h, w1, D = C.shape
    total = np.zeros(C.shape, np.int64)
    BIG = 32767
    for sx, sy in ((1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0), (1, -1), (0, -1), (-1, -1)):
        Lr = np.zeros(C.shape, np.int64)
        ys = range(h) if sy >= 0 else range(h - 1, -1, -1)
        xs = range(w1) if sx >= 0 else range(w1 - 1, -1, -1)
        for y in ys:
            for x in (xs if sy == 0 else [None]):
                if sy == 0:   # horizontal: sequential in x
                    px = x - sx
                    prev = Lr[y, px] if 0 <= px < w1 else np.zeros(D, np.int64)
                    prev = prev[None]
                    cur = C[y, x][None]
                else:         # the whole row depends on the previous row only
                    py = y - sy
                    prev = np.zeros((w1, D), np.int64)
                    if 0 <= py < h:
                        src = Lr[py]
                        if sx == 0:
                            prev = src
                        elif sx > 0:
                            prev[1:] = src[:-1]
                        else:
                            prev[:-1] = src[1:]
                    cur = C[y]
                mn = prev.min(1, keepdims=True)
                delta = mn + P2
                pad = np.full((prev.shape[0], 1), BIG, np.int64)
                dm = np.concatenate([pad, prev[:, :-1]], 1) + P1
                dq = np.concatenate([prev[:, 1:], pad], 1) + P1
                L = cur + np.minimum(np.minimum(prev, delta), np.minimum(dm, dq)) - delta
                if sy == 0:
                    Lr[y, x] = L[0]
                else:
                    Lr[y] = L
        assert Lr.min() >= 0
        total += Lr
    return np.minimum(total, 32767)


@pytest.mark.parametrize("kw", [dict(), dict(num_disparities=16, min_disparity=0, sad_window_size=5, p1=30, p2=90)])
def test_sgbm_cost_volume_and_path_sums_equal_independent_numpy(kw):
    dp = _params(**{**dict(num_disparities=32, sad_window_size=7), **kw})
    left, right = _pair(120, 56, 9)
    _, Cv, Sv = O.stereo_sgbm(left, right, dp, debug=True)
    C = _numpy_cost_volume(left, right, dp)
    assert np.array_equal(C, Cv.astype(np.int64))
    S = _numpy_aggregate(C, dp.p1, max(dp.p2, dp.p1 + 1))
    # S = saturate(saturate(pass 1) + pass 2) == min(32767, sum of the eight path costs) because L_r >= 0
    assert np.array_equal(S, Sv.astype(np.int64))


def test_sgbm_known_shift_and_invalid_value():
    dp = _params()
    left, right = _pair(400, 200, 17)
    disp = O.dense_stereo_reconstruction(left, right, dp)
    inv = (dp.min_disparity - 1) * 16
    assert disp.dtype == np.int16 and disp.shape == left.shape
    assert np.all(disp[:, :dp.min_disparity + dp.num_disparities] == inv)   # columns < minX1 cannot be matched
    valid = disp != inv
    assert valid.mean() > 0.5
    assert np.mean(np.abs(disp[valid] / 16.0 - 17) <= 0.5) > 0.97
    # MODE_SGBM (use_mode_HH: false) and StereoBM find the same shift
    d5 = O.dense_stereo_reconstruction(left, right, _params(use_mode_hh=0))
    v5 = d5 != inv
    assert v5.mean() > 0.5 and np.mean(np.abs(d5[v5] / 16.0 - 17) <= 0.5) > 0.97
    bm = O.dense_stereo_reconstruction(left, right, _params(use_sgbm=0))
    vb = bm != inv
    assert vb.mean() > 0.3 and np.mean(np.abs(bm[vb] / 16.0 - 17) <= 0.5) > 0.97
    # flat images: every cost ties, SGBM returns the smallest disparity everywhere it can match
    flat = np.full((64, 160), 90, np.uint8)
    df = O.stereo_sgbm(flat, flat, _params(speckle_window_size=0))
    assert np.all(df[:, 65:] == dp.min_disparity * 16)


def test_median_blur_16s_matches_scipy():
    rng = np.random.RandomState(0)
    img = rng.randint(-50, 2000, (37, 53)).astype(np.int16)
    for k in (3, 5):
        assert np.array_equal(O.median_blur_16s(img, k), ndimage.median_filter(img, size=k, mode="nearest"))


def _speckles_reference(img, new_val, max_size, max_diff):
    """independent statement: remove 4-connected components (|difference| <= max_diff between neighbours,
    both valid) of at most max_size pixels"""
    h, w = img.shape
    parent = np.arange(h * w)

    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i
    v = img.astype(np.int64)
    for y in range(h):
        for x in range(w):
            if v[y, x] == new_val:
                continue
            for yy, xx in ((y, x + 1), (y + 1, x)):
                if yy < h and xx < w and v[yy, xx] != new_val and abs(v[y, x] - v[yy, xx]) <= max_diff:
                    a, b = find(y * w + x), find(yy * w + xx)
                    if a != b:
                        parent[max(a, b)] = min(a, b)
    roots = np.array([find(i) for i in range(h * w)])
    sizes = np.bincount(roots, minlength=h * w)
    out = img.copy().reshape(-1)
    kill = (v.reshape(-1) != new_val) & (sizes[roots] <= max_size)
    out[kill] = new_val
    return out.reshape(h, w)


def test_filter_speckles_matches_component_definition():
    rng = np.random.RandomState(1)
    base = (rng.randint(0, 6, (40, 60)) * 40).astype(np.int16)
    base = np.kron(base[::4, ::4], np.ones((4, 4), np.int16))          # blocks of 16 px
    noise = rng.randint(0, 100, base.shape) < 15
    img = np.where(noise, rng.randint(0, 400, base.shape), base).astype(np.int16)
    img[rng.randint(0, 100, base.shape) < 10] = -16
    for max_size, max_diff in ((10, 16), (40, 48), (3, 0)):
        got = O.filter_speckles_16s(img, -16, max_size, max_diff)
        assert np.array_equal(got, _speckles_reference(img, -16, max_size, max_diff))


def test_reproject_image_to_3d():
    """cv::reprojectImageTo3D(handleMissingValues): [X Y Z W] = Q [x y d 1], points = XYZ / W, z = 10000 at
    the minimum disparity; tests/testStereoCamera.cpp:264-362 property: points reproject onto their pixel"""
    fx, cx, cy, b = 436.2, 364.4, 256.9, 0.11
    Q = np.array([[1, 0, 0, -cx], [0, 1, 0, -cy], [0, 0, 0, fx], [0, 0, 1 / b, 0]], np.float64)
    rng = np.random.RandomState(2)
    disp = rng.uniform(1, 60, (30, 50)).astype(np.float32)
    disp[rng.randint(0, 10, disp.shape) == 0] = 0.0     # missing values = the minimum of the image
    xyz = O.reproject_image_to_3d(disp, Q, True)
    miss = disp == 0
    assert np.all(xyz[miss][:, 2] == 10000.0)
    v, u = np.nonzero(~miss)
    p = xyz[v, u].astype(np.float64)
    z = fx * b / disp[v, u]
    assert np.allclose(p[:, 2], z, rtol=1e-6)
    assert np.allclose(fx * p[:, 0] / p[:, 2] + cx, u, atol=1e-3)
    assert np.allclose(fx * p[:, 1] / p[:, 2] + cy, v, atol=1e-3)
    # the float conversion happens before the division (Vec3f = Vec3d; Vec3f /= w)
    w = Q[3, 2] * disp[v, u].astype(np.float64)
    X = (u + Q[0, 3]).astype(np.float32).astype(np.float64)
    assert np.array_equal(xyz[v, u][:, 0], (X * (1.0 / w)).astype(np.float32))
