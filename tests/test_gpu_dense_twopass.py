"""GPU parity of the two-pass path aggregation (cv::StereoSGBM MODE_HH, four or more pairs per call): rows of a pass are
waves that hand their path costs to the next row through LDS / HBM (k_dense.hip, dense_aggregate_pass_kernel).  Every
int16 disparity must equal the oracle's (oracle/ocv_stereo.cpp), whatever the number of row bands, the width, the number
of disparities, the number of pairs in the call and the number of calls made on the context before (the hand-over
entries carry a 2-bit launch tag).  Reference: StereoMatcher::denseStereoReconstruction, StereoMatcher.cpp:32-121.

Run on the MI355X box:  python -m pytest tests -m gpu -x -q
"""
import os

import numpy as np
import pytest

import oracle_lib as O
from kimera_vio_amd import _abi as abi
from kimera_vio_amd import frontend as F
from kimera_vio_amd import synth
from test_gpu_parity import euroc_cams, euroc_params, gray

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def seq():
    z = np.load(os.path.join(G, "micro_euroc_f10_18.npz"))
    return dict(lefts=z["lefts"], rights=z["rights"])


@pytest.fixture(scope="module")
def ctx():
    L, R = euroc_cams()
    c = F.Context(L, R, euroc_params())
    yield c
    c.close()


@pytest.fixture(scope="module")
def rect_pairs(seq):
    L, R = euroc_cams()
    ocam = O.Camera(L, R)
    pairs = [(ocam.rectify_image(0, gray("left_img_0.png")), ocam.rectify_image(1, gray("right_img_0.png")))]
    for i in range(6):
        pairs.append((ocam.rectify_image(0, seq["lefts"][i]), ocam.rectify_image(1, seq["rights"][i])))
    return pairs


def _check(c, pairs, dp, roi=None):
    exp = [O.dense_stereo_reconstruction(l, r, dp, *(roi or ())) for l, r in pairs]
    got = c.dense_stereo_reconstruction([p[0] for p in pairs], [p[1] for p in pairs], dp)
    for k, (g, e) in enumerate(zip(got, exp)):
        assert np.array_equal(g, e), (k, len(pairs), np.count_nonzero(g != e), g.shape)
    return exp


def test_euroc_pairs_in_one_call_bit_exact(ctx, rect_pairs):
    """seven rectified EuRoC pairs (752 x 480: 30 row bands per pass) in one call, default DenseStereoParams"""
    dp = abi.dense_stereo_params_default()
    exp = _check(ctx, rect_pairs, dp)
    valid = exp[0] != (dp.min_disparity - 1) * 16
    assert valid.mean() > 0.3


def test_repeated_calls_and_changing_pair_counts(ctx, rect_pairs):
    """the launch tag cycles 1, 2, 3: five calls with the same number of pairs, then other counts (a change zeroes the
    hand-over buffer), then chunks (9 pairs = 8 + 1: the last pair takes the few-pairs path)"""
    dp = abi.dense_stereo_params_default()
    for rot in range(5):
        _check(ctx, [rect_pairs[(rot + k) % 7] for k in range(4)], dp)
    _check(ctx, rect_pairs[:5], dp)
    _check(ctx, rect_pairs[3:7], dp)
    _check(ctx, rect_pairs[:2], dp)
    _check(ctx, rect_pairs[1:7], dp)
    _check(ctx, (rect_pairs + rect_pairs)[:9], dp)


@pytest.mark.parametrize("kw", [dict(num_disparities=32), dict(num_disparities=16, min_disparity=0),
                                dict(sad_window_size=5, p1=60, p2=200), dict(uniqueness_ratio=10, disp_12_max_diff=1)])
def test_parameter_variants(ctx, rect_pairs, kw):
    dp = abi.dense_stereo_params_default()
    for k, v in kw.items():
        setattr(dp, k, v)
    _check(ctx, rect_pairs[:4], dp)


@pytest.mark.parametrize("w,h", [(323, 241), (120, 20), (130, 16), (140, 17), (200, 33)])
def test_other_image_sizes(w, h):
    """bands that are not full (H mod 16), a single band, one-row bands, a handful of matchable columns"""
    from kimera_vio_amd import workloads
    L, R = workloads.make_cameras(w, h)
    c = F.Context(L, R, euroc_params())
    try:
        roi = (list(c.rect.roi1), list(c.rect.roi2))
        pairs = []
        for k in range(5):
            tex = synth.base_texture(w + 80, h, 11 + k)
            base = np.clip(np.rint(tex[:h, 3:3 + w + 70]), 0, 255).astype(np.uint8)
            pairs.append((np.ascontiguousarray(base[:, :w]), np.ascontiguousarray(base[:, 7 + k:7 + k + w])))
        dp = abi.dense_stereo_params_default()
        _check(c, pairs, dp, roi)
        dp.num_disparities = 48
        _check(c, pairs[:4], dp, roi)
    finally:
        c.close()


def test_a_chunk_whose_hand_over_wait_ran_out_is_repeated_on_the_direction_sweeps():
    """the two-pass launch's waves wait for each other with bounded polls; when one runs out (preemption, a
    debugger) the call used to fail with KVFE_ERR_HIP.  The chunk is now repeated on the eight independent sweeps.
    KVFE_DENSE_FORCE_FALLBACK=1 treats every chunk's first attempt as timed out (a child process: the switch is read
    once): 5 pairs in one call still equal the oracle, and equal the two-pass result of this process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, os, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))\n"
        "import oracle_lib as O\n"
        "from kimera_vio_amd import _abi as abi, frontend as F\n"
        "from test_gpu_parity import euroc_cams, euroc_params\n"
        "z = np.load(os.path.join(%r, 'tests', 'golden', 'micro_euroc_f10_18.npz'))\n"
        "L, R = euroc_cams(); oc = O.Camera(L, R)\n"
        "pairs = [(oc.rectify_image(0, z['lefts'][i]), oc.rectify_image(1, z['rights'][i])) for i in range(5)]\n"
        "dp = abi.dense_stereo_params_default()\n"
        "c = F.Context(L, R, euroc_params())\n"
        "got = c.dense_stereo_reconstruction([p[0] for p in pairs], [p[1] for p in pairs], dp)\n"
        "bad = sum(int(not np.array_equal(g, O.dense_stereo_reconstruction(l, r, dp))) for g, (l, r) in zip(got, pairs))\n"
        "np.save(sys.argv[1], np.stack(got)); c.close(); print('pairs differing from the oracle:', bad)\n" % (root, root, root))
    import tempfile
    outs = []
    for force in ("1", "0"):
        with tempfile.NamedTemporaryFile(suffix=".npy", delete=False) as t:
            pass
        try:
            r = subprocess.run([sys.executable, "-c", code, t.name], env=dict(os.environ, KVFE_DENSE_FORCE_FALLBACK=force),
                               capture_output=True, text=True, timeout=600)
            assert r.returncode == 0 and "pairs differing from the oracle: 0" in r.stdout, (force, r.stdout[-800:], r.stderr[-1500:])
            outs.append(np.load(t.name))
        finally:
            os.unlink(t.name)
    assert np.array_equal(outs[0], outs[1])
