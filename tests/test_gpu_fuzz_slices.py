"""Fixed-seed slices of the randomised GPU-vs-oracle cross-checks (tools/fuzz_*.py) inside the driver-run
suite: random image sizes, feature counts, LK windows / pyramid depths / iteration caps, ANMS types, stereo
template sizes, RANSAC variants and thresholds, the three front-end types with several streams of different
cadence, dense-stereo parameter sets, and every component call.  Each tool prints one summary line that
counts mismatching configurations; the slice passes when that count is zero.  (The tools remain runnable by
hand with other seeds: `python tools/fuzz_frontend.py 200 5`.)"""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, n, seed, timeout=600):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(n), str(seed)], cwd=ROOT,
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (tool, r.stdout[-3000:], r.stderr[-3000:])
    return r.stdout


@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_frontend_slice(seed):
    """stereo front-end sequences: 10 random configurations x 5 frames per seed"""
    out = _run("fuzz_frontend.py", 10, seed)
    m = re.search(r"configs failed: (\d+) of (\d+)", out)
    assert m and int(m.group(2)) == 10, out[-2000:]
    assert int(m.group(1)) == 0, out[-4000:]
    assert out.count(" ok ") >= 8, out[-4000:]         # at most a couple of configurations refused at create


def test_fuzz_variants_slice():
    """mono / RGBD / multi-stream batches with different cadences, dense-stereo parameter sets"""
    out = _run("fuzz_variants.py", 10, 21)
    m = re.search(r"front-end configs failed: (\d+) of (\d+) ; dense configs failed: (\d+) of (\d+)", out)
    assert m, out[-2000:]
    assert int(m.group(1)) == 0 and int(m.group(3)) == 0, out[-4000:]


def test_fuzz_components_slice():
    """every component call (rectify, undistort, versors, LK, cornerSubPix, predictor, stereo search, sparse
    stereo, equalizeHist) on random sizes / windows"""
    out = _run("fuzz_components.py", 8, 31)
    m = re.search(r"mismatching checks: (\d+)", out)
    assert m and int(m.group(1)) == 0, out[-4000:]


def test_fuzz_ransac_slice():
    """5-point / 2-point mono, 3-point Arun / 1-point voting on random scenes: every field identical"""
    out = _run("fuzz_ransac.py", 12, 41)
    m = re.search(r"mismatching checks: (\d+)", out)
    assert m and int(m.group(1)) == 0, out[-4000:]


def test_fuzz_batched_slice():
    """round 6: batches of 5 - 24 streams with different sequences and cadences through every entry point (host, device
    with device_frames_persist 0 / 1, staged), random sizes / windows (24 = the four-points-per-wave tracking kernel) /
    pyramid depths / ANMS types: 8 configurations, every stream compared on every step"""
    out = _run("fuzz_batched.py", 8, 51, timeout=900)
    m = re.search(r"configs failed: (\d+) of (\d+)", out)
    assert m and int(m.group(2)) == 8, out[-2000:]
    assert int(m.group(1)) == 0, out[-4000:]
    assert out.count(" ok ") >= 6, out[-4000:]
