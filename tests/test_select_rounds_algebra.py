"""CPU statement of the round-parallel form of cv::goodFeaturesToTrack's greedy minimum-distance filter
(opencv imgproc featureselect.cpp; FeatureDetector.cpp:288 calls it through cv::GFTTDetector) that `select_kernel`'s
opt-in path runs (KVFE_SELECT_IMPL=1, kimera_vio_amd/csrc/k_detect.hip: lfmis rounds).

Sequential reference: walk the candidates in rank order (strongest first); accept one iff no ACCEPTED corner lies
closer than minDistance (dx^2 + dy^2 < md^2); stop after maxCorners acceptances.

Round-parallel form (exactly the same set, the lexicographically-first maximal independent set):
  test:    every undecided candidate r -- blocked pixel (inside the disc of a corner accepted in an earlier round)
           -> rejected; otherwise, if no candidate q < r within minDistance is still undecided or pending -> pending;
  commit:  pending -> accepted, its disc is marked in the blocked-pixel bitmap;
until nothing is undecided; the accepted ranks, ascending, cut at maxCorners.
The kernel's race rule is part of the statement: inside one test phase a candidate may read a neighbour's state from
before OR after that neighbour's own test (0 = undecided and 3 = pending both block), and a stale 'undecided' only
costs a round.  The emulation draws that choice at random."""
import numpy as np
import pytest


def _sequential(xy, md, max_corners):
    acc = []
    for r, (x, y) in enumerate(xy):
        ok = True
        for q in acc:
            dx, dy = x - xy[q][0], y - xy[q][1]
            if dx * dx + dy * dy < md * md:
                ok = False
                break
        if ok:
            acc.append(r)
            if max_corners > 0 and len(acc) == max_corners:
                break
    return acc


def _rounds(rng, xy, W, H, md, max_corners):
    n = len(xy)
    st = np.zeros(n, np.int8)                     # 0 undecided, 1 accepted, 2 rejected, 3 pending
    blocked = np.zeros((H, W), bool)
    gw = (W + md - 1) // md
    cells = {}
    for r, (x, y) in enumerate(xy):
        cells.setdefault((y // md) * gw + x // md, []).append(r)
    rounds = 0
    while (st == 0).any():
        rounds += 1
        order = rng.permutation(n)                # threads run in any order inside the phase
        for r in order:
            if st[r] != 0:
                continue
            x, y = xy[r]
            if blocked[y, x]:
                st[r] = 2
                continue
            wait = False
            for cy in range(max(y // md - 1, 0), y // md + 2):
                for cx in range(max(x // md - 1, 0), min(x // md + 1, gw - 1) + 1):
                    for q in cells.get(cy * gw + cx, ()):
                        if q < r and st[q] in (0, 3):
                            dx, dy = x - xy[q][0], y - xy[q][1]
                            if dx * dx + dy * dy < md * md:
                                wait = True
            if not wait:
                st[r] = 3
        for r in np.nonzero(st == 3)[0]:          # commit
            st[r] = 1
            x, y = xy[r]
            for yy in range(max(y - md + 1, 0), min(y + md - 1, H - 1) + 1):
                dy = yy - y
                h = int(np.floor(np.sqrt(md * md - dy * dy - 1e-9)))
                while h * h + dy * dy >= md * md:
                    h -= 1
                blocked[yy, max(x - h, 0):min(x + h, W - 1) + 1] = True
    acc = [int(r) for r in np.nonzero(st == 1)[0]]
    if max_corners > 0:
        acc = acc[:max_corners]
    return acc, rounds


@pytest.mark.parametrize("W,H,n,md,max_corners,seed", [
    (752, 480, 5000, 10, 2000, 1),       # a real frame's candidate count, nothing truncated
    (752, 480, 5000, 10, 150, 2),        # truncated at maxCorners
    (320, 200, 3000, 7, 0, 3),           # dense: long dependency chains
    (200, 120, 1500, 20, 0, 4),          # large radius
    (160, 100, 800, 1, 0, 5),            # minDistance 1: only exact duplicates collide (there are none)
    (400, 40, 2500, 5, 0, 6),            # a strip: chains along one axis
])
def test_rounds_equal_sequential_greedy(W, H, n, md, max_corners, seed):
    rng = np.random.default_rng(seed)
    idx = rng.choice(W * H, size=min(n, W * H), replace=False)     # distinct pixels, rank = position in the list
    xy = [(int(i % W), int(i // W)) for i in idx]
    want = _sequential(xy, md, max_corners)
    got, rounds = _rounds(rng, xy, W, H, md, max_corners)
    assert got == want
    assert rounds <= len(xy)


def test_rounds_on_a_monotone_chain():
    """worst case for the number of rounds: strengths fall along a line of candidates md - 1 apart -- every candidate
    waits for its stronger neighbour; the result is still the sequential one (every other candidate)"""
    rng = np.random.default_rng(0)
    md, W, H = 6, 400, 20
    xy = [(5 * k, 10) for k in range(70)]
    want = _sequential(xy, md, 0)
    got, rounds = _rounds(rng, xy, W, H, md, 0)
    assert got == want == list(range(0, 70, 2))
    assert rounds >= 2
