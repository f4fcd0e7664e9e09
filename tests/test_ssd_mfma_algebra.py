"""CPU statement of the algebra behind `ssd_search_mfma` (kimera_vio_amd/csrc/k_stereo.hip): the template search of
searchRightKeypointEpipolar (StereoMatcher.cpp:196-423, cv::matchTemplate TM_SQDIFF + first minimum) written as the
matrix product the kernel feeds to v_mfma_i32_16x16x64_i8 --

    C[oh][j] = sum_k' A[oh][k'] B[k'][j],  A[oh][k'] = S'[16 oh + k'],  B[k'][j] = T'[k' - j],  u = 16 oh + j

on images shifted to signed bytes (x - 128), with the kernel's operand addressing (lane (i, g) of A reads the 16-byte
chunk i + g + 4 s of a stripe row, lane (j, g) of B reads five dwords of the zero-padded template row and funnel-shifts
them by (4 - j) & 3 bytes), its LDS geometry (stripe bytes that were never staged hold garbage, an odd template height
runs one phantom row against a template row of zeros), its prefix sums of the shifted squares and its (ssd, offset) key.
numpy emulates the lanes; the result must be the brute-force first minimum of sum (T - S)^2 for every geometry class
the GPU tests run (tests/test_gpu_stereo_mfma_r3.py).  This pins the index arithmetic on the CPU; what the instruction
itself computes is checked on the GPU."""
import numpy as np
import pytest


def _search(rng, tc, tr, sc, sr, W, H, tcx, tcy, scx, scy):
    L = rng.integers(0, 256, (H, W), dtype=np.uint8)
    R = rng.integers(0, 256, (H, W), dtype=np.uint8)
    KS = (tc + 15 + 63) >> 6
    TPW = 16 * KS + 8
    rw, rh = sc - tc + 1, sr - tr + 1
    NJg, ndwg = (3 + rw + 15) >> 4, (3 + sc + 3) >> 2
    SPB = 16 * max(NJg + 3 + 4 * (KS - 1), (ndwg + 3) >> 2)
    sh0 = scx & 3
    x_al = scx - sh0
    ndw = (sh0 + sc + 3) >> 2
    assert ndw <= 64 and TPW <= 64            # the kernel's lane maps (StereoGeom::mfma_ok)
    # template rows: [16 zero bytes | template - 128 | zeros], one extra row of zeros
    X = np.zeros((tr + 1, TPW * 4), dtype=np.int64)
    X[:tr, 16:16 + tc] = L[tcy:tcy + tr, tcx:tcx + tc].astype(np.int64) - 128
    # stripe rows: staged dwords, garbage behind them and in the extra row
    S = rng.integers(-128, 128, (sr + 1, SPB)).astype(np.int64)
    for y in range(sr):
        S[y, :4 * ndw] = R.reshape(-1)[(scy + y) * W + x_al:(scy + y) * W + x_al + 4 * ndw].astype(np.int64) - 128
    t2 = int((X[:tr] ** 2).sum())
    NJ = (sh0 + rw + 15) >> 4
    best = None
    for oy in range(rh):
        P2 = np.concatenate([[0], np.cumsum((S[oy:oy + tr, :4 * ndw] ** 2).sum(0))])
        for mt in range((NJ + 15) // 16):
            A, B = {}, {}
            rows = tr + (tr & 1)                      # two rows per trip
            for lane in range(64):
                j, g = lane & 15, lane >> 4
                ia = min(16 * mt + j, NJ - 1)
                jc, bsh = (j + 3) >> 2, (4 - (j & 3)) & 3
                for y in range(rows):
                    for s in range(KS):
                        A[y, s, lane] = S[oy + y, 16 * (ia + g) + 64 * s:16 * (ia + g) + 64 * s + 16]
                        d = (4 * g + 4 - jc) + 16 * s
                        B[y, s, lane] = X[y, 4 * d:4 * d + 20][bsh:bsh + 16]
            C = np.zeros((16, 16), dtype=np.int64)     # D = A x B: row = A's lane & 15, column = B's lane & 15
            for y in range(rows):
                for s in range(KS):
                    for i in range(16):
                        for jj in range(16):
                            C[i, jj] += sum(int((A[y, s, i + 16 * g] * B[y, s, jj + 16 * g]).sum()) for g in range(4))
            for lane in range(64):                     # D layout: column = lane & 15, rows 4 (lane >> 4) + r
                j, g = lane & 15, lane >> 4
                for r in range(4):
                    oh = 16 * mt + 4 * g + r
                    u = 16 * oh + j
                    ox = u - sh0
                    if oh < NJ and 0 <= ox < rw:
                        ssd = t2 + int(P2[u + tc] - P2[u]) - 2 * int(C[4 * g + r, j])
                        key = (ssd, oy * rw + ox)
                        if best is None or key < best:
                            best = key
    T = L[tcy:tcy + tr, tcx:tcx + tc].astype(np.int64)
    brute = min((int(((T - R[scy + oy:scy + oy + tr, scx + ox:scx + ox + tc].astype(np.int64)) ** 2).sum()), oy * rw + ox)
                for oy in range(rh) for ox in range(rw))
    return best, brute


@pytest.mark.parametrize("cfg", [
    (101, 11, 201, 11, 752, 480, 300, 100, 150, 100),   # shipped geometry, stripe start = 2 mod 4
    (101, 11, 201, 11, 752, 480, 303, 100, 151, 100),   # template start = 3 mod 4, stripe start = 3 mod 4
    (103, 11, 203, 13, 752, 480, 301, 100, 98, 99),     # three bytes in the last template dword, three offset rows
    (41, 7, 165, 9, 400, 80, 50, 30, 3, 29),            # one K step
    (121, 5, 205, 5, 400, 50, 100, 10, 53, 10),         # three K steps
    (49, 3, 213, 7, 400, 50, 100, 10, 60, 8),           # template + 15 = 64: exactly one K step; 165 offsets x 5 rows
])
def test_mfma_formulation_equals_first_minimum_of_sqdiff(cfg):
    rng = np.random.default_rng(cfg[0] * 1000 + cfg[6])
    best, brute = _search(rng, *cfg)
    assert best == brute
