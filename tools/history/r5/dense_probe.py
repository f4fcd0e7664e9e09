"""round 5: kernel time of kvfe_dense_stereo_reconstruction per pair (HIP events inside libkvfe), n pairs per call --
the dense_stereo leg of bench.py without the rest.  KVFE_X_DENSE8=1 selects the eight direction sweeps (A/B while the
two-pass aggregation was brought up)."""
import os
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kimera_vio_amd import _abi as abi
from kimera_vio_amd import frontend as F
from kimera_vio_amd import workloads as WL


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (752, 480)
    wl = WL.build("c5", batch=2, unique=2, ring=(n + 1) // 2, width=W, height=H, seed_base=900)
    ctx = F.Context(wl.left, wl.right, wl.params, batch=1)
    pairs = []
    for i in range(n):
        l, r = wl.lefts[i // 2, i % 2], wl.rights[i // 2, i % 2]
        pairs.append((ctx.undistort_rectify_image(0, l), ctx.undistort_rectify_image(1, r)))
    dp = abi.dense_stereo_params_default()
    lefts, rights = [a for a, _ in pairs], [b for _, b in pairs]
    first = ctx.dense_stereo_reconstruction(lefts, rights, dp)
    ctx.dense_profile_read()
    vals = []
    for _ in range(7):
        disp = ctx.dense_stereo_reconstruction(lefts, rights, dp)
        ms, cnt = ctx.dense_profile_read()
        vals.append(ms / cnt)
    same = all(np.array_equal(a, b) for a, b in zip(first, disp))
    prof = None
    if "agprof" in os.environ.get("KVFE_LIB", "") or "agfree" in os.environ.get("KVFE_LIB", ""):
        prof = ctx.dense_debug_volume(2, (32,)).view(np.uint32).astype(np.int64)
    ctx.close()
    import zlib
    crc = zlib.crc32(b"".join(np.ascontiguousarray(d).tobytes() for d in disp))
    print("dense %dx%d n=%d %s: %.4f ms per pair (median of 7; min %.4f max %.4f)  repeat-identical %s  crc %08x" % (
        W, H, n, "eight sweeps" if os.environ.get("KVFE_X_DENSE8") else "two passes", statistics.median(vals),
        min(vals), max(vals), same, crc))
    if prof is not None:
        rows, dur, dmax, tin, tout, sin, sout, t0, t1, lturns, lsteps, zdur, zrows = prof[:13]
        steps = rows * (W - 65)
        print("   last launch: %d rows, row duration mean %.1f us max %.1f us, kernel span %.1f us; per step %.0f ns; "
              "steps that waited for the row before %.1f %% (%.2f turns per step), for the row after %.1f %% (%.2f); "
              "first rows %.1f us; loader: steps re-polled %d, turns %d" % (
                  rows, dur / max(rows, 1) / 100.0, dmax / 100.0, ((t1 - t0) & 0xffffffff) / 100.0,
                  dur * 10.0 / max(steps, 1), 100.0 * sin / max(steps, 1), tin / max(steps, 1),
                  100.0 * sout / max(steps, 1), tout / max(steps, 1), zdur / max(zrows, 1) / 100.0, lsteps, lturns))


if __name__ == "__main__":
    main()
