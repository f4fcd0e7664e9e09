"""Per-iteration latency of the cornerSubPix wave kernel: kvfe_corner_subpix on n corners that never
converge (eps = 0), 20 vs 100 iterations; the difference / 80 is one iteration of the slowest wave."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from kimera_vio_amd import frontend as F, params as P, synth
G = os.path.join(ROOT, "tests", "golden")
L = P.load_camera_params(os.path.join(G, "params_euroc", "LeftCameraParams.yaml"))
R = P.load_camera_params(os.path.join(G, "params_euroc", "RightCameraParams.yaml"))
p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=0)
img = synth.RigStream(L, R, seed=1).frame(0)[0]
c = F.Context(L, R, p)
rng = np.random.default_rng(0)
for n in (1, 64, 256 * 4, 256 * 7, 256 * 12, 256 * 24):
    pts = np.stack([rng.uniform(40, 700, n), rng.uniform(40, 440, n)], 1).astype(np.float32)
    res = {}
    for it in (20, 100):
        c.corner_subpix(img, pts, 10, -1, it, 0.0)
        t0 = time.perf_counter()
        for _ in range(5):
            c.corner_subpix(img, pts, 10, -1, it, 0.0)
        res[it] = (time.perf_counter() - t0) / 5
    print(f"n={n:6d}  20 it: {res[20]*1e6:9.1f} us  100 it: {res[100]*1e6:9.1f} us  per iteration: {(res[100]-res[20])/80*1e6:7.3f} us")
c.close()
