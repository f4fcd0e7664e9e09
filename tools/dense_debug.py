"""GPU-side debugging of the dense stereo path: compares C, the per-pass path sums and the final
disparity of libkvfe with the oracle on one rectified EuRoC pair and prints the first mismatches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from PIL import Image
import oracle_lib as O
from kimera_vio_amd import _abi as abi, frontend as F, params as P

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
L = P.load_camera_params(os.path.join(G, "sensorLeft.yaml"))
R = P.load_camera_params(os.path.join(G, "sensorRight.yaml"))
p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=0)
oc = O.Camera(L, R)
l = oc.rectify_image(0, np.array(Image.open(os.path.join(G, "left_img_0.png")).convert("L")))
r = oc.rectify_image(1, np.array(Image.open(os.path.join(G, "right_img_0.png")).convert("L")))
dp = abi.dense_stereo_params_default()
import json
for k, v in json.loads(os.environ.get("DENSE_KW", "{}")).items():
    setattr(dp, k, v)
t = time.time()
raw, Cv, Sv = O.stereo_sgbm(l, r, dp, debug=True)
print("oracle sgbm %.2fs" % (time.time() - t))
exp = O.dense_stereo_reconstruction(l, r, dp)
c = F.Context(L, R, p)
got = c.dense_stereo_reconstruction(l, r, dp)
gC = c.dense_debug_volume(2, Cv.shape)
gA = c.dense_debug_volume(0, Cv.shape).view(np.uint16).astype(np.int64)
def rep(name, a, b):
    bad = np.argwhere(a != b)
    print(name, "mismatches", len(bad), "of", a.size)
    for idx in bad[:8]:
        print("   at", tuple(idx), "got", a[tuple(idx)], "exp", b[tuple(idx)])
rep("C", gC, Cv)
S = np.minimum(32767, gA)
rep("S", S, Sv.astype(np.int64))
rep("disp", got, exp)
t = time.time()
for _ in range(3):
    c.dense_stereo_reconstruction([l] * 8, [r] * 8, dp)
ms, n = c.dense_profile_read()
print("gpu kernels: %.3f ms/pair over %d pairs; wall %.3f s" % (ms / max(n, 1), n, time.time() - t))
