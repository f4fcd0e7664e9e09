import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from PIL import Image
import oracle_lib as O
from kimera_vio_amd import _abi as abi, frontend as F, params as P
G = "/root/repo/tests/golden"
L = P.load_camera_params(os.path.join(G, "sensorLeft.yaml")); R = P.load_camera_params(os.path.join(G, "sensorRight.yaml"))
p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=0)
oc = O.Camera(L, R)
l = oc.rectify_image(0, np.array(Image.open(os.path.join(G, "left_img_0.png")).convert("L")))
r = oc.rectify_image(1, np.array(Image.open(os.path.join(G, "right_img_0.png")).convert("L")))
c = F.Context(L, R, p)
dp = abi.dense_stereo_params_default()
for n in (1, 2, 3, 8):
    c.dense_stereo_reconstruction([l] * n, [r] * n, dp); c.dense_profile_read()
    t = time.perf_counter()
    for _ in range(5): c.dense_stereo_reconstruction([l] * n, [r] * n, dp)
    wall = (time.perf_counter() - t) / 5
    ms, cnt = c.dense_profile_read()
    print("n=%d: kernels %.3f ms per call (%.3f ms/pair), wall %.3f ms per call" % (n, ms / 5, ms / cnt, wall * 1e3))
