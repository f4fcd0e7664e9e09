"""round 6: ONE rectified 752 x 480 pair per kvfe_dense_stereo_reconstruction call, 20 calls (rocprofv3 --kernel-trace --stats
around it: which kernel of the sequence the 0.68 ms are)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from kimera_vio_amd import _abi as abi, frontend as F
from test_gpu_parity import euroc_cams, euroc_params
z = np.load(os.path.join(ROOT, "tests", "golden", "micro_euroc_f10_18.npz"))
L, R = euroc_cams(); oc = O.Camera(L, R)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
pairs = [(oc.rectify_image(0, z["lefts"][i]), oc.rectify_image(1, z["rights"][i])) for i in range(n)]
dp = abi.dense_stereo_params_default()
c = F.Context(L, R, euroc_params())
for _ in range(20):
    c.dense_stereo_reconstruction([p[0] for p in pairs], [p[1] for p in pairs], dp)
print(c.dense_profile_read())
c.close()
