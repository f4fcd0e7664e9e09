mkdir -p gpurun_out; export TMPDIR=/tmp
cd tests
timeout 90 python - <<'PY' 2>&1 | tail -30
import sys, faulthandler
sys.path.insert(0, '..')
print("import", flush=True)
import numpy as np
import test_gpu_pyramid_r3 as T
print("ctx", flush=True)
c = T._ctx(752, 480, 2, win=8)
print("ctx ok", flush=True)
imgs = T._images(1, 480, 752, seed=1)
lv, cp = c.build_optical_flow_pyramid(imgs, with_level0_copy=True)
print("pyr ok", np.array_equal(cp, imgs), flush=True)
for g, e in zip(lv[0], T._oracle_levels(imgs[0], 2)):
    print(np.array_equal(g, e), flush=True)
PY
echo "rc=$?"
