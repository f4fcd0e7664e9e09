# round 5, call ad: host side only -- the PNG batch decode with the atomic hand-over pool (bench.py input_side; r5_v4: 0.83 k
# frames/s on one thread, 12.6 k on all 256 with 64 files per call)
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "cgroup: cpu.max=$(cat /sys/fs/cgroup/cpu.max 2>/dev/null) cfs_quota=$(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) cfs_period=$(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null) nproc=$(nproc) affinity=$(python -c 'import os; print(len(os.sched_getaffinity(0)))')"
timeout 200 python bench.py --legs input --steps 8 --warmup 3 --repeats 1 --no-stage-events --no-cpu-baseline > gpurun_out/ad_line.json 2> gpurun_out/ad_err.log
python - <<'PY'
import json
d=json.load(open('bench_detail.json')); print(d.get('input_side'))
PY
python - <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from kimera_vio_amd import dataprovider as DP
files = [open(os.path.join('tests', 'golden', n), 'rb').read() for n in ('left_img_0.png', 'right_img_0.png')]
for nf in (64, 128):
    fl = files * (nf // 2); out = np.empty((nf, 480, 752), np.uint8)
    for th in (8, 16, 32, 0):
        DP.decode_png_gray_batch(fl, out, th)
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < 0.4:
            DP.decode_png_gray_batch(fl, out, th); n += nf
        print('files per call %d threads %d: %.0f frames/s' % (nf, th, n / (time.perf_counter() - t0)), flush=True)
PY
