# round 5, call r: two-pass aggregation, quick check (parity tests + time at 8 / 4 pairs and 1280x720)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_dense_twopass.py -m gpu -q -x > gpurun_out/r_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r_tests.log
grep -E "^FAILED|^ERROR|Error|assert |ran out" gpurun_out/r_tests.log | head -20
timeout 120 python tools/r5/dense_probe.py 8 2>&1 | grep -v amdgpu.ids
timeout 120 python tools/r5/dense_probe.py 4 2>&1 | grep -v amdgpu.ids
timeout 120 python tools/r5/dense_probe.py 8 1280 720 2>&1 | grep -v amdgpu.ids
