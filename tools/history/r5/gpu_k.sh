# round 5, call k: two-pass path aggregation (rows of a pass as waves handing their costs down) -- parity, then time
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_dense_twopass.py -m gpu -q -x > gpurun_out/k_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/k_tests.log
grep -E "^FAILED|^ERROR|Error|assert |ran out" gpurun_out/k_tests.log | head -20
for n in 8 4; do
KVFE_X_DENSE8=1 timeout 120 python tools/r5/dense_probe.py $n 2>&1 | grep -v amdgpu.ids
timeout 120 python tools/r5/dense_probe.py $n 2>&1 | grep -v amdgpu.ids
done
timeout 120 python tools/r5/dense_probe.py 8 1280 720 2>&1 | grep -v amdgpu.ids
KVFE_X_DENSE8=1 timeout 120 python tools/r5/dense_probe.py 8 1280 720 2>&1 | grep -v amdgpu.ids
