# round 5, call t: deeper request windows in the rows that store to HBM and in the loaders
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/kimera_vio_amd/csrc
timeout 600 python -m pytest tests/test_gpu_dense_twopass.py -m gpu -q -x > gpurun_out/t_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/t_tests.log
grep -E "^FAILED|^ERROR|Error|assert |ran out" gpurun_out/t_tests.log | head -20
timeout 120 python tools/r5/dense_probe.py 8 2>&1 | grep -v amdgpu.ids
KVFE_LIB=$L/libkvfe_agprof.so timeout 120 python tools/r5/dense_probe.py 8 2>&1 | grep -v amdgpu.ids
timeout 120 python tools/r5/dense_probe.py 4 2>&1 | grep -v amdgpu.ids
timeout 120 python tools/r5/dense_probe.py 8 1280 720 2>&1 | grep -v amdgpu.ids
