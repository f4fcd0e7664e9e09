# round 3: the streaming pyramid kernel -- parity tests, then A/B of the pyramid stage (tile kernel vs pyr2, strip heights)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_pyramid_r3.py -x -q > gpurun_out/pyr_tests.log 2>&1; echo "pyr tests rc=$?"; tail -15 gpurun_out/pyr_tests.log
VARIANTS="KVFE_PYR_IMPL=0|KVFE_PYR_IMPL=1|KVFE_PYR_T2=2|KVFE_PYR_T2=4|KVFE_PYR_T2=8|KVFE_PYR_T2=16" REPEATS=1 bash tools/gpu_ab.sh
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests.log
tail -3 gpurun_out/gpu_tests.log
