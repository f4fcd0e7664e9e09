mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
timeout 60 python -c "
import sys; sys.path.insert(0,'tests')
import test_gpu_pyramid_r3 as T
c=T._ctx(752,480,2,win=8); print('sanity ok')" || { echo "SANITY FAILED"; exit 1; }
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && { grep -E "Error|FAILED|assert" gpurun_out/gpu_tests.log | head -20; exit 1; }
VARIANTS="${VARIANTS:-KVFE_MINEIG_IMPL=1|KVFE_MINEIG_IMPL=2}" REPEATS=1 bash tools/gpu_ab.sh
