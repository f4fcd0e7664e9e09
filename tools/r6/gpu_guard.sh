# round 6: the GPU suite with every device buffer of the library between unmapped guard ranges (KVFE_GUARD_ALLOC, kvfe_api.cpp):
# 1 = buffers END on a mapping's end (accesses past the end fault), 2 = buffers BEGIN on a mapping's start (accesses in front
# fault).  usage: bash tools/r6/gpu_guard.sh [pytest args]     (default: the whole -m gpu suite, minus the timing assertion)
mkdir -p gpurun_out; export TMPDIR=/tmp
for M in ${MODES:-1 2}; do
  KVFE_GUARD_ALLOC=$M timeout ${TMO:-1500} python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_staged_overlap.py "$@" > gpurun_out/guard_$M.log 2>&1
  echo "KVFE_GUARD_ALLOC=$M rc=$?"; tail -5 gpurun_out/guard_$M.log | cut -c1-300
done
