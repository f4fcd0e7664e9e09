# round 5, call s: the dense tests of the whole suite (the error word of the two-pass aggregation was read uninitialised by calls that take another path)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -k "dense or backproject or fuzz or c5" > gpurun_out/s_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s_tests.log
grep -E "^FAILED|^ERROR" gpurun_out/s_tests.log | head
