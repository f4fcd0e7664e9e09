mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
timeout 200 python -m pytest tests/test_gpu_pyramid_r3.py -x -q > gpurun_out/pyr_tests.log 2>&1; rc=$?; echo "pyr tests rc=$rc"; tail -3 gpurun_out/pyr_tests.log
[ $rc -ne 0 ] && exit 1
Q="--steps 10 --warmup 3 --repeats 1 --legs none --no-stage-events"
cd /tmp
for t in ${T2S:-2 4 8}; do
KVFE_PYR_T2=$t timeout 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3c_kt$t -o kt -- python $R/bench.py $Q > $R/gpurun_out/r3c_kt$t.log 2>&1; echo "kt rc=$?"
python $R/tools/rocpd_stats.py $(find $R/gpurun_out/r3c_kt$t -name "*.db" | head -1) | grep -E "pyr2"
done
