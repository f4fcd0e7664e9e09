set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests.log
tail -5 gpurun_out/gpu_tests.log
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
cat gpurun_out/bench.json
R=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -o kt -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-single-stream > $R/gpurun_out/prof_kt.log 2>&1; echo "kt rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -o f -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-single-stream --groups 1 > $R/gpurun_out/prof_fetch.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -o w -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-single-stream --groups 1 > $R/gpurun_out/prof_write.log 2>&1; echo "write rc=$?"
cd $R
find gpurun_out -type f | head -50
du -sh gpurun_out
