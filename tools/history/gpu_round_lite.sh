# one GPU call: parity tests, bench line, kernel trace and FETCH/WRITE PMC passes (no SQ passes)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests.log
tail -3 gpurun_out/gpu_tests.log
timeout 400 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
cut -c1-400 gpurun_out/bench.json
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-single-stream"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -o kt -- $B > $R/gpurun_out/prof_kt.log 2>&1; echo "kt rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -o f -- $B > $R/gpurun_out/prof_fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -o w -- $B > $R/gpurun_out/prof_write.log 2>&1; echo "write rc=$?"
cd $R
du -sh gpurun_out
