# quick GPU check: parity tests + bench (+ optional FETCH/WRITE pass)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests.log
tail -3 gpurun_out/gpu_tests.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_quick.json 2> gpurun_out/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_quick.json'))
print(d['value'], d['ms_per_step'], d['stage_ms_per_step_summed_over_groups'], d.get('single_stream'))
PY
if [ -n "$PMC" ]; then
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-single-stream"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -o f -- $B > $R/gpurun_out/prof_fetch.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -o w -- $B > $R/gpurun_out/prof_write.log 2>&1; echo "write rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -o kt -- $B > $R/gpurun_out/prof_kt.log 2>&1; echo "kt rc=$?"
cd $R
fi
