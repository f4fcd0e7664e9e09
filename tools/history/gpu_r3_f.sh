mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
timeout 60 python -c "
import sys; sys.path.insert(0,'tests')
import test_gpu_pyramid_r3 as T
c=T._ctx(752,480,2,win=8); print('sanity ok')" || { echo "SANITY FAILED"; exit 1; }
timeout 600 python -m pytest tests/test_gpu_replay_r3.py tests/test_gpu_rccl_r3.py tests/test_gpu_pipelined_r3.py -q > gpurun_out/new_tests.log 2>&1; echo "new tests rc=$?"; tail -25 gpurun_out/new_tests.log | cut -c1-300
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && { grep -E "Error|FAILED|assert" gpurun_out/gpu_tests.log | head -20; }
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench.json'))
print("value", d['value'], d['ms_per_step'], d['repeats']['values'])
print(d['stage_ms_per_step_summed_over_groups'])
print("roofline", d['roofline'])
for k in ('nominal','single_stream','c5','klt_max_level_4','kf_realistic'):
    if k in d: print(k, d[k]['value'], d[k]['ms_per_step'], d[k].get('check'))
if 'nominal' in d: print([ (x['kernel'], x['avg_launch_ms'], x['frac'], x['launches_with_work'], x['launches_sampled']) for x in d['nominal']['roofline_kernels']])
for k in ('dense_stereo','dense_stereo_c5'):
    if k in d: print(k, d[k]['value'], d[k]['roofline']['frac'], d[k]['roofline']['design_traffic_frac_of_peak'])
print(d.get('cpu_baseline'))
PY
