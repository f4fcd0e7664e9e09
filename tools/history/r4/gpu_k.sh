# round 4, call k: grouped cornerSubPix with a device-sized grid (an idle launch was ~30 us of empty workgroups on the critical path)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bench_configs.py -m gpu -q -x > gpurun_out/k_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/k_tests.log | cut -c1-200
timeout 300 python bench.py --legs kf_realistic,c5 --steps 30 --warmup 8 --repeats 3 --stage-event-stride 4 2> gpurun_out/k_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{})
print(d['value'], d['ms_per_step'], d['repeats']['values'], ' '.join('%s %.3f' % (k[:7], v) for k, v in st.items()))
for k in ('kf_realistic','c5'):
    if k in d: print('   ', k, {a: d[k].get(a) for a in ('value','ms_per_step') if a in d[k]}, ' '.join('%s %.3f' % (kk[:7], v) for kk, v in d[k].get('stage_ms_per_step_summed_over_groups',{}).items()))
"
