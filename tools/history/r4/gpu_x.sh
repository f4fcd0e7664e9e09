# round 4, call x: the step's records back to back behind an offset table, one DMA transfer sized from the last completed
# step's need (+ 25 % + 64 KB), the rest fetched on access when a step needs more
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
timeout 1200 python -m pytest tests/test_gpu_pipelined_r3.py tests/test_gpu_bench_configs.py tests/test_gpu_parity.py tests/test_gpu_components_r2.py -m gpu -q -x -k "not dense" > gpurun_out/x_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/x_tests.log | cut -c1-300
cd tests/cpp && make shim_check > /dev/null 2>&1; cd $R
timeout 400 python bench.py --legs outputs,spinonce,nominal --steps 40 --warmup 8 --repeats 5 --stage-event-stride 4 2> gpurun_out/x_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{})
print(d['value'], d['ms_per_step'], d['repeats']['values'], ' '.join('%s %.3f' % (k[:7], v) for k, v in st.items()))
for k in ('outputs_inclusive','single_stream_spinonce'):
    if k in d: print('   ', k, {a: d[k].get(a) for a in ('value','ms_per_step','no_readback_value','vs_no_readback','host_enqueue_ms_per_step') if a in d[k]})
for k in ('nominal','kf_realistic','c5'):
    if k in d: print('   ', k, {a: d[k].get(a) for a in ('value','ms_per_step') if a in d[k]}, d[k]['repeats']['values'])
"
