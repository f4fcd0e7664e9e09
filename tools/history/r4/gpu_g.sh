# round 4, call g: the pruned build -- full GPU suite, smoke(), the default bench line
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/g_tests.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/g_tests.log | cut -c1-220
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py --legs nominal,single_stream,kf_realistic,outputs,spinonce > gpurun_out/g_bench.json 2> gpurun_out/g_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/g_bench.json'))
print("value", d['value'], d['ms_per_step'], d['repeats']['values'])
for k in ('nominal','single_stream','kf_realistic','outputs_inclusive','single_stream_spinonce'):
    if k in d: print('   ', k, {a: d[k].get(a) for a in ('value','ms_per_step','ms_per_pair','no_readback_value','vs_no_readback','error') if a in d[k]})
print("stages", d.get('stage_ms_per_step_summed_over_groups'))
print("roofline", d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], "weighted", d['roofline_dense_weighted']['frac'])
for k in d['roofline_kernels']: print("   ", k['kernel'], k['frac'], k['avg_launch_ms'])
PY
