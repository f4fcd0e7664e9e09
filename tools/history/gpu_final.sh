# final check of a round: the whole GPU suite, smoke(), then the default bench line (all legs)
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests.log
tail -4 gpurun_out/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench.json'))
print("value", d['value'], d['ms_per_step'], d.get('repeats',{}).get('values'))
print("stages", d.get('stage_ms_per_step_summed_over_groups'))
for k in ('nominal','single_stream','c5','dense_stereo','dense_stereo_c5','pcie_inclusive','input_side','cpu_baseline'):
    v=d.get(k)
    if v: print(k, v.get('value', v.get('decode_all_threads')), v.get('ms_per_step', v.get('ms_per_pair')))
print("roofline", d.get('roofline'))
print("largest", d.get('largest_kernel'))
PY
