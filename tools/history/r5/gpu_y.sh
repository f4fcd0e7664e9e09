# round 5, call y: (1) packed rectification taps: parity of the paths that rectify, the launch alone against the float map
# (KVFE_RECT_FLOAT_MAP=1), (2) grouped cornerSubPix: chunk loop rolled (libkvfe_v1.so = f75f4c5) against unrolled
# (libkvfe_v2.so: 32 spilled registers, per-chunk window coordinates hoisted), same call
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/kimera_vio_amd/csrc
timeout 600 python -m pytest tests/test_gpu_components_r2.py tests/test_gpu_bench_configs.py -m gpu -q > gpurun_out/y_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/y_tests.log
grep -E "^FAILED|^ERROR" gpurun_out/y_tests.log | head -30
for V in libkvfe_v1.so libkvfe_v2.so libkvfe_v1.so libkvfe_v2.so; do
KVFE_LIB=$L/$V timeout 300 python bench.py --legs kf_realistic --steps 26 --warmup 10 --repeats 3 --stage-event-stride 2 --no-cpu-baseline > gpurun_out/y_line.json 2> gpurun_out/y_err.log
python - "$V" <<'PY'
import json,sys
d=json.load(open('bench_detail.json'))
v=d.get('kf_realistic',{}); st=v.get('stage_ms_per_step_summed_over_groups',{})
print(sys.argv[1], 'value', d['value'], '  kf_realistic', v.get('value'), v.get('repeats',{}).get('values'), 'subpix %.3f' % st.get('subpix_append', -1))
PY
done
for E in "" 1 "" 1; do
if [ -n "$E" ]; then export KVFE_RECT_FLOAT_MAP=1; else unset KVFE_RECT_FLOAT_MAP; fi
timeout 300 python bench.py --legs alone --steps 26 --warmup 10 --repeats 2 --stage-event-stride 2 --no-cpu-baseline > gpurun_out/y_line.json 2> gpurun_out/y_err.log
python - "float_map=${E:-0}" <<'PY'
import json,sys
d=json.load(open('bench_detail.json')); st=d.get('stage_ms_per_step_summed_over_groups',{})
a=d.get('dense_kernels_alone',{})
print(sys.argv[1], 'value', d['value'], 'rectify in the step %.4f' % st.get('rectify', -1), 'alone:', [(k['kernel'], k['avg_launch_ms'], k['frac']) for k in a.get('kernels', [])])
PY
done
unset KVFE_RECT_FLOAT_MAP
