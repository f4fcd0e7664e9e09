# round 4, call h: rectification -- LDS base folded into the tap addresses, 8 against 16 streams per block; ssd_tie_policy tests
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ssd_tie_policy.py tests/test_gpu_parity.py tests/test_gpu_components_r2.py -m gpu -q -x > gpurun_out/h_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/h_tests.log | cut -c1-200
run() {
env $1 timeout 300 python bench.py --legs ${2:-none} --steps 30 --warmup 8 --repeats 2 --stage-event-stride 4 $3 2> gpurun_out/h_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{})
print('[$1 $3]', d['value'], d['ms_per_step'], d['repeats']['values'], ' '.join('%s %.3f' % (k[:7], v) for k, v in st.items()))
"
}
run KVFE_RECT_NSUB=4
run KVFE_RECT_NSUB=8
run KVFE_RECT_NSUB=4
run KVFE_RECT_NSUB=8
run KVFE_RECT_NSUB=4 none --single-hip-stream
run KVFE_RECT_NSUB=8 none --single-hip-stream
