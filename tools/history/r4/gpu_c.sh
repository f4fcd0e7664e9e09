# round 4, call c: grouped cornerSubPix kernel -- parity suite, A/B against the one-corner-per-block kernel
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 120 python -c "
import sys; sys.path.insert(0,'tests')
import test_gpu_pyramid_r3 as T
c=T._ctx(752,480,2,win=8); print('sanity ok')" || { echo "SANITY FAILED"; exit 1; }
timeout 900 python -m pytest ${TESTS:-tests} -m gpu -q -x > gpurun_out/c_tests.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/c_tests.log | cut -c1-220
run() {
env $1 timeout 300 python bench.py --legs ${2:-none} --steps 30 --warmup 8 --repeats 2 --stage-event-stride 4 2> gpurun_out/c_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{})
print('[$1]', d['value'], d['ms_per_step'], d['repeats']['values'], ' '.join('%s %.3f' % (k[:7], v) for k, v in st.items()))
for k in ('single_stream','outputs_inclusive','single_stream_spinonce','kf_realistic','nominal','c5'):
    if k in d: print('   ', k, {a: d[k].get(a) for a in ('value','ms_per_step','ms_per_pair','no_readback_value','vs_no_readback','error') if a in d[k]}, ' '.join('%s %.3f' % (kk[:7], v) for kk, v in d[k].get('stage_ms_per_step_summed_over_groups',{}).items()))
if 'roofline' in d and d['roofline']: print('    roofline', d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], 'weighted', d.get('roofline_dense_weighted',{}).get('frac'))
"
}
run KVFE_SUBPIX_GROUP=0 kf_realistic
run KVFE_SUBPIX_GROUP=1 kf_realistic
[ -n "$SKIP_C5" ] || { run KVFE_SUBPIX_GROUP=0 c5; run KVFE_SUBPIX_GROUP=1 c5; }
