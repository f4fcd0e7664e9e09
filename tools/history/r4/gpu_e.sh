# round 4, call e: grouped cornerSubPix, KC = 56 -- parity on the bench configs, phase stats, A/B
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_bench_configs.py -m gpu -q -x > gpurun_out/e_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/e_tests.log | cut -c1-200
for leg in none kf_realistic; do
KVFE_SUBPIX_STATS=1 KVFE_SUBPIX_GROUP=1 timeout 300 python bench.py --legs $leg --steps 20 --warmup 5 --repeats 1 --no-stage-events 2>&1 >/dev/null | grep KVFE_SUBPIX_STATS
done
run() {
env $1 timeout 300 python bench.py --legs ${2:-none} --steps 30 --warmup 8 --repeats 2 --stage-event-stride 4 2> gpurun_out/e_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{})
print('[$1]', d['value'], d['ms_per_step'], d['repeats']['values'], ' '.join('%s %.3f' % (k[:7], v) for k, v in st.items()))
for k in ('single_stream','outputs_inclusive','single_stream_spinonce','kf_realistic','nominal','c5'):
    if k in d: print('   ', k, {a: d[k].get(a) for a in ('value','ms_per_step','ms_per_pair','no_readback_value','vs_no_readback','error') if a in d[k]}, ' '.join('%s %.3f' % (kk[:7], v) for kk, v in d[k].get('stage_ms_per_step_summed_over_groups',{}).items()))
"
}
run KVFE_SUBPIX_GROUP=0 kf_realistic
run KVFE_SUBPIX_GROUP=1 kf_realistic
