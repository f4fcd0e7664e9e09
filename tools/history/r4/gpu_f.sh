# round 4, call f: run walk inside the steady loop of mineig2 (A/B), two-kernel cornerSubPix slot (default = by count)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_bench_configs.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/f_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/f_tests.log | cut -c1-200
KVFE_SUBPIX_GROUP=1 timeout 600 python -m pytest tests/test_gpu_bench_configs.py tests/test_gpu_pipelined_r3.py -m gpu -q -x > gpurun_out/f_tests_g.log 2>&1; echo "pytest(group forced) rc=$?"; tail -3 gpurun_out/f_tests_g.log | cut -c1-200
run() {
env $1 timeout 300 python bench.py --legs ${2:-none} --steps 30 --warmup 8 --repeats 2 --stage-event-stride 4 2> gpurun_out/f_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{})
print('[$1]', d['value'], d['ms_per_step'], d['repeats']['values'], ' '.join('%s %.3f' % (k[:7], v) for k, v in st.items()))
for k in ('single_stream','outputs_inclusive','single_stream_spinonce','kf_realistic','nominal','c5'):
    if k in d: print('   ', k, {a: d[k].get(a) for a in ('value','ms_per_step','ms_per_pair','no_readback_value','vs_no_readback','error') if a in d[k]}, ' '.join('%s %.3f' % (kk[:7], v) for kk, v in d[k].get('stage_ms_per_step_summed_over_groups',{}).items()))
"
}
run KVFE_MINEIG_SKIP=0
run KVFE_MINEIG_SKIP=1 kf_realistic
run KVFE_MINEIG_SKIP=0
run KVFE_MINEIG_SKIP=1
