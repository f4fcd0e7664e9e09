# round 4, call a: first contact of the round's changes with the hardware
#   full GPU suite; KVFE_SELECT_IMPL=1 on the detection / sequence tests; A/B of the run-skipping min-eigenvalue kernel;
#   strip-height sweep; the output-side legs; a kernel trace of the default leg
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 120 python -c "
import sys; sys.path.insert(0,'tests')
import test_gpu_pyramid_r3 as T
c=T._ctx(752,480,2,win=8); print('sanity ok')" || { echo "SANITY FAILED"; exit 1; }
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/a_tests.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/a_tests.log | cut -c1-220
echo "== KVFE_SELECT_IMPL=1 on detection + sequences"
KVFE_SELECT_IMPL=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_configs.py tests/test_gpu_fuzz_slices.py -m gpu -q > gpurun_out/a_tests_sel.log 2>&1; echo "pytest(select rounds) rc=$?"; tail -12 gpurun_out/a_tests_sel.log | cut -c1-220
run() {
env $1 timeout 300 python bench.py --legs ${2:-none} --steps 30 --warmup 8 --repeats 2 --stage-event-stride 4 2> gpurun_out/a_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{})
print('[$1]', d['value'], d['ms_per_step'], d['repeats']['values'], ' '.join('%s %.3f' % (k[:7], v) for k, v in st.items()))
for k in ('single_stream','outputs_inclusive','single_stream_spinonce','kf_realistic','nominal'):
    if k in d: print('   ', k, {a: d[k].get(a) for a in ('value','ms_per_step','ms_per_pair','no_readback_value','vs_no_readback','host_enqueue_ms_per_step','error') if a in d[k]})
if 'roofline' in d and d['roofline']: print('    roofline', d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], 'weighted', d.get('roofline_dense_weighted',{}).get('frac'))
"
}
run KVFE_MINEIG_SKIP=0
run KVFE_MINEIG_SKIP=1
run KVFE_MINEIG_SKIP=0
run KVFE_MINEIG_SKIP=1
for r in 40 60 120; do run KVFE_MINEIG_ROWS=$r; done
echo "== legs"
run KVFE_X=0 single_stream,outputs,spinonce,kf_realistic
run KVFE_SELECT_IMPL=1 single_stream
echo "== kernel trace, default leg"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/a_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --legs none --steps 20 --warmup 5 --repeats 1 --no-stage-events > $GRAFT_REPO_ROOT/gpurun_out/a_kt.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT; db=$(find gpurun_out/a_kt -name "*.db" | head -1); python tools/rocpd_stats.py $db > gpurun_out/a_kt_stats.md; head -24 gpurun_out/a_kt_stats.md | cut -c1-150
