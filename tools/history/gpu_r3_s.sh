# round 3, call s: cornerSubPix waves at raised issue priority (s_setprio 3; KVFE_SUBPIX_PRIO=0 = default priority)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_pipelined_r3.py tests/test_gpu_bench_configs.py -m gpu -x -q > gpurun_out/s_tests.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/s_tests.log
run() {
env $1 timeout 300 python bench.py --legs $2 --steps 40 --warmup 8 --repeats 2 2> gpurun_out/s_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{}); print('[$1]', d['value'], d['ms_per_step'], d['repeats']['values'], [(k, d[k]['value']) for k in ('kf_realistic','single_stream') if k in d], ' '.join('%s %.3f' % (k[:9], v) for k, v in st.items() if k[:3] in ('sub','rec','ste','lk_')))"
}
run KVFE_SUBPIX_PRIO=0 kf_realistic
run KVFE_SUBPIX_PRIO=1 kf_realistic
run KVFE_SUBPIX_PRIO=0 none
run KVFE_SUBPIX_PRIO=1 none
