# round 4, call l (result: no difference within the noise on the headline, kf_realistic or c5 -- the simpler full grid stays):
# grouped cornerSubPix, groups handed out through a per-stream counter to a device-sized grid
# (libkvfe.so) against one block per group on the full grid (libkvfe_base.so), on one box
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/kimera_vio_amd/csrc
timeout 900 python -m pytest tests/test_gpu_bench_configs.py tests/test_gpu_pipelined_r3.py -m gpu -q -x > gpurun_out/l_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/l_tests.log | cut -c1-200
run() {
KVFE_LIB=$L/$1 timeout 300 python bench.py --legs ${2:-none} --steps 30 --warmup 8 --repeats 3 --stage-event-stride 4 2> gpurun_out/l_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{})
print('[$1]', d['value'], d['ms_per_step'], d['repeats']['values'], ' '.join('%s %.3f' % (k[:7], v) for k, v in st.items()))
for k in ('kf_realistic','c5'):
    if k in d: print('   ', k, {a: d[k].get(a) for a in ('value','ms_per_step') if a in d[k]}, ' '.join('%s %.3f' % (kk[:7], v) for kk, v in d[k].get('stage_ms_per_step_summed_over_groups',{}).items()))
"
}
run libkvfe_base.so kf_realistic,c5
run libkvfe.so kf_realistic,c5
run libkvfe_base.so none
run libkvfe.so none
