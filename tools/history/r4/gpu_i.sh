# round 4, call i: cornerSubPix (one corner per block) with __launch_bounds__(128, 3): 168 instead of 172 VGPRs = 3 instead of
# 2 waves per SIMD.  A/B of two builds on one box (libkvfe_base.so = without)
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/kimera_vio_amd/csrc
timeout 600 python -m pytest tests/test_gpu_bench_configs.py -m gpu -q -x > gpurun_out/i_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/i_tests.log | cut -c1-200
run() {
KVFE_LIB=$L/$1 KVFE_SUBPIX_GROUP=$3 timeout 300 python bench.py --legs ${2:-none} --steps 30 --warmup 8 --repeats 2 --stage-event-stride 4 2> gpurun_out/i_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{})
print('[$1 group=$3]', d['value'], d['ms_per_step'], d['repeats']['values'], ' '.join('%s %.3f' % (k[:7], v) for k, v in st.items()))
for k in ('kf_realistic','c5'):
    if k in d: print('   ', k, {a: d[k].get(a) for a in ('value','ms_per_step') if a in d[k]}, ' '.join('%s %.3f' % (kk[:7], v) for kk, v in d[k].get('stage_ms_per_step_summed_over_groups',{}).items()))
"
}
run libkvfe_base.so kf_realistic 0
run libkvfe.so kf_realistic 0
run libkvfe_base.so none 0
run libkvfe.so none 0
