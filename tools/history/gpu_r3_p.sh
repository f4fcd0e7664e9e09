# round 3, call p: cornerSubPix with the border branch of getRectSubPix served from the LDS stage: full suite, then
# libkvfe_base.so (previous commit) against libkvfe.so: main leg with stage times, kf_realistic, single stream.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; L=$PWD/kimera_vio_amd/csrc
timeout 60 python -c "
import sys; sys.path.insert(0,'tests')
import test_gpu_pyramid_r3 as T
c=T._ctx(752,480,2,win=8); print('sanity ok')" || { echo "SANITY FAILED"; exit 1; }
timeout 700 python -m pytest tests -m gpu -x -q > gpurun_out/p_tests.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 gpurun_out/p_tests.log
[ $rc -ne 0 ] && grep -E "Error|FAILED|assert" gpurun_out/p_tests.log | head -12
run() {
KVFE_LIB=$L/$1 timeout 300 python bench.py --legs $2 --steps 40 --warmup 8 --repeats 2 2> gpurun_out/p_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{}); print('$1', d['value'], d['ms_per_step'], d['repeats']['values'], [(k, d[k]['value']) for k in ('nominal','c5','kf_realistic','klt_max_level_4','single_stream') if k in d], ' '.join('%s %.3f' % (k[:9], v) for k, v in st.items()))"
}
run libkvfe_base.so none
run libkvfe.so none
run libkvfe_base.so kf_realistic,single_stream
run libkvfe.so kf_realistic,single_stream
