# A/B of two builds of libkvfe.so on ONE box: full GPU parity tests on the working-tree build, then the bench main leg
# alternating libkvfe_base.so (see tools/gpu_lk_ab.sh for how to build it) and libkvfe.so.  BENCH_ARGS: extra bench flags.
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/kimera_vio_amd/csrc
timeout 60 python -c "
import sys; sys.path.insert(0,'tests')
import test_gpu_pyramid_r3 as T
c=T._ctx(752,480,2,win=8); print('sanity ok')" || { echo "SANITY FAILED"; exit 1; }
timeout 900 python -m pytest tests -m gpu -x -q ${TEST_ARGS:-} > gpurun_out/gpu_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && { grep -E "Error|FAILED|assert" gpurun_out/gpu_tests.log | head -20; exit 1; }
run() {
KVFE_LIB=$L/$1 timeout 300 python bench.py --legs ${LEGS:-none} --steps 30 --warmup 8 --repeats 2 ${BENCH_ARGS:---stage-event-stride 0} | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{}); print('$1', d['value'], d['ms_per_step'], d['repeats']['values'], [(k, d[k]['value']) for k in ('nominal','single_stream','c5','kf_realistic') if k in d], ' '.join('%s %.3f' % (k[:9], v) for k, v in st.items()))"
}
for lib in libkvfe_base.so libkvfe.so libkvfe_base.so libkvfe.so; do [ -f $L/$lib ] && run $lib; done
