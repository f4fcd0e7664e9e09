# round 3, call n: cornerSubPix launch with the stream as the fast grid index and a strided corner loop (1 792 blocks
# instead of ~50 k, most of which had nothing to do): full suite, then side-stream priority on / off on one box.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
timeout 60 python -c "
import sys; sys.path.insert(0,'tests')
import test_gpu_pyramid_r3 as T
c=T._ctx(752,480,2,win=8); print('sanity ok')" || { echo "SANITY FAILED"; exit 1; }
timeout 700 python -m pytest tests -m gpu -x -q > gpurun_out/n_tests.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 gpurun_out/n_tests.log
[ $rc -ne 0 ] && grep -E "Error|FAILED|assert" gpurun_out/n_tests.log | head -12
run() {
env $1 timeout 300 python bench.py --legs $2 --steps 40 --warmup 8 --repeats 2 2> gpurun_out/n_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{}); print('[$1]', d['value'], d['ms_per_step'], d['repeats']['values'], [(k, d[k]['value']) for k in ('nominal','c5','kf_realistic','klt_max_level_4','single_stream') if k in d], 'roofline', d['roofline']['frac'], ' '.join('%s %.3f' % (k[:9], v) for k, v in st.items()))"
}
for rep in 1 2; do
  run KVFE_SIDE_PRIO=0 kf_realistic
  run KVFE_SIDE_PRIO=1 kf_realistic
done
run KVFE_SIDE_PRIO=0 single_stream,c5
