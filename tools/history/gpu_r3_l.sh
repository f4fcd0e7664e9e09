# round 3, call l: stereo outlier rejection at the head of the tail on the side stream (KVFE_RANSAC_TAIL=1) instead of on
# the main stream: parity subset with it on, then A/B of the main leg and the side legs on one box.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
timeout 60 python -c "
import sys; sys.path.insert(0,'tests')
import test_gpu_pyramid_r3 as T
c=T._ctx(752,480,2,win=8); print('sanity ok')" || { echo "SANITY FAILED"; exit 1; }
KVFE_RANSAC_TAIL=1 timeout 500 python -m pytest tests/test_gpu_pipelined_r3.py tests/test_gpu_parity.py tests/test_gpu_bench_configs.py tests/test_gpu_fuzz_slices.py tests/test_gpu_pnp.py tests/test_gpu_replay_r3.py -m gpu -x -q > gpurun_out/l_tests.log 2>&1; rc=$?
echo "pytest (ransac in the tail) rc=$rc"; tail -3 gpurun_out/l_tests.log
[ $rc -ne 0 ] && grep -E "Error|FAILED|assert" gpurun_out/l_tests.log | head -12
run() {
env $1 timeout 300 python bench.py --legs $2 --steps 30 --warmup 8 --repeats 2 2> gpurun_out/l_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{}); print('[$1]', d['value'], d['ms_per_step'], d['repeats']['values'], [(k, d[k]['value']) for k in ('nominal','c5','kf_realistic','klt_max_level_4') if k in d], ' '.join('%s %.3f' % (k[:9], v) for k, v in st.items()))"
}
for rep in 1 2 3; do
  run KVFE_X=0 none
  run KVFE_RANSAC_TAIL=1 none
done
run KVFE_X=0 nominal,c5,kf_realistic,klt4
run KVFE_RANSAC_TAIL=1 nominal,c5,kf_realistic,klt4
