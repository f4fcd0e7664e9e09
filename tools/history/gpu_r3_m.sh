# round 3, call m: side stream at high priority (KVFE_SIDE_PRIO=1), four waves per corner in cornerSubPix
# (KVFE_SUBPIX_WAVES=4), and the split tracking launch again on top of the rejection-in-the-tail default (with and
# without the priority): main leg + kf_realistic on one box.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
timeout 60 python -c "
import sys; sys.path.insert(0,'tests')
import test_gpu_pyramid_r3 as T
c=T._ctx(752,480,2,win=8); print('sanity ok')" || { echo "SANITY FAILED"; exit 1; }
run() {
env $1 timeout 300 python bench.py --legs $2 --steps 30 --warmup 8 --repeats 2 2> gpurun_out/m_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{}); print('[$1]', d['value'], d['ms_per_step'], d['repeats']['values'], [(k, d[k]['value']) for k in ('nominal','c5','kf_realistic','klt_max_level_4','single_stream') if k in d], ' '.join('%s %.3f' % (k[:9], v) for k, v in st.items()))"
}
for rep in 1 2; do
  run KVFE_X=0 kf_realistic
  run KVFE_SIDE_PRIO=1 kf_realistic
  run KVFE_SUBPIX_WAVES=4 kf_realistic
  run "KVFE_LK_SPLIT=1 KVFE_SIDE_PRIO=1" kf_realistic
  run "KVFE_LK_SPLIT=1" kf_realistic
done
