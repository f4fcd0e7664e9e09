# round 3, call g: the SSD search on the matrix cores (KVFE_SSD_IMPL=1) and the split tracking launch (KVFE_LK_SPLIT=1):
# the whole GPU suite with both on, isolating runs if it fails, then the bench main leg for the four combinations on
# this one box, the side legs once per combination, and a kernel trace with both on.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
timeout 60 python -c "
import sys; sys.path.insert(0,'tests')
import test_gpu_pyramid_r3 as T
c=T._ctx(752,480,2,win=8); print('sanity ok')" || { echo "SANITY FAILED"; exit 1; }
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/g_tests_default.log 2>&1; rc=$?
echo "pytest default (SSD on the matrix cores) rc=$rc"; tail -3 gpurun_out/g_tests_default.log
SSD=1; SPL=1
[ $rc -ne 0 ] && { SSD=0; grep -E "Error|FAILED|assert" gpurun_out/g_tests_default.log | head -12; }
KVFE_LK_SPLIT=1 timeout 400 python -m pytest tests/test_gpu_pipelined_r3.py tests/test_gpu_parity.py tests/test_gpu_bench_configs.py tests/test_gpu_fuzz_slices.py -m gpu -q -k "pipelined or split or sequence or bench or config or fuzz" > gpurun_out/g_tests_split.log 2>&1; r2=$?
echo "split rc=$r2"; tail -3 gpurun_out/g_tests_split.log; [ $r2 -ne 0 ] && { SPL=0; grep -E "Error|FAILED|assert" gpurun_out/g_tests_split.log | head -12; }
run() {  # ssd split legs steps
KVFE_SSD_IMPL=$1 KVFE_LK_SPLIT=$2 timeout 300 python bench.py --legs $3 --steps $4 --warmup 8 --repeats 2 2> gpurun_out/g_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{}); print('ssd=$1 split=$2', d['value'], d['ms_per_step'], d['repeats']['values'], [(k, d[k]['value']) for k in ('nominal','single_stream','c5','kf_realistic','klt_max_level_4') if k in d], ' '.join('%s %.3f' % (k[:9], v) for k, v in st.items()))"
}
for rep in 1 2; do
  run 0 0 none 30
  [ $SSD = 1 ] && run 1 0 none 30
  [ $SPL = 1 ] && run 0 1 none 30
  [ $SSD = 1 ] && [ $SPL = 1 ] && run 1 1 none 30
done
run 0 0 nominal,kf_realistic,single_stream,c5 20
[ $SSD = 1 ] && [ $SPL = 1 ] && run 1 1 nominal,kf_realistic,single_stream,c5 20
[ $SSD = 1 ] && [ $SPL = 0 ] && run 1 0 nominal,kf_realistic,single_stream,c5 20
[ $SSD = 0 ] && [ $SPL = 1 ] && run 0 1 nominal,kf_realistic,single_stream,c5 20
cd /tmp
KVFE_SSD_IMPL=$SSD KVFE_LK_SPLIT=$SPL timeout 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/g_kt -o kt -- python $R/bench.py --steps 20 --warmup 5 --repeats 1 --legs none --no-stage-events > $R/gpurun_out/g_kt.log 2>&1; echo "kt rc=$?"
cd $R
python tools/rocpd_stats.py $(find gpurun_out/g_kt -name "*.db" | head -1) 2>/dev/null | head -30
