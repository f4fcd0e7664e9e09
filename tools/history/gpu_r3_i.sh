# round 3, call i: few-stream fork swap (corner refinement on the main stream, chain on the side stream; default for
# B <= 4) and the tracking launch without its unread error pass: full suite, suite subset with the swap off, single-stream
# leg A/B, main leg A/B (KVFE_LK_ERR=1 = error pass computed), single-stream kernel trace.
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
timeout 60 python -c "
import sys; sys.path.insert(0,'tests')
import test_gpu_pyramid_r3 as T
c=T._ctx(752,480,2,win=8); print('sanity ok')" || { echo "SANITY FAILED"; exit 1; }
timeout 700 python -m pytest tests -m gpu -x -q > gpurun_out/i_tests.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 gpurun_out/i_tests.log
[ $rc -ne 0 ] && grep -E "Error|FAILED|assert" gpurun_out/i_tests.log | head -12
KVFE_FORK_SWAP=0 timeout 400 python -m pytest tests/test_gpu_pipelined_r3.py tests/test_gpu_parity.py tests/test_gpu_replay_r3.py -m gpu -q -k "pipelined or split or sequence or replay" > gpurun_out/i_tests_noswap.log 2>&1; echo "no-swap subset rc=$?"; tail -2 gpurun_out/i_tests_noswap.log
run() {
env $1 timeout 300 python bench.py --legs $2 --steps 30 --warmup 8 --repeats 2 2> gpurun_out/i_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{}); print('[$1]', d['value'], d['ms_per_step'], d['repeats']['values'], [(k, d[k]['value'], d[k].get('repeats',{}).get('values')) for k in ('single_stream',) if k in d], ' '.join('%s %.3f' % (k[:9], v) for k, v in st.items()))"
}
for rep in 1 2; do
  run KVFE_FORK_SWAP=0 single_stream
  run KVFE_FORK_SWAP=1 single_stream
  run KVFE_LK_ERR=1 none
  run KVFE_X=0 none
done
cd /tmp
timeout 120 rocprofv3 --kernel-trace -d $R/gpurun_out/i_kt_c2 -o kt -- python $R/bench.py --config c2 --steps 200 --warmup 20 --repeats 1 --legs none --no-stage-events > $R/gpurun_out/i_kt_c2.log 2>&1; echo "kt c2 rc=$?"
grep -o '"value": [0-9.]*' $R/gpurun_out/i_kt_c2.log | head -1
