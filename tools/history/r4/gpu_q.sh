# round 4, call q (experiment): the corner refinement on the main stream and the rectify / match / reject chain on the side
# (result: +1.9 % headline, +2.0 % kf_realistic, +1.1 % nominal, +0.2 % c5; kept as the rule for persisting device frames -- the KVFE_X_SWAP switch is not in the tree)
# stream for 64 streams too (fork_swap), without the chain join in front of the tracking launch when the frames persist
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
KVFE_X_SWAP=1 timeout 600 python -m pytest tests/test_gpu_pipelined_r3.py tests/test_gpu_bench_configs.py -m gpu -q -x -k "not dense" > gpurun_out/q_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/q_tests.log | cut -c1-300
for X in 0 1; do
KVFE_X_SWAP=$X timeout 300 python bench.py --legs nominal,kf_realistic,c5 --steps 30 --warmup 8 --repeats 3 --stage-event-stride 4 2> gpurun_out/q_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{})
print('[SWAP=$X]', d['value'], d['ms_per_step'], d['repeats']['values'], ' '.join('%s %.3f' % (k[:7], v) for k, v in st.items()))
for k in ('nominal','kf_realistic','c5'):
    if k in d: print('   ', k, {a: d[k].get(a) for a in ('value','ms_per_step') if a in d[k]}, ' '.join('%s %.3f' % (kk[:7], v) for kk, v in d[k].get('stage_ms_per_step_summed_over_groups',{}).items()))
"
done
