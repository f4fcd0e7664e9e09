# round 4, call aa (experiment): rectification on the side stream right behind the keyframe decision (beside the mono
# rejection, the min-eigenvalue launch and the selection) instead of beside cornerSubPix
# (result: rectification 0.167 -> 0.046 ms in the step, cornerSubPix 0.423 -> 0.401, step 1.076 -> 1.070 ms, real frames +1.6 %,
# reference cadence -2 %: kept for steps whose inputs force a keyframe on every stream -- the KVFE_X_RECT_EARLY switch is not in the tree)
mkdir -p gpurun_out; export TMPDIR=/tmp
KVFE_X_RECT_EARLY=1 timeout 600 python -m pytest tests/test_gpu_bench_configs.py tests/test_gpu_pipelined_r3.py -m gpu -q -x -k "not dense and not forced and not transfer" 2>&1 | tail -2
for X in 0 1 0 1; do
KVFE_X_RECT_EARLY=$X timeout 300 python bench.py --legs nominal,kf_realistic --steps 30 --warmup 8 --repeats 3 --stage-event-stride 4 2> gpurun_out/aa_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{})
print('[early=$X]', d['value'], d['ms_per_step'], d['repeats']['values'], 'roofline', d['roofline']['kernel'], d['roofline']['frac'], ' '.join('%s %.3f' % (k[:7], v) for k, v in st.items()))
for k in ('nominal','kf_realistic'):
    if k in d: print('   ', k, {a: d[k].get(a) for a in ('value','ms_per_step') if a in d[k]}, d[k]['repeats']['values'])
"
done
