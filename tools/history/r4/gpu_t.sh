# round 4, call t (experiment): four waves per corner in the one-corner-per-block cornerSubPix kernel at 64 streams
# (result: cornerSubPix 0.345 -> 0.386 ms in the step, step 1.148 -> 1.186 ms; two waves per corner stay -- the KVFE_X_NW switch is not in the tree)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
for X in 2 4; do
KVFE_X_NW=$X timeout 300 python bench.py --legs nominal,c5 --steps 30 --warmup 8 --repeats 3 --stage-event-stride 4 2> gpurun_out/t_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{})
print('[NW=$X]', d['value'], d['ms_per_step'], d['repeats']['values'], ' '.join('%s %.3f' % (k[:7], v) for k, v in st.items()))
for k in ('nominal','kf_realistic','c5'):
    if k in d: print('   ', k, {a: d[k].get(a) for a in ('value','ms_per_step') if a in d[k]}, ' '.join('%s %.3f' % (kk[:7], v) for kk, v in d[k].get('stage_ms_per_step_summed_over_groups',{}).items()))
"
done
