# round 4, call r (experiment): with the chain off the loop (fork_swap), start it only when the corner refinement is done
# (result: step 1.145 -> 1.263 ms, not kept -- the KVFE_X_SERIAL switch is not in the tree)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
for X in 0 1; do
KVFE_X_SERIAL=$X timeout 300 python bench.py --legs nominal,kf_realistic,c5 --steps 30 --warmup 8 --repeats 3 --stage-event-stride 4 2> gpurun_out/r_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{})
print('[SERIAL=$X]', d['value'], d['ms_per_step'], d['repeats']['values'], ' '.join('%s %.3f' % (k[:7], v) for k, v in st.items()))
for k in ('nominal','kf_realistic','c5'):
    if k in d: print('   ', k, {a: d[k].get(a) for a in ('value','ms_per_step') if a in d[k]}, ' '.join('%s %.3f' % (kk[:7], v) for kk, v in d[k].get('stage_ms_per_step_summed_over_groups',{}).items()))
"
done
