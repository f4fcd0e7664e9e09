# round 4, call u (experiment): the output records moved by the DMA engine (hipMemcpyAsync) instead of the copy kernel --
# does the tracking launch still end when the transfer ends?
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
for X in 0 1 0 1; do
KVFE_X_DMA=$X timeout 300 python bench.py --legs outputs --steps 30 --warmup 8 --repeats 3 --stage-event-stride 4 2> gpurun_out/u_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{})
print('[DMA=$X]', d['value'], d['ms_per_step'], d['repeats']['values'], ' '.join('%s %.3f' % (k[:7], v) for k, v in st.items()))
for k in ('outputs_inclusive',):
    if k in d: print('   ', k, {a: d[k].get(a) for a in ('value','ms_per_step','vs_no_readback') if a in d[k]})
"
done
cd /tmp
KVFE_X_DMA=1 timeout 200 rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/prof_u -o kt -- python $R/bench.py --steps 20 --warmup 5 --repeats 1 --legs none > $R/gpurun_out/prof_u.log 2>&1; echo "kt rc=$?"
