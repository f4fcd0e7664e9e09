# round 4, call p (experiment): why does track_finalize start only when the output transfer has ended?
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
for X in 0 3 4 5 0 3; do
KVFE_X_OUT=$X timeout 300 python bench.py --legs none --steps 30 --warmup 8 --repeats 3 --stage-event-stride 4 2> gpurun_out/p_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{})
print('[X=$X]', d['value'], d['ms_per_step'], d['repeats']['values'], ' '.join('%s %.3f' % (k[:7], v) for k, v in st.items()))
"
done
cd /tmp
for X in 3 4 5; do
KVFE_X_OUT=$X timeout 200 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_p$X -o kt -- python $R/bench.py --steps 20 --warmup 5 --repeats 1 --legs none > $R/gpurun_out/prof_p.log 2>&1; echo "kt rc=$?"
done
