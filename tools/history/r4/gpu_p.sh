# round 4, call p (experiment, the KVFE_X_OUT switch it used is not in the tree): in the kernel traces track_finalize starts
# 6 us after max(end of the tracking launch, end of the output transfer kernel).  Moving the transfer behind the tail's
# event on the tail's stream (2), binding the wait through a relay stream (3), both (4) change nothing; dropping the
# wait (5, racy) gains 0.7 %.  The tracking launch itself ends within +-5 us of the transfer's end in most steps: its
# completion appears to wait for the transfer's PCIe writes (not proven).  Worth <= 1 % of the step: left alone.
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
