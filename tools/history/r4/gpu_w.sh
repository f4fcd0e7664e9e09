# round 4, call w: timeline of the step after the fork swap and the DMA output transfer; five repeats (is only the first slow?)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
timeout 300 python bench.py --legs none --steps 40 --warmup 8 --repeats 5 --stage-event-stride 4 2> gpurun_out/w_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['repeats']['values'])"
timeout 300 python bench.py --legs none --steps 40 --warmup 40 --repeats 3 --stage-event-stride 4 2> gpurun_out/w_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('warmup 40:', d['value'], d['ms_per_step'], d['repeats']['values'])"
cd /tmp
timeout 200 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_w -o kt -- python $R/bench.py --steps 20 --warmup 5 --repeats 1 --legs none > $R/gpurun_out/prof_w.log 2>&1; echo "kt rc=$?"
