# round 4, call af (experiment): tracking launch with the stream as the FAST grid index (all streams' point r together, the empty
# ranks at the end of the grid) instead of the point index
# (result: 0.530 -> 0.533 ms, step 1.068 -> 1.079 ms: not kept)
mkdir -p gpurun_out; export TMPDIR=/tmp
L=$PWD/kimera_vio_amd/csrc
for V in libkvfe_base.so libkvfe_tr.so libkvfe_base.so libkvfe_tr.so; do
KVFE_LIB=$L/$V timeout 300 python bench.py --legs none --steps 30 --warmup 8 --repeats 3 --stage-event-stride 4 2> gpurun_out/af_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{})
print('[$V]', d['value'], d['ms_per_step'], d['repeats']['values'], 'lk %.4f' % st['lk_track'])
"
done
