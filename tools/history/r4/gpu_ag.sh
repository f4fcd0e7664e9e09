# round 4, call ag (experiment): with the rectification already done behind the keyframe decision, start the SSD search of the
# tracked keypoints only when the corner refinement is done (cornerSubPix alone on the chip)
# (result: cornerSubPix 0.405 -> 0.384 ms, tracking launch 0.527 -> 0.576 ms, step 1.067 -> 1.158 ms: not kept -- the KVFE_X_CHAIN_LATE switch is
# not in the tree)
mkdir -p gpurun_out; export TMPDIR=/tmp
for X in 0 1 0 1; do
KVFE_X_CHAIN_LATE=$X timeout 300 python bench.py --legs kf_realistic --steps 30 --warmup 8 --repeats 3 --stage-event-stride 4 2> gpurun_out/ag_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); st=d.get('stage_ms_per_step_summed_over_groups',{})
print('[late=$X]', d['value'], d['ms_per_step'], d['repeats']['values'], 'subpix %.3f lk %.3f' % (st['subpix_append'], st['lk_track']), 'kf_realistic', d['kf_realistic']['value'])
"
done
