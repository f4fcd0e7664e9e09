# LK kernel A/B on one box
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/kimera_vio_amd/csrc
run() {
KVFE_LIB=$L/$1 KVFE_LK_STREAM_MAJOR=$2 python bench.py --legs none --steps 30 --warmup 8 --repeats 2 --stage-event-stride 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step_summed_over_groups']; print('$1 sm=$2 c3', d['value'], 'lk', s.get('lk_track'), 'pyr', s.get('pyramid'), d['check'])"
}
run libkvfe.so 0
run libkvfe_rowtest.so 0
KVFE_LIB=$L/libkvfe_rowtestprof.so python bench.py --legs none --steps 10 --warmup 3 --repeats 1 --no-stage-events 2>&1 | grep "KVFE_LK_PROF one\|timeline"
