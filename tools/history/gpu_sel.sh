# select-kernel iteration loop: detection parity tests, phase stamps on the single EuRoC stream, stage timers
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "detect or anms or gftt or select or bench_configs or fuzz_frontend or sequence" 2>&1 | tail -3
KVFE_SELECT_PROF=1 python bench.py --config c2 --legs none --steps 100 --warmup 10 --repeats 1 --no-stage-events 2>&1 | grep -v "^{" | tail -1
python bench.py --config c2 --legs none --steps 200 --warmup 20 --repeats 2 --stage-event-stride 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c2', d['value'], d['stage_ms_per_step_summed_over_groups'])"
python bench.py --legs none --steps 20 --warmup 5 --repeats 2 --stage-event-stride 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c3', d['value'], d['stage_ms_per_step_summed_over_groups'])"
