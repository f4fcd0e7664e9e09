for m in 3 1 2; do echo "== rect mode $m (no side stream)"; KVFE_NO_SIDE_STREAM=1 KVFE_RECT_MODE=$m python bench.py --no-cpu-baseline --no-single-stream | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms_per_step_summed_over_groups'])"
done
