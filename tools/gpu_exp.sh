timeout 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for r in 1; do echo "== ransac $r"; python bench.py --ransac $r --no-cpu-baseline | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms_per_step_summed_over_groups'], d['check'], d['single_stream'])"
done
