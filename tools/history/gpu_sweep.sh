mkdir -p gpurun_out
for g in 1 2 4 8; do
  echo "== groups $g"; python bench.py --groups $g --no-cpu-baseline --no-single-stream 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d['stage_ms_per_step_summed_over_groups'])
"
done
echo "== groups 1 no side"; KVFE_NO_SIDE_STREAM=1 python bench.py --groups 1 --no-cpu-baseline --no-single-stream | cut -c1-200
echo "== groups 2 no stage events"; python bench.py --groups 2 --no-stage-events --no-cpu-baseline --no-single-stream | cut -c1-200
echo "== groups 1 no stage events"; python bench.py --groups 1 --no-stage-events --no-cpu-baseline --no-single-stream --steps 200 | cut -c1-200
rocm-smi --showclocks | head -30
